#!/bin/bash
OUT=gpurun_out/r6h; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['ms_per_step'], d['kernels_ms'])"; }
for a in 0 1 2 3 0 1 2 3; do
  [ $a = 0 ] && unset EARHIP_LIB || export EARHIP_LIB=$PWD/libear_amd/lib_abl$a/libearhip.so
  python bench.py --stream-only --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c4_abl$a.json; line $OUT/c4_abl$a.json "C4 ablation=$a"
  python bench.py --config C3 --stream-only --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c3_abl$a.json; line $OUT/c3_abl$a.json "C3 ablation=$a"
done
