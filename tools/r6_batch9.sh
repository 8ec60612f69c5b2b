#!/bin/bash
OUT=gpurun_out/r6i; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $OUT/all.log
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['plan'], d['parity']['pass'])"; }
for i in 1 2 3 4; do
  python bench.py --config C2 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c2_$i.json; line $OUT/c2_$i.json "C2 default(512) run $i"
  EARHIP_H2_WGS=512 python bench.py --config C2 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c2w_$i.json; line $OUT/c2w_$i.json "C2 512 wgs=512 run $i"
done
