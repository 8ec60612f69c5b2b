#!/bin/bash
# round-6 batch 3 (GPU box): K2 with folded butterflies (tests + A/B timing), host-stream chunk sweep, config 2 on 512-sample tiles
OUT=gpurun_out/r6c; mkdir -p $OUT
python -m pytest tests/test_gpu_render.py tests/test_gpu_block_convolver.py tests/test_gpu_fft.py -x -q -m gpu > $OUT/k2_tests.log 2>&1; echo "k2-related tests rc=$?"; tail -3 $OUT/k2_tests.log
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for rep in 1 2; do
  for lib in new old; do
    [ $lib = old ] && export EARHIP_LIB=$PWD/libear_amd/lib_ab/libearhip.so || unset EARHIP_LIB
    python bench.py --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c4_$lib.json; line $OUT/c4_$lib.json "C4 k2=$lib"
    python bench.py --config C3 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c3_$lib.json; line $OUT/c3_$lib.json "C3 k2=$lib"
  done
done
unset EARHIP_LIB
python bench.py --config C5 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c5.json; line $OUT/c5.json "C5"
EARHIP_H2_TILE=512 python bench.py --config C2 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c2_512.json; line $OUT/c2_512.json "C2 tile512"
python bench.py --config C2 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c2.json; line $OUT/c2.json "C2 default"
python tools/host_stream_sweep.py 2>&1 | tee $OUT/host_sweep.txt
