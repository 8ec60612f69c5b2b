#!/bin/bash
# K2 with the wave index as a scalar and the stores last: tests, phases, timing (default / folded butterflies), flakiness of both
OUT=gpurun_out/r6f; mkdir -p $OUT
python -m pytest tests/test_gpu_render.py tests/test_gpu_block_convolver.py tests/test_gpu_fft.py tests/test_gpu_smoke.py -x -q -m gpu > $OUT/k2_tests.log 2>&1; echo "k2-related tests rc=$?"; tail -3 $OUT/k2_tests.log
EARHIP_LIB=$PWD/libear_amd/lib_prof/libearhip.so python tools/k2_phases.py 1024 1024 2>/dev/null | tee $OUT/k2_phases.txt
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for rep in 1 2; do
  for lib in default fold; do
    [ $lib = fold ] && export EARHIP_LIB=$PWD/libear_amd/lib_fold/libearhip.so || unset EARHIP_LIB
    python bench.py --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c4_$lib.json; line $OUT/c4_$lib.json "C4 k2=$lib"
    python bench.py --config C3 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c3_$lib.json; line $OUT/c3_$lib.json "C3 k2=$lib"
  done
done
unset EARHIP_LIB
python bench.py --config C5 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c5.json; line $OUT/c5.json "C5"
for lib in default fold; do
  [ $lib = fold ] && export EARHIP_LIB=$PWD/libear_amd/lib_fold/libearhip.so || unset EARHIP_LIB
  echo "flakiness, k2=$lib:"
  for i in $(seq 1 50); do python -m pytest tests/test_gpu_render.py -x -q -m gpu -k concurrent_threads 2>&1 | grep -E "AssertionError:|passed|failed" | sed "s/ in .*//" | tail -2; done | sort | uniq -c
done
