#!/bin/bash
OUT=gpurun_out/r6l; mkdir -p $OUT
python -m pytest tests/test_gpu_render.py tests/test_gpu_render_full.py tests/test_gpu_gain_interp.py tests/test_gpu_smoke.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
EARHIP_LIB=$PWD/libear_amd/lib_bprof/libearhip.so python tools/build_phases.py adm 2>/dev/null
EARHIP_LIB=$PWD/libear_amd/lib_bprof/libearhip.so python tools/build_phases.py moving 2>/dev/null
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for sc in adm moving bursty-moving bursty-adm; do
  python bench.py --scene $sc --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/$sc.json; line $OUT/$sc.json "$sc"
done
