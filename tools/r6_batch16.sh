#!/bin/bash
OUT=gpurun_out/r6q; mkdir -p $OUT
python -m pytest tests/test_gpu_render_full.py tests/test_gpu_render.py tests/test_gpu_smoke.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for rep in 1 2; do
for sc in moving bursty-moving; do
  python bench.py --scene $sc --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/${sc}.json; line $OUT/${sc}.json "$sc"
  EARHIP_BUILD_2K=1 python bench.py --scene $sc --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/${sc}_2k.json; line $OUT/${sc}_2k.json "$sc 2k"
done
done
for tpw in 1 2 8; do EARHIP_HBUILD_TPW=$tpw python bench.py --scene moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/mv_tpw$tpw.json; line $OUT/mv_tpw$tpw.json "moving tpw=$tpw"; done
