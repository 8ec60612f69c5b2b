#!/bin/bash
OUT=gpurun_out/r6p; mkdir -p $OUT
python -m pytest tests/test_gpu_render_full.py tests/test_gpu_render.py tests/test_gpu_smoke.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for rep in 1 2; do
for two in 1 0; do
for sc in moving bursty-moving; do
  EARHIP_BUILD_2K=$two python bench.py --scene $sc --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/${sc}_$two.json; line $OUT/${sc}_$two.json "$sc 2k=$two"
done
done
done
for tpw in 1 2 4 8; do EARHIP_HBUILD_TPW=$tpw python bench.py --scene moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/mv_tpw$tpw.json; line $OUT/mv_tpw$tpw.json "moving 2k tpw=$tpw"; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/k0prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --scene moving --stream-only --steps 20 --warmup 5 > /tmp/k0prof.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/k0prof/p_results.db 2>/dev/null | grep -E "hinge_classify|hinge_build|hinge_gate|level_probe|gain_mix_hg" | head -8
