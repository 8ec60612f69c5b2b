#!/bin/bash
# hinge kernel A/B: phase sums (diagnostic build), the two moving scenes' bench lines twice, the hinge parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6hg
EARHIP_LIB=$PWD/libear_amd/lib_hgprof/libearhip.so python tools/hg_phases.py moving 2>&1 | tail -11 | head -10
for s in moving bursty-moving moving bursty-moving; do
  python bench.py --scene $s --steps 80 --warmup 20 --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$s', d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])"
done
python -m pytest tests -q -m gpu -x -k "hinge or moving" 2>&1 | tail -3
