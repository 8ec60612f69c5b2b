#!/bin/bash
# paired lists beyond a ramp share of one half: speed and distance from the CPU path (the planner's rule forbids them there)
cd $GRAFT_REPO_ROOT
for pr in "960,470" "960,500" "960,600" "960,800"; do
  for pairs in 0 1; do
    EARHIP_BENCH_ADM=$pr EARHIP_MFMA=5 EARHIP_P2_PAIRS=$pairs python bench.py --scene adm --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('period,ramp $pr pairs=$pairs', d['ms_per_step'], d['kernels_ms'], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "$pr pairs=$pairs: no line"
  done
done
