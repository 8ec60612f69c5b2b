#!/bin/bash
# ramp-then-hold curves (ADM interpolationLength < block duration) at different ramp shares: planner against forced kernels
cd $GRAFT_REPO_ROOT
for pr in "960,96" "960,240" "960,400" "960,470" "960,500" "960,600" "960,800" "480,200" "480,260" "480,400"; do
  for k in 3; do
    EARHIP_BENCH_ADM=$pr EARHIP_MFMA=$k python bench.py --scene adm --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('period,ramp $pr MFMA=$k', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'][:20], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "period,ramp $pr MFMA=$k: no line"
  done
done
