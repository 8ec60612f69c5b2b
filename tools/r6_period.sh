#!/bin/bash
# always-ramping curves at other update periods: the planner's choice against the forced list / hinge kernels
cd $GRAFT_REPO_ROOT
for per in 240 360 480 720 960 1920; do
  for k in 3; do
    EARHIP_BENCH_MOVING_PERIOD=$per EARHIP_MFMA=$k python bench.py --scene moving --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('period $per MFMA=$k', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'][:20], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])"
  done
done
