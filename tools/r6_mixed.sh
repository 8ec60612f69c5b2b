#!/bin/bash
# how many objects off the grid the grid kernel's exact path is worth: the planner's own choice against the forced list kernels
cd $GRAFT_REPO_ROOT
for every in 128 64 56 33; do
  for k in 3 5 6; do
    EARHIP_BENCH_MIXED_EVERY=$every EARHIP_MFMA=$k python bench.py --scene mixed --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('one in $every forced=[$k]', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'][:20], d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])"
  done
done
