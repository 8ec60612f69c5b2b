#!/bin/bash
# always-ramping objects with some objects on STEP metadata (jumpPosition: a new gain vector every 20 ms, no ramp) among them:
# such pairs are the hinge kernel's exact path; how many are worth it?
cd $GRAFT_REPO_ROOT
for every in 100 50 25 12 6; do
  for k in 3 5 6; do
    EARHIP_BENCH_MIXED_BASE=moving EARHIP_BENCH_MIXED_ODD=960,0 EARHIP_BENCH_MIXED_EVERY=$every EARHIP_MFMA=$k python bench.py --scene mixed --steps 40 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('moving + one stepper in $every MFMA=$k', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'][:16], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "one in $every MFMA=$k: no line"
  done
done
