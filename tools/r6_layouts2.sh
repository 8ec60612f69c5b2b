#!/bin/bash
# static gains + a few ADM-style movers off the grid on small layouts: planner against the grid kernel forced
cd $GRAFT_REPO_ROOT
for lay in 0+5+0 4+5+0; do
for every in 512 128 32; do
  for k in 3 4 5; do
    EARHIP_BENCH_MIXED_BASE=static EARHIP_BENCH_MIXED_EVERY=$every EARHIP_MFMA=$k python bench.py --layout $lay --scene mixed --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lay static + one in $every MFMA=$k', d['ms_per_step'], d['kernels_ms']['gain_mix'], d['roofline']['kernel'][:16], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "$lay one in $every MFMA=$k: no line"
  done
done
done
bash tools/r6_layouts.sh | grep "MFMA=3"
