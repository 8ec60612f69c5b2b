#!/bin/bash
# ADM-style objects (a block every 20 ms, 5 ms ramps, then hold) with always-ramping objects (a new target every 5 ms) among them
cd $GRAFT_REPO_ROOT
for every in 2 4 8 32; do
  for k in 3 5 6; do
    EARHIP_BENCH_MIXED_BASE=adm EARHIP_BENCH_MIXED_ODD=240,240 EARHIP_BENCH_MIXED_EVERY=$every EARHIP_MFMA=$k python bench.py --scene mixed --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('adm + one fast mover in $every MFMA=$k', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'][:20], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "one in $every MFMA=$k: no line"
  done
done
