#!/bin/bash
# the planner's choices at other object counts (fitted at 1024): 256 and 2048 objects, 24 channels
cd $GRAFT_REPO_ROOT
for M in 256 2048; do
for sc in "moving:EARHIP_BENCH_MOVING_PERIOD=240" "adm:EARHIP_BENCH_ADM=960,600" "adm:EARHIP_BENCH_ADM=960,500" "mixed:EARHIP_BENCH_MIXED_BASE=static EARHIP_BENCH_MIXED_ODD=240,240 EARHIP_BENCH_MIXED_EVERY=8" "mixed:EARHIP_BENCH_MIXED_BASE=static EARHIP_BENCH_MIXED_EVERY=64"; do
  scene=${sc%%:*}; envs=${sc#*:}
  for k in 3 4 5 6; do
    env $envs EARHIP_MFMA=$k python bench.py --objects $M --scene $scene --steps 40 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('M=$M $scene [$envs] MFMA=$k', d['ms_per_step'], d['kernels_ms']['gain_mix'], d['roofline']['kernel'][:16], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "M=$M $scene [$envs] MFMA=$k: no line"
  done
done
done
