#!/bin/bash
# the planner's cost comparison was fitted on 24 channels: its choices on 5 and 10 channels (one / two column tiles)
cd $GRAFT_REPO_ROOT
for lay in 0+5+0 4+5+0; do
for sc in "moving:EARHIP_BENCH_MOVING_PERIOD=240" "moving:EARHIP_BENCH_MOVING_PERIOD=960" "adm:EARHIP_BENCH_ADM=960,600" "adm:EARHIP_BENCH_ADM=960,240" "mixed:EARHIP_BENCH_MIXED_BASE=static EARHIP_BENCH_MIXED_ODD=240,240 EARHIP_BENCH_MIXED_EVERY=8"; do
  scene=${sc%%:*}; envs=${sc#*:}
  for k in 3 5 6; do
    env $envs EARHIP_MFMA=$k python bench.py --layout $lay --scene $scene --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lay $scene [$envs] MFMA=$k', d['ms_per_step'], d['kernels_ms']['gain_mix'], d['roofline']['kernel'][:16], d['roofline']['plan'].get('tile_samples'), d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])" 2>/dev/null || echo "$lay $scene [$envs] MFMA=$k: no line"
  done
done
done
