#!/bin/bash
# the same question for curves that HOLD most of the time: static gains + a few objects on ADM metadata off the grid
cd $GRAFT_REPO_ROOT
for every in 1024 512 256 128; do
  for k in 3; do
    EARHIP_BENCH_MIXED_BASE=static EARHIP_BENCH_MIXED_EVERY=$every EARHIP_MFMA=$k python bench.py --scene mixed --steps 60 --warmup 10 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('static + one in $every, MFMA=$k', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'][:20], d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])"
  done
done
