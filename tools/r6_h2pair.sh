#!/bin/bash
# the 8-wave grid kernel as one launch (both forms inside) against the pair of launches: same box, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "--config C4" "--config C2" "--config C3" "--config C5" "--scene bursty" "--objects 128"; do
  for pair in 0 1; do
    EARHIP_H2_PAIR=$pair python bench.py $cfg --steps 100 --warmup 20 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg pair=$pair', d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['plan']['form'], d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])"
  done
done
done
python -m pytest tests -q -m gpu -x -k "h2 or grid or tile or wide or level or bursty or quiet" 2>&1 | grep -E "passed|failed" | tail -1
