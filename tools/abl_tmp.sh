#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT && timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for round in 1 2 3; do for v in narrow wide; do
  if [ $v = narrow ]; then export EARHIP_H2_NARROW=1; else unset EARHIP_H2_NARROW; fi
  echo "$v: $(timeout 200 python $ROOT/bench.py --steps 30 --warmup 5 --stream-only 2>/dev/null | python $ROOT/tools/benchline.py)"
done; done
