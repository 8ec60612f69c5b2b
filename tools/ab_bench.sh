#!/bin/bash
# Same-box A/B of library builds (box-to-box clocks differ by several percent, so variants
# are only comparable inside one gpurun call).  Put the builds under libear_amd/lib/<name>.so
# (they travel with the snapshot), then on the box:  bash tools/ab_bench.sh "<bench args>" a b [c ...]
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cp $ROOT/libear_amd/lib/libearhip.so /tmp/libearhip_keep.so
for round in 1 2 3; do for v in "$@"; do
  cp $ROOT/libear_amd/lib/$v.so $ROOT/libear_amd/lib/libearhip.so
  echo "$v: $(timeout 200 python $ROOT/bench.py --steps 100 --warmup 10 --stream-only $ARGS 2>/dev/null | python $ROOT/tools/benchline.py)"
done; done
cp /tmp/libearhip_keep.so $ROOT/libear_amd/lib/libearhip.so
