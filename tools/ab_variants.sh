#!/bin/bash
# Same-box A/B of library builds under libear_amd/lib_<NAME>/ (tools/build_variant.sh): on the box,
#   bash tools/ab_variants.sh "<bench args>" base NAME1 NAME2 ...    ("base" = libear_amd/lib)
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for v in "$@"; do
  lib=$ROOT/libear_amd/lib_$v/libearhip.so; [ "$v" = base ] && lib=$ROOT/libear_amd/lib/libearhip.so
  echo "$v: $(EARHIP_LIB=$lib timeout 200 python $ROOT/bench.py --steps 60 --warmup 5 --stream-only $ARGS 2>/dev/null | python $ROOT/tools/benchline.py)"
done; done
