#!/bin/bash
# HBM-traffic counters only (the rd / wr / fetch / write passes of tools/profile_passes.sh) for one bench shape:
#   bash tools/traffic_passes.sh <tag> [bench args...]  -> gpurun_out/traffic_<tag>_summary.txt
set -u
TAG=${1:-t}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/traffic_$TAG
mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
EXTRA="$*"
run() { local name=$1; shift
  timeout 150 rocprofv3 "$@" -d $OUT/$name -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --stream-only $EXTRA > $OUT/$name.log 2>&1; }
run stats --kernel-trace --stats
run rd --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wr --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python3 $ROOT/tools/rocprof_summary.py $OUT/{stats,rd,wr,fetch,write}/p_results.db > $ROOT/gpurun_out/traffic_${TAG}_summary.txt 2>&1
wc -l $ROOT/gpurun_out/traffic_${TAG}_summary.txt
