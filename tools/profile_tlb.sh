#!/bin/bash
# TLB / L1 counters for the gain stage (separate PMC pass), see tools/profile_passes.sh
TAG=${1:-tlb}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum -d $OUT/tlb -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-blocks 0 "$@" > $OUT/tlb.log 2>&1
rocprofv3 --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum -d $OUT/tcp -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-blocks 0 "$@" > $OUT/tcp.log 2>&1
