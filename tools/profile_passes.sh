#!/bin/bash
# Collect the rocprofv3 passes behind profiles/: kernel trace + stats, then PMC counters in
# separate passes (never combined with tracing domains other than --kernel-trace).
# usage (on the GPU box): bash tools/profile_passes.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_$TAG   # the sqlite results stay on the box (gpurun_out/ is capped at 64 MiB)
mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout 150 rocprofv3 "$@" -d $OUT/$name -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --stream-only $EXTRA > $OUT/$name.log 2>&1
}
EXTRA="$*"
# the stats pass profiles the DEFAULT bench command (stream steps + block-mode + parity + CPU legs) WITHOUT its secondary
# workloads: those are child processes (the profiler would follow them and mix their kernels into the headline's stats)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- python3 $ROOT/bench.py --no-secondary $EXTRA > $OUT/stats.log 2>&1
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
run sq2 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_LDS
run ta --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
run rd --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wr --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python3 $ROOT/tools/rocprof_summary.py $OUT/{stats,sq1,sq2,ta,rd,wr,fetch,write}/p_results.db > $ROOT/gpurun_out/prof_${TAG}_summary.txt 2>&1
wc -l $ROOT/gpurun_out/prof_${TAG}_summary.txt
