#!/bin/bash
# Quick SQ-counter passes over the stream-only bench (kernel under tuning).
# usage (on the GPU box): bash tools/profile_quick.sh <tag> [bench args...]; env (EARHIP_*) is inherited
set -u
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_$TAG
mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
EXTRA="$*"
run() { local name=$1; shift
  timeout 150 rocprofv3 "$@" -d $OUT/$name -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --stream-only $EXTRA > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log; }
run stats --kernel-trace --stats
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
run sq2 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_LDS
run sq3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16
run ta --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
DBS=""
for n in stats sq1 sq2 sq3 ta; do [ -f $OUT/$n/p_results.db ] && DBS="$DBS $OUT/$n/p_results.db"; done
python3 $ROOT/tools/rocprof_summary.py $DBS > $ROOT/gpurun_out/prof_${TAG}_summary.txt 2>&1
wc -l $ROOT/gpurun_out/prof_${TAG}_summary.txt
