#!/bin/bash
OUT=gpurun_out/r6g; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['ms_per_step'], d['kernels_ms'])"; }
EARHIP_LIB=$PWD/libear_amd/lib_prof/libearhip.so python bench.py --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c4_prof.json; line $OUT/c4_prof.json "C4 prof-lib"
for R in 3 5 7 9 11 13 17; do
  EARHIP_RUN=$R python bench.py --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c4_r$R.json; line $OUT/c4_r$R.json "C4 run=$R"
done
for R in 5 9 13 17 21 25 31; do
  EARHIP_RUN=$R python bench.py --config C3 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c3_r$R.json; line $OUT/c3_r$R.json "C3 run=$R"
done
python bench.py --config C3 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c3.json; line $OUT/c3.json "C3 default-run"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/k2prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --brief --steps 20 --warmup 5 > /tmp/k2prof.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/k2prof/p_results.db 2>/dev/null | grep -E "decorrelate|gain_mix_h2|seg_prep|level_probe" | head -12
