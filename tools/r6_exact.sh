#!/bin/bash
# exact-path search through the header: GPU suite, the scenes that use the exact paths, the bursty chunks' launch times
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r6_exact_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/r6_exact_tests.log | tail -1
for s in bursty-moving adm levels-adm moving; do
  python bench.py --scene $s --steps 80 --warmup 20 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$s', d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['max_channel_rel_rms_vs_cpu'], d['parity']['pass'])"
done
OUT=/tmp/prof_x; mkdir -p $OUT; R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- python3 $R/bench.py --no-secondary --scene bursty-moving > $OUT/stats.log 2>&1
python3 $R/tools/rocprof_summary.py $OUT/stats/p_results.db 2>&1 | grep -E "k_gain_mix_hg" | head -8
