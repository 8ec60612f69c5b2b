#!/bin/bash
OUT=gpurun_out/r6n; mkdir -p $OUT
python -m pytest tests/test_gpu_render_full.py -x -q -m gpu -k "two_kernel_list_builder or seed_sweep and adm or bursty_audio or both_forms or call_lengths" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for rep in 1 2; do
for two in 1 0; do
  EARHIP_BUILD_2K=$two python bench.py --scene adm --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/adm_$two.json; line $OUT/adm_$two.json "adm 2k=$two"
  EARHIP_BUILD_2K=$two python bench.py --scene panned-adm --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/padm_$two.json; line $OUT/padm_$two.json "panned-adm 2k=$two"
  EARHIP_BUILD_2K=$two EARHIP_HINGE=0 python bench.py --scene moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/mvp_$two.json; line $OUT/mvp_$two.json "moving on piece lists 2k=$two"
done
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/k0prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --scene adm --stream-only --steps 20 --warmup 5 > /tmp/k0prof.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/k0prof/p_results.db 2>/dev/null | grep -E "piece_classify|piece_build|level_probe|gain_mix_p2" | head -8
