#!/bin/bash
# hinge kernel's robust form: tests, bursty-moving with it and with the hand-over, moving scene unchanged?
OUT=gpurun_out/r6e; mkdir -p $OUT
python -m pytest tests/test_gpu_render_full.py -x -q -m gpu -k "seed_sweep and moving or two_renderers or bursty_audio or both_forms or quiet" -s > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -E "seed sweep|bursty audio|both forms|passed|failed|Error" $OUT/tests.log | tail -20
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['plan'].get('form'), d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'])"; }
for rep in 1 2; do
  python bench.py --scene bursty-moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/bm_robust.json; line $OUT/bm_robust.json "bursty-moving robust"
  EARHIP_HG_ROBUST=0 python bench.py --scene bursty-moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/bm_standby.json; line $OUT/bm_standby.json "bursty-moving standby"
  python bench.py --scene moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/moving.json; line $OUT/moving.json "moving robust-allowed"
  EARHIP_HG_ROBUST=0 python bench.py --scene moving --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/moving_sb.json; line $OUT/moving_sb.json "moving standby-armed"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
