#!/bin/bash
OUT=gpurun_out/r6k; mkdir -p $OUT
python -m pytest tests/test_gpu_render_full.py tests/test_gpu_smoke.py -x -q -m gpu -k "seed_sweep and moving or two_renderers or bursty_audio or both_forms or quiet or long_calls or smoke" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
line() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['plan'].get('form'), d['parity']['pass'], d['parity']['max_channel_rel_rms_vs_cpu'], 'block', (d.get('block_mode') or {}).get('ms_per_block'), [ (c['source'], c['blocks_per_call'], c['frac']) for c in (d.get('host_stream') or {}).get('calls', [])])"; }
python bench.py --scene bursty-moving --steps 20 --warmup 5 --cpu-blocks 0 --no-secondary 2>/dev/null | tail -1 > $OUT/bm.json; line $OUT/bm.json "bursty-moving"
python bench.py --scene moving --steps 20 --warmup 5 --cpu-blocks 0 --no-secondary 2>/dev/null | tail -1 > $OUT/mv.json; line $OUT/mv.json "moving"
python bench.py --steps 20 --warmup 5 --cpu-blocks 0 --no-secondary 2>/dev/null | tail -1 > $OUT/c4.json; line $OUT/c4.json "headline"
