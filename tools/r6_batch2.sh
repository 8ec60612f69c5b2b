#!/bin/bash
# round-6 measurement batch 2 (GPU box): new host-pipeline / comm / bench tests, the host_stream leg, config 2 tile runs
OUT=gpurun_out/r6b; mkdir -p $OUT
python -m pytest tests -x -q -m gpu -k "long_calls_from_host or single_rank_exchange or one_gpu_line or libears_own_calling" -s > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -12 $OUT/tests.log
python bench.py --no-secondary --steps 20 --warmup 5 2> $OUT/default.err | tail -1 > $OUT/default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6b/default.json'))
print('headline', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
print('block_mode', d['block_mode']['ms_per_block'], d['block_mode']['pinned_buffers']['ms_per_block'])
hs=d['host_stream']; print('h2d', hs['h2d_GBps_measured'], 'pass', hs.get('pass'))
for c in hs['calls']: print(c)
PY
for runs in 0 1 0 1; do
  EARHIP_H2_RUNS=$runs python bench.py --config C2 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c2_runs$runs.json
  python -c "
import json; d=json.load(open('$OUT/c2_runs$runs.json')); print('C2 runs=$runs', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'])"
done
for runs in 0 1; do
  EARHIP_H2_RUNS=$runs python bench.py --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c4_runs$runs.json
  python -c "
import json; d=json.load(open('$OUT/c4_runs$runs.json')); print('C4 runs=$runs', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'])"
  EARHIP_H2_RUNS=$runs python bench.py --config C3 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c3_runs$runs.json
  python -c "
import json; d=json.load(open('$OUT/c3_runs$runs.json')); print('C3 runs=$runs', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['parity']['pass'])"
done
EARHIP_DEBUG_TIMING=1 python tools/host_stream_rate.py 64 2>&1 | tail -4
