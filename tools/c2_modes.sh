#!/bin/bash
# Config 2's gain kernel runs in one of two modes per PROCESS (0.227 / 0.246 ms, round 4): N fresh processes, each
# printing K1's time beside the device addresses of its buffers.   bash tools/c2_modes.sh [N] [extra bench args]
N=${1:-10}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $N); do
  python3 $ROOT/bench.py --config C2 --brief --steps 60 --warmup 10 "$@" 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); b=d['config']['buffers']
        print(round(d['kernels_ms']['gain_mix'],4), d['ms_per_step'], d['roofline']['frac'], b['input'], b['outputs'][0], d['parity']['max_channel_rel_rms_vs_cpu'])
"
done
