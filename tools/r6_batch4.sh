#!/bin/bash
# config 2: 256-sample tiles (default, 4 waves) against 512-sample tiles (8 waves), six fresh processes each (two buffer modes)
OUT=gpurun_out/r6d; mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  for tile in 256 512; do
    EARHIP_H2_TILE=$tile python bench.py --config C2 --brief --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c2_t${tile}_$i.json
    python -c "
import json; d=json.load(open('$OUT/c2_t${tile}_$i.json')); print('C2 tile $tile run $i: K1', d['kernels_ms']['gain_mix'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['plan'], d['parity']['pass'])"
  done
done
