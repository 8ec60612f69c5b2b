#!/bin/bash
cd $GRAFT_REPO_ROOT
for wgs in 256 384 512 768; do
  for rep in 1 2; do
  EARHIP_H2_WGS=$wgs python bench.py --config C2 --steps 100 --warmup 20 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C2 wgs=$wgs', d['ms_per_step'], d['kernels_ms']['gain_mix'], d['roofline']['frac'], d['parity']['pass'])"
  done
done
for wgs in 256 512; do
  EARHIP_H2_WGS=$wgs python bench.py --objects 128 --steps 100 --warmup 20 --no-secondary --brief 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('obj128 wgs=$wgs', d['ms_per_step'], d['kernels_ms']['gain_mix'], d['roofline']['frac'], d['parity']['pass'])"
done
