#!/bin/bash
# Collect the bench lines of every configuration / scene for profiles/ (on the GPU box):
#   bash tools/collect_round.sh <tag>     -> gpurun_out/<tag>_bench_<name>.json
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT && mkdir -p gpurun_out
run() { # name, bench args...
  local name=$1; shift
  timeout 400 python bench.py "$@" 2> gpurun_out/${TAG}_bench_${name}.err | tail -1 > gpurun_out/${TAG}_bench_${name}.json
  python tools/benchline.py < gpurun_out/${TAG}_bench_${name}.json | sed "s/^/$name: /"
}
# the driver's entry point first: a round must not end with smoke() red
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1 && echo "smoke: ok" || { echo "smoke: FAILED"; tail -5 gpurun_out/${TAG}_smoke.txt; }
run C4
run C2 --config C2
run C3 --config C3
run C5 --config C5
run adm --scene adm
run static --scene static
run moving --scene moving
run mixed --scene mixed
run panned --scene panned
run panned_adm --scene panned-adm
run levels --scene levels
run levels_adm --scene levels-adm
run bursty --scene bursty
run bursty_adm --scene bursty-adm
run bursty_moving --scene bursty-moving
run obj512 --objects 512 --brief
run obj256 --objects 256 --brief
run obj128 --objects 128 --brief
run refbench --config refbench --steps 50 --warmup 5
EARHIP_MFMA=1 run f32_exact --brief
run blocks513 --blocks 513 --brief
run blocks257 --blocks 257 --brief
run blocks129 --blocks 129 --brief
