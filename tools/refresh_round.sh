cd $GRAFT_REPO_ROOT
bash tools/collect_round.sh r05 > gpurun_out/r05_collect.txt 2>&1
bash tools/profile_passes.sh r05_final > gpurun_out/r05_pp_final.log 2>&1
bash tools/profile_passes.sh r05_adm --scene adm > gpurun_out/r05_pp_adm.log 2>&1
bash tools/profile_passes.sh r05_moving --scene moving > gpurun_out/r05_pp_moving.log 2>&1
bash tools/traffic_passes.sh r05_C2 --config C2 > gpurun_out/r05_tp_C2.log 2>&1
bash tools/traffic_passes.sh r05_C3 --config C3 > gpurun_out/r05_tp_C3.log 2>&1
bash tools/traffic_passes.sh r05_C5 --config C5 > gpurun_out/r05_tp_C5.log 2>&1
timeout 600 python bench.py 2> gpurun_out/r05_bench_default.err | tail -1 > gpurun_out/r05_bench_default.json
