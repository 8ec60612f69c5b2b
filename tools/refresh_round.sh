cd $GRAFT_REPO_ROOT
bash tools/collect_round.sh r06 > gpurun_out/r06_collect.txt 2>&1
bash tools/profile_passes.sh r06_final > gpurun_out/r06_pp_final.log 2>&1
bash tools/profile_passes.sh r06_adm --scene adm > gpurun_out/r06_pp_adm.log 2>&1
bash tools/profile_passes.sh r06_moving --scene moving > gpurun_out/r06_pp_moving.log 2>&1
bash tools/profile_passes.sh r06_bursty_moving --scene bursty-moving > gpurun_out/r06_pp_bursty_moving.log 2>&1
bash tools/traffic_passes.sh r06_C2 --config C2 > gpurun_out/r06_tp_C2.log 2>&1
bash tools/traffic_passes.sh r06_C3 --config C3 > gpurun_out/r06_tp_C3.log 2>&1
bash tools/traffic_passes.sh r06_C5 --config C5 > gpurun_out/r06_tp_C5.log 2>&1
( time timeout 600 python bench.py ) 2> gpurun_out/r06_bench_default.err | tail -1 > gpurun_out/r06_bench_default.json
