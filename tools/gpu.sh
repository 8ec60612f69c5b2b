#!/bin/bash
# tools/gpu.sh NAME TIMEOUT 'command' — run a command on an MI355X box (gpurun), log under gpurun_out/NAME_call.log
cd /root/repo || exit 1
name=$1; to=$2; shift 2
mkdir -p gpurun_out/$name
/usr/local/graft/bin/gpurun --timeout "$to" -- "mkdir -p gpurun_out/$name; $*" > gpurun_out/${name}_call.log 2>&1
tail -4 gpurun_out/${name}_call.log | grep -E "status=|GPU-minutes"
