#!/bin/bash
# round-5 closing evidence: kernel stats + EDT counter passes, counter passes of the path kernel, the driver's bench call
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05z
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05z/steps20.json 2> gpurun_out/r05z/steps20.err
tail -c 400 gpurun_out/r05z/steps20.err
timeout 500 bash tools/profile_round.sh r05 2>&1 | tail -5
timeout 800 bash tools/pmc_trace_r3.sh r05 2>&1 | tail -8
