#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06ak
( while true; do echo "T $(date +%s.%N)"; rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|busy" ; sleep 0.5; done ) > gpurun_out/r06ak/smi.txt 2>&1 &
SMI=$!
bash tools/gpu.sh r06ak lanes:gate:20:2:KIMI_BENCH_INFLIGHT=20 
kill $SMI
