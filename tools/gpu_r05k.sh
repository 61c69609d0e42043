#!/bin/bash
# Round-5 GPU call K: the rocprofv3 evidence of the round (kernel stats of one bench step, EDT and path-kernel counter passes: one
# --pmc set per run, kernel-trace only), c2soma with its CPU baseline.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== kernel stats + EDT pmc"; bash tools/profile_round.sh r05 2>&1 | tail -3
echo "== path kernel pmc"; bash tools/pmc_trace_r3.sh r05 2>&1 | tail -8
python tools/summarize_profiles.py r05 r05 2>&1 | tail -5
echo "== c2soma"; KIMI_BENCH_INFLIGHT=1 timeout 900 python bench.py --workload c2soma --steps 2 --warmup 0 > gpurun_out/prof_r05/bench_c2soma.json 2> gpurun_out/prof_r05/bench_c2soma.err; tail -c 1500 gpurun_out/prof_r05/bench_c2soma.json; tail -3 gpurun_out/prof_r05/bench_c2soma.err
