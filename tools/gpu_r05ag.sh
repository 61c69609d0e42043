#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05ag
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05ag/steps20.json 2> gpurun_out/r05ag/steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05ag/steps20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["single_volume_ms"], d["speedup_latency"], d["speedup_throughput"], d["roofline"]["launches"])
PY
timeout 400 bash tools/pmc_trace_r3.sh r05 2>&1 | tail -7
