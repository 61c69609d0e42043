#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05w
KIMI_BENCH_LANE_PHASES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05w/phases.json 2> gpurun_out/r05w/phases.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05w/phases.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["single_volume_ms"])
print(json.dumps(d["phases_under_load"]))
PY
