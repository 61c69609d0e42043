#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05ad
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r05ad/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r05ad/pytest.log | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05ad/steps20.json 2> gpurun_out/r05ad/steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05ad/steps20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["single_volume_ms"], d["phases_s"], d["speedup_latency"], d["speedup_throughput"])
PY
