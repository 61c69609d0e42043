#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05ah
timeout 600 python bench.py > gpurun_out/r05ah/default.json 2> gpurun_out/r05ah/default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05ah/default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["volumes_in_flight"], d["single_volume_ms"], d["speedup_latency"], d["speedup_throughput"])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
