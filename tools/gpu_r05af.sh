#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05af
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05af/steps20.json 2> gpurun_out/r05af/steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05af/steps20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["single_volume_ms"], d["sweep"]["events"], d["sweep"]["fell_back_to_heap"], d["phases_s"]["paths"])
print([(c["longest_voxels"], c["longest_Mcyc"]) for c in d["chains_under_load"]][:6])
PY
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r05af/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r05af/pytest.log | tail -3
