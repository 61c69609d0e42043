#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05aa
timeout 600 python -m pytest tests/test_gpu_trace.py -q -x -k "soma" 2>&1 | tail -3
for w in c2soma c2soma2; do
timeout 900 python bench.py --workload $w --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05aa/$w.json 2> gpurun_out/r05aa/$w.err
done
KH_SOMA_LANES=1 timeout 900 python bench.py --workload c2soma2 --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05aa/c2soma2_serial.json 2> gpurun_out/r05aa/c2soma2_serial.err
python - <<'PY'
import json
for n in ("c2soma", "c2soma2", "c2soma2_serial"):
    try:
        d = json.loads(open("gpurun_out/r05aa/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["single_volume_ms"], d["phases_s"])
    except Exception as e:
        print(n, "failed", e)
PY
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05aa/steps20.json 2> gpurun_out/r05aa/steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05aa/steps20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["single_volume_ms"], json.dumps(d["roofline"])[:700])
PY
