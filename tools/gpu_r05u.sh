#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05u
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05u/base.json 2> gpurun_out/r05u/base.err
KIMI_HIP_LIB=$PWD/kimimaro_amd/libkimi_hip_w4.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05u/w4.json 2> gpurun_out/r05u/w4.err
python - <<'PY'
import json
for n in ("base", "w4"):
    try:
        d = json.loads(open("gpurun_out/r05u/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"])
    except Exception as e:
        print(n, "failed", e)
PY
