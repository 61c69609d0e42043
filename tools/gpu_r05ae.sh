#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
export GPU_MAX_HW_QUEUES=24
mkdir -p gpurun_out/r05ae
KH_BIG_LDS_LABELS=32 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05ae/big32.json 2> gpurun_out/r05ae/big32.err
KH_BIG_LDS_LABELS=32 KH_BIG_THREADS=256 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05ae/big32t256.json 2> gpurun_out/r05ae/big32t256.err
KH_BIG_LDS_LABELS=64 KH_BIG_LDS_HEAP=0 KH_BIG_THREADS=256 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05ae/big64t256nolds.json 2> gpurun_out/r05ae/big64t256nolds.err
python - <<'PY'
import json
for n in ("big32", "big32t256", "big64t256nolds"):
    try:
        d = json.loads(open("gpurun_out/r05ae/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["single_volume_ms"], d["hbm_reserved_peak_gb"])
        print("   ", [(c["longest_voxels"], c["longest_Mcyc"]) for c in d["chains_under_load"]][:6])
    except Exception as e:
        print(n, "failed", e); print(open("gpurun_out/r05ae/%s.err" % n).read()[-600:])
PY
