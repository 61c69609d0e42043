#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05v
for n in 16 48; do
KH_BIG_LDS_LABELS=$n timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05v/big$n.json 2> gpurun_out/r05v/big$n.err
done
python - <<'PY'
import json
for n in ("big16", "big48"):
    try:
        d = json.loads(open("gpurun_out/r05v/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"])
        print("   ", [(c["longest_voxels"], c["longest_Mcyc"]) for c in d["chains_under_load"]][:6])
    except Exception as e:
        print(n, "failed", e)
PY
