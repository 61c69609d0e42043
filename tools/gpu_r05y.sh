#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05y
for n in 128 256; do
KH_EDF_THREADS=$n KIMI_BENCH_LANE_PHASES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05y/edf$n.json 2> gpurun_out/r05y/edf$n.err
done
python - <<'PY'
import json
for n in (128, 256):
    d = json.loads(open("gpurun_out/r05y/edf%d.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["single_volume_ms"], d["phases_s"]["edf_root"], d["phases_s"]["edf_daf"])
    print(json.dumps(d["phases_under_load"]["mean_s"]))
PY
