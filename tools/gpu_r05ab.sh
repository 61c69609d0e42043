#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p gpurun_out/r05ab
for s in 0.06 0.12 0.25; do
KIMI_BENCH_STAGGER_S=$s KIMI_BENCH_LANE_PHASES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05ab/st$s.json 2> gpurun_out/r05ab/st$s.err
done
python - <<'PY'
import json
for n in ("0.06", "0.12", "0.25"):
    try:
        d = json.loads(open("gpurun_out/r05ab/st%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["phases_under_load"]["volume_s"])
    except Exception as e:
        print(n, "failed", e)
PY
