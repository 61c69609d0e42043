#!/bin/bash
# Round-4 closing call: the whole GPU suite and the driver's bench call on the final tree.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04z
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== full suite"; timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1; tail -6 $OUT/tests.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench (driver's call)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_s20.json 2> $OUT/bench_s20.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_s20.json")); print(d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"], d.get("speedup_latency"), d.get("speedup_throughput"), d["roofline"]["frac"], d["roofline"]["traffic_source"], d["roofline_trace"]["frac"], d["roofline_trace"]["traffic_source"])
except Exception as e: print("failed", e); print(open("$OUT/bench_s20.err").read()[-2000:])
PY
