#!/bin/bash
# Round-4 GPU call G: staged sssp (searches) -- parity gate + bench.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04g
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_kat.py tests/test_gpu_c3.py -x -q -m gpu > $OUT/t.txt 2>&1; rc=$?; tail -3 $OUT/t.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -60 $OUT/t.txt; exit 1; }
for cfg in "s20 20 5 0" "s36l12 36 2 12"; do
  set -- $cfg
  echo "== $1"; KIMI_BENCH_INFLIGHT=$4 timeout 900 python bench.py --steps $2 --warmup $3 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); print("$1", d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"], d["phases_s"]); print("   ", d["chains"]["sum_Mcyc"], d["chains"]["longest"][0])
except Exception as e: print("$1 failed", e); print(open("$OUT/$1.err").read()[-1500:])
PY
done
