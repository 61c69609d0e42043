#!/bin/bash
# Round-5 GPU call A: candidate spill + ghosts with roll-back -- gate (reference vectors, f4), c2 in the three ghost modes against the
# pooled oracle, then c3 / c2soma single-volume benches with ghosts on and off.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r05a
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_cube.py tests/test_gpu_post.py -x -q -m gpu > $OUT/t.txt 2>&1; rc=$?; tail -3 $OUT/t.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -80 $OUT/t.txt; }
echo "== c2 ghost modes"; timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -s -k "c2_ghost or c2_full" > $OUT/t2.txt 2>&1; tail -8 $OUT/t2.txt
for cfg in "c3_ghosts c3 1" "c3_noghosts c3 0" "c2soma_ghosts c2soma 1"; do
  set -- $cfg
  echo "== $1"; KH_GHOSTS=$3 KIMI_BENCH_INFLIGHT=1 timeout 600 python bench.py --workload $2 --steps 2 --warmup 0 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); print("$1", "single_ms", d["single_volume_ms"], "ms/step", d["ms_per_step"], d["sweep"]); print("   ", d["chains"]["sum_Mcyc"]); [print("   ", c) for c in d["chains"]["longest"]]; print("   phases", d["phases_s"]); print("   roofline", d["roofline"]["frac"], d["roofline"]["launches"])
except Exception as e: print("$1 failed", e); print(open("$OUT/$1.err").read()[-2500:])
PY
done
