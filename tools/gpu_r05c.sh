#!/bin/bash
# Round-5 GPU call C: the decision-bit heap emulation -- gate on the reference vectors through the heap alone and through the sweep,
# c2 in the three ghost modes against the pooled oracle, cycle split on c3, c3 single-volume bench, c3 parity.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r05c
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_cube.py tests/test_gpu_budget.py -x -q -m gpu > $OUT/t.txt 2>&1; rc=$?; tail -5 $OUT/t.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; grep -n "Error\|assert\|FAILED" $OUT/t.txt | head -40; tail -60 $OUT/t.txt; exit 1; }
echo "== c2 ghost modes"; timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -s -k "c2_ghost or c2_full" > $OUT/t2.txt 2>&1; tail -6 $OUT/t2.txt
echo "== profile split"; KH_PROFILE=1 timeout 600 python tools/dev_gpu_check.py c3 > $OUT/prof.txt 2>&1; grep -n "skeletonize\|paths \|worst label\|pop/push/fire\|labels with bails\|pushes total" $OUT/prof.txt
for cfg in "c3_bigw 1" "c3_smallw 0"; do
  set -- $cfg
  echo "== $1"; KH_BIG_LDS_HEAP=$2 KIMI_BENCH_INFLIGHT=1 timeout 600 python bench.py --workload c3 --steps 2 --warmup 0 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); print("$1", "single_ms", d["single_volume_ms"], "ms/step", d["ms_per_step"]); print("   ", d["chains"]["sum_Mcyc"]); [print("   ", c) for c in d["chains"]["longest"]]; print("   roofline", d["roofline"]["frac"], d["roofline"]["launches"])
except Exception as e: print("$1 failed", e); print(open("$OUT/$1.err").read()[-2500:])
PY
done
echo "== c3 parity"; timeout 1500 python -m pytest tests/test_gpu_c3.py -x -q -m gpu > $OUT/t3.txt 2>&1; tail -3 $OUT/t3.txt
