#!/bin/bash
# Round-5 GPU call J: round-4 heap back in (the decision-bit heap measured no faster: tests/experiments/bitheap_r5.patch), new CCL link
# pass, 256 labels in the big-LDS launch -- gate, single-volume c3 with the per-label cycle dump, the driver's bench configuration.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r05j
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== gate"; timeout 1200 python -m pytest tests/test_gpu_trace.py tests/test_gpu_cube.py tests/test_gpu_post.py tests/test_gpu_ccl.py tests/test_gpu_edt.py tests/test_gpu_kat.py tests/test_gpu_budget.py -x -q -m gpu > $OUT/t.txt 2>&1; rc=$?; tail -4 $OUT/t.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; grep -n "Error\|assert\|FAILED" $OUT/t.txt | head -40; exit 1; }
for cfg in "c3_split256 256" "c3_split128 128"; do
  set -- $cfg
  echo "== $1"; KH_SPLIT_SLOTS=$2 KIMI_BENCH_DUMP_TASKS=$OUT/$1_tasks.npz KIMI_BENCH_INFLIGHT=1 timeout 600 python bench.py --workload c3 --steps 2 --warmup 0 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); print("$1", "single_ms", d["single_volume_ms"], "ms/step", d["ms_per_step"]); print("   ", d["chains"]["sum_Mcyc"]); [print("   ", c) for c in d["chains"]["longest"][:2]]; print("   roofline", d["roofline"]["frac"], d["roofline"]["launches"]); print("   phases", d["phases_s"])
except Exception as e: print("$1 failed", e); print(open("$OUT/$1.err").read()[-2500:])
PY
done
echo "== driver configuration"; timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s20.json 2> $OUT/bench_s20.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_s20.json")); print("s20", d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"], d["sweep"])
except Exception as e: print("s20 failed", e); print(open("$OUT/bench_s20.err").read()[-2500:])
PY
