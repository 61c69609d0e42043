#!/bin/bash
# Round-4 GPU call A: gate (sweep vectors + small volumes), c3 parity, bench A/B of the pending-deadline filter,
# counter passes of the path kernel, kernel stats, then the whole GPU suite.   gpurun -- bash tools/gpu_r04a.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04a
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py -x -q -m gpu > $OUT/t_trace.txt 2>&1; rc=$?; tail -3 $OUT/t_trace.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -40 $OUT/t_trace.txt; exit 1; }
echo "== c3 parity"; timeout 900 python -m pytest tests/test_gpu_c3.py -x -q -m gpu > $OUT/t_c3.txt 2>&1; rc=$?; tail -3 $OUT/t_c3.txt
[ $rc -ne 0 ] && { echo "C3 FAILED"; tail -40 $OUT/t_c3.txt; }
echo "== bench filter on"; timeout 600 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/bench_filter.json 2> $OUT/bench_filter.err; python - <<PY
import json
for f in ("bench_filter",):
    try:
        d = json.load(open("$OUT/%s.json" % f)); print(f, d["ms_per_step"], d["single_volume_ms"], d["hbm_reserved_peak_gb"], d["sweep"], d["phases_s"])
    except Exception as e: print(f, "failed", e)
PY
echo "== bench filter off"; KH_SWEEP_FILTER=0 timeout 600 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/bench_nofilter.json 2> $OUT/bench_nofilter.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_nofilter.json")); print("nofilter", d["ms_per_step"], d["single_volume_ms"], d["sweep"], d["phases_s"])
except Exception as e: print("nofilter failed", e)
PY
echo "== 64-thread workgroups"; KH_TRACE_THREADS=64 KH_SWEEP_LDS_LEVELS=2048 timeout 600 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/bench_t64.json 2> $OUT/bench_t64.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_t64.json")); print("t64", d["ms_per_step"], d["single_volume_ms"], d["phases_s"])
except Exception as e: print("t64 failed", e)
PY
echo "== pmc"; bash tools/pmc_trace_r3.sh r04a 2>&1 | tail -12
echo "== kernel stats"; bash tools/profile_round.sh r04a nopmc 2>&1 | tail -5
echo "== full suite"; timeout 1800 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1; tail -15 $OUT/tests.txt
