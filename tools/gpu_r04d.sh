#!/bin/bash
# Round-4 GPU call D: level window + threads per label x lanes (staggered runs).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04d
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
show() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.load(open(f))
    print(f.split("/")[-1], "ms/step", d["ms_per_step"], "single", d["single_volume_ms"], "value", d["value"], "hbm", d["hbm_reserved_peak_gb"],
          "lanes", d.get("lanes"), d["volumes_in_flight"], "fallbacks", d["sweep"]["fell_back_to_heap"], "levels", d["sweep"]["levels"], "why", d["chains"]["bail_reasons_or"])
    print("   phases", d["phases_s"])
    print("   solo sums", d["chains"]["sum_Mcyc"], "loaded", [(c["longest_Mcyc"], c["sum_Mcyc_inval"], c["sum_Mcyc_rail"]) for c in (d.get("chains_under_load") or [])][:3])
except Exception as e:
    print(f, "failed", e)
    try: print(open(f.replace(".json", ".err")).read()[-1500:])
    except Exception: pass
PY
}
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_cube.py -x -q -m gpu > $OUT/t_trace.txt 2>&1; rc=$?; tail -3 $OUT/t_trace.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -40 $OUT/t_trace.txt; exit 1; }
echo "== gate t64"; KH_TRACE_THREADS=64 timeout 900 python -m pytest tests/test_gpu_trace.py -x -q -m gpu > $OUT/t_trace64.txt 2>&1; rc=$?; tail -3 $OUT/t_trace64.txt
[ $rc -ne 0 ] && { echo "GATE64 FAILED"; tail -40 $OUT/t_trace64.txt; }
run() { name=$1; steps=$2; shift; shift; echo "== $name"; env "$@" timeout 900 python bench.py --steps $steps --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; show $OUT/$name.json; }
run thr4_t256 16 KH_HEAP_PRIO=1
run thr4_t64 16 KH_HEAP_PRIO=1 KH_TRACE_THREADS=64
run thr8_t64 32 KH_HEAP_PRIO=1 KH_TRACE_THREADS=64 KIMI_BENCH_INFLIGHT=8
run thr8_t256 32 KH_HEAP_PRIO=1 KIMI_BENCH_INFLIGHT=8
run thr6_t128 24 KH_HEAP_PRIO=1 KH_TRACE_THREADS=128 KIMI_BENCH_INFLIGHT=6
echo "== kat + c3 parity (t64)"; KH_TRACE_THREADS=64 timeout 1200 python -m pytest tests/test_gpu_kat.py tests/test_gpu_c3.py -x -q -m gpu > $OUT/t_c3.txt 2>&1; tail -3 $OUT/t_c3.txt
