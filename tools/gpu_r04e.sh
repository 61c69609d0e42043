#!/bin/bash
# Round-4 GPU call E: chunk recycling + smaller heaps (memory per lane), default lane counts, process lanes at width.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04e
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
show() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.load(open(f))
    print(f.split("/")[-1], "ms/step", d["ms_per_step"], "single", d["single_volume_ms"], "value", d["value"], "hbm", d["hbm_reserved_peak_gb"],
          "lanes", d.get("lanes"), d["volumes_in_flight"], "fallbacks", d["sweep"]["fell_back_to_heap"], "levels", d["sweep"]["levels"])
    c = d["chains"]; print("   why", c["bail_reasons_or"], "arena", c["labels_bailing_for_arena"], "lists/levels", c["labels_bailing_for_lists_or_levels"], "retraced", c["labels_retraced_for_scratch"])
    print("   phases", d["phases_s"])
except Exception as e:
    print(f, "failed", e)
    try: print(open(f.replace(".json", ".err")).read()[-1500:])
    except Exception: pass
PY
}
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_cube.py tests/test_gpu_budget.py tests/test_gpu_lanes.py -x -q -m gpu > $OUT/t_trace.txt 2>&1; rc=$?; tail -3 $OUT/t_trace.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -40 $OUT/t_trace.txt; exit 1; }
run() { name=$1; steps=$2; warm=$3; shift; shift; shift; echo "== $name"; env "$@" timeout 900 python bench.py --steps $steps --warmup $warm --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; show $OUT/$name.json; }
run default8 8 1 KH_HEAP_PRIO=1
run driver20 20 5 KH_HEAP_PRIO=1
run thr12_s36 36 2 KH_HEAP_PRIO=1 KIMI_BENCH_INFLIGHT=12
run proc12_s36 36 2 KH_HEAP_PRIO=1 KIMI_BENCH_INFLIGHT=12 KIMI_BENCH_LANES=process
run noprio20 20 5 KH_HEAP_PRIO=0
echo "== kat + c3 parity"; timeout 1200 python -m pytest tests/test_gpu_kat.py tests/test_gpu_c3.py -x -q -m gpu > $OUT/t_c3.txt 2>&1; tail -3 $OUT/t_c3.txt
