#!/bin/bash
# Round-4 GPU call B: gate, then lane experiments (threads vs processes, heap-wave priority, lane counts), EDF pre-check.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04b
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
show() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.load(open(f))
    print(f.split("/")[-1], "ms/step", d["ms_per_step"], "single", d["single_volume_ms"], "value", d["value"], "hbm", d["hbm_reserved_peak_gb"],
          "lanes", d.get("lanes"), d["volumes_in_flight"], "fallbacks", d["sweep"]["fell_back_to_heap"], "events", d["sweep"]["events"])
    print("   phases", d["phases_s"])
    print("   chains", json.dumps(d.get("chains"))[:1500])
except Exception as e:
    print(f, "failed", e)
    try: print(open(f.replace(".json", ".err")).read()[-1500:])
    except Exception: pass
PY
}
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_lanes.py -x -q -m gpu > $OUT/t_trace.txt 2>&1; rc=$?; tail -3 $OUT/t_trace.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -40 $OUT/t_trace.txt; exit 1; }
run() { name=$1; shift; echo "== $name"; env "$@" timeout 900 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; show $OUT/$name.json; }
run thr4 KIMI_BENCH_LANES=thread
run thr4_prio KIMI_BENCH_LANES=thread KH_HEAP_PRIO=1
run proc4 KIMI_BENCH_LANES=process
run proc4_prio KIMI_BENCH_LANES=process KH_HEAP_PRIO=1
run proc6_prio KIMI_BENCH_LANES=process KH_HEAP_PRIO=1 KIMI_BENCH_INFLIGHT=6
run proc8_prio KIMI_BENCH_LANES=process KH_HEAP_PRIO=1 KIMI_BENCH_INFLIGHT=8
echo "== c3 parity"; timeout 900 python -m pytest tests/test_gpu_c3.py -x -q -m gpu > $OUT/t_c3.txt 2>&1; tail -3 $OUT/t_c3.txt
