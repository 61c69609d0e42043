#!/bin/bash
# Round-4 GPU call I: sweep on typed (global / LDS) pointers, loads of a phase in flight together -- gate, probe, bench.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04i
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== gate"; timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_cube.py -x -q -m gpu > $OUT/t.txt 2>&1; rc=$?; tail -3 $OUT/t.txt
[ $rc -ne 0 ] && { echo "GATE FAILED"; tail -60 $OUT/t.txt; exit 1; }
KIMI_HIP_LIB=$REPO/kimimaro_amd/libkimi_hip_probe.so KH_TRACE_THREADS=64 timeout 300 python tools/trace_only.py c3 > $OUT/probe_t64.txt 2>&1
python - <<PY
import re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(int))
for line in open("$OUT/probe_t64.txt"):
    if not line.startswith("SWCYC"): continue
    kv = dict(re.findall(r"(\w+)=(\d+)", line))
    b = (kv["blk"], kv["nf"])
    for k, v in kv.items():
        if k not in ("blk", "nf"): acc[b][k] += int(v)
for b, v in sorted(acc.items(), key=lambda x: int(x[0][0])):
    lv = max(v["lev"], 1)
    print("threads 64 block", b, "levels", v["lev"], "events", v["ev"], "cycles/level:", {k: round(v[k] / lv) for k in ("commit", "next", "A", "cascA", "B", "cascB", "pairs")})
PY
for cfg in "s20 20 5 0" "s36l12 36 2 12"; do
  set -- $cfg
  echo "== $1"; KIMI_BENCH_INFLIGHT=$4 timeout 900 python bench.py --steps $2 --warmup $3 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); print("$1", d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"], d["sweep"]["fell_back_to_heap"]); print("   ", d["chains"]["sum_Mcyc"], d["chains"]["longest"][0]["Mcyc_inval"])
except Exception as e: print("$1 failed", e); print(open("$OUT/$1.err").read()[-1500:])
PY
done
echo "== c3 + kat"; timeout 1200 python -m pytest tests/test_gpu_kat.py tests/test_gpu_c3.py -x -q -m gpu > $OUT/t2.txt 2>&1; tail -3 $OUT/t2.txt
