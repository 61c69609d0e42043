#!/bin/bash
# Round-4 GPU call C: what stretches a volume when four are in flight?  chain cycles under load, kernel durations under load
# (kernel trace), SQ / instruction-cache counters under load.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04c
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace under load"
KH_HEAP_PRIO=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --steps 8 --warmup 0 --no-cpu-baseline > $OUT/bench_kt.json 2> $OUT/bench_kt.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_kt.json")); print("ms/step", d["ms_per_step"], "single", d["single_volume_ms"]); print("loaded", json.dumps(d.get("chains_under_load"))); print("solo", json.dumps(d["chains"]["longest"][:2]), d["chains"]["sum_Mcyc"])
except Exception as e: print("failed", e); print(open("$OUT/bench_kt.err").read()[-1500:])
PY
head -8 $OUT/kt/*/kt_kernel_stats.csv 2>/dev/null || find $OUT/kt -name "*stats*"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE"; do
  tag=$(echo $SET | cut -d' ' -f1)
  echo "== pmc $SET"
  KH_HEAP_PRIO=1 timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- python $REPO/bench.py --steps 8 --warmup 0 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob("$OUT/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, v in acc.items():
    if "trace_paths" in k or "edf" in k: print(k, {a: "%.3g" % b for a, b in v.items()})
PY
done
