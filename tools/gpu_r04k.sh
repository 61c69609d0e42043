#!/bin/bash
# kernel timeline of the driver's call (10 lanes x 64 threads)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04k
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --steps 20 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import csv, glob, collections, json
try:
    d = json.load(open("$OUT/bench.json")); print("ms/step", d["ms_per_step"], d["volumes_in_flight"])
except Exception as e: print("bench failed", e)
p = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
def short(n): return n.split("(")[0].replace("void ", "").replace("kh::", "")[:22]
big = [r for r in rows if any(k in r["Kernel_Name"] for k in ("trace_paths", "edf_batch", "ccl_link"))]
big.sort(key=lambda r: int(r["Start_Timestamp"]))
with open("$OUT/timeline.txt", "w") as f:
    for r in big:
        f.write("%-22s q=%-3s start %8.3f dur %7.3f\n" % (short(r["Kernel_Name"]), r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e9, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9))
print(open("$OUT/timeline.txt").read()[-6000:])
PY
