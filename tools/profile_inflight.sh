#!/bin/bash
# Kernel trace of the default bench configuration (4 volumes in flight): how long does one volume's path kernel run when
# the other lanes load the GPU?   Usage (under gpurun): bash tools/profile_inflight.sh <tag>   -> gpurun_out/prof_<tag>_inflight/
set -u
TAG=${1:-r02b}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_inflight
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- \
  python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python - <<PY
import csv
rows = []
with open("$OUT/ktrace/kt_kernel_trace.csv") as f:
    for r in csv.DictReader(f):
        if "trace_paths_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), int(r["Grid_Size_X"])))
rows.sort()
t0 = rows[0][0]
with open("$OUT/trace_kernel_timeline.txt", "w") as o:
    o.write("# trace_paths_kernel launches of bench.py --steps 4 --warmup 1 (4 volumes in flight): start s | end s | duration s | stream/queue | grid\n")
    for s, e, q, g in rows:
        o.write("%.3f | %.3f | %.3f | %s | %d\n" % ((s - t0) / 1e9, (e - t0) / 1e9, (e - s) / 1e9, q, g))
print(open("$OUT/trace_kernel_timeline.txt").read())
PY
