#!/bin/bash
# phase cycles of the sweep's level loop (KH_SWEEP_PROBE build) for four labels of c3, one wave and four waves per label
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04h
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
for T in 64 256; do
  KIMI_HIP_LIB=$REPO/kimimaro_amd/libkimi_hip_probe.so KH_TRACE_THREADS=$T timeout 300 python tools/trace_only.py c3 > $OUT/probe_t$T.txt 2>&1
  grep -c SWCYC $OUT/probe_t$T.txt; grep TRACEONLY $OUT/probe_t$T.txt
done
python - <<PY
import re, collections
for T in (64, 256):
    acc = collections.defaultdict(lambda: collections.defaultdict(int))
    for line in open("$OUT/probe_t%d.txt" % T):
        if not line.startswith("SWCYC"): continue
        kv = dict(re.findall(r"(\w+)=(\d+)", line))
        b = (kv["blk"], kv["nf"])
        for k, v in kv.items():
            if k not in ("blk", "nf"): acc[b][k] += int(v)
        acc[b]["calls"] += 1
    for b, v in sorted(acc.items(), key=lambda x: int(x[0][0])):
        lv = max(v["lev"], 1)
        print("threads", T, "block", b, "calls", v["calls"], "certified", v["ok"], "levels", v["lev"], "events", v["ev"],
              "cycles/level:", {k: round(v[k] / lv) for k in ("commit", "next", "A", "cascA", "B", "cascB", "pairs")})
PY
