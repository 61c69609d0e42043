"""SWCYC lines of a -DKH_SWEEP_PROBE build (csrc/trace.hip) -> cycles per level and per deadline event, by label."""
import collections
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(int))
for line in open(sys.argv[1]):
    if not line.startswith("SWCYC"):
        continue
    kv = dict(re.findall(r"(\w+)=(\d+)", line))
    if "nf" not in kv or "d_casc" not in kv:      # (a line cut short by another writer)
        continue
    b = (kv["blk"], kv["nf"])
    for k, v in kv.items():
        if k not in ("blk", "nf"):
            acc[b][k] += int(v)
    acc[b]["calls"] += 1
for b, v in sorted(acc.items(), key=lambda x: int(x[0][0])):
    lv = max(v["lev"], 1)
    dn = max(v["dn"], 1)
    print("block", b, "calls", v["calls"], "certified", v["ok"], "levels", v["lev"], "events", v["ev"],
          "cycles/level:", {k: round(v[k] / lv) for k in ("commit", "next", "A", "cascA", "B", "cascB", "pairs")})
    print("      thread 0's deadline events", v["dn"], "cycles/event:",
          {k: round(v[k] / dn) for k in ("d_own", "d_alive", "d_rank", "d_sched", "d_casc")})
