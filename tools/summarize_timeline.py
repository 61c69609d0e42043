"""rocprofv3 --kernel-trace CSV of a bench call -> when which kernels ran: per kernel family the first start, the last end and the
busy time, and for the LAST round of path kernels (the timed steps) each launch's start and end relative to the round's first kernel."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        fam = ("paths" if "trace_paths" in name else "edf" if "edf_batch" in name else "ccl" if "ccl_" in name else
               "edt" if "edt_" in name else "pdrf" if "pdrf" in name else "prep" if "kh::" in name else "other")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
rows.sort()
t0 = rows[0][0]
paths = [r for r in rows if r[2] == "paths"]
# the last round = the path kernels that start after the longest gap between consecutive path-kernel starts... simpler: the last N
# launches where N = launches of one round (all launches within 12 s of the last one's start)
last = paths[-1][0]
rnd = [r for r in paths if last - r[0] < 12e9]
r0 = min(r[0] for r in rows if r[0] >= rnd[0][0] - 4e9 and r[2] in ("ccl", "edt", "prep", "edf"))
print("round: %d path-kernel launches; round starts %.3f s after the first kernel of the run" % (len(rnd), (r0 - t0) / 1e9))
for fam in ("ccl", "edt", "prep", "edf", "pdrf", "paths"):
    sel = [r for r in rows if r[2] == fam and r[0] >= r0]
    if not sel:
        continue
    print("  %-6s n=%4d first start %6.3f s  last end %6.3f s  sum of durations %7.3f s" % (
        fam, len(sel), (min(r[0] for r in sel) - r0) / 1e9, (max(r[1] for r in sel) - r0) / 1e9, sum(r[1] - r[0] for r in sel) / 1e9))
print("  path kernels (start .. end, s):", " ".join("%.2f..%.2f" % ((a - r0) / 1e9, (b - r0) / 1e9) for a, b, _ in sorted(rnd)))
edf = sorted(r for r in rows if r[2] == "edf" and r[0] >= r0)
print("  edf kernels (start .. end, s):", " ".join("%.2f..%.2f" % ((a - r0) / 1e9, (b - r0) / 1e9) for a, b, _ in edf))
