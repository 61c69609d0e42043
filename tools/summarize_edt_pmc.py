"""gpurun_out/<tag>/edtpmc_<name>/pass_*/ (tools/gpu.sh edtpmc) -> one line per EDT kernel: SQ counters per launch and per voxel.

  python tools/summarize_edt_pmc.py <dir> [nvox]"""
import collections
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
nvox = float(sys.argv[2]) if len(sys.argv) > 2 else 512.0 ** 3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(src, "pass_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "edt_" not in name:
                continue
            k = re.sub(r"\(.*", "", name).replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("End_Timestamp"):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {}
for k, ctr in acc.items():
    e = {c: sum(v) / len(v) for c, v in ctr.items()}
    if dur.get(k):
        e["ms_per_launch_under_pmc"] = sum(dur[k]) / len(dur[k])
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in e:
            e[c + "_per_wave_row"] = e[c] / (nvox / 64.0)     # wave-instructions per 64 voxels
    if "SQ_WAVE_CYCLES" in e:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
            if c in e:
                e[c + "_frac_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
    if "SQ_WAVE_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
        e["mean_waves_resident_per_busy_cycle"] = e["SQ_WAVE_CYCLES"] / e["SQ_BUSY_CYCLES"]
    out[k] = e
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
for k, e in out.items():
    print(k[:70])
    print("   ", {a: round(b, 4) for a, b in e.items() if "per_wave_row" in a or "frac" in a or a in ("ms_per_launch_under_pmc", "SQ_LDS_BANK_CONFLICT", "mean_waves_resident_per_busy_cycle")})
