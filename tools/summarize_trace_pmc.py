"""gpurun_out/prof_<tag>/pmc_trace_*/ (tools/pmc_trace_r3.sh) -> profiles/<prefix>_c3_trace_pmc.json: per-launch counters of
trace_paths_kernel on the bench volume c3 (one volume at a time).

  python tools/summarize_trace_pmc.py <tag> [prefix]"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else tag
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(src, "pmc_trace_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "trace_paths_kernel" not in name and "edf_batch_kernel" not in name and "ccl_link" not in name:
                continue
            k = re.sub(r"\(.*", "", name).replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
algo = None
for log in sorted(glob.glob(os.path.join(src, "pmc_trace_*.log"))):
    m = re.search(r"algorithmic bytes of the path kernel \(SURVEY 8d\) (\d+)", open(log).read())
    if m:
        algo = int(m.group(1))
out = {"source": "rocprofv3 --pmc <one set per run> --kernel-trace -- python tools/trace_only.py c3 (tools/pmc_trace_r3.sh); " + pre,
       "units": "per launch (mean over the launches of the run); FETCH_SIZE / WRITE_SIZE are reported in KiB and converted to bytes",
       "correction": "FETCH_SIZE = memory-side read requests x 64 B; on gfx950 a full 128-B line request is tallied as 64 B (MI355X_MICROARCH.md, "
                     "HBM section; calibrated in round 1 on the streaming edt_x_kernel): x2 = fetch_bytes_corrected, exact if every miss of these "
                     "scattered narrow accesses fills a whole 128-B line (uncalibrated for this pattern: the raw figure is given too).  WRITE_SIZE "
                     "as reported.  Infinity-Cache hits are counted (these are the L2's fabric-side requests).  *_per_volume = mean per launch x "
                     "launches of the run (one c3 volume; the largest 128 labels run as a launch of their own on a second stream).",
       "algorithmic_bytes_path_kernel": algo, "kernels": {}}
for k, ctr in acc.items():
    e = {c: sum(v) / len(v) for c, v in ctr.items()}
    e["launches_seen"] = max(len(v) for v in ctr.values())
    nl = {c: len(v) for c, v in ctr.items()}
    if "FETCH_SIZE" in e:
        e["fetch_bytes_raw"] = e["FETCH_SIZE"] * 1024.0
        e["fetch_bytes_corrected"] = e["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"] * 1024.0
    if "fetch_bytes_corrected" in e and "write_bytes" in e:
        e["hbm_bytes_corrected"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        e["hbm_bytes_corrected_per_volume"] = e["fetch_bytes_corrected"] * nl["FETCH_SIZE"] + e["write_bytes"] * nl["WRITE_SIZE"]
        e["hbm_bytes_raw_per_volume"] = e["fetch_bytes_raw"] * nl["FETCH_SIZE"] + e["write_bytes"] * nl["WRITE_SIZE"]
        if algo and "trace_paths" in k:
            e["traffic_over_algorithmic"] = e["hbm_bytes_corrected_per_volume"] / algo
            e["traffic_over_algorithmic_raw"] = e["hbm_bytes_raw_per_volume"] / algo
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
        e["mean_waves_resident_per_busy_cycle"] = e["SQ_WAVE_CYCLES"] / e["SQ_BUSY_CYCLES"]
    if dur.get(k):
        e["ms_per_launch_under_pmc"] = sum(dur[k]) / len(dur[k])
    out["kernels"][k] = e
dst = os.path.join(ROOT, "profiles", pre + "_c3_trace_pmc.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
for k, e in out["kernels"].items():
    print(k[:60], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in (
        "hbm_bytes_corrected", "traffic_over_algorithmic", "l2_hit_rate", "ms_per_launch_under_pmc", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU")})
