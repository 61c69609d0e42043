"""Turns gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small committed files under profiles/.

  python tools/summarize_profiles.py <tag> <round-prefix>      e.g.  r01b r01b
"""
import collections
import csv
import json
import os
import shutil
import sys

tag, pre = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
V = 512 ** 3
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2   # bytes per component id in the EDT launches (u16 since round 2)

shutil.copy(os.path.join(src, "ktrace", "kt_kernel_stats.csv"), os.path.join(dst, pre + "_c3_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, pre + "_c3_bench_under_rocprof.json"))

# per (kernel, grid) launch statistics of the EDT kernels: full-volume launches vs the 2-D border planes
acc = collections.defaultdict(list)
with open(os.path.join(src, "ktrace", "kt_kernel_trace.csv")) as f:
    for r in csv.DictReader(f):
        if "edt_" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0]
            acc[(name, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(dst, pre + "_c3_edt_dispatches.txt"), "w") as f:
    f.write("# EDT kernel dispatches of `rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 0 "
            "--no-cpu-baseline` (%s, workload c3).\n# Full-volume launches (512^3) vs the 2-D border-plane launches of "
            "fix_borders that share the kernels.\n# kernel | grid size | launches | avg us | min us | max us\n" % pre)
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s | %d | %d | %.1f | %.1f | %.1f\n" % (name, grid, len(v), sum(v) / len(v), min(v), max(v)))

# PMC: bytes per launch of the full-volume EDT kernels
def pmc(counter):
    a = collections.defaultdict(list)
    rows = []
    with open(os.path.join(src, "pmc_" + counter, "pmc_counter_collection.csv")) as f:
        rd = csv.DictReader(f)
        for r in rd:
            if "edt_" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 2 ** 20:
                a[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
                rows.append(r)
    with open(os.path.join(dst, "%s_c3_edt_pmc_%s.csv" % (pre, counter)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rd.fieldnames)
        w.writeheader()
        w.writerows(rows)
    return {k: sum(v) / len(v) for k, v in a.items()}

fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python tools/edt_only.py c3 ; " + pre,
       "units": "bytes per launch; FETCH_SIZE / WRITE_SIZE are reported in KiB",
       "correction": "FETCH_SIZE on gfx950 counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section): x2.  Calibrated in this "
                     "access pattern on edt_x_kernel, which reads the 512^3 label volume exactly once (L * 134,217,728 B).  WRITE_SIZE "
                     "needs no correction (every pass writes 536,870,912 B).",
       "kernels": {}}
for k in sorted(fetch):
    alg = (L + 4) * V if "edt_x" in k else (L + 8) * V
    hbm = 2.0 * fetch[k] + write[k]
    out["kernels"][k] = {"FETCH_SIZE_raw_bytes": fetch[k], "WRITE_SIZE_bytes": write[k], "hbm_bytes_corrected": hbm,
                         "algorithmic_bytes": alg, "traffic_over_algorithmic": round(hbm / alg, 4)}
json.dump(out, open(os.path.join(dst, pre + "_c3_edt_pmc.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
print(open(os.path.join(dst, pre + "_c3_edt_dispatches.txt")).read())
