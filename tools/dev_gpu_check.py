"""Developer timing probe (not part of the product or the tests)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from kimimaro_amd.engine import Engine
import kimimaro_amd
from shapes import voronoi_labels

eng = Engine()
eng.profile = bool(int(os.environ.get("KH_PROFILE", "0")))
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    shape, nl, pts, seed, an = (512, 512, 100), 333, 12, 2, (16, 16, 40)
else:
    shape, nl, pts, seed, an = (512, 512, 512), 2124, 16, 3, (16, 16, 40)
t = time.time()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
lab, an = bench.make_volume(which)
shape = lab.shape
print("gen", time.time() - t, flush=True)
d = eng.to_device(lab)
n = lab.size
out = eng.empty(n, torch.float32); ws = eng.empty(2 * n, torch.float32)
for lb, dd in ((4, d),):
    for _ in range(2):
        eng.edt(dd, lb, shape, an, False, out, ws)
    eng.sync()
    t = time.time()
    for _ in range(5):
        eng.edt(dd, lb, shape, an, False, out, ws)
    eng.sync()
    dt = (time.time() - t) / 5
    print("EDT L=%d: %.3f ms  -> %.1f GB/s algorithmic (%d B/vox)" % (lb, dt * 1e3, n * (3 * lb + 20) / dt / 1e9, 3 * lb + 20), flush=True)
timings = []
t0 = time.perf_counter()
sk = kimimaro_amd.skeletonize(lab, anisotropy=an, dust_threshold=1000, fix_borders=False, progress=False, _engine=eng, _timings=timings)
eng.sync()
t1 = time.perf_counter()
print("skeletonize: %d skeletons in %.3f s -> %.1f labels/s" % (len(sk), t1 - t0, len(sk) / (t1 - t0)))
prev = timings[0][1]
for name, ts in timings[1:]:
    print("  %-14s %.3f s" % (name, ts - prev)); prev = ts
print("verts", sum(s.vertices.shape[0] for s in sk.values()), "arena MB", getattr(eng, "last_arena_bytes", 0) / 1e6)
import kimimaro_amd.engine as E
tk = eng.last_tasks
if tk is not None:
    print("kcyc target/rail/inval sums:", tk["cyc_target"].sum(), tk["cyc_rail"].sum(), tk["cyc_inval"].sum())
    i = np.argmax(tk["cyc_inval"].astype(np.int64) + tk["cyc_rail"])
    print("worst label: count", tk["count"][i], "paths", tk["n_paths"][i], "kcyc", tk["cyc_target"][i], tk["cyc_rail"][i], tk["cyc_inval"][i], "pushes", tk["stat_heap_pushes"][i], "settled", tk["stat_settled"][i])
    print("kcyc pop/push/fire sums:", tk["cyc_pop"].astype(np.int64).sum(), tk["cyc_push"].astype(np.int64).sum(), tk["cyc_fire"].astype(np.int64).sum())
    print("worst label kcyc pop/push/fire:", tk["cyc_pop"][i], tk["cyc_push"][i], tk["cyc_fire"][i])
    print("sweep: calls", tk["stat_sweep_calls"].sum(), "bails", tk["stat_sweep_bails"].sum(), "levels", tk["stat_sweep_levels"].astype(np.int64).sum(),
          "events", tk["stat_sweep_events"].astype(np.int64).sum(), "why", np.bitwise_or.reduce(tk["stat_sweep_why"]))
    print("worst label sweep: calls", tk["stat_sweep_calls"][i], "bails", tk["stat_sweep_bails"][i], "levels", tk["stat_sweep_levels"][i], "events", tk["stat_sweep_events"][i])
    b = np.flatnonzero(tk["stat_sweep_bails"])
    for why in (1, 2, 4, 8, 16, 32):
        bb = np.flatnonzero(tk["stat_sweep_why"] & why)
        print("  why", why, "labels", bb.size, "counts", tk["count"][bb][:12], "nlev", tk["nlev"][bb][:12], "shift", tk["ev_shift"][bb][:12], "maxnev", tk["cyc_pop"][bb][:12], "blocks used", tk["cyc_push"][bb][:12], "of", tk["ev_chunks"][bb][:12], "bails", tk["stat_sweep_bails"][bb][:12], "calls", tk["stat_sweep_calls"][bb][:12])
    nb_ = np.flatnonzero(tk["stat_sweep_bails"] == 0)
    tot_ = tk["cyc_target"].astype(np.int64) + tk["cyc_rail"] + tk["cyc_inval"]
    if nb_.size:
        j = nb_[np.argmax(tot_[nb_])]
        print("slowest label WITHOUT bails: count", tk["count"][j], "paths", tk["n_paths"][j], "kcyc target/rail/inval", tk["cyc_target"][j], tk["cyc_rail"][j], tk["cyc_inval"][j],
              "levels", tk["stat_sweep_levels"][j], "events", tk["stat_sweep_events"][j], "-> ms at 2.4 GHz: %.1f" % (tot_[j] * 1024 / 2.4e6))
        print("sum over labels without bails: kcyc target/rail/inval", tk["cyc_target"][nb_].astype(np.int64).sum(), tk["cyc_rail"][nb_].astype(np.int64).sum(), tk["cyc_inval"][nb_].astype(np.int64).sum())
    print("labels with bails:", b.size, "their voxels", tk["count"][b].sum(), "of", tk["count"].sum(), "largest such", tk["count"][b].max() if b.size else 0)
    print("pushes total", tk["stat_heap_pushes"].astype(np.int64).sum(), "settled total", tk["stat_settled"].astype(np.int64).sum(), "paths", tk["n_paths"].sum())
if tk is not None:
    small = tk["count"] < 32768
    for nm, sel in (("small (<32768 vox)", small), ("big", ~small)):
        alloc = tk["ev_chunks"][sel].astype(np.int64) * (8 << tk["ev_shift"][sel].astype(np.int64))
        used = tk["cyc_push"][sel].astype(np.int64) * (8 << tk["ev_shift"][sel].astype(np.int64))
        print("arena %s: labels %d, allocated %.2f GB, used (max over calls) %.2f GB, voxels %d, sum nlev %d, max used/alloc %.2f" % (
            nm, int(sel.sum()), alloc.sum() / 1e9, used.sum() / 1e9, int(tk["count"][sel].sum()), int(tk["nlev"][sel].sum()),
            float((used / np.maximum(alloc, 1)).max()) if sel.any() else 0))
