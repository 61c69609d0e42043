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
lab = voronoi_labels(shape, nl, seed=seed, pts_per_label=pts, step=24.0, anisotropy=an)
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
print("verts", sum(s.vertices.shape[0] for s in sk.values()))
import kimimaro_amd.engine as E
tk = E.LAST_TASKS
if tk is not None:
    print("kcyc target/rail/inval sums:", tk["cyc_target"].sum(), tk["cyc_rail"].sum(), tk["cyc_inval"].sum())
    i = np.argmax(tk["cyc_inval"].astype(np.int64) + tk["cyc_rail"])
    print("worst label: count", tk["count"][i], "paths", tk["n_paths"][i], "kcyc", tk["cyc_target"][i], tk["cyc_rail"][i], tk["cyc_inval"][i], "pushes", tk["stat_heap_pushes"][i], "settled", tk["stat_settled"][i])
    print("kcyc pop/push/fire sums:", tk["cyc_pop"].astype(np.int64).sum(), tk["cyc_push"].astype(np.int64).sum(), tk["cyc_fire"].astype(np.int64).sum())
    print("worst label kcyc pop/push/fire:", tk["cyc_pop"][i], tk["cyc_push"][i], tk["cyc_fire"][i])
    print("pushes of the 128 largest labels (large-LDS kernel):", tk["stat_heap_pushes"][:128].astype(np.int64).sum())
    print("pushes total", tk["stat_heap_pushes"].astype(np.int64).sum(), "settled total", tk["stat_settled"].astype(np.int64).sum(), "paths", tk["n_paths"].sum())
