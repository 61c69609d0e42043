"""Developer probe: the reference's KATs at their own sizes, timed (not a test)."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kimimaro_amd
from kimimaro_amd.engine import Engine
import kimimaro_amd.engine as E
eng = Engine()
TP = {"scale": 1.5, "const": 300, "pdrf_scale": 100000, "pdrf_exponent": 4, "soma_acceptance_threshold": 3500,
      "soma_detection_threshold": 750, "soma_invalidation_const": 300, "soma_invalidation_scale": 2}
def stats():
    tk = eng.last_tasks
    return dict(calls=int(tk["stat_sweep_calls"].sum()), bails=int(tk["stat_sweep_bails"].sum()), why=int(np.bitwise_or.reduce(tk["stat_sweep_why"])), levels=int(tk["stat_sweep_levels"].sum()), nlev=tk["nlev"].tolist()[:4], pushes=int(tk["stat_heap_pushes"].sum()))
which = sys.argv[1:] or ["square_main"]
for w in which:
    if w.startswith("square"):
        n = 1000
        labels = np.ones((n, n), dtype=np.uint8)
        if w == "square_anti": labels[-1, 0] = 0; labels[0, -1] = 0
        else: labels[0, 0] = 0; labels[-1, -1] = 0
        t = time.time(); skels = kimimaro_amd.skeletonize(labels, teasar_params=TP, fix_borders=False, _engine=eng); eng.sync()
        print(w, "%.2f s" % (time.time() - t), skels[1].vertices.shape, skels[1].edges.shape, abs(skels[1].cable_length() - 999 * np.sqrt(2)), stats(), flush=True)
    elif w == "cube":
        n = 128
        labels = np.ones((n, n, n), dtype=np.uint8); labels[0, 0, 0] = 0; labels[-1, -1, -1] = 0
        t = time.time(); skels = kimimaro_amd.skeletonize(labels, fix_borders=False, _engine=eng); eng.sync()
        print("cube 128: %.2f s" % (time.time() - t), skels[1].vertices.shape, skels[1].edges.shape, abs(skels[1].cable_length() - 127 * np.sqrt(3)), stats(), flush=True)
    else:
        labels = np.zeros((256, 256, 256), dtype=np.uint8); labels[64:196, 64:196, :] = 128
        kw = dict(teasar_params={"const": 250, "scale": 10, "pdrf_exponent": 4, "pdrf_scale": 100000}, anisotropy=(40, 32, 20), dust_threshold=1000, fix_branching=True, fix_borders=True)
        t = time.time(); skels = kimimaro_amd.skeletonize(labels, _engine=eng, **kw); eng.sync()
        v = skels[128].voxel_space().vertices
        print("fix_borders_z 256: %.2f s" % (time.time() - t), v.shape, v[:3].tolist(), bool(np.all(v[:, 2] == np.arange(256))), stats(), flush=True)
