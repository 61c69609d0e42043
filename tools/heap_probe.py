"""Developer probe: the heap emulation alone (sweep switched off), PROF counters of the pop pipeline.
usage: python tools/heap_probe.py [c2|c3|mini]   (prints ticks / pops / pushes / cycles of the worst label and in total)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import kimimaro_amd  # noqa: E402
import kimimaro_amd.engine as E  # noqa: E402
from kimimaro_amd.engine import Engine  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
lab, an = bench.make_volume(which)
for sweep in ((False, True) if "--both" in sys.argv else (False,)):
    eng = Engine()
    eng.sweep = sweep
    eng.profile = True
    for rep in range(2):
        t0 = time.perf_counter()
        sk = kimimaro_amd.skeletonize(lab, anisotropy=an, dust_threshold=1000, fix_borders=True, progress=False, _engine=eng)
        eng.sync()
        dt = time.perf_counter() - t0
    tk = E.LAST_TASKS
    tot = tk["cyc_inval"].astype(np.int64)
    i = int(np.argmax(tot))
    ticks, pops, stalls = (tk[k].astype(np.int64) * 1024 for k in ("cyc_pop", "cyc_push", "cyc_fire"))
    pushes = tk["stat_heap_pushes"].astype(np.int64)
    print("sweep=%s  %s: %d skeletons in %.3f s" % (sweep, which, len(sk), dt))
    print("  worst label: %d voxels, inval %.3f Gcyc, pushes %d, ticks %d, pops %d, push-stall iterations %d" % (
        tk["count"][i], tot[i] * 1024 / 1e9, pushes[i], ticks[i], pops[i], stalls[i]))
    if pops[i]:
        print("  worst label: %.2f ticks / pop, %.0f cycles / pop+push pair, %.0f cycles / tick (upper bound: all of inval)" % (
            ticks[i] / pops[i], tot[i] * 1024 / max(pushes[i], 1), tot[i] * 1024 / max(ticks[i], 1)))
    srv = tk["cyc_fire"].astype(np.int64) * 1024
    j = int(np.argmax(srv))
    print("  rounds %d; heap-server cycles: total %.1f G, worst label %.3f G (%d voxels, %d pushes -> %.0f cycles / pair)" % (
        eng.last_rounds, srv.sum() / 1e9, srv[j] / 1e9, tk["count"][j], pushes[j], srv[j] / max(pushes[j], 1)))
    print("  all labels: pushes %d, ticks %d, pops %d, inval %.1f Gcyc, heap calls via sweep bails %d" % (
        pushes.sum(), ticks.sum(), pops.sum(), tot.sum() * 1024 / 1e9, int(tk["stat_sweep_bails"].sum())))
