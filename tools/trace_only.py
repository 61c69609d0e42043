"""Developer / profiling probe: ONE skeletonize of the bench volume (default c3) on one engine -- the command the
rocprofv3 counter passes of the path kernel wrap (tools/pmc_trace_r3.sh)."""
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

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
lab, an = bench.make_volume(which)
eng = Engine()
t0 = time.perf_counter()
sk = kimimaro_amd.skeletonize(lab, anisotropy=an, dust_threshold=1000, fix_borders=True, progress=False, _engine=eng)
eng.sync()
tk = eng.last_tasks
nf = int(tk["count"].astype(np.int64).sum())
settled = int(tk["stat_settled"].astype(np.int64).sum())
print("TRACEONLY %s: %d skeletons, %.3f s, Nf %d, settled %d, algorithmic bytes of the path kernel (SURVEY 8d) %d" % (
    which, len(sk), time.perf_counter() - t0, nf, settled, (4 + 9) * nf + 10 * nf + 12 * nf + 12 * nf + 12 * settled + 2 * nf))
