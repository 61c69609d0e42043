"""Developer probe: the 60 reference vectors of roll_invalidation_ball_inside_component through the heap emulation alone."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kimimaro_amd import ops
from kimimaro_amd.engine import Engine
eng = Engine(); eng.sweep = os.environ.get("SWEEP", "0") == "1"
ops._engine = eng
z = np.load(os.path.join(ROOT, "tests", "golden", "invalidation_ball.npz"))
unpack = lambda b, shape: np.unpackbits(b)[: int(np.prod(shape))].reshape(shape, order="F").astype(np.uint8)
bad = 0
for i in range(int(z["n"])):
    shape = tuple(int(v) for v in z["shape_%d" % i])
    m = np.asfortranarray(unpack(z["mask_%d" % i], shape))
    path = z["path_%d" % i]
    dbf = np.zeros(shape, np.float32, order="F")
    dbf[path[:, 0], path[:, 1], path[:, 2]] = z["dbfpath_%d" % i]
    scale, const = z["sc_%d" % i]
    try:
        cnt, out = ops.roll_invalidation_ball_inside_component(m, dbf, scale, const, z["an_%d" % i], path)
    except Exception as e:
        print(i, "EXC", repr(e)[:200]); bad += 1; continue
    ok = cnt == int(z["count_%d" % i]) and np.array_equal(out, unpack(z["after_%d" % i], shape))
    bad += not ok
    print(i, "ok" if ok else "BAD", "count", cnt, "want", int(z["count_%d" % i]), "sources", len(path), "voxels", int(np.count_nonzero(unpack(z["mask_%d" % i], shape))), flush=True)
print("bad", bad)
