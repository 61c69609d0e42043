"""Reader of tests/golden/trace_paths.npz: inputs and path lists of the REFERENCE's own trace() (kimimaro/trace.py,
run in the build container by tests/golden/make_golden.py with the compiled reference skeletontricks and the oracle's
restatements standing in for its absent third-party imports)."""
import ast
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases():
    z = np.load(os.path.join(G, "trace_paths.npz"))
    for i in range(int(z["n"])):
        shape = tuple(int(v) for v in z["shape_%d" % i])
        mask = np.unpackbits(z["mask_%d" % i])[: int(np.prod(shape))].reshape(shape, order="F").astype(np.uint8)
        kw = dict(ast.literal_eval(str(z["kw_%d" % i])))
        extra = dict(ast.literal_eval(str(z["extra_%d" % i])))
        lens = z["lens_%d" % i]
        verts = z["verts_%d" % i].astype(np.int64)
        paths, pos = [], 0
        for n in lens.tolist():
            paths.append(verts[pos:pos + n])
            pos += n
        yield i, np.asfortranarray(mask), tuple(float(a) for a in z["an_%d" % i]), kw, extra, paths


def tie_cases():
    """tests/golden/trace_paths_ties.npz: the runs of the reference's trace() that the oracle does NOT reproduce, because
    numpy's unstable argsort in CachedTargetFinder (skeletontricks.pyx:1001-1006) broke a DAF tie the other way (SURVEY
    0-7a).  Yields (i, mask, anisotropy, kw, extra, reference paths, index of the first path that differs)."""
    z = np.load(os.path.join(G, "trace_paths_ties.npz"))
    for i in range(int(z["n"])):
        shape = tuple(int(v) for v in z["shape_%d" % i])
        mask = np.unpackbits(z["mask_%d" % i])[: int(np.prod(shape))].reshape(shape, order="F").astype(np.uint8)
        kw = dict(ast.literal_eval(str(z["kw_%d" % i])))
        extra = dict(ast.literal_eval(str(z["extra_%d" % i])))
        verts = z["verts_%d" % i].astype(np.int64)
        paths, pos = [], 0
        for n in z["lens_%d" % i].tolist():
            paths.append(verts[pos:pos + n])
            pos += n
        yield i, np.asfortranarray(mask), tuple(float(a) for a in z["an_%d" % i]), kw, extra, paths, int(z["first_diff_%d" % i])
