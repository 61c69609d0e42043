"""Generates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference):

  * compiles the reference's ext/skeletontricks where it lies (oracle/build_ref.py -> oracle/_ref)
    and records its OUTPUTS on seeded inputs;
  * loads the reference's kimimaro/trace.py with stub modules for its missing third-party imports
     and records compute_pdrf / find_soma_root outputs;
  * records scipy.ndimage.distance_transform_edt outputs for single-label volumes.

Only data (inputs + expected outputs) is written; no reference source is copied.
Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import build_ref  # noqa: E402
from shapes import random_walk_tube  # noqa: E402

REF = "/root/reference"


def load_reference_trace(st):
    for name in ("dijkstra3d", "edt", "fill_voids"):
        sys.modules.setdefault(name, types.ModuleType(name))
    ost = types.ModuleType("osteoid")
    ost.Skeleton = object
    sys.modules.setdefault("osteoid", ost)
    pkg = types.ModuleType("kimimaro")
    pkg.__path__ = []
    pkg.skeletontricks = st
    sys.modules["kimimaro"] = pkg
    sys.modules["kimimaro.skeletontricks"] = st
    spec = importlib.util.spec_from_file_location("kimimaro.trace", os.path.join(REF, "kimimaro", "trace.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def edt_like(mask, an, rng):
    """A plausible positive float32 field (values only matter at the path vertices)."""
    from scipy import ndimage
    return np.asfortranarray(ndimage.distance_transform_edt(mask, sampling=an).astype(np.float32))


def gen_ball(st):
    rng = np.random.default_rng(20260926)
    cases = {}
    n = 0
    for t in range(60):
        shape = (int(rng.integers(14, 30)), int(rng.integers(14, 30)), int(rng.integers(10, 26)))
        an = [(1, 1, 1), (16, 16, 40), (4, 4, 40), (1, 2, 3), (40, 32, 20)][t % 5]
        if t < 4:
            # SURVEY B-8 shadow case: a 3x3 tube, a big and a small ball next to each other
            shape, an = (40, 5, 5), (1, 1, 1)
            m = np.zeros(shape, np.uint8, order="F")
            m[:, 1:4, 1:4] = 1
            path = np.array([(5, 2, 2), (8, 2, 2)] if t % 2 == 0 else [(8, 2, 2), (5, 2, 2)])
            dbf = np.zeros(shape, np.float32, order="F")
            dbf[5, 2, 2], dbf[8, 2, 2] = 12.0, 2.5
            scale, const = 1.0, 0.0
        else:
            m = random_walk_tube(shape, 500 + t, steps=30, step=2.5, radius=(1.2, 4.0))
            dbf = edt_like(m, an, rng)
            idx = np.flatnonzero(m.ravel(order="F"))
            k = int(rng.integers(1, 14))
            if t % 2 == 0:  # a contiguous run of voxels (like a real path)
                start = int(rng.integers(0, max(1, idx.size - k)))
                sel = idx[start:start + k]
            else:
                sel = rng.choice(idx, min(k, idx.size), replace=False)
            sx, sy = shape[0], shape[1]
            path = np.stack([sel % sx, (sel // sx) % sy, sel // (sx * sy)], axis=1)
            scale = [1.5, 4.0, 0.5, 2.0][t % 4]
            const = [0.0, 3.0 * an[0], 30.0, 0.7][(t // 4) % 4]
        before = m.copy(order="F")
        cnt, _ = st.roll_invalidation_ball_inside_component(
            m, dbf, scale, const, an, [tuple(int(v) for v in p) for p in path])
        sxx, syy = shape[0], shape[1]
        cases["shape_%d" % n] = np.array(shape)
        cases["an_%d" % n] = np.array(an, np.float32)
        cases["mask_%d" % n] = np.packbits(before.ravel(order="F"))
        cases["path_%d" % n] = path.astype(np.int32)
        cases["dbfpath_%d" % n] = dbf[path[:, 0], path[:, 1], path[:, 2]].astype(np.float32)
        cases["sc_%d" % n] = np.array([scale, const], np.float32)
        cases["count_%d" % n] = np.array(cnt)
        cases["after_%d" % n] = np.packbits(m.ravel(order="F"))
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "invalidation_ball.npz"), **cases)
    print("invalidation_ball:", n)


def gen_cube(st):
    # the reference's own fixture recipe (automated_test.py:710-747, seed 0xDECAFBAD) with a random DBF
    rng = np.random.default_rng(seed=0xDECAFBAD)
    cases = {}
    for t in range(100):
        shape = tuple(int(s) for s in rng.integers(8, 24, size=3))
        labels = np.asfortranarray((rng.random(shape) < 0.8).astype(np.uint8))
        dbf = np.asfortranarray(rng.uniform(0.0, 2.0, size=shape).astype(np.float32))
        n_path = int(rng.integers(1, 4))
        path = [tuple(int(rng.integers(0, s)) for s in shape) for _ in range(n_path)]
        radius = float(rng.uniform(0.5, 3.0))
        scale = float(rng.uniform(0.0, 1.5))
        an = tuple(float(rng.uniform(0.5, 4.0)) for _ in range(3))
        before = labels.copy(order="F")
        cnt, out = st.roll_invalidation_cube(labels, dbf, path, scale, radius, anisotropy=an)
        cases["shape_%d" % t] = np.array(shape)
        cases["an_%d" % t] = np.array(an, np.float32)
        cases["mask_%d" % t] = np.packbits(before.ravel(order="F"))
        pa = np.array(path, np.int32)
        cases["dbfpath_%d" % t] = dbf[pa[:, 0], pa[:, 1], pa[:, 2]]  # the cube only reads DBF at the path
        cases["path_%d" % t] = pa
        cases["sc_%d" % t] = np.array([scale, radius], np.float32)
        cases["count_%d" % t] = np.array(cnt)
        cases["after_%d" % t] = np.packbits(out.ravel(order="F"))
    # the reference's exact-count cases (automated_test.py:650-708)
    cases["n"] = np.array(100)
    np.savez_compressed(os.path.join(HERE, "invalidation_cube.npz"), **cases)
    print("invalidation_cube: 100")


def gen_finder(st):
    rng = np.random.default_rng(7)
    cases = {}
    for t in range(8):
        shape = (int(rng.integers(6, 14)), int(rng.integers(6, 14)), int(rng.integers(4, 10)))
        mask = np.asfortranarray((rng.random(shape) < 0.6).astype(np.uint8))
        daf = np.asfortranarray(rng.permutation(mask.size).reshape(shape).astype(np.float32))  # tie free
        finder = st.CachedTargetFinder(mask, daf)
        seq = []
        m = mask.copy(order="F")
        while True:
            tgt = finder.find_target(m)
            if tgt is None:
                break
            seq.append([int(v) for v in tgt])
            # kill a random half of the remaining voxels incl. the target
            m[tuple(seq[-1])] = 0
            kill = rng.random(shape) < 0.3
            m[kill] = 0
            cases.setdefault("kills_%d" % t, []).append(np.packbits(m.ravel(order="F")))
        cases["kills_%d" % t] = np.stack(cases["kills_%d" % t]) if "kills_%d" % t in cases else np.zeros((0, 1), np.uint8)
        cases["shape_%d" % t] = np.array(shape)
        cases["mask_%d" % t] = np.packbits(mask.ravel(order="F"))
        cases["daf_%d" % t] = daf.ravel(order="F")
        cases["seq_%d" % t] = np.array(seq, np.int32).reshape(-1, 3)
        # first_label / zero2inf / inf2zero
        cases["first_%d" % t] = np.array(st.first_label(mask) or (-1, -1, -1))
    cases["n"] = np.array(8)
    np.savez_compressed(os.path.join(HERE, "target_finder.npz"), **cases)
    print("target_finder: 8")


def gen_legacy(st):
    """the legacy skeletontricks.find_target(labels, PDRF) (skeletontricks.pyx:331-367) and first_label (:307-326) of the
    compiled reference: fields with many ties (the FIRST maximum of the x-outermost scan wins), negative values, -inf
    inside the mask, an empty mask."""
    rng = np.random.default_rng(331)
    cases = {}
    n = 0
    for t in range(14):
        shape = (int(rng.integers(5, 40)), int(rng.integers(5, 34)), int(rng.integers(3, 30)))
        dens = [0.6, 0.05, 1.0, 0.3][t % 4]
        mask = np.asfortranarray((rng.random(shape) < dens).astype(np.uint8))
        kind = t % 5
        if kind == 0:
            f = rng.random(shape) * 100                                  # tie free
        elif kind == 1:
            f = np.floor(rng.random(shape) * 6)                          # heavy ties
        elif kind == 2:
            f = -np.floor(rng.random(shape) * 5) - 1                     # all negative, ties
        elif kind == 3:
            f = np.floor(rng.random(shape) * 4)
            f[rng.random(shape) < 0.5] = -np.inf                         # -inf inside the mask
        else:
            f = np.full(shape, 7.0)                                      # one plateau
        f = np.asfortranarray(f.astype(np.float32))
        if t == 9:
            mask[...] = 0                                                # empty mask -> (-1, -1, -1)
        if t == 11:
            f[...] = -np.inf                                             # nothing above -inf -> (-1, -1, -1)
        cases["shape_%d" % n] = np.array(shape)
        cases["mask_%d" % n] = np.packbits(mask.ravel(order="F"))
        cases["field_%d" % n] = f.ravel(order="F")
        cases["target_%d" % n] = np.array(st.find_target(mask, f), np.int64)
        fl = st.first_label(mask)
        cases["first_%d" % n] = np.array(fl if fl is not None else (-1, -1, -1), np.int64)
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "legacy_targets.npz"), **cases)
    print("legacy_targets:", n)


def gen_pdrf(trace):
    rng = np.random.default_rng(11)
    cases = {}
    n = 0
    for expo in (1, 2, 4, 16, 3, 5):        # 3, 5: the np.power branch (trace.py:346-347); appended, so the first 16 cases stay as they were
        for scale in (5000, 100000):
            for zero_daf in (False, True):
                shape = (9, 7, 5)
                dbf = np.asfortranarray(rng.uniform(1, 400, shape).astype(np.float32))
                bg = rng.random(shape) < 0.3
                dbf[bg] = np.inf  # zero2inf'ed background
                daf = np.asfortranarray(rng.uniform(0, 9000, shape).astype(np.float32))
                daf[bg] = 0
                dbf_max = np.max(dbf[~bg])
                max_daf = np.float32(0) if zero_daf else np.max(daf)
                daf_in = daf.copy(order="F")
                out = trace.compute_pdrf(dbf_max, scale, expo, dbf, daf, max_daf)
                cases["dbf_%d" % n] = dbf.ravel(order="F")
                cases["daf_in_%d" % n] = daf_in.ravel(order="F")
                cases["daf_out_%d" % n] = daf.ravel(order="F")
                cases["par_%d" % n] = np.array([dbf_max, scale, expo, max_daf], np.float64)
                cases["out_%d" % n] = out.ravel(order="F")
                n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "pdrf.npz"), **cases)
    print("pdrf:", n)


def gen_border(st):
    from scipy import ndimage
    rng = np.random.default_rng(5)
    cases = {}
    n = 0
    for t in range(24):
        shape = (int(rng.integers(8, 40)), int(rng.integers(8, 40)))
        wx, wy = [(1, 1), (16, 16), (16, 40), (40, 16), (32, 20)][t % 5]
        k = int(rng.integers(1, 6))
        # random blobs: rectangles / discs (lots of ties)
        cc = np.zeros(shape, np.uint32, order="F")
        for lab in range(1, k + 1):
            x0, y0 = int(rng.integers(0, shape[0] - 3)), int(rng.integers(0, shape[1] - 3))
            x1, y1 = int(rng.integers(x0 + 2, shape[0] + 1)), int(rng.integers(y0 + 2, shape[1] + 1))
            cc[x0:x1, y0:y1] = lab
        # make labels connected-component-like: relabel components
        lab2, nl = ndimage.label(cc > 0, structure=np.ones((3, 3)))
        # keep original rectangles as distinct labels (components of equal value)
        out = np.zeros(shape, np.uint32, order="F")
        nxt = 1
        for v in np.unique(cc[cc > 0]):
            comp, nc = ndimage.label(cc == v, structure=np.ones((3, 3)))
            for c in range(1, nc + 1):
                out[comp == c] = nxt
                nxt += 1
        cc = np.asfortranarray(out)
        # black-border multi-label EDT by brute force (small planes)
        dt = np.zeros(shape, np.float32, order="F")
        padded = np.pad(cc, 1)
        for v in range(1, nxt):
            m = padded == v
            d = ndimage.distance_transform_edt(m, sampling=(wx, wy))[1:-1, 1:-1]
            dt[cc == v] = d[cc == v]
        res = st.find_border_targets(dt, cc, wx, wy)
        keys = np.array(list(res.keys()), np.int64)
        vals = np.array([[int(res[key][0]), int(res[key][1])] for key in res.keys()], np.int64).reshape(-1, 2)
        cases["cc_%d" % n] = cc
        cases["dt_%d" % n] = dt
        cases["w_%d" % n] = np.array([wx, wy], np.float32)
        cases["keys_%d" % n] = keys
        cases["vals_%d" % n] = vals
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "border_targets.npz"), **cases)
    print("border_targets:", n)


def gen_edt():
    from scipy import ndimage
    cases = {}
    n = 0
    for seed, an in [(1, (1, 1, 1)), (2, (16, 16, 40)), (3, (8, 8, 40)), (4, (40, 32, 20))]:
        m = random_walk_tube((36, 30, 26), seed)
        d = ndimage.distance_transform_edt(m, sampling=an).astype(np.float32)
        cases["mask_%d" % n] = np.packbits(m.ravel(order="F"))
        cases["shape_%d" % n] = np.array(m.shape)
        cases["an_%d" % n] = np.array(an, np.float32)
        cases["dt_%d" % n] = d.ravel(order="F").astype(np.float16 if False else np.float32)
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "edt_scipy.npz"), **cases)
    print("edt_scipy:", n)


# ---------------------------------------------------------------------------------------------------------------
# trace_paths.npz: the reference's OWN trace() (kimimaro/trace.py:36-267, loaded from /root/reference) run on small
# shapes.  Its skeletontricks is the compiled reference (oracle/_ref: the real heap invalidation, the real
# CachedTargetFinder, zero2inf / inf2zero); its missing third-party imports are stubbed with the oracle's
# restatements (edt.edt -> oracle.edt, dijkstra3d.* -> oracle, fill_voids.fill -> scipy stand-in, osteoid.Skeleton ->
# oracle.skeleton.Skeleton).  What is pinned is therefore the CONTROL FLOW of trace() / compute_paths() (target
# stacks, soma branch, max_paths, rail edits, invalidation calls) as the reference executes it.  Only inputs and the
# resulting path lists are stored.
def load_reference_trace_full(st):
    import scipy.ndimage
    import oracle as K
    from oracle.skeleton import Skeleton

    edt = types.ModuleType("edt")
    edt.edt = lambda labels, anisotropy=(1, 1, 1), black_border=False, voxel_graph=None, parallel=1: K.edt(labels, anisotropy, bool(black_border))
    fv = types.ModuleType("fill_voids")

    def fill(labels, in_place=True, return_fill_count=True):
        filled = scipy.ndimage.binary_fill_holes(labels)
        n = int(np.count_nonzero(filled)) - int(np.count_nonzero(labels))
        out = np.asfortranarray(filled.astype(labels.dtype))
        return (out, n) if return_fill_count else out
    fv.fill = fill
    d3 = types.ModuleType("dijkstra3d")
    d3.euclidean_distance_field = lambda labels, source, anisotropy=(1, 1, 1), free_space_radius=0, voxel_graph=None, return_max_location=False: \
        K.euclidean_distance_field(labels, source, anisotropy, free_space_radius)
    class Parents:   # dijkstra3d.parental_field's result, as the oracle represents it (field + its distance field)
        def __init__(self, field, source):
            self.field, self.source = np.asfortranarray(field), tuple(int(v) for v in source)
            self.dist = K.field_distances(field, source)

        def __setitem__(self, key, value):   # trace.py:220 clears the root's parent (a no-op here) whatever the mode
            assert tuple(int(v) for v in key) == self.source and value == 0
    d3.parental_field = lambda field, source, voxel_graph=None: Parents(field, source)
    d3.path_from_parents = lambda parents, target: K.path_to_source(parents.field, parents.dist, parents.source, tuple(int(v) for v in target))
    d3.railroad = lambda field, source, voxel_graph=None: K.railroad(field, tuple(int(v) for v in source))
    ost = types.ModuleType("osteoid")
    ost.Skeleton = Skeleton
    for name, mod in (("edt", edt), ("fill_voids", fv), ("dijkstra3d", d3), ("osteoid", ost)):
        sys.modules[name] = mod
    pkg = types.ModuleType("kimimaro")
    pkg.__path__ = []
    pkg.skeletontricks = st
    sys.modules["kimimaro"] = pkg
    sys.modules["kimimaro.skeletontricks"] = st
    spec = importlib.util.spec_from_file_location("kimimaro.trace", os.path.join(REF, "kimimaro", "trace.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_trace_paths(st):
    import oracle as K
    from oracle import pipeline as P
    from shapes import soma_shape, voronoi_labels
    trace = load_reference_trace_full(st)
    captured = []
    real_from_path = sys.modules["osteoid"].Skeleton.from_path

    def spy(path):
        captured.append(np.asarray(path, dtype=np.int64).reshape(-1, 3).copy())
        return real_from_path(path)
    sys.modules["osteoid"].Skeleton.from_path = staticmethod(spy)
    cases = {}
    ties = {"n": 0}
    n = 0
    skipped = 0
    specs = []
    for t in range(30):
        an = [(1, 1, 1), (16, 16, 40), (40, 32, 20)][t % 3]
        kw = dict(scale=[1.5, 4, 0.5][t % 3], const=[an[0] * 2, an[0] * 0.5, an[0] * 5][(t // 3) % 3], pdrf_scale=100000, pdrf_exponent=4,
                  fix_branching=(t % 4 != 3))
        if t % 5 == 4:
            kw["max_paths"] = 3
        specs.append(("tube", 700 + t, an, kw))
    for hole in (False, True):
        specs.append(("soma", hole, (1, 1, 1), dict(scale=1.5, const=3, pdrf_scale=100000, pdrf_exponent=4, soma_detection_threshold=6,
                                                     soma_acceptance_threshold=10, soma_invalidation_scale=1.0, soma_invalidation_const=2,
                                                     fix_branching=True)))
    for kind, seed, an, kw in specs:
        if kind == "tube":
            m = random_walk_tube((40, 36, 30), seed, steps=45, step=3.0, radius=(1.3, 4.5))
            cc, _ = K.connected_components(m)
            big = np.argmax(np.bincount(cc.ravel())[1:]) + 1
            m = np.asfortranarray((cc == big).astype(np.uint8))
        else:
            m = np.asfortranarray(soma_shape(hole=seed).astype(np.uint8))
        dbf = K.edt(m, an, black_border=bool(np.all(m)))
        idx = np.flatnonzero(m.ravel(order="F"))
        rng = np.random.default_rng(1000 + n)
        extra = {}
        if kind == "tube" and n % 3 == 1:     # manual targets / forced root (intake.py:486-492)
            pts = K.locs_to_pts(rng.choice(idx, 3, replace=False), m.shape)
            extra = dict(root=tuple(int(v) for v in pts[0]), manual_targets_before=[tuple(int(v) for v in pts[1])],
                         manual_targets_after=[tuple(int(v) for v in pts[2])])
        del captured[:]
        call = dict(kw)
        call.update({k: (list(v) if isinstance(v, list) else v) for k, v in extra.items()})
        skel = trace.trace(m.astype(bool).copy(order="F"), dbf.copy(order="F"), anisotropy=an, **call)
        ref_paths = [p.copy() for p in captured]
        # the oracle on the same input: tie orders the reference leaves to numpy's unstable argsort can differ
        call2 = dict(kw)
        call2.update({k: (list(v) if isinstance(v, list) else v) for k, v in extra.items()})
        mine = P.trace(m.astype(bool).copy(order="F"), dbf.copy(order="F"), anisotropy=an, return_paths=True, **call2)
        same = len(mine) == len(ref_paths) and all(np.array_equal(np.asarray(a), b) for a, b in zip(mine, ref_paths))
        if not same:
            skipped += 1
            k = next((i for i, (a, b) in enumerate(zip(mine, ref_paths)) if not np.array_equal(np.asarray(a), b)), min(len(mine), len(ref_paths)))
            print("  case %s/%s: oracle differs from the reference run (%d vs %d paths, first difference at path %d: targets %s vs %s) -- stored as a documented difference"
                  % (kind, seed, len(mine), len(ref_paths), k, np.asarray(mine[k])[-1].tolist() if k < len(mine) else None, ref_paths[k][-1].tolist() if k < len(ref_paths) else None))
            # the documented difference: numpy's unstable argsort in CachedTargetFinder broke a DAF tie the other way
            # (SURVEY 0-7a).  Stored with the reference's own paths, so that a test can show what the difference is: equal
            # paths up to path k, and at k two targets with the SAME distance from the root.
            t = ties["n"]
            ties["mask_%d" % t] = np.packbits(m.ravel(order="F"))
            ties["shape_%d" % t] = np.array(m.shape)
            ties["an_%d" % t] = np.array(an, np.float32)
            ties["kw_%d" % t] = np.array(repr(sorted(kw.items())))
            ties["extra_%d" % t] = np.array(repr(sorted(extra.items())))
            ties["first_diff_%d" % t] = np.array(k)
            ties["lens_%d" % t] = np.array([len(p) for p in ref_paths], np.int64)
            ties["verts_%d" % t] = np.concatenate(ref_paths).astype(np.int32) if ref_paths else np.zeros((0, 3), np.int32)
            ties["n"] = t + 1
            continue
        cases["mask_%d" % n] = np.packbits(m.ravel(order="F"))
        cases["shape_%d" % n] = np.array(m.shape)
        cases["an_%d" % n] = np.array(an, np.float32)
        cases["kw_%d" % n] = np.array(repr(sorted(kw.items())))
        cases["extra_%d" % n] = np.array(repr(sorted(extra.items())))
        cases["npaths_%d" % n] = np.array(len(ref_paths))
        cases["lens_%d" % n] = np.array([len(p) for p in ref_paths], np.int64)
        cases["verts_%d" % n] = np.concatenate(ref_paths).astype(np.int32) if ref_paths else np.zeros((0, 3), np.int32)
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "trace_paths.npz"), **cases)
    ties["n"] = np.array(ties["n"])
    np.savez_compressed(os.path.join(HERE, "trace_paths_ties.npz"), **ties)
    print("trace_paths:", n, "stored,", skipped, "stored as documented differences (trace_paths_ties.npz)")




# ---------------------------------------------------------------------------------------------------------------
# f4: kimimaro/post.py (postprocess = remove_dust -> remove_loops -> join_close_components -> remove_ticks).
# The reference's own file is imported with stand-ins for the packages this image lacks: fastremap.unique -> np.unique,
# osteoid.Skeleton -> tests/golden/osteoid_standin.py, kimimaro.skeletontricks -> the compiled reference (oracle/_ref).
def load_reference_post(st):
    import osteoid_standin
    fr = types.ModuleType("fastremap")
    fr.unique = lambda a, return_counts=False: np.unique(np.asarray(a), return_counts=return_counts)
    ost = types.ModuleType("osteoid")
    ost.Skeleton = osteoid_standin.Skeleton
    ost.Bbox = osteoid_standin.Bbox
    sys.modules["fastremap"] = fr
    sys.modules["osteoid"] = ost
    pkg = types.ModuleType("kimimaro")
    pkg.__path__ = []
    pkg.skeletontricks = st
    sys.modules["kimimaro"] = pkg
    sys.modules["kimimaro.skeletontricks"] = st
    spec = importlib.util.spec_from_file_location("kimimaro.post", os.path.join(REF, "kimimaro", "post.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, osteoid_standin.Skeleton


def random_forest(rng, ntrees, nodes, spread, loops=0, tick_every=0):
    """vertices (float coordinates: no exact ties between branch lengths), edges, radii of a few random trees, with
    `loops` extra edges closing cycles inside trees and short side branches ("ticks")."""
    verts, edges, radii = [], [], []
    for t in range(ntrees):
        base = len(verts)
        origin = rng.uniform(0, spread, 3)
        n = int(nodes[t % len(nodes)])
        pos = [origin]
        par = [-1]
        for i in range(1, n):
            # mostly extend the latest tip (long neurite-like runs), sometimes branch off an earlier node
            p = i - 1 if rng.random() < 0.85 else int(rng.integers(0, i))
            step = rng.normal(0, 1, 3)
            step = step / np.linalg.norm(step) * rng.uniform(20, 60)
            if tick_every and i % tick_every == 0:
                p = int(rng.integers(0, i))
                step = step / np.linalg.norm(step) * rng.uniform(3, 15)
            pos.append(pos[p] + step)
            par.append(p)
        for i in range(n):
            verts.append(pos[i])
            radii.append(rng.uniform(5, 40))
            if par[i] >= 0:
                edges.append((base + par[i], base + i))
        for _ in range(loops):
            a, b = rng.choice(n, 2, replace=False)
            edges.append((base + int(a), base + int(b)))
    return np.array(verts, np.float32), np.array(edges, np.uint32), np.array(radii, np.float32)


def gen_post(st):
    post, Skel = load_reference_post(st)
    cases = {}
    n = 0

    def store(fn, args, vin, ein, rin, out):
        nonlocal n
        cases["fn_%d" % n] = np.array(fn)
        cases["args_%d" % n] = np.array(repr(args))
        cases["vin_%d" % n], cases["ein_%d" % n], cases["rin_%d" % n] = vin, ein, rin
        cases["vout_%d" % n] = np.asarray(out.vertices, np.float32)
        cases["eout_%d" % n] = np.asarray(out.edges, np.int64).reshape(-1, 2)
        cases["rout_%d" % n] = np.asarray(out.radii, np.float32)
        n += 1

    rng = np.random.default_rng(20240)
    for t in range(36):
        kind = t % 6
        ntrees = [1, 3, 2, 4, 1, 3][kind]
        loops = [0, 0, 1, 2, 3, 1][kind]
        ticks = [6, 0, 5, 7, 4, 6][kind]
        spread = [400, 150, 600, 250, 300, 2000][kind]
        v, e, r = random_forest(rng, ntrees, [60, 25, 90, 40], spread, loops=loops, tick_every=ticks)
        mk = lambda: Skel(v.copy(), e.copy(), r.copy(), segid=7)
        dust = [0, 300, 1500][t % 3]
        tick = [0, 40, 120][(t // 3) % 3]
        store("postprocess", (dust, tick), v, e, r, post.postprocess(mk(), dust_threshold=dust, tick_threshold=tick))
        store("remove_dust", (dust,), v, e, r, post.remove_dust(mk().consolidate(), dust))
        store("remove_loops", (), v, e, r, post.remove_loops(mk().consolidate()))
        if loops == 0:
            store("remove_ticks", (tick,), v, e, r, post.remove_ticks(mk().consolidate(), tick))
        rad = [None, 80.0, 500.0][t % 3]
        store("join_close_components", (rad, bool(t % 2)), v, e, r,
              post.join_close_components(mk(), radius=rad, restrict_by_radius=bool(t % 2)))
    # skeletons of a chunked volume, merged per label (what post.py is for): the oracle pipeline on four overlapping
    # quadrants of a small dense volume with fix_borders, then the reference's postprocess
    from oracle import pipeline as P
    from shapes import voronoi_labels
    an = (16, 16, 40)
    lab = voronoi_labels((96, 96, 40), 10, 77, pts_per_label=5, anisotropy=an)
    parts = {}
    for ox in (0, 47):
        for oy in (0, 47):
            sub = np.asfortranarray(lab[ox:ox + 49, oy:oy + 49, :])
            sk = P.skeletonize(sub, anisotropy=an, dust_threshold=50, fix_borders=True, fix_branching=True)
            for k, s in sk.items():
                vv = s.vertices + np.array([ox * an[0], oy * an[1], 0], np.float32)
                parts.setdefault(k, []).append(Skel(vv, s.edges, s.radii, segid=k))
    for k in sorted(parts):
        merged = Skel.simple_merge(parts[k])
        if merged.empty():
            continue
        v, e, r = merged.vertices.copy(), merged.edges.copy(), merged.radii.copy()
        store("postprocess", (500, 900), v, e, r, post.postprocess(Skel(v.copy(), e.copy(), r.copy(), segid=k), 500, 900))
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "post.npz"), **cases)
    print("post:", n, "cases")


def gen_ball_graph(st):
    """roll_invalidation_ball_inside_component(..., voxel_connectivity_graph=) of the COMPILED reference
    (skeletontricks.pyx:373-418 -> dijkstra_invalidation.hpp:126-191) on 40 tubes / blobs with random connectivity graphs:
    a fraction of the 26 bits of every voxel cleared at random (asymmetrically, as the reference reads only the CURRENT voxel's
    word), plus planes of walls; some objects touch the x faces of their array (the degenerate corner entries there are gated by
    the corner's bit)."""
    rng = np.random.default_rng(20260927)
    cases = {}
    n = 0
    for t in range(40):
        shape = (int(rng.integers(8, 22)), int(rng.integers(10, 24)), int(rng.integers(8, 20)))
        an = [(1, 1, 1), (16, 16, 40), (4, 4, 40), (1, 2, 3), (40, 32, 20)][t % 5]
        if t % 4 == 3:
            m = np.ones(shape, np.uint8, order="F")          # a solid block: every x face voxel has degenerate corners
        else:
            m = random_walk_tube(shape, 900 + t, steps=30, step=2.5, radius=(1.5, 4.0))
        dbf = edt_like(m, an, rng)
        vcg = np.full(shape, (1 << 26) - 1, dtype=np.uint32, order="F")
        drop = [0.0, 0.05, 0.2, 0.5][t % 4]
        for b in range(26):
            vcg &= ~(np.asfortranarray(rng.random(shape) < drop).astype(np.uint32) << np.uint32(b))
        if t % 3 == 0:                                       # a wall: nobody steps in +z across the middle plane
            zc = shape[2] // 2
            plus_z = sum(1 << b for b in (4, 13, 12, 11, 10, 21, 20, 19, 18))   # the bits of the entries with dz = +1
            vcg[:, :, zc] &= np.uint32(~plus_z & 0xFFFFFFFF)
        idx = np.flatnonzero(m.ravel(order="F"))
        k = int(rng.integers(1, 10))
        start = int(rng.integers(0, max(1, idx.size - k)))
        sel = idx[start:start + k] if t % 2 == 0 else rng.choice(idx, min(k, idx.size), replace=False)
        sx, sy = shape[0], shape[1]
        path = np.stack([sel % sx, (sel // sx) % sy, sel // (sx * sy)], axis=1)
        scale = [1.5, 4.0, 0.5, 2.0][t % 4]
        const = [0.0, 3.0 * an[0], 30.0, 0.7][(t // 4) % 4]
        before = m.copy(order="F")
        cnt, _ = st.roll_invalidation_ball_inside_component(
            m, dbf, scale, const, an, [tuple(int(v) for v in p) for p in path], voxel_connectivity_graph=vcg)
        cases["shape_%d" % n] = np.array(shape)
        cases["an_%d" % n] = np.array(an, np.float32)
        cases["mask_%d" % n] = np.packbits(before.ravel(order="F"))
        cases["graph_%d" % n] = vcg.ravel(order="F").copy()
        cases["path_%d" % n] = path.astype(np.int32)
        cases["dbfpath_%d" % n] = dbf[path[:, 0], path[:, 1], path[:, 2]].astype(np.float32)
        cases["sc_%d" % n] = np.array([scale, const], np.float32)
        cases["count_%d" % n] = np.array(cnt)
        cases["after_%d" % n] = np.packbits(m.ravel(order="F"))
        n += 1
    # Round 5: cases in which a voxel is reached ONLY through a corner entry that degenerated into a yz diagonal at an x face of
    # the array (dijkstra_invalidation.hpp:116-123), which the graph gates by the CORNER's bit (:182-190) although the diagonal's
    # own bit is clear.  First the hand-derived one (shape (1, 2, 2), one allowed corner: the reference invalidates 2 voxels), then
    # thin slabs (sx = 1..3: every voxel on an x face) whose graphs have all four yz-diagonal bits cleared everywhere.
    rng2 = np.random.default_rng(20260928)
    yz_diag = sum(1 << b for b in (17, 13, 16, 12))
    for t in range(13):
        if t == 0:
            shape, an = (1, 2, 2), (1, 1, 1)
            m = np.ones(shape, np.uint8, order="F")
            vcg = np.zeros(shape, np.uint32, order="F")
            vcg[0, 0, 0] = 1 << 18
            path = np.array([[0, 0, 0]])
            dbf = np.full(shape, 10.0, np.float32, order="F")
            scale, const = 1.0, 0.0
        else:
            shape = (int(rng2.integers(1, 4)), int(rng2.integers(4, 12)), int(rng2.integers(4, 12)))
            an = [(1, 1, 1), (16, 16, 40), (4, 4, 40), (1, 2, 3)][t % 4]
            m = np.asfortranarray((rng2.random(shape) < 0.9).astype(np.uint8))
            dbf = edt_like(m, an, rng2)
            vcg = np.full(shape, (1 << 26) - 1, dtype=np.uint32, order="F") & np.uint32(~yz_diag & 0xFFFFFFFF)
            drop = [0.3, 0.6, 0.8][t % 3]
            for b in range(18):                              # faces and edges mostly closed, corners stay open
                vcg &= ~(np.asfortranarray(rng2.random(shape) < drop).astype(np.uint32) << np.uint32(b))
            for b in range(18, 26):
                vcg &= ~(np.asfortranarray(rng2.random(shape) < 0.15).astype(np.uint32) << np.uint32(b))
            idx = np.flatnonzero(m.ravel(order="F"))
            sel = rng2.choice(idx, min(int(rng2.integers(1, 4)), idx.size), replace=False)
            sx, sy = shape[0], shape[1]
            path = np.stack([sel % sx, (sel // sx) % sy, sel // (sx * sy)], axis=1)
            scale, const = [4.0, 8.0][t % 2], 6.0 * max(an)
        before = m.copy(order="F")
        cnt, _ = st.roll_invalidation_ball_inside_component(
            m, dbf, scale, const, an, [tuple(int(v) for v in p) for p in path], voxel_connectivity_graph=vcg)
        cases["shape_%d" % n] = np.array(shape)
        cases["an_%d" % n] = np.array(an, np.float32)
        cases["mask_%d" % n] = np.packbits(before.ravel(order="F"))
        cases["graph_%d" % n] = vcg.ravel(order="F").copy()
        cases["path_%d" % n] = path.astype(np.int32)
        cases["dbfpath_%d" % n] = dbf[path[:, 0], path[:, 1], path[:, 2]].astype(np.float32)
        cases["sc_%d" % n] = np.array([scale, const], np.float32)
        cases["count_%d" % n] = np.array(cnt)
        cases["after_%d" % n] = np.packbits(m.ravel(order="F"))
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "invalidation_ball_graph.npz"), **cases)
    print("invalidation_ball_graph:", n)


def gen_avocado(st):
    """kimimaro.skeletontricks.find_avocado_fruit (skeletontricks.pyx:905-992) and get_mapping (:490-525) of the compiled reference on
    seeded volumes: random label soups (every tie and every ray that runs into the background or the array's edge), nested shells
    (a pit inside a fruit, inside another), coordinates on faces, edges and corners."""
    rng = np.random.default_rng(0xA70CAD0)
    cases, n = {}, 0
    for t in range(120):
        shp = tuple(int(v) for v in rng.integers(1, 9, 3))
        if t % 3 == 0:
            shp = tuple(max(v, 3) for v in shp)
            lab = np.full(shp, 3, dtype=np.uint32, order="F")
            lab[1:-1, 1:-1, 1:-1] = 2
            if min(shp) >= 5:
                lab[2:-2, 2:-2, 2:-2] = 1
            if t % 6 == 0:
                lab[tuple(int(rng.integers(0, v)) for v in shp)] = 0
        else:
            lab = np.asfortranarray(rng.integers(0, int(rng.integers(2, 6)), shp).astype(np.uint32))
        pts = np.array([[int(rng.integers(0, v)) for v in shp] for _ in range(6)], dtype=np.int64)
        out = np.array([[int(v) for v in st.find_avocado_fruit(lab, int(p[0]), int(p[1]), int(p[2]))] for p in pts], dtype=np.int64)
        orig = np.asfortranarray(rng.integers(0, 6, shp).astype(np.uint32))
        cc = np.asfortranarray(rng.integers(0, 5, shp).astype(np.uint32))
        mp = st.get_mapping(orig, cc)
        keys = np.array(sorted(int(k) for k in mp), dtype=np.int64)
        cases["lab%d" % n] = lab
        cases["pts%d" % n] = pts
        cases["fruit%d" % n] = out
        cases["orig%d" % n] = orig
        cases["cc%d" % n] = cc
        cases["mapk%d" % n] = keys
        cases["mapv%d" % n] = np.array([int(mp[k]) for k in keys], dtype=np.int64)
        n += 1
    cases["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "avocado.npz"), **cases)
    print("avocado:", n)


if __name__ == "__main__" and "avocado" in sys.argv[1:]:
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    gen_avocado(st)
elif __name__ == "__main__" and "ball_graph" in sys.argv[1:]:
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    gen_ball_graph(st)
elif __name__ == "__main__" and "post" in sys.argv[1:]:
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    gen_post(st)
elif __name__ == "__main__" and "pdrf" in sys.argv[1:]:
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    gen_pdrf(load_reference_trace(st))
elif __name__ == "__main__" and "legacy" in sys.argv[1:]:
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    gen_legacy(st)
elif __name__ == "__main__" and "trace_paths" in sys.argv[1:]:
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    gen_trace_paths(st)
elif __name__ == "__main__":
    st = build_ref.load()
    assert st is not None, "needs /root/reference"
    trace = load_reference_trace(st)
    gen_ball(st)
    gen_ball_graph(st)
    gen_cube(st)
    gen_finder(st)
    gen_legacy(st)
    gen_pdrf(trace)
    gen_border(st)
    gen_edt()
    gen_avocado(st)


