"""Function-level mirrors of the reference's hot-path call sites, running on the MI355X.
Same names, argument meaning and error behaviour as the modules kimimaro imports, so the parity tests
read like the reference's own tests (automated_test.py).  numpy in, numpy out; every call uploads,
runs the HIP kernel(s) through the C ABI, and downloads."""
from __future__ import annotations

import numpy as np

from . import _abi
from .engine import Engine

_engine = None


def engine():
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine


def edt(labels, anisotropy=(1, 1, 1), black_border=False, parallel=1, voxel_graph=None):
    """edt.edt as called at kimimaro/intake.py:178-183."""
    if voxel_graph is not None:
        raise NotImplementedError("voxel_graph")
    eng = engine()
    lab = np.asarray(labels)
    shape0 = lab.shape
    ndim = max(1, min(lab.ndim, 3))
    if lab.dtype == bool:
        lab = lab.view(np.uint8)
    if lab.dtype.kind not in "ui" or lab.dtype.itemsize > 4:
        lab = lab.astype(np.uint32)
    while lab.ndim < 3:
        lab = lab[..., np.newaxis]
    lab = np.asfortranarray(lab)
    an = list(anisotropy) + [1.0] * (3 - len(anisotropy))
    out = eng.edt(eng.to_device(lab), lab.dtype.itemsize, lab.shape, an, black_border, ndim=ndim)
    return out.cpu().numpy().reshape(lab.shape, order="F").reshape(shape0, order="F")


def roll_invalidation_cube(labels, DBF, path, scale, const, anisotropy=(1, 1, 1), invalid_vertices={}):
    """kimimaro.skeletontricks.roll_invalidation_cube (skeletontricks.pyx:766-836).
    Accepts C or F contiguous `labels` (mutated in place and returned, like the reference: `out is labels`),
    coerces the DBF layout without touching the caller's array, raises ValueError when non contiguous."""
    is_f = labels.flags.f_contiguous
    if not (is_f or labels.flags.c_contiguous):
        raise ValueError(
            "roll_invalidation_cube: `labels` must be C- or F-contiguous. "
            "Got shape=({0}, {1}, {2}), strides=({3}, {4}, {5}).".format(*labels.shape, *labels.strides))
    if is_f and not DBF.flags.f_contiguous:
        DBF = np.asfortranarray(DBF)
    elif (not is_f) and not DBF.flags.c_contiguous:
        DBF = np.ascontiguousarray(DBF)
    s0, s1, s2 = labels.shape
    if is_f:
        sx, sy, sz = s0, s1, s2
        w = (anisotropy[0], anisotropy[1], anisotropy[2])
        locs = [c[0] + s0 * c[1] + s0 * s1 * c[2] for c in path if tuple(c) not in invalid_vertices]
    else:
        sx, sy, sz = s2, s1, s0
        w = (anisotropy[2], anisotropy[1], anisotropy[0])
        locs = [c[2] + s2 * c[1] + s2 * s1 * c[0] for c in path if tuple(c) not in invalid_vertices]
    if len(locs) == 0:
        return 0, labels
    eng = engine()
    t = eng.torch
    flat = labels.reshape(-1, order="F" if is_f else "C").view(np.uint8)
    d_mask = t.from_numpy(np.ascontiguousarray(flat)).to(eng.device)
    d_dbf = t.from_numpy(np.ascontiguousarray(DBF.reshape(-1, order="F" if is_f else "C").astype(np.float32, copy=False))).to(eng.device)
    d_path = t.from_numpy(np.asarray(locs, dtype=np.int64)).to(eng.device)
    d_cnt = t.zeros(1, dtype=t.int64, device=eng.device)
    _abi.check(eng.lib.kh_invalidate_cube(eng.ptr(d_mask), eng.ptr(d_dbf), sx, sy, sz, float(w[0]), float(w[1]), float(w[2]),
                                          eng.ptr(d_path), len(locs), np.float32(scale), np.float32(const),
                                          eng.ptr(d_cnt), eng.stream()))
    flat[...] = d_mask.cpu().numpy()
    return int(d_cnt.item()), labels


def roll_invalidation_ball_inside_component(labels, DBF, scale, const, anisotropy=(1, 1, 1), path=(), voxel_connectivity_graph=None):
    """kimimaro.skeletontricks.roll_invalidation_ball_inside_component (skeletontricks.pyx:373-418), call site
    kimimaro/trace.py:253-259: `labels` (uint8 / bool, Fortran order) is the mask of ONE object and is zeroed in place
    inside the rolling ball of every path vertex (radius scale * DBF[v] + const); returns (invalidated, labels)."""
    if voxel_connectivity_graph is not None:
        raise NotImplementedError("voxel_connectivity_graph")
    if not labels.flags.f_contiguous:
        raise ValueError("roll_invalidation_ball_inside_component: labels must be Fortran ordered (skeletontricks.pyx:398)")
    eng = engine()
    lab = labels.view(np.uint8)
    dbf = np.asfortranarray(DBF, dtype=np.float32)
    pts = np.asarray(path, dtype=np.int64).reshape(-1, 3)
    if pts.shape[0] == 0:
        return 0, labels
    sx, sy = lab.shape[0], lab.shape[1]
    locs = pts[:, 0] + sx * (pts[:, 1] + sy * pts[:, 2])
    f = np.float32
    radii = (f(scale) * dbf.reshape(-1, order="F")[locs]).astype(np.float32) + f(const)     # f32 ops, pyx:393-395
    ctx = eng.single_object(lab, anisotropy, rmax=float(radii.max()), dbf=dbf)
    d_alive = eng.torch.from_numpy(np.ascontiguousarray(lab.reshape(-1, order="F"))).to(eng.device)
    cnt, _ = eng.invalidate_ball(ctx, d_alive, locs, scale, const, anisotropy)
    lab.reshape(-1, order="F")[...] = d_alive.cpu().numpy()
    return cnt, labels
