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
    if voxel_graph is not None:
        # edt.edt(voxel_graph=): walls between voxels (kh_edt_graph_cells / kh_edt_graph_sample; the package is absent from the
        # reference tree: PARITY UNPINNED, include/kimi_hip.h)
        if ndim < 3 and black_border:
            raise NotImplementedError("edt(voxel_graph=, black_border=True) on fewer than three axes")
        vg = np.asarray(voxel_graph)
        while vg.ndim < 3:
            vg = vg[..., np.newaxis]
        if tuple(vg.shape) != tuple(lab.shape):
            raise ValueError("voxel_graph must have the shape of the labels")
        d_graph = eng.to_device(np.asfortranarray(vg.astype(np.uint32)))
        out = eng.edt_graph(eng.to_device(lab), lab.dtype.itemsize, d_graph, lab.shape, an, black_border)
        return out.cpu().numpy().reshape(lab.shape, order="F").reshape(shape0, order="F")
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
    inside the rolling ball of every path vertex (radius scale * DBF[v] + const); returns (invalidated, labels).
    voxel_connectivity_graph (uint32, the labels' shape, cc3d's bit layout; skeletontricks.pyx:380,405-416 ->
    dijkstra_invalidation.hpp:126-191): a direction whose bit is clear in the word of the voxel being expanded is not followed."""
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
    ctx = eng.single_object(lab, anisotropy, rmax=float(radii.max()), dbf=dbf, voxel_graph=voxel_connectivity_graph)
    d_alive = eng.torch.from_numpy(np.ascontiguousarray(lab.reshape(-1, order="F"))).to(eng.device)
    cnt, _ = eng.invalidate_ball(ctx, d_alive, locs, scale, const, anisotropy)
    lab.reshape(-1, order="F")[...] = d_alive.cpu().numpy()
    return cnt, labels


# ---------------------------------------------------------------------------------------------------------------------
# Function-level mirrors of the modules kimimaro/trace.py imports: same names, argument meaning
# and return values as at the call sites cited, every computation a HIP kernel reached through the C ABI.  The path
# loop of skeletonize() does not go through these (it has them fused on the device); they exist so that the reference's
# own vectors can be fed to the HIP implementation of every step.

def _f3(a, dtype=None):
    a = np.asarray(a)
    while a.ndim < 3:
        a = a[..., np.newaxis]
    return np.asfortranarray(a if dtype is None else a.astype(dtype, copy=False))


def _loc(pt, shape):
    return int(pt[0]) + shape[0] * (int(pt[1]) + shape[1] * int(pt[2]))


def _pts(locs, shape):
    locs = np.asarray(locs, dtype=np.int64)
    return np.stack([locs % shape[0], (locs // shape[0]) % shape[1], locs // (shape[0] * shape[1])], axis=1)


def zero2inf(field):
    """kimimaro.skeletontricks.zero2inf (skeletontricks.pyx:203-224): in place, returns the same array."""
    eng = engine()
    flat = field.reshape(-1, order="F")
    d = eng.torch.from_numpy(np.ascontiguousarray(flat)).to(eng.device)
    _abi.check(eng.lib.kh_zero2inf(eng.ptr(d), flat.size, eng.stream()))
    flat[...] = d.cpu().numpy()
    return field


def inf2zero(field):
    """kimimaro.skeletontricks.inf2zero (skeletontricks.pyx:177-198): in place, returns the same array."""
    eng = engine()
    flat = field.reshape(-1, order="F")
    d = eng.torch.from_numpy(np.ascontiguousarray(flat)).to(eng.device)
    _abi.check(eng.lib.kh_inf2zero(eng.ptr(d), flat.size, eng.stream()))
    flat[...] = d.cpu().numpy()
    return field


def fill(img, in_place=True, return_fill_count=True):
    """fill_voids.fill as called at kimimaro/trace.py:109."""
    eng = engine()
    m = _f3(img)
    d_mask = eng.torch.from_numpy(np.ascontiguousarray((m != 0).astype(np.uint8).reshape(-1, order="F"))).to(eng.device)
    d_out, n = eng.fill_voids(d_mask, m.shape)
    out = d_out.cpu().numpy().reshape(m.shape, order="F").astype(img.dtype).reshape(img.shape, order="F")
    if in_place:
        img[...] = out
        out = img
    return (out, n) if return_fill_count else out


def compute_pdrf(dbf_max, pdrf_scale, pdrf_exponent, DBF, DAF, max_daf):
    """kimimaro.trace.compute_pdrf (kimimaro/trace.py:315-356): DBF has been through zero2inf, DAF is normalised in
    place (like the reference).  Power-of-two exponents take the repeated-squaring branch (:343-345); any other
    exponent the np.power branch (:346-347): the device computes the base and the tail, numpy itself the power (its
    rounding is the host libm's -- the only function that reproduces the reference's bits on a given machine)."""
    eng = engine()
    f = np.float32
    M = f(1 / (f(dbf_max) ** 1.01))
    t = eng.torch
    d_dbf = t.from_numpy(np.ascontiguousarray(np.asarray(DBF, dtype=np.float32).reshape(-1, order="F"))).to(eng.device)
    daf_flat = DAF.reshape(-1, order="F")
    d_daf = t.from_numpy(np.ascontiguousarray(daf_flat)).to(eng.device)
    d_out = eng.empty(d_dbf.numel(), t.float32)
    call = lambda stage: _abi.check(eng.lib.kh_pdrf_field(eng.ptr(d_dbf), eng.ptr(d_daf), d_dbf.numel(), M, stage, f(pdrf_scale),
                                                          f(max_daf), eng.ptr(d_out), eng.stream()))
    if _abi.is_pow2_exponent(pdrf_exponent):
        call(int(pdrf_exponent).bit_length() - 1)
    else:
        call(_abi.PDRF_BASE)
        base = d_out.cpu().numpy()
        with np.errstate(all="ignore"):
            np.power(base, pdrf_exponent, out=base)
        d_out.copy_(t.from_numpy(base))
        call(_abi.PDRF_FINISH)
    daf_flat[...] = d_daf.cpu().numpy()
    return d_out.cpu().numpy().reshape(np.asarray(DBF).shape, order="F")


def euclidean_distance_field(labels, source, anisotropy=(1, 1, 1), free_space_radius=0, voxel_graph=None,
                             return_max_location=False):
    """dijkstra3d.euclidean_distance_field as called at kimimaro/trace.py:139-145, 302-307: geodesic distance inside
    the mask from `source`, +inf elsewhere; with return_max_location also the (x, y, z) of the largest finite value."""
    eng = engine()
    lab = _f3(labels)
    # voxel_graph: the directions a voxel's word does not allow leave the neighbour masks the search works from
    # (kh_apply_voxel_graph; one-way edges for an asymmetric graph, like the invalidation reads it)
    ctx = eng.single_object(lab, anisotropy, voxel_graph=voxel_graph)
    shape = ctx["shape"]
    t = eng.torch
    task = ctx["task"]
    task["root"] = _loc(source, shape)
    task["fsr"] = np.float32(free_space_radius)
    d_task = t.from_numpy(task.view(np.uint8).reshape(-1).copy()).to(eng.device)
    d_field = t.full((ctx["nvox"] + 4,), float("inf"), dtype=t.float32, device=eng.device)   # (+ padding: the searches read rows of three words)
    P = eng.ptr
    _abi.check(eng.lib.kh_edf_batch(P(d_task), 1, 2, P(ctx["d_lists"]), P(ctx["d_nbr"]), shape[0], shape[1], shape[2],
                                    float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]), P(d_field),
                                    P(ctx["d_qstate"]), P(ctx["d_queues"]), eng.stream()))
    out = d_field[:ctx["nvox"]].cpu().numpy().reshape(shape, order="F").reshape(np.asarray(labels).shape, order="F")
    if not return_max_location:
        return out
    done = d_task.cpu().numpy().view(_abi.LABEL_T)
    return out, tuple(int(v) for v in _pts([int(done["max_loc"][0])], shape)[0])


class _Search:
    """an object's device context + its weight field for kh_path_search"""

    def __init__(self, field, voxel_graph=None):
        eng = engine()
        self.eng = eng
        self.host_shape = np.asarray(field).shape
        f = _f3(field, np.float32)
        self.shape = f.shape
        self.graph = voxel_graph is not None
        self.ctx = eng.single_object(np.isfinite(f), (1, 1, 1), voxel_graph=voxel_graph)
        t = eng.torch
        flat = np.concatenate([f.reshape(-1, order="F"), np.full(4, np.inf, dtype=np.float32)])     # (+ padding: rows of three words)
        self.d_field = t.from_numpy(np.ascontiguousarray(flat)).to(eng.device)
        self.d_dist = t.full((f.size + 4,), float("inf"), dtype=t.float32, device=eng.device)

    def run(self, mode, source, target=0):
        eng, ctx, t, P = self.eng, self.ctx, self.eng.torch, self.eng.ptr
        cap = 4 * ctx["count"] + 1024
        d_path = eng.empty(cap, t.int32)
        d_n = t.zeros(1, dtype=t.int32, device=eng.device)
        sx, sy, sz = self.shape
        _abi.check(eng.lib.kh_path_search(P(ctx["d_task"]), mode, P(ctx["d_lists"]), P(ctx["d_nbr"]), sx, sy, sz, 1.0, 1.0, 1.0,
                                          P(self.d_field), P(self.d_dist), P(ctx["d_qstate"]), P(ctx["d_queues"]), int(source),
                                          int(target), P(d_path), cap, P(d_n), int(self.graph), eng.stream()))
        status = int(ctx["d_task"].cpu().numpy().view(_abi.LABEL_T)["status"][0])
        if status:
            raise _abi.KimiHipError("kh_path_search: %s" % _abi.describe_status(status))
        n = int(d_n.item())
        return _pts(d_path[:n].cpu().numpy().view(np.uint32), self.shape)


def railroad(field, source, voxel_graph=None):
    """dijkstra3d.railroad(field, source) as called at kimimaro/trace.py:240-242: the path from `source` to the nearest
    zero-weight voxel, rail end first, as an (n, 3) array."""
    s = _Search(field, voxel_graph)
    return s.run(0, _loc(source, s.shape))


def parental_field(field, source, voxel_graph=None):
    """dijkstra3d.parental_field(field, source) as called at kimimaro/trace.py:155: the parents ARRAY of the weighted Dijkstra
    from `source` (uint32, the field's shape, Fortran order; entry = linear index of the predecessor + 1, 0 = none), which the
    caller may edit (`parents[tuple(root)] = 0`, trace.py:220) before handing it to path_from_parents (kh_parental_field)."""
    s = _Search(field, voxel_graph)
    eng, ctx, t, P = s.eng, s.ctx, s.eng.torch, s.eng.ptr
    sx, sy, sz = s.shape
    d_par = eng.empty(sx * sy * sz, t.int32)
    _abi.check(eng.lib.kh_parental_field(P(ctx["d_task"]), P(ctx["d_lists"]), P(ctx["d_nbr"]), sx, sy, sz, P(s.d_field), P(s.d_dist),
                                         P(ctx["d_qstate"]), P(ctx["d_queues"]), _loc(source, s.shape), P(d_par), int(s.graph),
                                         eng.stream()))
    status = int(ctx["d_task"].cpu().numpy().view(_abi.LABEL_T)["status"][0])
    if status:
        raise _abi.KimiHipError("kh_parental_field: %s" % _abi.describe_status(status))
    return d_par.cpu().numpy().view(np.uint32).reshape(s.shape, order="F").reshape(s.host_shape, order="F")


def path_from_parents(parents, target):
    """dijkstra3d.path_from_parents(parents, target) as called at kimimaro/trace.py:244: the pointer chase from `target`
    to the voxel whose entry is 0, returned source -> target as an (n, 3) array (kh_path_from_parents)."""
    eng = engine()
    t, P = eng.torch, eng.ptr
    par = _f3(parents, np.uint32)
    shape = par.shape
    d_par = t.from_numpy(np.ascontiguousarray(par.reshape(-1, order="F")).view(np.int32)).to(eng.device)
    cap = par.size
    d_path = eng.empty(cap, t.int32)
    d_n = t.zeros(1, dtype=t.int32, device=eng.device)
    _abi.check(eng.lib.kh_path_from_parents(P(d_par), par.size, _loc(target, shape), P(d_path), cap, P(d_n), eng.stream()))
    n = int(d_n.item())
    if n == 0:
        raise _abi.KimiHipError("kh_path_from_parents: the parents of the target do not lead to a voxel without a parent")
    return _pts(d_path[:n].cpu().numpy().view(np.uint32), shape)


def dijkstra(field, source, target, voxel_graph=None):
    """dijkstra3d.dijkstra(field, source, target) as called at kimimaro/trace.py:385 (point_to_point): the cheapest path
    from `source` to `target` where entering a voxel costs its field value, as an (n, 3) array source first.  The same
    search as parental_field + path_from_parents (one weighted Dijkstra from the source, predecessor walk from the
    target); ties between equally cheap paths follow the canonical predecessor rule of DESIGN.md 3.3."""
    s = _Search(field, voxel_graph)
    src = _loc(source, s.shape)
    s.run(1, src)                                   # (the predecessor WALK crosses float-absorption plateaus, which a
    return s.run(2, src, _loc(target, s.shape))     # parents array cannot express: DESIGN.md 3.3)


def first_label(labels):
    """kimimaro.skeletontricks.first_label (skeletontricks.pyx:307-326): (x, y, z) of the first non-zero voxel in the
    z / y / x raster, None when there is none."""
    eng = engine()
    lab = _f3(labels)
    t = eng.torch
    d = t.from_numpy(np.ascontiguousarray((lab != 0).astype(np.uint8).reshape(-1, order="F"))).to(eng.device)
    d_out = t.zeros(1, dtype=t.int64, device=eng.device)
    _abi.check(eng.lib.kh_first_label(eng.ptr(d), lab.size, eng.ptr(d_out), eng.stream()))
    loc = int(d_out.cpu().numpy().view(np.uint64)[0])
    if loc == 0xFFFFFFFFFFFFFFFF:
        return None
    return tuple(int(v) for v in _pts([loc], lab.shape)[0])


def find_target(labels, PDRF):
    """the legacy kimimaro.skeletontricks.find_target(labels, PDRF) (skeletontricks.pyx:331-367): the first voxel in the
    reference's scan order (x outermost, z innermost) holding the maximum of PDRF over the mask; (-1, -1, -1) when the
    mask is empty (or holds only -inf / NaN)."""
    eng = engine()
    lab = _f3(labels)
    t = eng.torch
    d_lab = t.from_numpy(np.ascontiguousarray((lab != 0).astype(np.uint8).reshape(-1, order="F"))).to(eng.device)
    d_f = t.from_numpy(np.ascontiguousarray(_f3(PDRF, np.float32).reshape(-1, order="F"))).to(eng.device)
    d_out = t.zeros(1, dtype=t.int64, device=eng.device)
    sx, sy, sz = lab.shape
    _abi.check(eng.lib.kh_find_target(eng.ptr(d_lab), eng.ptr(d_f), sx, sy, sz, eng.ptr(d_out), eng.stream()))
    key = int(d_out.cpu().numpy().view(np.uint64)[0])
    if key == 0:
        return (-1, -1, -1)
    scan = 0xFFFFFFFF - (key & 0xFFFFFFFF)
    return (int(scan // (sy * sz)), int((scan // sz) % sy), int(scan % sz))


class CachedTargetFinder:
    """kimimaro.skeletontricks.CachedTargetFinder (skeletontricks.pyx:995-1045): find_target(labels) returns the
    still-valid voxel with the largest DAF (ties: the larger index), None when none is left."""

    def __init__(self, labels, daf):
        eng = engine()
        self.eng = eng
        lab = _f3(labels)
        self.shape = lab.shape
        self.ctx = eng.single_object(lab, (1, 1, 1))
        t = eng.torch
        d_daf = t.from_numpy(np.ascontiguousarray(_f3(daf, np.float32).reshape(-1, order="F"))).to(eng.device)
        self.d_ldaf = eng.empty(max(self.ctx["count"], 1), t.float32)
        _abi.check(eng.lib.kh_gather_f32(eng.ptr(d_daf), eng.ptr(self.ctx["d_lists"]), self.ctx["count"], eng.ptr(self.d_ldaf),
                                         eng.stream()))

    def find_target(self, labels):
        eng, t = self.eng, self.eng.torch
        d_alive = t.from_numpy(np.ascontiguousarray(_f3(labels).view(np.uint8).reshape(-1, order="F"))).to(eng.device)
        d_out = t.zeros(1, dtype=t.int64, device=eng.device)
        _abi.check(eng.lib.kh_target_max(eng.ptr(self.ctx["d_lists"]), eng.ptr(self.d_ldaf), eng.ptr(d_alive), self.ctx["count"],
                                         eng.ptr(d_out), eng.stream()))
        key = int(d_out.cpu().numpy().view(np.uint64)[0])
        if key == 0:
            return None
        return tuple(int(v) for v in _pts([key & 0xFFFFFFFF], self.shape)[0])
