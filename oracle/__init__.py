"""oracle -- CPU restatement of the kimimaro hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (kimimaro_amd/) never does: it must fail loudly when
the HIP library is missing rather than fall back to anything here.

numpy-facing wrappers over oracle/libkimi_oracle.so (built by oracle/build.py
from oracle/kimi_oracle.c).  All volumes are Fortran ordered, coordinates are
(x, y, z), linear index = x + sx*(y + sy*z).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB
        if not os.path.exists(path) or (
                os.path.exists(_build.SRC) and os.path.getmtime(path) < os.path.getmtime(_build.SRC)):
            path = _build.build()
        L = C.CDLL(path)
        i64, f32, u64, vp = C.c_int64, C.c_float, C.c_uint64, C.c_void_p
        L.ko_weights26.argtypes = [f32, f32, f32, vp]
        L.ko_edt.argtypes = [vp, C.c_int, i64, i64, i64, f32, f32, f32, C.c_int, vp]
        L.ko_edt_nd.argtypes = [vp, C.c_int, C.c_int, i64, i64, i64, f32, f32, f32, C.c_int, vp]
        L.ko_edf.argtypes = [vp, i64, i64, i64, f32, f32, f32, u64, f32, vp, vp, vp]
        L.ko_set_voxel_graph.argtypes = [vp]
        L.ko_set_voxel_graph.restype = None
        L.ko_pdrf.argtypes = [vp, vp, i64, f32, C.c_int, f32, f32, vp]
        L.ko_target_order.argtypes = [vp, vp, i64, vp]
        L.ko_target_order.restype = i64
        L.ko_railroad.argtypes = [vp, i64, i64, i64, u64, vp, vp, vp, vp]
        L.ko_parental_field.argtypes = [vp, i64, i64, i64, u64, vp, vp]
        L.ko_path_from_parents.argtypes = [vp, i64, u64, vp, vp]
        L.ko_field_distances.argtypes = [vp, i64, i64, i64, u64, vp]
        L.ko_path_to_source.argtypes = [vp, vp, i64, i64, i64, u64, u64, vp, vp]
        L.ko_invalidate_ball.argtypes = [vp, i64, i64, i64, f32, f32, f32, vp, vp, i64, vp, vp]
        L.ko_invalidate_ball_graph.argtypes = [vp, i64, i64, i64, f32, f32, f32, vp, vp, i64, vp, vp, vp]
        L.ko_ball_radii.argtypes = [vp, vp, i64, f32, f32, vp]
        L.ko_invalidate_cube.argtypes = [vp, vp, i64, i64, i64, f32, f32, f32, vp, i64, f32, f32, vp]
        L.ko_zero2inf.argtypes = [vp, i64]
        L.ko_inf2zero.argtypes = [vp, i64]
        L.ko_first_label.argtypes = [vp, i64]
        L.ko_first_label.restype = i64
        L.ko_ccl26.argtypes = [vp, C.c_int, i64, i64, i64, vp]
        L.ko_ccl26.restype = i64
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


_ERR = {1: "invalid argument", 2: "out of memory", 3: "no path / no rail reachable",
        4: "float-absorption plateau (not restated)"}


def _check(rc):
    if rc != 0:
        raise OracleError(_ERR.get(rc, "error %d" % rc))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(a, dtype=None):
    """3D Fortran-contiguous view/copy."""
    a = np.asarray(a)
    while a.ndim < 3:
        a = a[..., np.newaxis]
    if dtype is not None and a.dtype != dtype:
        a = a.astype(dtype, order="F")
    return np.asfortranarray(a)


def loc_of(pt, shape):
    return int(pt[0]) + shape[0] * (int(pt[1]) + shape[1] * int(pt[2]))


def pt_of(loc, shape):
    loc = int(loc)
    sx, sy = shape[0], shape[1]
    return (loc % sx, (loc // sx) % sy, loc // (sx * sy))


def locs_to_pts(locs, shape):
    locs = np.asarray(locs, dtype=np.int64)
    sx, sy = shape[0], shape[1]
    return np.stack([locs % sx, (locs // sx) % sy, locs // (sx * sy)], axis=1)


def weights26(anisotropy):
    w = np.zeros(26, dtype=np.float32)
    lib().ko_weights26(float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]), _p(w))
    return w


def edt(labels, anisotropy=(1, 1, 1), black_border=False):
    """edt.edt(labels, anisotropy=, black_border=) -> float32 F-ordered (intake.py:178-183)."""
    labels = np.asarray(labels)
    nd = labels.ndim
    if labels.dtype == bool:
        labels = labels.view(np.uint8)
    if labels.dtype.itemsize not in (1, 2, 4, 8) or labels.dtype.kind not in "ui":
        labels = labels.astype(np.uint32)
    lab = _f3(labels)
    an = list(np.asarray(anisotropy, dtype=np.float32)) + [np.float32(1)] * (3 - len(anisotropy))
    out = np.zeros(lab.shape, dtype=np.float32, order="F")
    _check(lib().ko_edt_nd(_p(lab), lab.dtype.itemsize, max(1, min(nd, 3)), lab.shape[0], lab.shape[1], lab.shape[2],
                           an[0], an[1], an[2], int(bool(black_border)), _p(out)))
    return out.reshape(labels.shape, order="F") if nd < 3 else out


_GRAPH_BIT = (1, 0, 3, 2, 5, 4, 9, 7, 8, 6, 17, 13, 16, 12, 15, 11, 14, 10, 25, 24, 23, 21, 22, 20, 19, 18)   # ko_graph_bit
_DIRS = ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1), (-1, -1, 0), (-1, 1, 0), (1, -1, 0), (1, 1, 0),
         (0, -1, -1), (0, -1, 1), (0, 1, -1), (0, 1, 1), (-1, 0, -1), (-1, 0, 1), (1, 0, -1), (1, 0, 1), (-1, -1, -1), (1, -1, -1),
         (-1, 1, -1), (-1, -1, 1), (1, 1, -1), (1, -1, 1), (-1, 1, 1), (1, 1, 1))                                 # dijkstra_invalidation.hpp:60-124


def edt_graph(labels, graph, anisotropy=(1, 1, 1), black_border=False):
    """edt.edt(labels, anisotropy=, black_border=, voxel_graph=) (kimimaro/intake.py:174-183, trace.py:112-117).  The `edt` package is
    absent from the reference tree -- PARITY UNPINNED by reference code.  Restated from the package's published method: the image
    is doubled along every axis (voxel (x, y, z) at cell (2x, 2y, 2z)); the cell between a voxel and its +x / +y / +z neighbour is
    foreground iff the voxel is and bit 0 / 2 / 4 of its graph word (cc3d's layout) is set; the remaining cells of a voxel's
    2 x 2 x 2 block follow the voxel; a binary transform with HALF the pitch is sampled at the voxel cells.  Without a black border
    the cells behind the last voxel of an axis follow the voxel (no wall at the end of the array, like at its start)."""
    lab = _f3(np.asarray(labels))
    g = _f3(np.asarray(graph)).astype(np.uint32)
    fg = lab != 0
    sx, sy, sz = lab.shape
    cells = np.zeros((2 * sx, 2 * sy, 2 * sz), dtype=np.uint8, order="F")
    px, py, pz = fg & ((g & 1) != 0), fg & ((g & 4) != 0), fg & ((g & 16) != 0)
    if not black_border:
        px[-1, :, :] = fg[-1, :, :]
        py[:, -1, :] = fg[:, -1, :]
        pz[:, :, -1] = fg[:, :, -1]
    cells[0::2, 0::2, 0::2] = fg
    cells[1::2, 0::2, 0::2] = px
    cells[0::2, 1::2, 0::2] = py
    cells[0::2, 0::2, 1::2] = pz
    for sl in ((1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)):
        cells[sl[0]::2, sl[1]::2, sl[2]::2] = fg
    an = [np.float32(a) / np.float32(2) for a in (list(anisotropy) + [1.0] * (3 - len(anisotropy)))]
    fine = edt(cells, an, black_border)
    return np.asfortranarray(fine[0::2, 0::2, 0::2])


def color_connectivity_graph(labels, graph):
    """cc3d.color_connectivity_graph(voxel_graph, connectivity=26) followed by `cc_labels *= all_labels > 0` (kimimaro/utility.py:73-75;
    cc3d is absent from the reference tree -- PARITY UNPINNED): two foreground voxels are joined iff they are 26-neighbours and the
    graph word of the later one in the raster allows the step to the earlier one.  Ids 1..N by first appearance in the F-order raster."""
    import scipy.sparse
    import scipy.sparse.csgraph
    lab = _f3(np.asarray(labels))
    g = _f3(np.asarray(graph)).astype(np.uint32)
    fg = lab != 0
    sx, sy, sz = lab.shape
    idx = np.arange(lab.size, dtype=np.int64).reshape(lab.shape, order="F")
    rows, cols = [], []
    for k, (dx, dy, dz) in enumerate(_DIRS):
        if not (dz < 0 or (dz == 0 and (dy < 0 or (dy == 0 and dx < 0)))):
            continue
        src = tuple(slice(max(0, -d), n - max(0, d)) for d, n in zip((dx, dy, dz), (sx, sy, sz)))      # voxels that have the neighbour
        dst = tuple(slice(max(0, d), n - max(0, -d)) for d, n in zip((dx, dy, dz), (sx, sy, sz)))
        ok = fg[src] & fg[dst] & (((g[src] >> _GRAPH_BIT[k]) & 1) != 0)
        rows.append(idx[src][ok])
        cols.append(idx[dst][ok])
    rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    adj = scipy.sparse.coo_matrix((np.ones(rows.size, np.uint8), (rows, cols)), shape=(lab.size, lab.size))
    _, comp = scipy.sparse.csgraph.connected_components(adj, directed=False)
    flat_fg = fg.ravel(order="F")
    comp = comp[flat_fg]
    _, first, inv = np.unique(comp, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))             # component -> rank of its first appearance
    out = np.zeros(lab.size, dtype=np.uint32)
    out[flat_fg] = order[inv] + 1
    return out.reshape(lab.shape, order="F"), int(first.size)


@contextlib.contextmanager
def voxel_graph(graph):
    """voxel_graph= of the dijkstra3d calls (kimimaro/trace.py:139-145,155,240-242): inside the block every search of this module
    (euclidean_distance_field, railroad, field_distances / parental_field and the predecessor walks) steps from a voxel only in the
    directions its word allows (cc3d's bit layout, the word of the voxel being expanded -- ko_edge in kimi_oracle.c).  None = no
    graph.  dijkstra3d's source is absent: PARITY UNPINNED by reference code."""
    if graph is None:
        yield
        return
    g = _f3(graph, np.uint32)
    lib().ko_set_voxel_graph(_p(g))
    try:
        yield
    finally:
        lib().ko_set_voxel_graph(None)


def euclidean_distance_field(mask, source, anisotropy=(1, 1, 1), free_space_radius=0):
    """dijkstra3d.euclidean_distance_field(..., return_max_location=True) (trace.py:139-145)."""
    m = _f3(mask, np.uint8)
    out = np.empty(m.shape, dtype=np.float32, order="F")
    ml = C.c_uint64(0)
    mv = C.c_float(0)
    _check(lib().ko_edf(_p(m), m.shape[0], m.shape[1], m.shape[2],
                        float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]),
                        loc_of(source, m.shape), np.float32(free_space_radius), _p(out), C.byref(ml), C.byref(mv)))
    return out, pt_of(ml.value, m.shape)


def compute_pdrf(dbf_max, pdrf_scale, pdrf_exponent, DBF, DAF, max_daf):
    """kimimaro/trace.py:315-356.  DAF is mutated like in the reference."""
    f = np.float32
    M = f(1 / (f(dbf_max) ** 1.01))  # numpy scalar arithmetic, as trace.py:336
    assert DBF.flags.f_contiguous and DAF.flags.f_contiguous
    e = pdrf_exponent
    if not (int(e) == e and int(e) > 0 and (int(e) & (int(e) - 1)) == 0 and e < 2 ** 16):
        # trace.py:346-347, the np.power branch.  numpy's float32 power is the host's (glibc powf or a SIMD kernel,
        # by CPU features) and differs from C's powf in the last bit on ~10 % of the elements, so the restatement is
        # numpy's own call, statement for statement (ko_pdrf's powf line is kept for reference only).
        with np.errstate(all="ignore"):
            out = np.empty(DBF.shape, dtype=np.float32, order="F")
            np.multiply(DBF, M, out=out)
            np.subtract(f(1), out, out=out)
            np.power(out, e, out=out)
            out *= f(pdrf_scale)
            if max_daf != 0:
                DAF *= (1 / f(max_daf))
                out += DAF
        return np.asfortranarray(out)
    out = np.empty(DBF.shape, dtype=np.float32, order="F")
    _check(lib().ko_pdrf(_p(DBF), _p(DAF), DBF.size, M, int(pdrf_exponent),
                         f(pdrf_scale), f(max_daf), _p(out)))
    return out


def target_order(mask, daf):
    m = _f3(mask, np.uint8)
    d = _f3(daf, np.float32)
    order = np.empty(int(np.count_nonzero(m)), dtype=np.uint32)
    n = lib().ko_target_order(_p(m), _p(d), m.size, _p(order))
    assert n == order.size
    return order


def railroad(field, target, return_stats=False):
    """dijkstra3d.railroad(field, target) -> (n,3) path, rail end first (trace.py:240-242)."""
    f = _f3(field, np.float32)
    d = np.empty(f.shape, dtype=np.float32, order="F")
    path = np.empty(f.size, dtype=np.uint64)
    n = C.c_int64(0)
    settled = C.c_int64(0)
    _check(lib().ko_railroad(_p(f), f.shape[0], f.shape[1], f.shape[2], loc_of(target, f.shape),
                             _p(d), _p(path), C.byref(n), C.byref(settled)))
    pts = locs_to_pts(path[: n.value], f.shape)
    if return_stats:
        return pts, settled.value
    return pts


def parental_field(field, source):
    f = _f3(field, np.float32)
    d = np.empty(f.shape, dtype=np.float32, order="F")
    parents = np.zeros(f.shape, dtype=np.uint32, order="F")
    _check(lib().ko_parental_field(_p(f), f.shape[0], f.shape[1], f.shape[2],
                                   loc_of(source, f.shape), _p(d), _p(parents)))
    return parents


def field_distances(field, source):
    """distances of dijkstra3d.parental_field's search (trace.py:155)."""
    f = _f3(field, np.float32)
    d = np.empty(f.shape, dtype=np.float32, order="F")
    _check(lib().ko_field_distances(_p(f), f.shape[0], f.shape[1], f.shape[2], loc_of(source, f.shape), _p(d)))
    return d


def path_to_source(field, dist, source, target):
    """dijkstra3d.path_from_parents (trace.py:244) as a predecessor walk on `dist`: source -> ... -> target."""
    f = _f3(field, np.float32)
    d = _f3(dist, np.float32)
    path = np.empty(f.size, dtype=np.uint64)
    n = C.c_int64(0)
    _check(lib().ko_path_to_source(_p(f), _p(d), f.shape[0], f.shape[1], f.shape[2], loc_of(source, f.shape),
                                   loc_of(target, f.shape), _p(path), C.byref(n)))
    return locs_to_pts(path[: n.value], f.shape)


def path_from_parents(parents, target):
    p = _f3(parents, np.uint32)
    path = np.empty(p.size, dtype=np.uint64)
    n = C.c_int64(0)
    _check(lib().ko_path_from_parents(_p(p), p.size, loc_of(target, p.shape), _p(path), C.byref(n)))
    return locs_to_pts(path[: n.value], p.shape)


def roll_invalidation_ball_inside_component(labels, DBF, scale, const, anisotropy, path,
                                            return_stats=False, voxel_connectivity_graph=None):
    """skeletontricks.pyx:373-418.  labels (uint8/bool, F order) is mutated in place.
    voxel_connectivity_graph: optional uint32 array of the labels' shape (F order), cc3d's bit layout."""
    assert labels.flags.f_contiguous and DBF.flags.f_contiguous
    lab = labels.view(np.uint8)
    path = np.asarray(path, dtype=np.int64).reshape(-1, 3)
    sx, sy, sz = lab.shape
    locs = (path[:, 0] + sx * (path[:, 1] + sy * path[:, 2])).astype(np.uint64)
    radii = np.empty(locs.size, dtype=np.float32)
    lib().ko_ball_radii(_p(DBF), _p(locs), locs.size, np.float32(scale), np.float32(const), _p(radii))
    cnt = C.c_int64(0)
    ops = C.c_int64(0)
    if voxel_connectivity_graph is not None:
        vcg = np.asfortranarray(voxel_connectivity_graph, dtype=np.uint32)
        assert vcg.shape == lab.shape
        _check(lib().ko_invalidate_ball_graph(_p(lab), sx, sy, sz, float(anisotropy[0]), float(anisotropy[1]),
                                              float(anisotropy[2]), _p(locs), _p(radii), locs.size, _p(vcg),
                                              C.byref(cnt), C.byref(ops)))
    else:
        _check(lib().ko_invalidate_ball(_p(lab), sx, sy, sz, float(anisotropy[0]), float(anisotropy[1]),
                                        float(anisotropy[2]), _p(locs), _p(radii), locs.size,
                                        C.byref(cnt), C.byref(ops)))
    if return_stats:
        return cnt.value, labels, ops.value
    return cnt.value, labels


def roll_invalidation_cube(labels, DBF, path, scale, const, anisotropy=(1, 1, 1)):
    """skeletontricks.pyx:766-836 (F-ordered labels only in the oracle)."""
    assert labels.flags.f_contiguous
    DBF = np.asfortranarray(DBF, dtype=np.float32)
    lab = labels.view(np.uint8)
    path = np.asarray(path, dtype=np.int64).reshape(-1, 3)
    sx, sy, sz = lab.shape
    locs = (path[:, 0] + sx * (path[:, 1] + sy * path[:, 2])).astype(np.uint64)
    cnt = C.c_int64(0)
    _check(lib().ko_invalidate_cube(_p(lab), _p(DBF), sx, sy, sz, float(anisotropy[0]),
                                    float(anisotropy[1]), float(anisotropy[2]), _p(locs), locs.size,
                                    np.float32(scale), np.float32(const), C.byref(cnt)))
    return cnt.value, labels


def zero2inf(f):
    lib().ko_zero2inf(_p(f), f.size)
    return f


def inf2zero(f):
    lib().ko_inf2zero(_p(f), f.size)
    return f


def first_label(mask):
    m = _f3(mask, np.uint8)
    i = lib().ko_first_label(_p(m), m.size)
    return None if i < 0 else pt_of(i, m.shape)


def connected_components(labels):
    """26-connected multi-label CCL (cc3d.connected_components, utility.py:74-77)."""
    lab = _f3(labels)
    if lab.dtype == bool:
        lab = lab.view(np.uint8)
    out = np.zeros(lab.shape, dtype=np.uint32, order="F")
    n = lib().ko_ccl26(_p(lab), lab.dtype.itemsize, lab.shape[0], lab.shape[1], lab.shape[2], _p(out))
    if n < 0:
        raise OracleError("ccl: out of memory")
    return out, int(n)
