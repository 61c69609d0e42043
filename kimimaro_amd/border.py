"""fix_borders: kimimaro/intake.py:544-585 compute_border_targets (row f2 of SURVEY.md section 8).

For each of the six faces of the volume: 2-D connected components of the face, 2-D EDT with
black_border=True, and per component the pixel of maximum distance (ties resolved by the
coordinate-frame-free rules of skeletontricks.compute_tiebreaker_maxima).  These become forced
targets; the last one of each label is its root (intake.py:486-488), so that skeletons of adjacent
chunks meet at the same face voxel.

Host side (faces are 2-D, ~1e5 pixels): numpy + two host helpers of libkimi_hip.so (CCL and
find_border_targets restated in C++); the 2-D EDT runs on the MI355X (kh_edt_nd, ndim = 2).
The sets of the reference are kept as Python sets of int tuples, inserted in the same order, so that
CPython's iteration order -- which decides the root (intake.py:583,488) -- is reproduced.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import numpy as np

from . import _abi


def _ccl2d(plane):
    lib = _abi.lib()
    p = np.asfortranarray(plane)
    if p.dtype.itemsize not in (1, 2, 4, 8) or p.dtype.kind not in "ui":
        p = p.astype(np.uint64)
    out = np.zeros(p.shape, dtype=np.uint32, order="F")
    n = lib.kh_host_ccl26(p.ctypes.data_as(C.c_void_p), p.dtype.itemsize, p.shape[0], p.shape[1], 1,
                          out.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise MemoryError("kh_host_ccl26 failed")
    return out, int(n)


def find_border_targets(dt, cc, wx, wy, nlab):
    """skeletontricks.find_border_targets (skeletontricks.pyx:591-647) -> ordered {cc id: (x, y)}."""
    lib = _abi.lib()
    dt = np.asfortranarray(dt, dtype=np.float32)
    cc = np.asfortranarray(cc, dtype=np.uint32)
    xy = np.zeros(2 * (nlab + 1), dtype=np.float32)
    order = np.zeros(nlab + 1, dtype=np.int32)
    n = lib.kh_host_find_border_targets(dt.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p),
                                        dt.shape[0], dt.shape[1], np.float32(wx), np.float32(wy), nlab,
                                        xy.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise MemoryError("kh_host_find_border_targets failed")
    return {int(l): (xy[2 * l], xy[2 * l + 1]) for l in order[:n]}


def compute_border_targets(cc_labels, anisotropy, eng=None, edt2d=None, faces=None, shape=None):
    """kimimaro/intake.py:544-585.  `eng`: Engine (2-D EDT on the GPU); `edt2d`: alternative callable
    edt(labels2d, anisotropy, black_border) used by the oracle pipeline.  `faces`: the six faces
    [z=0, z=-1, y=0, y=-1, x=0, x=-1] when the component volume lives on the device (cc_labels None)."""
    sx, sy, sz = cc_labels.shape if cc_labels is not None else shape
    if faces is None:
        faces = (cc_labels[:, :, 0], cc_labels[:, :, -1], cc_labels[:, 0, :], cc_labels[:, -1, :],
                 cc_labels[0, :, :], cc_labels[-1, :, :])
    # the six faces in the order of intake.py:551-558 (z = 0, z = max, y = 0, y = max, x = 0, x = max); a face pixel
    # (p, q) lies on the two axes that span the face, the third coordinate is the face's own
    extent = (sx, sy, sz)
    planes = []
    for fixed_axis, span in ((2, (0, 1)), (1, (0, 2)), (0, (1, 2))):
        for side in (0, 1):
            at = 0 if side == 0 else extent[fixed_axis] - 1

            def place(p, q, fixed_axis=fixed_axis, span=span, at=at):
                pt = [0, 0, 0]
                pt[span[0]], pt[span[1]], pt[fixed_axis] = p, q, at
                return tuple(pt)
            planes.append((faces[len(planes)], span, place))
    target_list = defaultdict(set)
    for plane, dims, rotatefn in planes:
        wx, wy = anisotropy[dims[0]], anisotropy[dims[1]]
        plane = np.copy(plane, order="F")
        cc_plane, n = _ccl2d(plane)
        if n == 0:
            continue
        if edt2d is not None:
            dt_plane = edt2d(cc_plane, (wx, wy), True)
        else:
            d = eng.to_device(cc_plane)
            dt_plane = eng.edt(d, 4, (cc_plane.shape[0], cc_plane.shape[1], 1), (wx, wy, 1.0), True, ndim=2)
            dt_plane = dt_plane.cpu().numpy().reshape(cc_plane.shape, order="F")
        plane_targets = find_border_targets(dt_plane, cc_plane, wx, wy, n)
        # get_mapping(plane, cc_plane): cc id -> label of the 3-D component on this face
        lab_of = np.zeros(n + 1, dtype=plane.dtype)       # (every pixel of a component holds the same label: any write wins)
        lab_of[cc_plane.reshape(-1, order="F")] = plane.reshape(-1, order="F")
        for label, pt in plane_targets.items():
            target_list[int(lab_of[label])].add(rotatefn(int(pt[0]), int(pt[1])))
    none = np.array([], np.uint32)      # (shared: a label without targets is looked up per component, thousands of times per volume)
    none.setflags(write=False)
    out = defaultdict(lambda: none)
    for label, pts in target_list.items():
        out[label] = np.array(list(pts), dtype=np.uint32)
    return out
