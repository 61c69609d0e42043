"""kimimaro_amd.trace.trace -- mirror of kimimaro.trace.trace (kimimaro/trace.py:36-194) for ONE
binary object, running on the MI355X.  Used by the parity tests: same arguments, same return value
(a Skeleton in voxel space with radii and the anisotropy transform)."""
from __future__ import annotations

import numpy as np

from .engine import Engine, NONE32
from .intake import TRACE_DEFAULTS, paths_of
from .skeleton import Skeleton


def trace(labels, DBF, scale=10, const=10, anisotropy=(1, 1, 1),
          soma_detection_threshold=1100, soma_acceptance_threshold=4000,
          pdrf_scale=5000, pdrf_exponent=16, soma_invalidation_scale=0.5, soma_invalidation_const=0,
          fix_branching=True, manual_targets_before=None, manual_targets_after=None, root=None,
          max_paths=None, voxel_graph=None, return_paths=False, _engine=None, _return_raw=False):
    eng = _engine or Engine()
    labels = np.asarray(labels)
    while labels.ndim < 3:
        labels = labels[..., np.newaxis]
        DBF = np.asarray(DBF)[..., np.newaxis]
    shape = labels.shape
    cc = np.asfortranarray((labels != 0).astype(np.uint32))
    dbf = np.asfortranarray(DBF, dtype=np.float32)
    d_cc = eng.to_device(cc)
    d_dbf = eng.to_device(dbf)
    counts, dbf_max, first_index, xmin, xmax = eng.label_stats(d_cc, 4, d_dbf, shape, 1)
    if counts[1] == 0:
        return [] if return_paths else Skeleton()
    loc = lambda p: int(p[0]) + shape[0] * (int(p[1]) + shape[1] * int(p[2]))
    mtb = [loc(p) for p in (manual_targets_before or [])]
    mta = [loc(p) for p in (manual_targets_after or [])]
    dmax = np.float32(dbf_max[1])
    soma_mode = False
    d_graph = None
    if voxel_graph is not None:
        # trace.py:139-145,155,167,240-242,257: the graph goes to every dijkstra3d search and to the invalidation
        vg = np.asarray(voxel_graph)
        while vg.ndim < 3:
            vg = vg[..., np.newaxis]
        if tuple(vg.shape) != tuple(shape):
            raise ValueError("voxel_graph must have the shape of the labels")
        d_graph = eng.to_device(np.asfortranarray(vg.astype(np.uint32)))
    if dmax > soma_detection_threshold:  # kimimaro/trace.py:108-119
        # fill_voids.fill (kh_fill_voids, row f3) + crop re-EDT, both on the GPU
        d_filled, nfilled = eng.fill_voids((d_cc != 0).to(eng.torch.uint8), shape)
        if nfilled > 0:
            d_cc = d_filled.to(eng.torch.int32)
            cc = eng.to_host_volume(d_cc, shape)
            # kimimaro/trace.py:112-117 (with a graph: edt.edt(voxel_graph=), kh_edt_graph_*, PARITY UNPINNED)
            d_dbf = eng.edt(d_cc, 4, shape, anisotropy, bool(np.all(cc))) if d_graph is None else \
                eng.edt_graph(d_cc, 4, d_graph, shape, anisotropy, bool(np.all(cc)))
            dbf = d_dbf.cpu().numpy().reshape(shape, order="F")
            counts, dbf_max, first_index, xmin, xmax = eng.label_stats(d_cc, 4, d_dbf, shape, 1)
            dmax = np.float32(dbf_max[1])
        soma_mode = bool(dmax > soma_acceptance_threshold)
    r = NONE32 if root is None else loc(root)
    soma = None
    if soma_mode:  # trace.py:123-127,134
        import scipy.ndimage
        if root is not None:
            mtb.insert(0, loc(root))
        maxima = (dbf == dmax)  # find_soma_root, trace.py:269-289
        com = np.asarray(scipy.ndimage.center_of_mass(maxima), dtype=np.float32)
        coords = np.vstack(np.where(maxima)).T
        sroot = coords[np.argmin(np.sum((coords - com) ** 2, axis=1))].astype(np.uint32)
        r = loc(sroot)
        soma_radius = dmax * soma_invalidation_scale + soma_invalidation_const
        soma = {"soma_mode": [1], "fsr": [np.float32(dbf[tuple(sroot)])], "soma_radius": [np.float32(soma_radius)],
                "soma_scale": [np.float32(soma_invalidation_scale)], "soma_const": [np.float32(soma_invalidation_const)]}
    params = dict(TRACE_DEFAULTS)
    params.update(scale=scale, const=const, pdrf_scale=pdrf_scale, pdrf_exponent=pdrf_exponent)
    # the reference runs on the array it is given: the x faces of THAT array are where its neighbour enumeration degenerates
    # (dijkstra_invalidation.hpp:116-123), not the object's own extent (skeletonize crops every label to its box first)
    res = eng.run_labels(d_cc, 4, d_dbf, shape, anisotropy, 1, [1], counts[1:2], dbf_max[1:2], first_index[1:2],
                         [0], [shape[0] - 1], [r], [mtb], [mta], params, fix_branching=fix_branching, max_paths=max_paths,
                         return_fields=_return_raw, soma=soma, voxel_graph=d_graph)
    if _return_raw:
        return res
    paths = paths_of(res, 0, shape)
    if return_paths:
        return paths
    skel = Skeleton.simple_merge([Skeleton.from_path(p) for p in paths if len(p) > 0]).consolidate()
    verts = skel.vertices.flatten().astype(np.uint32)
    dz = dbf.copy(order="F")
    dz[dz == 0] = np.inf  # zero2inf, trace.py:138 (radii are read from the inf-patched DBF)
    skel.radii = dz[verts[::3], verts[1::3], verts[2::3]]
    skel.transform = np.array([[anisotropy[0], 0, 0, 0], [0, anisotropy[1], 0, 0], [0, 0, anisotropy[2], 0]],
                              dtype=np.float32)
    return skel


def point_to_point(binary_img, start, end, anisotropy=(1, 1, 1), pdrf_scale=100000, pdrf_exponent=4):
    """kimimaro.trace.point_to_point (kimimaro/trace.py:358-390): one centerline from `start` to `end` through a binary
    image -- EDT with black_border, DAF from `start`, PDRF, then the cheapest path end -> start over the PDRF
    (dijkstra3d.dijkstra); radii are the (zero2inf'ed) DBF at the vertices, as in the reference.  Every step is the HIP
    kernel behind the function-level mirror of the module the reference calls (kimimaro_amd.ops)."""
    from . import ops
    img = np.asfortranarray(binary_img)
    DBF = ops.edt(img, anisotropy=anisotropy, black_border=True)
    dbf_max = np.max(DBF)
    DBF = ops.zero2inf(np.asfortranarray(DBF))
    DAF, target = ops.euclidean_distance_field(img, start, anisotropy=anisotropy, return_max_location=True)
    DAF = ops.inf2zero(np.asfortranarray(DAF))
    PDRF = ops.compute_pdrf(dbf_max, pdrf_scale, pdrf_exponent, DBF, DAF, DAF[target])
    path = ops.dijkstra(PDRF, end, start)
    skel = Skeleton.from_path(path)
    verts = skel.vertices.flatten().astype(np.uint32)
    skel.radii = DBF[verts[::3], verts[1::3], verts[2::3]]
    return skel
