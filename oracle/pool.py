"""oracle.pool -- oracle.pipeline.skeletonize with the per-component work spread over a forked process pool.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Same results as oracle.pipeline.skeletonize(labels, ...) for the options it takes; the one difference in the
arithmetic is the DBF of a component, which is computed on the component's bounding box grown by one voxel instead
of on the whole volume -- the distance of a voxel to the nearest voxel of ANOTHER label (or background) is decided
inside that box, so the values are identical (black_border=False; single-label volumes are not taken here).
Used by the full-size parity test of the bench workload (tests/test_gpu_c3.py): 3.4 k components in ~30 s on the
GPU box's host cores.
"""
from __future__ import annotations

import os
from collections import defaultdict

import numpy as np
import scipy.ndimage

import oracle as K
from oracle import pipeline as P
from oracle import border as _border

_G = {}


def _one(segid):
    cc, an, params, fix_branching, slices, targets = (_G[k] for k in ("cc", "an", "params", "fb", "slices", "targets"))
    slc = slices[segid - 1]
    lo = np.array([s.start for s in slc], dtype=np.int64)
    glo = [max(0, s.start - 1) for s in slc]
    ghi = [min(n, s.stop + 1) for s, n in zip(slc, cc.shape)]
    grown = np.asfortranarray(cc[tuple(slice(a, b) for a, b in zip(glo, ghi))])
    # DBF on the box grown by one voxel, then cut back to the bounding box itself: the reference traces on exactly
    # that crop (intake.py:450-467), and the crop's x extent is visible in the result (dijkstra_invalidation.hpp:116-123)
    inner = tuple(slice(int(s.start - g), int(s.stop - g)) for s, g in zip(slc, glo))
    dbf_g = K.edt(grown, an, black_border=False)
    labels = np.asfortranarray(grown[inner] == segid)
    dbf = np.asfortranarray(np.where(labels, dbf_g[inner], 0.0).astype(np.float32))
    mtb, root = [], None
    if len(targets.get(segid, ())) > 0:                           # intake.py:486-488
        mtb = [tuple(int(v) for v in (np.asarray(p, dtype=np.int64) - lo)) for p in targets[segid]]
        root = mtb.pop()
    skel = P.trace(labels, dbf, anisotropy=an, fix_branching=fix_branching, manual_targets_before=mtb, root=root, **params)
    if skel.empty():
        return segid, None
    verts = skel.vertices + lo.astype(np.float32)
    return segid, (verts, skel.edges, skel.radii)


def skeletonize_pool(all_labels, teasar_params=P.DEFAULT_TEASAR_PARAMS, anisotropy=(1, 1, 1), dust_threshold=1000,
                     fix_branching=True, fix_borders=True, workers=None, only=None):
    """kimimaro/intake.py:58-221 (serial path) with the loop of skeletonize_subset (:434-517) on a pool.
    only: optional iterable of connected-component ids to restrict the work to (a sample)."""
    import multiprocessing as mp
    from oracle.skeleton import Skeleton
    an = np.array(anisotropy, dtype=np.float32)
    all_labels = P.format_labels(all_labels)
    cc, remapping = P.compute_cc_labels(all_labels)
    counts = np.bincount(cc.ravel(order="K"))
    segids = [i for i in range(1, counts.size) if counts[i] > dust_threshold]
    if only is not None:
        keep = set(int(v) for v in only)
        segids = [s for s in segids if s in keep]
    targets = {}
    if fix_borders:
        bt = _border.compute_border_targets(cc, an, K.edt, K.connected_components)
        targets = {int(k): v for k, v in bt.items()}
    slices = [(s and s[::-1]) for s in scipy.ndimage.find_objects(cc.T)]
    _G.update(cc=cc, an=an, params=dict(teasar_params), fb=fix_branching, slices=slices, targets=targets)
    workers = workers or (os.cpu_count() or 1)
    order = sorted(segids, key=lambda s: -counts[s])           # big components first
    K.lib()                                                      # build / load the shared object before forking
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            results = list(pool.imap_unordered(_one, order, chunksize=1))
    else:
        results = [_one(s) for s in order]
    per_label = defaultdict(list)
    transform = np.array([[an[0], 0, 0, 0], [0, an[1], 0, 0], [0, 0, an[2], 0]], dtype=np.float32)
    for segid, res in sorted(results, key=lambda r: r[0]):       # component order of intake.py:444
        if res is None:
            continue
        verts, edges, radii = res
        orig = remapping[segid]
        per_label[orig].append(Skeleton(np.multiply(verts, an, dtype=np.float32), edges, radii, orig, transform, "physical"))
    return {k: Skeleton.simple_merge(v).consolidate() for k, v in per_label.items()}, cc, counts
