"""kimimaro_amd.skeletonize -- host-side mirror of kimimaro.skeletonize (kimimaro/intake.py:58-221)
driving the HIP kernels.  Same signature, same defaults, same return type ({label: Skeleton}).

What runs where
  host (numpy)         format_labels / object mask / early outs (intake.py:147-160), connected
                       components (row f1, host C++ for now), border targets (row f2), Skeleton assembly.
  MI355X (libkimi_hip) whole-volume EDT (intake.py:174-185), per-label statistics, and for every
                       connected component the complete TEASAR trace (kimimaro/trace.py:36-267):
                       find_root, DAF, PDRF, target finder, railroad, rolling invalidation.

There is no CPU fallback: without libkimi_hip.so + an MI355X this raises HipUnavailableError.
"""
from __future__ import annotations

import os

from collections import defaultdict

import numpy as np

from . import _abi
from .engine import Engine, NONE32
from .skeleton import Skeleton

DEFAULT_TEASAR_PARAMS = {  # kimimaro/intake.py:47-56
    "scale": 1.5,
    "const": 300,
    "pdrf_scale": 100000,
    "pdrf_exponent": 4,
    "soma_acceptance_threshold": 3500,
    "soma_detection_threshold": 750,
    "soma_invalidation_const": 300,
    "soma_invalidation_scale": 2,
}

# kimimaro/trace.py:38-43 -- the defaults trace() falls back to for keys missing from teasar_params
TRACE_DEFAULTS = {
    "scale": 10, "const": 10, "soma_detection_threshold": 1100, "soma_acceptance_threshold": 4000,
    "pdrf_scale": 5000, "pdrf_exponent": 16, "soma_invalidation_scale": 0.5, "soma_invalidation_const": 0,
    "max_paths": None,
}


class DimensionError(Exception):
    pass


def format_labels(labels, in_place=False):
    """The input as a Fortran-ordered array with exactly three axes, as kimimaro/intake.py:315-342 prepares it: bool
    volumes are reinterpreted as uint8, 1-D / 2-D inputs get trailing axes of extent 1, trailing singleton axes
    beyond the third are dropped, and a fourth non-trivial axis is a DimensionError (same message).  in_place avoids
    the copy when the array already is Fortran ordered."""
    vol = np.asfortranarray(labels) if in_place else np.array(labels, order="F", copy=True)
    if vol.dtype == np.bool_:
        vol = vol.view(np.uint8)
    given = vol.shape
    if vol.ndim > 3 and any(extent != 1 for extent in given[3:]):
        raise DimensionError(
            "Input labels may be no more than three non-trivial dimensions. Got: {}".format(given))
    return vol.reshape((given + (1, 1, 1))[:3], order="F")


def apply_object_mask(all_labels, object_ids):
    """kimimaro/intake.py:519-535."""
    if object_ids is None:
        return all_labels
    keep = np.isin(all_labels, np.asarray(list(object_ids), dtype=all_labels.dtype))
    all_labels[~keep] = 0
    return all_labels


def compute_cc_labels(all_labels):
    """kimimaro/utility.py:58-83 -> (cc_labels uint32 F-order, N, {cc id: original id}).
    Row f1: 26-connected multi-label CCL, host C++ helper in libkimi_hip.so for now."""
    lib = _abi.lib()
    lab = all_labels
    if lab.dtype.kind not in "ui" or lab.dtype.itemsize not in (1, 2, 4, 8):
        lab = lab.astype(np.uint64)
    lab = np.asfortranarray(lab)
    cc = np.zeros(lab.shape, dtype=np.uint32, order="F")
    import ctypes as C
    n = lib.kh_host_ccl26(lab.ctypes.data_as(C.c_void_p), lab.dtype.itemsize, lab.shape[0], lab.shape[1],
                          lab.shape[2], cc.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise MemoryError("kh_host_ccl26 failed")
    flat_cc = cc.reshape(-1, order="F")
    idx = np.flatnonzero(flat_cc)
    uniq, first_idx = np.unique(flat_cc[idx], return_index=True)
    orig = all_labels.reshape(-1, order="F")[idx[first_idx]]
    remap = {int(u): orig[i].item() for i, u in enumerate(uniq)}  # skeletontricks.get_mapping :490-525
    return cc, int(n), remap


def compute_cc_labels_device(eng, all_labels, d_graph=None):
    """kimimaro/utility.py:58-83 on the MI355X (kh_ccl26; with a voxel graph kh_ccl26_graph): (device cc volume, N, {cc id: original id})."""
    d_cc, n, rep = eng.ccl(all_labels, d_graph)
    orig = all_labels.reshape(-1, order="F")[rep[1:].astype(np.int64)] if n else []
    remap = {i + 1: orig[i].item() for i in range(n)}  # skeletontricks.get_mapping :490-525
    return d_cc, n, remap


class LazyVolume:
    """The component volume lives in HBM; the few host-side consumers (border faces, extra-target lookups,
    soma crops) pull what they need, the whole array only if a soma label asks for its crop."""

    def __init__(self, eng, d_cc, shape, host=None):
        self.eng, self.d, self.shape, self._host = eng, d_cc, tuple(shape), host

    def host(self):
        if self._host is None:
            self._host = self.eng.to_host_volume(self.d, self.shape)
        return self._host

    def faces(self):
        if self._host is not None:
            c = self._host
            return (c[:, :, 0], c[:, :, -1], c[:, 0, :], c[:, -1, :], c[0, :, :], c[-1, :, :])
        e, d, s = self.eng, self.d, self.shape
        return (e.face(d, s, 2, 0), e.face(d, s, 2, s[2] - 1), e.face(d, s, 1, 0), e.face(d, s, 1, s[1] - 1),
                e.face(d, s, 0, 0), e.face(d, s, 0, s[0] - 1))

    def release_device(self):
        """drop this object's reference to the u32 component volume in HBM (0.5 GB at 512^3): skeletonize_cc calls it once the u16
        copy serves every remaining sweep and no soma label will ask for a crop -- the volumes in flight are bounded by memory.
        Callers that want the memory back must not keep a reference of their own (pass the volume through this object only)."""
        self.d = None

    def __getitem__(self, pt):
        if self._host is not None:
            return self._host[pt]
        x, y, z = (int(v) for v in pt)
        return int(self.d[x + self.shape[0] * (y + self.shape[1] * z)].item()) & 0xFFFFFFFF


def _points_to_labels(pts, cc_labels):
    mapping = defaultdict(list)
    for pt in pts:
        pt = tuple(int(v) for v in pt)
        mapping[int(cc_labels[pt])].append(pt)
    return mapping


def _loc(pt, shape):
    return int(pt[0]) + shape[0] * (int(pt[1]) + shape[1] * int(pt[2]))


def skeletonize(all_labels, teasar_params=DEFAULT_TEASAR_PARAMS, anisotropy=(1, 1, 1),
                object_ids=None, dust_threshold=1000, progress=True, fix_branching=True,
                in_place=False, fix_borders=True, parallel=1, parallel_chunk_size=100,
                extra_targets_before=[], extra_targets_after=[], fill_holes=False,
                fix_avocados=False, voxel_graph=None, _timings=None, _engine=None):
    """Skeletonize all non-zero labels of a 2D/3D label image on the MI355X.

    Arguments and return value as kimimaro.skeletonize (kimimaro/intake.py:58-141).  `parallel` and
    `parallel_chunk_size` (process pool of the reference, intake.py:344-408) have no meaning here:
    one process drives one GPU; multi-GPU runs shard the connected components round robin over the
    ranks of torch.distributed (see kimimaro_amd.distributed).
    """
    eng = _engine or Engine()  # raises HipUnavailableError without a GPU: no CPU fallback
    anisotropy = np.array(anisotropy, dtype=np.float32)

    all_labels = format_labels(all_labels, in_place=in_place)
    all_labels = apply_object_mask(all_labels, object_ids)
    if all_labels.size <= dust_threshold:
        return {}
    minlabel, maxlabel = all_labels.min(), all_labels.max()
    if minlabel == 0 and maxlabel == 0:
        return {}

    d_graph = None
    if voxel_graph is not None:
        # kimimaro/intake.py:162,174-183,467: the graph decides the components (cc3d.color_connectivity_graph), puts walls into the
        # transform (edt.edt(voxel_graph=)) and goes to every search and to the invalidation of every label (trace(voxel_graph=)).
        # cc3d and edt are absent from the reference tree: kh_ccl26_graph / kh_edt_graph_* restate their published behaviour,
        # PARITY UNPINNED (include/kimi_hip.h).
        if fix_avocados:
            raise NotImplementedError("skeletonize(voxel_graph=, fix_avocados=True): the avocado pass re-labels components, which a "
                                      "graph of the ORIGINAL voxels does not describe")
        vg = np.asarray(voxel_graph)
        vg = vg.reshape((vg.shape + (1, 1, 1))[:3], order="F") if vg.ndim < 3 else vg
        if tuple(vg.shape) != tuple(all_labels.shape):
            raise ValueError("voxel_graph must have the shape of the labels")
        d_graph = eng.to_device(np.asfortranarray(vg.astype(np.uint32)))
    d_cc, nlabels, remapping = compute_cc_labels_device(eng, all_labels, d_graph)  # row f1 on the GPU
    if fill_holes:
        fill_all_holes_device(eng, d_cc, all_labels.shape, nlabels)              # intake.py:168-169
    avocado = None
    if fix_avocados:
        # kimimaro/intake.py:187-193 run BEFORE everything else that looks at the components: it edits them (and renumbers them)
        d_cc, nlabels, remapping, d_dbf_av = engage_avocado_protection_device(
            eng, d_cc, all_labels.shape, nlabels, remapping, anisotropy, bool(minlabel == maxlabel),
            teasar_params.get("soma_detection_threshold", 0))
        avocado = d_dbf_av
    cc = LazyVolume(eng, d_cc, all_labels.shape)
    del d_cc                   # (the volume is reached through `cc` from here on, which lets go of it as soon as it can)
    before = _points_to_labels(extra_targets_before, cc)
    after = _points_to_labels(extra_targets_after, cc)

    return skeletonize_cc(eng, cc, nlabels, remapping, teasar_params, anisotropy, dust_threshold,
                          fix_branching, fix_borders, before, after, black_border=(minlabel == maxlabel),
                          timings=_timings, d_dbf=avocado, d_graph=d_graph)


def _avocado_fruit_from_lines(xl, yl, zl, cx, cy, cz, background=0):
    """kimimaro.skeletontricks.find_avocado_fruit (skeletontricks.pyx:905-992) on the three axis-parallel lines of the label volume
    through (cx, cy, cz) (host copies): six rays from the voxel, each ends at the background or at the first other label, which it
    reports; the rays towards smaller coordinates stop BEFORE index 0 (`range(c, 0, -1)`).  Fewer than three reports: (label, label).
    The most frequent report -- the smallest label among equally frequent ones (np.unique order) -- is the fruit if at most one
    report disagrees with it (none when there are exactly three reports)."""
    label = int(xl[cx])
    rays = (xl[cx:], xl[cx:0:-1], yl[cy:], yl[cy:0:-1], zl[cz:], zl[cz:0:-1])
    changes = []
    for ray in rays:
        stop = np.flatnonzero((ray == background) | (ray != label))
        if stop.size and int(ray[stop[0]]) != background:
            changes.append(int(ray[stop[0]]))
    if len(changes) < 3:
        return label, label
    uniq, cts = np.unique(changes, return_counts=True)
    k = int(np.argmax(cts))
    if len(changes) - int(cts[k]) > (1 if len(changes) > 3 else 0):
        return label, label
    return label, int(uniq[k])


def engage_avocado_protection_device(eng, d_cc, shape, nlabels, remapping, anisotropy, black_border, soma_detection_threshold):
    """kimimaro/intake.py:600-704 (fix_avocados=True) on the component volume resident in HBM: a nucleus that carries a label of its
    own inside its cell ("pit" in "fruit") is merged into the cell, holes filled, up to 20 passes for nested ones; then the
    components are renumbered (fastremap.renumber: by first appearance) and mapped back to the original labels
    (skeletontricks.get_mapping, skeletontricks.pyx:490-525).  Device work: the transforms (kh_edt), the bounding boxes
    (kh_label_stats), the 2-D fills of the six faces of a crop and its 3-D fill (kh_fill_voids_nd); selections, arg-max and the
    relabelling are device-side tensor plumbing; the host reads three lines of labels per candidate (find_avocado_fruit's rays)
    and keeps the reference's sets -- INCLUDING their iteration order: the candidates of a pass are a Python set built from the
    sorted unique labels, and the reference edits the volume in that set's order.
    Returns (component volume u32 on the device, number of components, {component: original label}, its EDT)."""
    t = eng.torch
    sx, sy, sz = (int(v) for v in shape)
    nvox = sx * sy * sz
    d_cc = d_cc.clone()                      # (the caller's volume may be shared)
    orig = d_cc.clone()
    v = d_cc.view(sz, sy, sx)                # torch C order (z, y, x) == Fortran order (x, y, z)
    eng._narrow = None                       # the u16 copy kh_ccl26 made no longer matches what is edited here
    d_dbf = eng.edt(d_cc, 4, shape, anisotropy, black_border)
    thr = float(np.float32(soma_detection_threshold / 2.5))     # numpy compares the f32 field with the scalar in float32
    unchanged = set()
    for _ in range(20):
        vals = t.unique(d_cc[d_dbf > thr]).cpu().numpy().view(np.uint32)       # sorted, like fastremap.unique
        candidates = set([0] + [int(x) for x in vals]) if bool((d_dbf <= thr).any()) else set(int(x) for x in vals)
        candidates -= unchanged
        candidates.discard(0)
        order = [label for label in candidates if label != 0]
        changed, unchanged_now = set(), set()          # (the sets of ONE pass, intake.py:650-651)
        if order:
            counts, _, _, xmin, xmax = eng.label_stats(d_cc, 4, d_dbf, shape, nlabels)
            yz = eng.last_yz_extent
            dv = d_dbf.view(sz, sy, sx)
            for label in order:
                lo = (int(xmin[label]), int(yz[label, 0]), int(yz[label, 2]))
                hi = (int(xmax[label]) + 1, int(yz[label, 1]) + 1, int(yz[label, 3]) + 1)
                sub = v[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
                binimg = (sub == label)
                # paint_walls (:655-666): a 2-D fill on each of the six faces, in the reference's order
                for face in ((-1, 0), (-1, -1), (1, 0), (1, -1), (2, 0), (2, -1)):      # (torch axis, index): z, z, y, y, x, x
                    ax = 0 if face[0] == -1 else face[0]
                    plane = binimg.select(ax, face[1])
                    p2 = plane.to(t.uint8).contiguous()
                    filled, nfill = eng.fill_voids(p2.view(-1), (p2.shape[1], p2.shape[0], 1), ndim=2)
                    if nfill:
                        plane.copy_(filled.view(p2.shape).bool())
                prod = binimg * dv[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
                k = int(t.argmax(prod.reshape(-1)).item())            # first maximum in the raster (x fastest), like argmax(arr.T)
                nx, ny = hi[0] - lo[0], hi[1] - lo[1]
                cx, cy, cz = lo[0] + k % nx, lo[1] + (k // nx) % ny, lo[2] + k // (nx * ny)
                lines = [a.cpu().numpy().view(np.uint32) for a in (v[cz, cy, :], v[cz, :, cx].contiguous(), v[:, cy, cx].contiguous())]
                pit, fruit = _avocado_fruit_from_lines(lines[0], lines[1], lines[2], cx, cy, cz)
                if pit == fruit and pit not in changed:
                    unchanged_now.add(pit)
                else:
                    unchanged_now.discard(pit)
                    unchanged_now.discard(fruit)
                    changed.add(pit)
                    changed.add(fruit)
                    binimg |= (sub == fruit)
                cshape = (hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2])
                filled, _ = eng.fill_voids(binimg.to(t.uint8).contiguous().view(-1), cshape)
                fb = filled.view(cshape[2], cshape[1], cshape[0]).bool()
                sub[fb] = fruit                               # cc_labels[slc] *= ~binimg; cc_labels[slc] += fruit * binimg
        unchanged |= unchanged_now
        if len(changed) == 0:
            break
        d_dbf = eng.edt(d_cc, 4, shape, anisotropy, black_border)
    # fastremap.renumber: 1..N by first appearance in memory; get_mapping: the LAST run start of a component in the raster names it
    flat = d_cc.to(t.int64) & 0xFFFFFFFF
    idx = t.arange(nvox, device=eng.device, dtype=t.int64)
    top = int(flat.max().item()) + 1
    first = t.full((top,), nvox, dtype=t.int64, device=eng.device).scatter_reduce(0, flat, idx, reduce="amin")
    present = t.nonzero(first < nvox).reshape(-1)
    present = present[present != 0]
    by_first = present[t.argsort(first[present], stable=True)]
    lut = t.zeros(top, dtype=t.int64, device=eng.device)
    lut[by_first] = t.arange(1, by_first.numel() + 1, device=eng.device, dtype=t.int64)
    new = lut[flat]
    starts = t.ones(nvox, dtype=t.bool, device=eng.device)
    starts[1:] = new[1:] != new[:-1]
    sidx = t.nonzero(starts).reshape(-1)
    n_new = int(by_first.numel())
    last = t.full((n_new + 1,), -1, dtype=t.int64, device=eng.device).scatter_reduce(0, new[sidx], sidx, reduce="amax")
    last_h = last.cpu().numpy()
    src = (orig.to(t.int64) & 0xFFFFFFFF)[last.clamp(min=0)].cpu().numpy()
    adjusted = {}
    for new_cc in range(0, n_new + 1):
        if last_h[new_cc] >= 0 and int(src[new_cc]) in remapping:
            adjusted[new_cc] = remapping[int(src[new_cc])]
    return new.to(t.int32), n_new, adjusted, d_dbf       # (renumbering does not move a voxel: the last transform stands)


def fill_all_holes_device(eng, d_cc, shape, nlabels):
    """kimimaro/intake.py:747-795 on the component volume resident in HBM (modified in place): the holes of every
    component are filled with kh_fill_voids on the component's bounding box, in ascending label order; a component
    that gets swallowed is not processed itself any more.  Bounding boxes are those before any filling, as in the
    reference (find_objects is called once, intake.py:767).  Returns the number of voxels filled."""
    eng._narrow = None     # the component volume is edited in place below: its u16 copy no longer matches
    t = eng.torch
    nvox = int(shape[0]) * int(shape[1]) * int(shape[2])
    counts, _, _, xmin, xmax = eng.label_stats(d_cc, 4, t.zeros(nvox, dtype=t.float32, device=eng.device), shape, nlabels)
    yz = eng.last_yz_extent
    in_set = np.ones(nlabels + 1, dtype=bool)
    in_set[0] = False
    v = d_cc.view(shape[2], shape[1], shape[0])      # torch C order (z, y, x) == F order (x, y, z)
    filled_total = 0
    for label in range(1, nlabels + 1):
        if not in_set[label] or counts[label] == 0:
            continue
        lo = (int(xmin[label]), int(yz[label, 0]), int(yz[label, 2]))
        hi = (int(xmax[label]) + 1, int(yz[label, 1]) + 1, int(yz[label, 3]) + 1)
        cshape = (hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2])
        sub = v[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
        d_filled, n = eng.fill_voids((sub == label).to(t.uint8).contiguous().view(-1), cshape)
        if n == 0:
            continue
        filled_total += n
        fb = d_filled.view(cshape[2], cshape[1], cshape[0]).bool()
        for other in t.unique(sub[fb]).cpu().numpy():
            if other != label and other > 0:
                in_set[int(other)] = False
        sub[fb] = label
    return filled_total


def shard_components(cc_segids, counts, rank, world):
    """The components rank `rank` of `world` traces.  The reference deals them round robin (kimimaro/intake.py:388-389:
    cc_segids[i::parallel]); the cost of a component grows with its voxel count and the tail of the size distribution is
    heavy, so here they are dealt largest first to the rank with the least voxels so far (LPT; ties -> lowest rank).
    Deterministic: every rank computes the same assignment from the same counts."""
    order = sorted(cc_segids, key=lambda s: (-int(counts[s]), s))
    load = [0] * world
    mine = []
    for s in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += int(counts[s])
        if r == rank:
            mine.append(s)
    return sorted(mine)


def skeletonize_cc(eng, cc_labels, nlabels, remapping, teasar_params, anisotropy, dust_threshold,
                   fix_branching, fix_borders, before, after, black_border, timings=None,
                   rank=0, world=1, d_cc=None, d_dbf=None, d_graph=None):
    """Everything after the connected components (intake.py:174-221 + skeletonize_subset :434-517)."""
    try:
        return _skeletonize_cc(eng, cc_labels, nlabels, remapping, teasar_params, anisotropy, dust_threshold, fix_branching,
                               fix_borders, before, after, black_border, timings, rank, world, d_cc, d_dbf, d_graph)
    finally:
        eng._narrow = None      # the u16 copy of this volume's ids (2 B / voxel of HBM) is not kept alive past the call


def _skeletonize_cc(eng, cc_labels, nlabels, remapping, teasar_params, anisotropy, dust_threshold,
                    fix_branching, fix_borders, before, after, black_border, timings, rank, world, d_cc, d_dbf=None, d_graph=None):
    import time as _time

    def _mark(name):
        if timings is not None:
            eng.sync_stream()
            timings.append((name, _time.perf_counter()))

    _mark("start")
    if not isinstance(cc_labels, LazyVolume):
        cc_labels = LazyVolume(eng, d_cc, cc_labels.shape, host=cc_labels)
    shape = cc_labels.shape
    label_bytes = 4
    if d_cc is None:
        d_cc = cc_labels.d
    if d_cc is None:
        d_cc = eng.to_device(cc_labels.host())
        cc_labels.d = d_cc
    d_lab, label_bytes = eng.narrow(d_cc)          # u16 ids when there are < 65536 components (utility.py:79 refit)
    if d_dbf is None and d_graph is not None:
        d_dbf = eng.edt_graph(d_lab, label_bytes, d_graph, shape, anisotropy, black_border)  # intake.py:174-185 with voxel_graph
    if d_dbf is None:         # (fix_avocados hands over the transform of the components it left behind)
        d_dbf = eng.edt(d_lab, label_bytes, shape, anisotropy, black_border)  # intake.py:174-185
    counts, dbf_max, first_index, xmin, xmax = eng.label_stats(d_lab, label_bytes, d_dbf, shape, nlabels)
    yz = eng.last_yz_extent
    bbox = lambda sid: ((int(xmin[sid]), int(yz[sid, 0]), int(yz[sid, 2])),
                        (int(xmax[sid]) + 1, int(yz[sid, 1]) + 1, int(yz[sid, 3]) + 1))  # find_objects, utility.py:85-102
    _mark("edt+stats")

    # intake.py:198-201
    cc_segids = [sid for sid in range(1, nlabels + 1) if counts[sid] > dust_threshold]
    border_targets = defaultdict(list)
    if fix_borders:
        from .border import compute_border_targets
        border_targets = compute_border_targets(None, anisotropy, eng=eng, faces=cc_labels.faces(), shape=shape)  # intake.py:207

    _mark("border_targets")
    params = dict(TRACE_DEFAULTS)
    params.update(teasar_params)
    if world > 1:
        cc_segids = shard_components(cc_segids, counts, rank, world)

    soma_jobs = []
    segids, roots, tb, ta = [], [], [], []
    for segid in cc_segids:
        # intake.py:454-456: bounding boxes of volume <= 1 are skipped (never true above dust_threshold >= 1)
        if counts[segid] <= 1 and dust_threshold < 1:
            continue
        mtb, mta, root = [], [], NONE32
        if len(border_targets[segid]) > 0:                      # intake.py:486-488
            mtb = [_loc(p, shape) for p in border_targets[segid]]
            root = mtb.pop()
        if segid in before and len(before[segid]) > 0:
            mtb.extend(_loc(p, shape) for p in before[segid])
        if segid in after and len(after[segid]) > 0:
            mta.extend(_loc(p, shape) for p in after[segid])
        if dbf_max[segid] > params["soma_detection_threshold"] and _needs_soma_path(
                eng, d_cc, shape, bbox(segid), segid, float(dbf_max[segid]), params):
            soma_jobs.append((segid, root, mtb, mta))  # traced one by one on their crop, below
            continue
        segids.append(segid)
        roots.append(root)
        tb.append(mtb)
        ta.append(mta)

    if not soma_jobs and label_bytes == 2 and d_lab is not d_cc and os.environ.get("KH_KEEP_CC", "0") != "1":
        # nothing reads the u32 ids any more (the faces are taken, no soma crop will be asked for): the u16 copy serves from here on
        d_cc = None
        cc_labels.release_device()
        if getattr(eng, "_narrow", None) is not None:
            eng._narrow = (None, eng._narrow[1])
    sel = np.asarray(segids, dtype=np.int64)
    asm = Assembler(shape, anisotropy, remapping)
    eng.run_labels(d_lab, label_bytes, d_dbf, shape, anisotropy, nlabels, sel, counts[sel] if len(sel) else [],
                         dbf_max[sel] if len(sel) else [], first_index[sel] if len(sel) else [],
                         xmin[sel] if len(sel) else [], xmax[sel] if len(sel) else [], roots, tb, ta, params,
                         fix_branching=fix_branching, max_paths=params.get("max_paths"), timings=timings, consume=asm.add,
                         voxel_graph=d_graph)
    out = asm.finish()
    _mark("assemble")
    if soma_jobs:
        _trace_soma_labels(eng, soma_jobs, d_cc, d_dbf, shape, anisotropy, remapping, params, fix_branching, bbox, out, d_graph)
        _mark("soma_labels")
    return out


def _trace_soma_labels(eng, jobs, d_cc, d_dbf, shape, anisotropy, remapping, params, fix_branching, bbox, out, d_graph=None):
    """Labels that enter the soma branch of kimimaro/trace.py:108-134 (internal voids to fill, or DBF max above
    soma_acceptance_threshold) leave the shared-volume batch -- filling voids changes which voxels belong to
    the label -- and are traced on their bounding-box crop, exactly like intake.py:450-517.  The reference gives every
    label a process of its pool (intake.py:344-408); here several somas of one volume run side by side, each on a host
    thread with an Engine and a HIP stream of its own (`Engine.soma_lanes`, kimimaro_amd.lanes.Lanes); the skeletons are
    merged in the order of the labels either way."""
    from .trace import trace as trace_one
    sx, sy = shape[0], shape[1]
    an = np.asarray(anisotropy, dtype=np.float32)
    unloc = lambda l: (l % sx, (l // sx) % sy, l // (sx * sy))
    kw = {k: params[k] for k in ("scale", "const", "pdrf_scale", "pdrf_exponent", "soma_detection_threshold",
                                 "soma_acceptance_threshold", "soma_invalidation_scale", "soma_invalidation_const")}

    def one(e, i):
        segid, root, mtb, mta = jobs[i]
        lo, hi = bbox(segid)
        minpt = np.array(lo, dtype=np.int64)
        labels = e.crop(d_cc, shape, lo, hi) == segid
        dbf = np.where(labels, e.crop(d_dbf, shape, lo, hi, np.float32), 0.0).astype(np.float32)
        tr = lambda ls: [tuple(int(v) for v in (np.array(unloc(l)) - minpt)) for l in ls]
        skel = trace_one(labels, dbf, anisotropy=an, fix_branching=fix_branching, manual_targets_before=tr(mtb),
                         manual_targets_after=tr(mta), root=(None if root == NONE32 else tr([root])[0]),
                         max_paths=params.get("max_paths"), _engine=e,
                         voxel_graph=(None if d_graph is None else e.crop(d_graph, shape, lo, hi)), **kw)    # intake.py:467
        if not skel.empty():
            skel.vertices += minpt.astype(skel.vertices.dtype)
        return skel

    width = min(int(getattr(eng, "soma_lanes", 1)), len(jobs))
    if width > 1:
        eng.sync_stream()              # the component volume and its EDT are complete before another stream reads them
        results = (skel for _, skel in eng.soma_lane_pool(width).run(one, len(jobs), width=width))
    else:
        results = (one(eng, i) for i in range(len(jobs)))
    for (segid, _, _, _), skel in zip(jobs, results):
        if skel.empty():
            continue
        orig = remapping[segid]
        skel.id = orig
        skel.vertices = np.multiply(skel.vertices, an, dtype=np.float32)
        skel.space = "physical"
        out[orig] = Skeleton.simple_merge([out[orig], skel]).consolidate() if orig in out else skel.consolidate()


def _needs_soma_path(eng, d_cc, shape, box, segid, dbf_max, params):
    """kimimaro/trace.py:108-119 for a label whose DBF max exceeds soma_detection_threshold: does it take the
    soma branch?  True if the DBF max is already above soma_acceptance_threshold, or if the label has
    internal voids (fill_voids.fill would change it and its DBF; kh_fill_voids on the label's crop).  A label
    without voids below the acceptance threshold continues unchanged in the reference, so it stays in the
    shared-volume batch."""
    if dbf_max > params["soma_acceptance_threshold"]:
        return True
    t = eng.torch
    lo, hi = box
    cshape = (hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2])
    v = d_cc.view(shape[2], shape[1], shape[0])[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
    d_mask = (v == int(segid)).to(t.uint8).contiguous().view(-1)   # torch C order (z, y, x) == F order (x, y, z)
    _, nfilled = eng.fill_voids(d_mask, cshape)
    return nfilled > 0


def paths_of(res, slot, shape):
    """list of (n,3) integer voxel paths of task `slot`."""
    sx, sy = shape[0], shape[1]
    v0, l0, l1 = res["voff"][slot], res["loff"][slot], res["loff"][slot + 1]
    out = []
    pos = v0
    for n in res["lens"][l0:l1]:
        locs = res["verts"][pos:pos + n].astype(np.int64)
        out.append(np.stack([locs % sx, (locs // sx) % sy, locs // (sx * sy)], axis=1))
        pos += n
    return out


def consolidate_paths(locs, lens, radii, shape):
    """Skeleton.from_path per path + simple_merge + consolidate (kimimaro/trace.py:182-184) for one label, on
    linear voxel indices: returns (vertices (n,3) f32 sorted lexicographically by (x,y,z) like
    np.unique(axis=0), edges (m,2) u32 sorted/unique without self loops, radii of the first occurrences).
    Same result as kimimaro_amd.skeleton.Skeleton.consolidate (vertices no edge refers to dropped), ~10x cheaper
    (1-D unique on a key)."""
    sx, sy, sz = shape
    x, y, z = locs % sx, (locs // sx) % sy, locs // (sx * sy)
    key = (x * sy + y) * sz + z                      # row-lexicographic order of (x, y, z)
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)
    n = locs.size
    starts = np.cumsum(lens)[:-1]
    eidx = np.arange(n - 1)
    if starts.size:
        keep = np.ones(n - 1, dtype=bool)
        keep[starts - 1] = False                     # no edge across two paths
        eidx = eidx[keep]
    a, b = inv[eidx], inv[eidx + 1]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    ok = lo != hi
    ekey = np.unique(lo[ok] * np.int64(ukey.size) + hi[ok])
    edges = np.stack([ekey // ukey.size, ekey % ukey.size], axis=1)
    used = np.zeros(ukey.size, dtype=bool)
    used[edges.ravel()] = True
    if not used.all():      # a one-vertex path off every other path: no edge refers to it (consolidate drops it)
        first, edges = first[used], (np.cumsum(used) - 1)[edges]
    verts = np.stack([x[first], y[first], z[first]], axis=1).astype(np.float32)
    return verts, edges.astype(np.uint32), radii[first]


def _ranges(starts, counts):
    """concatenation of starts[i] + arange(counts[i]) over i, without a Python loop"""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    before = np.cumsum(counts) - counts
    return np.repeat(np.asarray(starts, dtype=np.int64) - before, counts) + np.arange(total, dtype=np.int64)


def consolidate_paths_flat(res, shape):
    """consolidate_paths for EVERY label of a result group in ONE native call outside the interpreter
    (kh_host_consolidate_paths, include/kimi_hip.h: with twenty volumes in flight the lanes reach this point together and what
    holds the interpreter lock is paid twenty times in a row).  Returns None for a group without vertices, else the slots' arrays
    back to back: verts (N,3) f32, radii (N) f32, edges (M,2) u32 with indices local to the slot, vstart / estart [nslots+1].
    Same arrays as consolidate_paths_flat_numpy (tests/test_host.py compares them)."""
    import ctypes as C
    from . import _abi
    sx, sy, sz = shape
    voff = np.ascontiguousarray(res["voff"], dtype=np.int64)
    loff = np.ascontiguousarray(res["loff"], dtype=np.int64)
    nslots = voff.size - 1
    locs = np.ascontiguousarray(res["verts"], dtype=np.uint32)
    n = int(locs.size)
    if n == 0:
        return None
    lens = np.ascontiguousarray(res["lens"], dtype=np.uint32)
    radii = np.ascontiguousarray(res["radii"], dtype=np.float32)
    oV, oR, oE = np.empty((n, 3), np.float32), np.empty(n, np.float32), np.empty((n, 2), np.uint32)
    vstart, estart = np.empty(nslots + 1, np.int64), np.empty(nslots + 1, np.int64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    got = _abi.lib().kh_host_consolidate_paths(nslots, P(voff), P(loff), P(locs), P(lens), P(radii), int(sx), int(sy), int(sz),
                                               P(oV), P(oR), P(oE), P(vstart), P(estart))
    if got < 0:
        raise MemoryError("kh_host_consolidate_paths failed")
    return {"verts": oV[:got], "radii": oR[:got], "edges": oE[:int(estart[-1])], "vstart": vstart, "estart": estart, "voff": voff}


def consolidate_paths_flat_numpy(res, shape):
    """the numpy form of consolidate_paths_flat (one sort over all path vertices instead of three np.unique calls per label); kept
    as the statement the native call is tested against.  consolidate_paths for EVERY label of a result group in one go (one sort over all path vertices instead of three
    np.unique calls per label: the per-label numpy overhead, 170 us x 3.4 k labels, was most of the assembly time of a
    512^3 volume).  Returns None for a group without vertices, else the slots' arrays back to back:
    verts (N,3) f32, radii (N) f32, edges (M,2) u32 with indices local to the slot, vstart / estart [nslots+1]."""
    sx, sy, sz = shape
    voff = np.asarray(res["voff"], dtype=np.int64)
    nslots = voff.size - 1
    locs = res["verts"].astype(np.int64)
    n = locs.size
    if n == 0:
        return None
    V = np.int64(sx) * sy * sz
    slot_of = np.repeat(np.arange(nslots, dtype=np.int64), np.diff(voff))
    x, y, z = locs % sx, (locs // sx) % sy, locs // (sx * sy)
    key = slot_of * V + (x * sy + y) * sz + z                 # slot, then row-lexicographic order of (x, y, z)
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)
    nu = ukey.size
    uslot = ukey // V
    ustart = np.searchsorted(uslot, np.arange(nslots + 1, dtype=np.int64))     # unique vertices of slot s: [ustart[s], ustart[s+1])
    # consecutive pairs inside a path are edges: drop the pair that straddles two paths (path ends, incl. label ends)
    lens = res["lens"].astype(np.int64)
    path_end = np.cumsum(lens) - 1
    keep = np.ones(max(n - 1, 0), dtype=bool)
    keep[path_end[path_end < n - 1]] = False
    eidx = np.flatnonzero(keep)
    a, b = inv[eidx], inv[eidx + 1]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    ok = lo != hi
    ekey = np.unique(lo[ok] * np.int64(nu) + hi[ok])           # sorted by (lo, hi): grouped by slot, rows sorted like np.unique(axis=0)
    elo, ehi = ekey // nu, ekey % nu
    used = np.zeros(nu, dtype=bool)
    used[elo] = True
    used[ehi] = True
    # vertices no edge refers to are dropped (Skeleton.consolidate); local index = rank among the slot's used vertices
    cum = np.concatenate([[0], np.cumsum(used)])
    rank = cum[1:] - 1
    base = cum[ustart[:-1]] if nu else np.zeros(nslots, np.int64)
    fu = first[used]
    verts_all = np.stack([x[fu], y[fu], z[fu]], axis=1).astype(np.float32)
    radii_all = res["radii"][fu]
    vstart = cum[ustart]                                       # used vertices of slot s: [vstart[s], vstart[s+1])
    eslot = uslot[elo]
    estart = np.searchsorted(eslot, np.arange(nslots + 1, dtype=np.int64))
    edges_all = np.stack([rank[elo] - base[eslot], rank[ehi] - base[eslot]], axis=1).astype(np.uint32)
    return {"verts": verts_all, "radii": radii_all, "edges": edges_all, "vstart": vstart, "estart": estart, "voff": voff}


def consolidate_paths_batch(res, shape):
    """consolidate_paths_flat slot by slot: yields (slot, vertices (n,3) f32, edges (m,2) u32, radii) for the slots that have
    vertices, the same arrays the per-label function returns."""
    f = consolidate_paths_flat(res, shape)
    if f is None:
        return
    vstart, estart, voff = f["vstart"], f["estart"], f["voff"]
    for s in range(voff.size - 1):
        if voff[s + 1] == voff[s]:
            continue
        yield s, f["verts"][vstart[s]:vstart[s + 1]], f["edges"][estart[s]:estart[s + 1]], f["radii"][vstart[s]:vstart[s + 1]]


class Assembler:
    """Skeleton assembly: kimimaro/trace.py:182-192 + intake.py:506-517, 587-593.  Results arrive in groups of
    labels (Engine.run_labels hands them over as the groups finish on the GPU): `add` consolidates the paths of a group's
    components, `finish` merges the components of every original label.  Nothing here loops over components in Python: with
    twenty volumes in flight the lanes reach this point together and every millisecond of interpreter time is paid twenty times
    in a row (0.8 s of a 7.8 s round before this form)."""

    def __init__(self, shape, anisotropy, remapping):
        self.shape = shape
        self.remapping = remapping
        self.an = np.asarray(anisotropy, dtype=np.float32)
        an = self.an
        self.transform = np.array([[an[0], 0, 0, 0], [0, an[1], 0, 0], [0, 0, an[2], 0]], dtype=np.float32)
        self.skeletons = defaultdict(list)       # original label -> [(component id, verts, edges, radii)]: per-component hand-over (tests)
        self.groups = []                         # (component ids of the group's slots, consolidate_paths_flat of the group)

    def add(self, res):
        flat = consolidate_paths_flat(res, self.shape)
        if flat is not None:
            self.groups.append((np.asarray(res["tasks"]["segid"], dtype=np.int64), flat))

    def _parts(self):
        """the components that have edges (Skeleton.empty() ones are dropped, intake.py:506), in arrival order, as arrays: component
        id, original label (as a code into `labels`), and where their vertices / edges lie in the concatenated arrays"""
        seg, v0, nv, e0, ne, Vs, Rs, Es = [], [], [], [], [], [], [], []
        vbase = ebase = 0
        for segids, f in self.groups:
            cnt_e = np.diff(f["estart"])
            keep = np.flatnonzero(cnt_e > 0)
            seg.append(segids[keep])
            v0.append(f["vstart"][keep] + vbase)
            nv.append(np.diff(f["vstart"])[keep])
            e0.append(f["estart"][keep] + ebase)
            ne.append(cnt_e[keep])
            Vs.append(f["verts"]); Rs.append(f["radii"]); Es.append(f["edges"])
            vbase += f["verts"].shape[0]
            ebase += f["edges"].shape[0]
        for orig, parts in self.skeletons.items():            # (hand-over per component: the same arrays, one part at a time)
            for comp, verts, edges, radii in parts:
                if edges.shape[0] == 0:
                    continue
                seg.append(np.array([-1 - len(self._extra)], dtype=np.int64))
                self._extra.append((orig, comp))
                v0.append(np.array([vbase])); nv.append(np.array([verts.shape[0]]))
                e0.append(np.array([ebase])); ne.append(np.array([edges.shape[0]]))
                Vs.append(np.asarray(verts, dtype=np.float32)); Rs.append(np.asarray(radii, dtype=np.float32))
                Es.append(np.asarray(edges, dtype=np.uint32))
                vbase += verts.shape[0]
                ebase += edges.shape[0]
        if not seg:
            return None
        cat = lambda xs, dt: np.concatenate(xs).astype(dt, copy=False)
        return (cat(seg, np.int64), cat(v0, np.int64), cat(nv, np.int64), cat(e0, np.int64), cat(ne, np.int64),
                np.concatenate(Vs), np.concatenate(Rs), np.concatenate(Es))

    def finish(self):
        """one Skeleton per original label, in the order in which the labels' first components arrived.  The components of a
        label are disjoint voxel sets, so Skeleton.simple_merge(...).consolidate() (intake.py:587-593) is a concatenation
        re-sorted lexicographically by vertex: done on integer keys for all labels in ONE native call outside the interpreter
        (kh_host_merge_components; same result as np.unique(vertices, axis=0) + edge remap per label)."""
        import ctypes as C
        from . import _abi
        sx, sy, sz = self.shape
        self._extra = []
        got = self._parts()
        if got is None:
            return {}
        seg, v0, nv, e0, ne, Vall, Rall, Eall = got
        # original label of every part, as a code; the dict of component ids is read once, not once per component
        keys = np.fromiter(self.remapping.keys(), dtype=np.int64, count=len(self.remapping)) if len(self.remapping) else np.zeros(0, np.int64)
        vals = list(self.remapping.values())
        comp_of = seg.copy()
        label_objs = []
        code_of_obj = {}
        if keys.size:
            ks = np.argsort(keys, kind="stable")
            pos = np.searchsorted(keys[ks], np.maximum(seg, 0))
            pos = np.minimum(pos, keys.size - 1)
            idx_in_vals = ks[pos]
        else:
            idx_in_vals = np.zeros(seg.size, dtype=np.int64)
        # code per distinct original label VALUE (several component ids map to one label)
        val_code = np.empty(len(vals), dtype=np.int64)
        for i, v in enumerate(vals):
            c = code_of_obj.get(v)
            if c is None:
                c = code_of_obj[v] = len(label_objs)
                label_objs.append(v)
            val_code[i] = c
        code = val_code[idx_in_vals] if len(vals) else np.zeros(seg.size, dtype=np.int64)
        for j in np.flatnonzero(seg < 0):                       # per-component hand-over: (label, component id) given directly
            orig, comp = self._extra[-1 - int(seg[j])]
            c = code_of_obj.get(orig)
            if c is None:
                c = code_of_obj[orig] = len(label_objs)
                label_objs.append(orig)
            code[j] = c
            comp_of[j] = comp
        ucode, first_idx = np.unique(code, return_index=True)
        label_order = ucode[np.argsort(first_idx, kind="stable")]                # labels by first arrival
        rank_of_code = np.empty(len(label_objs), dtype=np.int64)
        rank_of_code[label_order] = np.arange(label_order.size)
        order = np.lexsort((comp_of, rank_of_code[code]))                        # label by label, components by id (intake.py:444)
        nvs, nes = nv[order], ne[order]
        V = np.ascontiguousarray(Vall[_ranges(v0[order], nvs)], dtype=np.float32)
        R = np.ascontiguousarray(Rall[_ranges(v0[order], nvs)], dtype=np.float32)
        E = np.ascontiguousarray(Eall[_ranges(e0[order], nes)], dtype=np.uint32)
        vstart = np.concatenate([[0], np.cumsum(nvs)]).astype(np.int64)
        estart = np.concatenate([[0], np.cumsum(nes)]).astype(np.int64)
        pol = np.concatenate([[0], np.cumsum(np.bincount(rank_of_code[code], minlength=label_order.size))]).astype(np.int64)
        oV, oR, oE = np.empty_like(V), np.empty_like(R), np.empty_like(E)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        if _abi.lib().kh_host_merge_components(int(label_order.size), P(pol), P(vstart), P(estart), P(V), P(R), P(E), int(sy), int(sz),
                                               np.float32(self.an[0]), np.float32(self.an[1]), np.float32(self.an[2]),
                                               P(oV), P(oR), P(oE)) != 0:
            raise MemoryError("kh_host_merge_components failed")
        merged = {}
        va, ea = vstart[pol].tolist(), estart[pol].tolist()       # first vertex / edge of every label
        wrap, tf = Skeleton.wrap, self.transform
        for li, c in enumerate(label_order.tolist()):
            # copies: the public arrays own their memory (a kept Skeleton does not pin the volume's buffers)
            a, b, e0_, e1_ = va[li], va[li + 1], ea[li], ea[li + 1]
            merged[label_objs[c]] = wrap(oV[a:b].copy(), oE[e0_:e1_].copy(), oR[a:b].copy(), label_objs[c], tf.copy(), "physical")
        return merged


def assemble(res, shape, anisotropy, remapping):
    asm = Assembler(shape, anisotropy, remapping)
    asm.add(res)
    return asm.finish()
