"""oracle.pipeline -- CPU restatement of kimimaro.trace.trace / kimimaro.skeletonize
on top of oracle/libkimi_oracle.so.   TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows kimimaro/trace.py:36-267 and kimimaro/intake.py:58-221,434-517 line by
line (same control flow, same per-label crop + mask, same LIFO target stacks),
so that the HIP product path -- which works on whole-volume arrays and batches
labels -- is checked against an independently structured implementation.

Not restated (rows f2/f3 of SURVEY.md section 8 are handled where noted):
The Skeleton operations and the border targets are restated in oracle/skeleton.py and oracle/border.py (nothing here
imports the product).  soma mode is restated with a fill_voids stand-in (scipy.ndimage.binary_fill_holes) and a documented
guess of dijkstra3d's free_space_radius (source absent); fill_holes is restated with the same stand-in;
voxel_graph raises
NotImplementedError.
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np
import scipy.ndimage

import oracle as K
from oracle.skeleton import Skeleton
from oracle import border as _border

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


class DimensionError(Exception):
    pass


def trace(labels, DBF, scale=10, const=10, anisotropy=(1, 1, 1),
          soma_detection_threshold=1100, soma_acceptance_threshold=4000,
          pdrf_scale=5000, pdrf_exponent=16,
          soma_invalidation_scale=0.5, soma_invalidation_const=0,
          fix_branching=True, manual_targets_before=None, manual_targets_after=None,
          root=None, max_paths=None, voxel_graph=None, stats=None, return_paths=False, _vcg=None):
    """kimimaro/trace.py:36-194.  voxel_graph (uint32 per voxel, cc3d's bit layout): handed to every dijkstra3d search and to the
    invalidation as the reference does (trace.py:139-145,155,167,240-242,257) and to the soma branch's re-EDT (K.edt_graph)."""
    if voxel_graph is not None:
        voxel_graph = np.asfortranarray(voxel_graph, dtype=np.uint32)
        with K.voxel_graph(voxel_graph):
            return trace(labels, DBF, scale, const, anisotropy, soma_detection_threshold, soma_acceptance_threshold, pdrf_scale,
                         pdrf_exponent, soma_invalidation_scale, soma_invalidation_const, fix_branching, manual_targets_before,
                         manual_targets_after, root, max_paths, None, stats, return_paths, _vcg=voxel_graph)
    manual_targets_before = list(manual_targets_before or [])
    manual_targets_after = list(manual_targets_after or [])
    dbf_max = np.max(DBF)
    labels = np.asfortranarray(labels)
    if labels.dtype == bool:
        labels = labels.view(np.uint8)
    labels = np.asfortranarray(labels, dtype=np.uint8).copy(order="F")
    DBF = np.asfortranarray(DBF, dtype=np.float32).copy(order="F")

    if dbf_max > soma_detection_threshold:  # trace.py:108-119
        filled = scipy.ndimage.binary_fill_holes(labels)  # stand-in for fill_voids.fill (6-connected background)
        num_voxels_filled = int(np.count_nonzero(filled)) - int(np.count_nonzero(labels))
        if num_voxels_filled > 0:
            labels = np.asfortranarray(filled.astype(np.uint8))
            DBF = K.edt(labels, anisotropy, black_border=bool(np.all(labels))) if _vcg is None else \
                K.edt_graph(labels, _vcg, anisotropy, black_border=bool(np.all(labels)))       # trace.py:112-117
        dbf_max = np.max(DBF)
        soma_mode = bool(dbf_max > soma_acceptance_threshold)
    else:
        soma_mode = False

    soma_radius = 0.0
    if soma_mode:  # trace.py:123-127
        if root is not None:
            manual_targets_before.insert(0, tuple(int(v) for v in root))
        root = find_soma_root(DBF, dbf_max)
        soma_radius = dbf_max * soma_invalidation_scale + soma_invalidation_const
    elif root is None:
        root = find_root(labels, anisotropy)
    if root is None:
        return Skeleton() if not return_paths else []
    root = tuple(int(v) for v in root)

    free_space_radius = 0 if not soma_mode else DBF[root]  # trace.py:134
    DBF = K.zero2inf(DBF)  # trace.py:138
    DAF, target = K.euclidean_distance_field(labels, root, anisotropy, free_space_radius)  # :139-145
    DAF = K.inf2zero(DAF)  # :146
    order = K.target_order(labels, DAF)  # CachedTargetFinder.__init__ :147
    finder = _TargetFinder(order, labels.shape)
    PDRF = K.compute_pdrf(dbf_max, pdrf_scale, pdrf_exponent, DBF, DAF, DAF[target])  # :148

    if not fix_branching:
        parents = (PDRF, K.field_distances(PDRF, root))  # :155 (the parental field is its distance field here)
    else:
        parents = PDRF

    if soma_mode:  # trace.py:160-168
        _, labels = K.roll_invalidation_ball_inside_component(
            labels, DBF, soma_invalidation_scale, soma_invalidation_const, anisotropy, [root], voxel_connectivity_graph=_vcg)
    elif len(manual_targets_before) == 0:  # :171-172
        manual_targets_before.append(target)

    paths = compute_paths(root, labels, DBF, finder, parents, scale, const, anisotropy,
                          fix_branching, manual_targets_before, manual_targets_after,
                          max_paths, stats, soma_mode=soma_mode, soma_radius=soma_radius, vcg=_vcg)
    if return_paths:
        return paths

    skel = Skeleton.simple_merge([Skeleton.from_path(p) for p in paths if len(p) > 0]).consolidate()
    verts = skel.vertices.flatten().astype(np.uint32)
    skel.radii = DBF[verts[::3], verts[1::3], verts[2::3]]
    skel.transform = np.array([[anisotropy[0], 0, 0, 0], [0, anisotropy[1], 0, 0],
                               [0, 0, anisotropy[2], 0]], dtype=np.float32)
    return skel


class _TargetFinder:
    """CachedTargetFinder.find_target, skeletontricks.pyx:1008-1045."""

    def __init__(self, order, shape):
        self.order = order
        self.head = 0
        self.shape = shape

    def find_target(self, labels):
        flat = labels.ravel(order="F")
        o = self.order
        h = self.head
        n = o.size
        # chunked scan for the first index whose mask byte is still set
        while h < n:
            chunk = o[h:h + 4096]
            hit = np.flatnonzero(flat[chunk])
            if hit.size:
                h += int(hit[0])
                self.head = h
                return K.pt_of(o[h], self.shape)
            h += chunk.size
        self.head = n
        return None


def find_soma_root(DBF, dbf_max):
    """kimimaro/trace.py:269-289."""
    maxima = (DBF == dbf_max)
    com = np.asarray(scipy.ndimage.center_of_mass(maxima), dtype=np.float32)
    coords = np.vstack(np.where(maxima)).T
    root = np.argmin(np.sum((coords - com) ** 2, axis=1))
    return tuple(int(v) for v in coords[root].astype(np.uint32))


def compute_paths(root, labels, DBF, finder, parents, scale, const, anisotropy,
                  fix_branching, manual_targets_before, manual_targets_after, max_paths, stats=None,
                  soma_mode=False, soma_radius=0.0, vcg=None):
    """kimimaro/trace.py:196-267."""
    paths = []
    valid_labels = int(np.count_nonzero(labels))
    root = tuple(root)
    if max_paths is None:
        max_paths = valid_labels
    if len(manual_targets_before) + len(manual_targets_after) >= max_paths:
        return []
    if fix_branching:
        parents[root] = 0  # initial rail, :220
    while (valid_labels > 0 or manual_targets_before or manual_targets_after) and len(paths) < max_paths:
        if manual_targets_before:
            target = manual_targets_before.pop()
        elif valid_labels == 0:
            target = manual_targets_after.pop()
        else:
            target = finder.find_target(labels)
        target = tuple(int(v) for v in target)
        if fix_branching:
            path, settled = K.railroad(parents, target, return_stats=True)
        else:
            path = K.path_to_source(parents[0], parents[1], root, target)  # :244
            settled = 0
        if soma_mode:  # trace.py:246-251
            dist_to_soma_root = np.linalg.norm(np.asarray(anisotropy, dtype=np.float32) * (np.asarray(path) - np.asarray(root)), axis=1)
            path = np.concatenate((path[:1, :], path[dist_to_soma_root > soma_radius, :]))
        ops = 0
        if valid_labels > 0:
            invalidated, labels, ops = K.roll_invalidation_ball_inside_component(
                labels, DBF, scale, const, anisotropy, path, return_stats=True, voxel_connectivity_graph=vcg)
            valid_labels -= invalidated
        if fix_branching:
            parents[path[:, 0], path[:, 1], path[:, 2]] = 0.0  # :261-263
        if stats is not None:
            stats["paths"] = stats.get("paths", 0) + 1
            stats["settled"] = stats.get("settled", 0) + int(settled)
            stats["heap_pushes"] = stats.get("heap_pushes", 0) + int(ops)
            stats["path_vertices"] = stats.get("path_vertices", 0) + int(len(path))
        paths.append(path)
    return paths


def find_root(labels, anisotropy):
    """kimimaro/trace.py:291-308."""
    any_voxel = K.first_label(labels)
    if any_voxel is None:
        return None
    _, target = K.euclidean_distance_field(labels, any_voxel, anisotropy)
    return target


# ---------------------------------------------------------------------------
# intake.py restatement


def format_labels(labels):
    """kimimaro/intake.py:315-342: a Fortran-ordered copy with exactly three axes (missing axes are appended with
    extent 1, surplus trailing axes of extent 1 are dropped, anything else is an error); bool becomes uint8."""
    lab = np.array(labels, order="F")
    if lab.dtype == bool:
        lab = lab.view(np.uint8)
    given = lab.shape
    if lab.ndim < 3:
        lab = lab.reshape(given + (1,) * (3 - lab.ndim), order="F")
    elif lab.ndim > 3:
        if any(n != 1 for n in given[3:]):
            raise DimensionError(
                "Input labels may be no more than three non-trivial dimensions. Got: {}".format(given))
        lab = lab.reshape(given[:3], order="F")
    return lab


def compute_cc_labels(all_labels, voxel_graph=None):
    """kimimaro/utility.py:58-83: returns (cc_labels, {cc id: original id})."""
    cc, n = K.connected_components(all_labels) if voxel_graph is None else K.color_connectivity_graph(all_labels, voxel_graph)
    first = np.zeros(n + 1, dtype=np.int64)
    flat_cc = cc.ravel(order="F")
    idx = np.flatnonzero(flat_cc)
    # first occurrence of each component -> original label there
    uniq, first_idx = np.unique(flat_cc[idx], return_index=True)
    orig = all_labels.ravel(order="F")[idx[first_idx]]
    remap = {int(u): orig[i].item() for i, u in enumerate(uniq)}
    return cc, remap


def find_objects(cc_labels):
    """kimimaro/utility.py:85-102."""
    all_slices = scipy.ndimage.find_objects(cc_labels.T)
    return [(s and s[::-1]) for s in all_slices]


def fill_all_holes(cc_labels):
    """kimimaro/intake.py:747-795: fill the holes of every component (fill_voids.fill stand-in:
    scipy.ndimage.binary_fill_holes, 6-connected background) in ascending label order; a component that gets
    swallowed is not processed itself any more.  Bounding boxes are those of the labels before any filling."""
    import scipy.ndimage
    labels_set = set(int(l) for l in np.unique(cc_labels))
    labels_set.discard(0)
    all_slices = find_objects(cc_labels)
    for label in sorted(labels_set.copy()):
        if label not in labels_set:
            continue
        slices = all_slices[label - 1]
        if slices is None:
            continue
        binary = scipy.ndimage.binary_fill_holes(cc_labels[slices] == label)
        if int(binary.sum()) == int((cc_labels[slices] == label).sum()):
            continue
        sub = set(int(l) for l in np.unique(cc_labels[slices] * binary))
        sub.discard(label)
        labels_set -= sub
        cc_labels[slices] = cc_labels[slices] * ~binary + label * binary
    return cc_labels


def find_avocado_fruit(labels, cx, cy, cz, background=0):
    """kimimaro.skeletontricks.find_avocado_fruit (skeletontricks.pyx:905-992): six rays from (cx, cy, cz) along the axes; each stops at
    the background or reports the first other label it meets (the rays towards smaller coordinates never look at index 0:
    `range(c, 0, -1)`).  Fewer than three reports: no decision.  The most frequent report (np.unique order breaks ties: the smallest
    label) is the fruit if at most one report disagrees (none when there are exactly three).  Returns (pit, fruit)."""
    sx, sy, sz = labels.shape[:3]
    if cx >= sx or cy >= sy or cz >= sz:
        raise ValueError("<{},{},{}> must be be contained within shape <{},{},{}>".format(cx, cy, cz, sx, sy, sz))
    label = labels[cx, cy, cz]
    lines = (labels[cx:sx, cy, cz], labels[cx:0:-1, cy, cz], labels[cx, cy:sy, cz], labels[cx, cy:0:-1, cz],
             labels[cx, cy, cz:sz], labels[cx, cy, cz:0:-1])
    changes = []
    for line in lines:
        for v in line:
            if v == background:
                break
            if v != label:
                changes.append(v)
                break
    if len(changes) < 3:
        return (label, label)
    allowed = 1 if len(changes) > 3 else 0
    uniq, cts = np.unique(changes, return_counts=True)
    k = int(np.argmax(cts))
    if len(changes) - cts[k] > allowed:
        return (label, label)
    return (label, uniq[k])


def _fill2d(img):
    import scipy.ndimage
    return scipy.ndimage.binary_fill_holes(img)


def engage_avocado_protection_single_pass(cc_labels, all_dbf, candidates):
    """kimimaro/intake.py:642-704 (fill_voids.fill stand-in: scipy.ndimage.binary_fill_holes, in 2-D on the six faces of the crop
    and in 3-D on the crop).  `candidates`: an iterable in the order the reference's set iterates."""
    import scipy.ndimage
    candidates = [label for label in candidates if label != 0]
    unchanged, changed = set(), set()
    if len(candidates) == 0:
        return cc_labels, unchanged, changed
    slcs = find_objects(cc_labels)
    for label in candidates:
        slc = slcs[label - 1]
        offset = np.array([s.start for s in slc], dtype=np.int64)
        binimg = (cc_labels[slc] == label)
        for face in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), 0, slice(None)),
                     (slice(None), -1, slice(None)), (0, slice(None), slice(None)), (-1, slice(None), slice(None))):
            binimg[face] = _fill2d(binimg[face])                                   # paint_walls, :655-666
        prod = binimg * all_dbf[slc]
        coord = np.array(np.unravel_index(np.argmax(prod.T), prod.shape, order="F")) + offset      # argmax, :596-599
        pit, fruit = find_avocado_fruit(cc_labels, int(coord[0]), int(coord[1]), int(coord[2]))
        pit, fruit = int(pit), int(fruit)
        if pit == fruit and pit not in changed:
            unchanged.add(pit)
        else:
            unchanged.discard(pit)
            unchanged.discard(fruit)
            changed.add(pit)
            changed.add(fruit)
            binimg |= (cc_labels[slc] == fruit)
        binimg = scipy.ndimage.binary_fill_holes(binimg)
        sub = cc_labels[slc]
        sub *= ~binimg
        sub += np.asarray(fruit, dtype=cc_labels.dtype) * binimg
    return cc_labels, unchanged, changed


def get_mapping(orig_labels, cc_labels):
    """kimimaro.skeletontricks.get_mapping (skeletontricks.pyx:490-525): the raster (x fastest) is walked once; at every voxel whose
    component label differs from the PREVIOUS voxel's, remap[cc] = original label there -- the last such place wins."""
    o = orig_labels.ravel(order="F")
    c = cc_labels.ravel(order="F")
    remap = {}
    if o.size == 0:
        return remap
    starts = np.flatnonzero(np.concatenate([[True], c[1:] != c[:-1]]))
    for i in starts:            # ascending: later run starts overwrite earlier ones
        remap[int(c[i])] = o[i].item()
    return remap


def engage_avocado_protection(cc_labels, all_dbf, remapping, soma_detection_threshold, edtfn):
    """kimimaro/intake.py:600-640.  The candidates of a pass are a Python set built from the sorted unique labels (fastremap.unique)
    minus the labels that did not change in an earlier pass; the reference iterates that set, so does this."""
    orig_cc_labels = np.copy(cc_labels, order="F")
    unchanged = set()
    for _ in range(20):
        vals = np.unique(cc_labels * (all_dbf > soma_detection_threshold / 2.5))
        candidates = set(int(v) for v in vals)
        candidates -= unchanged
        candidates.discard(0)
        cc_labels, unchanged_this_cycle, changes = engage_avocado_protection_single_pass(cc_labels, all_dbf, candidates)
        unchanged |= unchanged_this_cycle
        if len(changes) == 0:
            break
        all_dbf = edtfn(cc_labels)
    # fastremap.renumber: new ids 1..N in the order of first appearance in memory (0 stays 0)
    flat = cc_labels.ravel(order="F")
    uniq, first = np.unique(flat, return_index=True)
    keep = uniq != 0
    order = np.argsort(first[keep], kind="stable")
    lut = np.zeros(int(uniq.max()) + 1 if uniq.size else 1, dtype=cc_labels.dtype)
    lut[uniq[keep][order]] = np.arange(1, int(keep.sum()) + 1, dtype=cc_labels.dtype)
    cc_labels = np.asfortranarray(lut[cc_labels])
    cc_remapping = get_mapping(orig_cc_labels, cc_labels)
    adjusted = {}
    for new_cc, cc in cc_remapping.items():
        if cc in remapping:
            adjusted[new_cc] = remapping[cc]
    return cc_labels, all_dbf, adjusted


def skeletonize(all_labels, teasar_params=DEFAULT_TEASAR_PARAMS, anisotropy=(1, 1, 1),
                object_ids=None, dust_threshold=1000, progress=False, fix_branching=True,
                in_place=False, fix_borders=True, parallel=1, parallel_chunk_size=100,
                extra_targets_before=[], extra_targets_after=[], fill_holes=False,
                fix_avocados=False, voxel_graph=None, stats=None):
    """kimimaro/intake.py:58-221 + skeletonize_subset :434-517 (serial path)."""
    anisotropy = np.array(anisotropy, dtype=np.float32)
    all_labels = format_labels(all_labels)
    if object_ids is not None:
        all_labels = all_labels * np.isin(all_labels, list(object_ids)).astype(all_labels.dtype)
    if all_labels.size <= dust_threshold:
        return {}
    minlabel, maxlabel = all_labels.min(), all_labels.max()
    if minlabel == 0 and maxlabel == 0:
        return {}
    if voxel_graph is not None:
        # intake.py:162,174-183,467 with a graph: cc3d.color_connectivity_graph + edt.edt(voxel_graph=) (both packages absent from
        # the reference tree: PARITY UNPINNED, oracle/__init__.py) and the cropped graph to every trace()
        if fix_avocados:
            raise NotImplementedError("skeletonize(voxel_graph=, fix_avocados=True)")
        voxel_graph = np.asfortranarray(np.asarray(voxel_graph).reshape(all_labels.shape, order="F"), dtype=np.uint32)
    cc_labels, remapping = compute_cc_labels(all_labels, voxel_graph)
    if fill_holes:
        cc_labels = fill_all_holes(cc_labels)                                    # intake.py:168-169

    before = _points_to_labels(extra_targets_before, cc_labels)
    after = _points_to_labels(extra_targets_after, cc_labels)

    if voxel_graph is not None:
        all_dbf = K.edt_graph(cc_labels, voxel_graph, anisotropy, black_border=(minlabel == maxlabel))
    else:
        all_dbf = K.edt(cc_labels, anisotropy, black_border=(minlabel == maxlabel))  # intake.py:174-185
    if fix_avocados:                                                             # intake.py:187-193
        cc_labels, all_dbf, remapping = engage_avocado_protection(
            cc_labels, all_dbf, remapping, teasar_params.get("soma_detection_threshold", 0),
            lambda lab: K.edt(lab, anisotropy, black_border=(minlabel == maxlabel)))

    counts = np.bincount(cc_labels.ravel(order="F"))
    cc_segids = [i for i in range(1, counts.size) if counts[i] > dust_threshold]
    all_slices = find_objects(cc_labels)

    border_targets = defaultdict(list)
    if fix_borders:
        border_targets = _border.compute_border_targets(cc_labels, anisotropy, K.edt, K.connected_components)

    skeletons = defaultdict(list)
    for segid in cc_segids:
        slices = all_slices[segid - 1]
        if slices is None:
            continue
        minpt = np.array([s.start for s in slices], dtype=np.int64)
        vol = np.prod([s.stop - s.start for s in slices])
        if vol <= 1:
            continue
        labels = (cc_labels[slices] == segid)
        dbf = np.where(labels, all_dbf[slices], 0.0)
        mtb, mta, root = [], [], None

        def translate(targets):
            t = np.array(targets, dtype=np.int64).reshape(-1, 3) - minpt
            return [tuple(int(v) for v in p) for p in t]

        if len(border_targets[segid]) > 0:
            mtb = translate(border_targets[segid])
            root = mtb.pop()
        if segid in before and len(before[segid]) > 0:
            mtb.extend(translate(before[segid]))
        if segid in after and len(after[segid]) > 0:
            mta.extend(translate(after[segid]))

        skel = trace(labels, dbf, anisotropy=anisotropy, fix_branching=fix_branching,
                     manual_targets_before=mtb, manual_targets_after=mta, root=root,
                     voxel_graph=(None if voxel_graph is None else voxel_graph[slices]),          # intake.py:467
                     stats=stats, **teasar_params)
        if skel.empty():
            continue
        skel.vertices += minpt.astype(skel.vertices.dtype)
        orig = remapping[segid]
        skel.id = orig
        skel.vertices = np.multiply(skel.vertices, anisotropy, dtype=np.float32)
        skel.space = "physical"
        skeletons[orig].append(skel)

    merged = {}
    for segid, skels in skeletons.items():  # intake.py:587-593
        merged[segid] = Skeleton.simple_merge(skels).consolidate()
    return merged


def _points_to_labels(pts, cc_labels):
    mapping = defaultdict(list)
    for pt in pts:
        pt = tuple(int(v) for v in pt)
        mapping[int(cc_labels[pt])].append(pt)
    return mapping
