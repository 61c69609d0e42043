#!/usr/bin/env python
"""How exposed is a bench volume to the one tie the reference leaves unspecified in CachedTargetFinder?

kimimaro's CachedTargetFinder sorts the DAF with numpy's default (unstable) argsort (skeletontricks.pyx:1001-1006), so
which of two valid voxels with EQUAL DAF is handed out first is a property of the CPU / numpy build, not of the algorithm
(SURVEY 0-7a).  Oracle and HIP path break the tie canonically (descending DAF, then descending index).  This tool counts,
for every connected component of a bench workload, the target selections in which the chosen voxel had a still-valid
rival with the same DAF -- i.e. the selections where a different numpy could have sent the reference another way.

  python tools/daf_tie_exposure.py [c2|c3|mini] [--workers N] [--flip]

--flip (round 6): every component with such a selection is traced a second time with the OPPOSITE tie rule (descending DAF, ties by
ASCENDING index) and the two skeletons are compared: how many components -- and how many of the volume's skeletons (original
labels) -- come out differently when the one unspecified tie goes the other way.

CPU only (drives the oracle; test / measurement infrastructure like everything under oracle/).  Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_STATE = {"daf": None, "ties": 0, "selections": 0, "flip": False}


def _patch():
    import oracle as K
    from oracle import pipeline as P
    real_order = K.target_order

    def target_order(mask, daf):
        _STATE["daf"] = np.asarray(daf).ravel(order="F")
        order = real_order(mask, daf)
        if _STATE["flip"]:
            # the same descending DAF, every run of equal values reversed: ties by ascending index
            d = _STATE["daf"][order]
            start = np.flatnonzero(np.concatenate([[True], d[1:] != d[:-1]]))
            end = np.concatenate([start[1:], [d.size]])
            order = order.copy()
            for a, b in zip(start, end):
                if b - a > 1:
                    order[a:b] = order[a:b][::-1]
        return order

    real_find = P._TargetFinder.find_target

    def find_target(self, labels):
        tgt = real_find(self, labels)
        if tgt is None:
            return tgt
        _STATE["selections"] += 1
        flat = labels.ravel(order="F")
        o, h, daf = self.order, self.head, _STATE["daf"]
        d0 = daf[o[h]]
        k = h + 1
        while k < o.size and daf[o[k]] == d0:      # the rest of the equal-DAF run, in the canonical order
            if flat[o[k]]:
                _STATE["ties"] += 1
                break
            k += 1
        return tgt

    K.target_order = target_order
    P.K.target_order = target_order
    P._TargetFinder.find_target = find_target


def _one(segid):
    from oracle import pool
    _STATE["ties"] = 0
    _STATE["selections"] = 0
    _STATE["flip"] = False
    _, res = pool._one(segid)
    ties, sel = _STATE["ties"], _STATE["selections"]
    changed = None
    if ties > 0 and "--flip" in sys.argv:
        _STATE["flip"] = True
        _, res2 = pool._one(segid)
        _STATE["flip"] = False
        same = (res is None) == (res2 is None) and (res is None or all(
            a.shape == b.shape and np.array_equal(a, b) for a, b in zip(res, res2)))
        changed = (not same, 0 if res is None else int(res[0].shape[0]), 0 if res2 is None else int(res2[0].shape[0]))
    return segid, ties, sel, changed


def main():
    import multiprocessing as mp
    import scipy.ndimage
    import bench
    import oracle as K
    from oracle import border as _border, pipeline as P, pool
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c3"
    workers = int(sys.argv[sys.argv.index("--workers") + 1]) if "--workers" in sys.argv else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    lab, an = bench.make_volume(name)
    an = np.array(an, dtype=np.float32)
    lab = P.format_labels(lab)
    cc, remapping = P.compute_cc_labels(lab)
    counts = np.bincount(cc.ravel(order="K"))
    segids = [i for i in range(1, counts.size) if counts[i] > 1000]
    bt = _border.compute_border_targets(cc, an, K.edt, K.connected_components)
    slices = [(s and s[::-1]) for s in scipy.ndimage.find_objects(cc.T)]
    pool._G.update(cc=cc, an=an, params=dict(P.DEFAULT_TEASAR_PARAMS), fb=True, slices=slices,
                   targets={int(k): v for k, v in bt.items()})
    _patch()
    K.lib()
    order = sorted(segids, key=lambda s: -counts[s])
    with mp.get_context("fork").Pool(workers) as p:
        res = list(p.imap_unordered(_one, order, chunksize=1))
    flips = [(s, c) for s, _, _, c in res if c is not None]
    res = [(s, t, n) for s, t, n, _ in res]
    exposed = [(s, t, n) for s, t, n in res if t > 0]
    out = {"workload": name, "components": len(res), "target_selections": int(sum(n for _, _, n in res)),
           "selections_with_an_equal_daf_rival": int(sum(t for _, t, _ in res)),
           "components_with_such_a_selection": len(exposed),
           "voxels_of_those_components": int(sum(int(counts[s]) for s, _, _ in exposed)),
           "voxels": int(sum(int(counts[s]) for s in segids)),
           "largest_exposed": sorted(((int(counts[s]), int(s), int(t)) for s, t, _ in exposed), reverse=True)[:8],
           "seconds": round(time.perf_counter() - t0, 1)}
    if "--flip" in sys.argv:
        diff = [(s, c) for s, c in flips if c[0]]
        out["opposite_tie_rule"] = {
            "components_retraced": len(flips), "components_whose_skeleton_differs": len(diff),
            "skeletons_that_differ": len(set(remapping[s] for s, _ in diff)), "skeletons": len(set(remapping[s] for s in segids)),
            "vertices_canonical_vs_opposite": [[int(counts[s]), int(c[1]), int(c[2])] for s, c in sorted(diff, key=lambda x: -counts[x[0]])[:12]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
