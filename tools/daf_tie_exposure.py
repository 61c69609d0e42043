#!/usr/bin/env python
"""How exposed is a bench volume to the one tie the reference leaves unspecified in CachedTargetFinder?

kimimaro's CachedTargetFinder sorts the DAF with numpy's default (unstable) argsort (skeletontricks.pyx:1001-1006), so
which of two valid voxels with EQUAL DAF is handed out first is a property of the CPU / numpy build, not of the algorithm
(SURVEY 0-7a).  Oracle and HIP path break the tie canonically (descending DAF, then descending index).  This tool counts,
for every connected component of a bench workload, the target selections in which the chosen voxel had a still-valid
rival with the same DAF -- i.e. the selections where a different numpy could have sent the reference another way.

  python tools/daf_tie_exposure.py [c2|c3|mini] [--workers N]

CPU only (drives the oracle; test / measurement infrastructure like everything under oracle/).  Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_STATE = {"daf": None, "ties": 0, "selections": 0}


def _patch():
    import oracle as K
    from oracle import pipeline as P
    real_order = K.target_order

    def target_order(mask, daf):
        _STATE["daf"] = np.asarray(daf).ravel(order="F")
        return real_order(mask, daf)

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
    pool._one(segid)
    return segid, _STATE["ties"], _STATE["selections"]


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
    cc, _ = P.compute_cc_labels(lab)
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
    exposed = [(s, t, n) for s, t, n in res if t > 0]
    out = {"workload": name, "components": len(res), "target_selections": int(sum(n for _, _, n in res)),
           "selections_with_an_equal_daf_rival": int(sum(t for _, t, _ in res)),
           "components_with_such_a_selection": len(exposed),
           "voxels_of_those_components": int(sum(int(counts[s]) for s, _, _ in exposed)),
           "voxels": int(sum(int(counts[s]) for s in segids)),
           "largest_exposed": sorted(((int(counts[s]), int(s), int(t)) for s, t, _ in exposed), reverse=True)[:8],
           "seconds": round(time.perf_counter() - t0, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
