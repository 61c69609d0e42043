"""All-cores CPU baseline of bench.py (test/measurement infrastructure, like everything under oracle/).

Run as a child process by bench.py's cpu_baseline leg, with a hard timeout, so that nothing here can hang or
crash the bench line:  python oracle/cpu_pool_baseline.py <cc_labels.npy> <json args>
It maps the component volume read-only, forks one worker per host core and lets them run the per-label work
of bench.cpu_baseline (EDT on the label's bounding box grown by one voxel + the full TEASAR trace, all through
oracle/kimi_oracle.c) on a seeded sample of the labels.  Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_CC = None
_SLICES = None
_AN = None
_PARAMS = None


def _task(sid):
    import oracle
    from oracle import pipeline as P
    if sid == 0:
        return 0
    slc = _SLICES[sid - 1][::-1]
    glo = [max(0, s.start - 1) for s in slc]
    grown = tuple(slice(g, min(n, s.stop + 1)) for g, s, n in zip(glo, slc, _CC.shape))
    inner = tuple(slice(s.start - g, s.stop - g) for s, g in zip(slc, glo))
    crop = np.asfortranarray(_CC[grown])
    dbf = oracle.edt(crop, _AN, black_border=False)[inner]
    mask = np.asfortranarray(crop[inner] == sid)
    dbf = np.asfortranarray(np.where(mask, dbf, 0.0).astype(np.float32))
    P.trace(mask, dbf, anisotropy=_AN, fix_branching=True, **_PARAMS)
    return 1


def main():
    global _CC, _SLICES, _AN, _PARAMS
    import multiprocessing as mp
    import scipy.ndimage
    path = sys.argv[1]
    args = json.loads(sys.argv[2])
    _CC = np.load(path, mmap_mode="r")
    _AN = tuple(args["anisotropy"])
    _PARAMS = args["params"]
    ncores = os.cpu_count() or 1
    counts = np.bincount(np.asarray(_CC).ravel(order="K"))
    segids = [i for i in range(1, counts.size) if counts[i] > args["dust_threshold"]]
    rng = np.random.default_rng(0)
    rng.shuffle(segids)
    nsample = int(min(len(segids), max(2 * ncores, args["one_core_rate"] * ncores * args["budget_s"])))
    _SLICES = scipy.ndimage.find_objects(np.asarray(_CC).T)
    import oracle  # noqa: F401  build / load the shared object before forking
    with mp.get_context("fork").Pool(ncores) as pool:
        pool.map(_task, [0] * (2 * ncores), chunksize=1)  # every worker up
        t0 = time.perf_counter()
        done = sum(pool.imap_unordered(_task, segids[:nsample], chunksize=1))
        dt = time.perf_counter() - t0
    print(json.dumps({"value": done / dt, "unit": "labels/s", "cores": ncores, "kind": "port",
                      "sample": "%d of %d labels in %.1f s wall on a %d-process pool, same per-label work as the "
                                "1-core line" % (done, len(segids), dt, ncores)}))


if __name__ == "__main__":
    main()
