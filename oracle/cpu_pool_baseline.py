"""All-cores CPU baseline of bench.py (test/measurement infrastructure, like everything under oracle/).

Run as a child process by bench.py's cpu_baseline leg, with a hard timeout, so that nothing here can hang or
crash the bench line:  python oracle/cpu_pool_baseline.py <cc_labels.npy> <json args>
It maps the component volume read-only, forks one worker per usable host core and lets them run the per-label work
of bench.cpu_baseline (EDT on the label's bounding box grown by one voxel + the full TEASAR trace, all through
oracle/kimi_oracle.c), the way kimimaro's own parallel path deals components to a process pool
(kimimaro/intake.py:344-408) -- except that the components are handed out LARGEST FIRST (the reference deals
them round robin; largest-first is the better schedule for a pool, so the baseline is not handicapped).

Two legs, so that the GPU figures can be compared like for like:
  latency     ONE volume's components through the pool: wall clock of a single skeletonize() on all cores.
  throughput  K volumes' components through the SAME pool at once (K = the GPU run's volumes in flight): the CPU
              counterpart of the pipelined GPU figure -- the tail of one volume is filled by the others.
Prints one JSON object: {"latency": {...}, "throughput": {...}, "cores": ..., "affinity": ..., "cgroup_cpus": ...}.
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


def usable_cores():
    """(cores this process may run on, cgroup CPU quota in cores or None)"""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    return aff, quota


def main():
    global _CC, _SLICES, _AN, _PARAMS
    import multiprocessing as mp
    import scipy.ndimage
    path = sys.argv[1]
    args = json.loads(sys.argv[2])
    _CC = np.load(path, mmap_mode="r")
    _AN = tuple(args["anisotropy"])
    _PARAMS = args["params"]
    aff, quota = usable_cores()
    ncores = max(1, int(min(aff, quota) if quota else aff))
    inflight = max(1, int(args.get("volumes_in_flight", 1)))
    counts = np.bincount(np.asarray(_CC).ravel(order="K"))
    segids = [i for i in range(1, counts.size) if counts[i] > args["dust_threshold"]]
    segids.sort(key=lambda s: (-int(counts[s]), s))          # largest first
    # bounded: when one volume would take the pool longer than the budget, every k-th component of the size-sorted
    # list is taken (same size distribution, the largest one included)
    est_wall = len(segids) / max(args["one_core_rate"], 1e-9) / ncores
    stride = max(1, int(np.ceil(est_wall / max(args["budget_s"], 1e-9))))
    sample = segids[::stride]
    _SLICES = scipy.ndimage.find_objects(np.asarray(_CC).T)
    import oracle  # noqa: F401  build / load the shared object before forking
    out = {"cores": ncores, "affinity": aff, "cgroup_cpus": quota, "os_cpu_count": os.cpu_count(), "unit": "labels/s",
           "kind": "port", "order": "largest component first"}
    with mp.get_context("fork").Pool(ncores) as pool:
        pool.map(_task, [0] * (2 * ncores), chunksize=1)  # every worker up
        t0 = time.perf_counter()
        done = sum(pool.imap_unordered(_task, sample, chunksize=1))
        dt = time.perf_counter() - t0
        out["latency"] = {"value": done / dt, "wall_s": dt,
                          "sample": "%d of %d components of ONE volume (every %d-th of the size-sorted list) on a %d-process "
                                    "pool, same per-label work as the 1-core line" % (done, len(segids), stride, ncores)}
        if inflight > 1:
            many = [s for s in sample for _ in range(inflight)]      # stays largest first
            t0 = time.perf_counter()
            done = sum(pool.imap_unordered(_task, many, chunksize=1))
            dt = time.perf_counter() - t0
            out["throughput"] = {"value": done / dt, "wall_s": dt, "volumes": inflight,
                                 "sample": "%d components = %d volumes' worth through the same pool at once" % (done, inflight)}
        else:
            out["throughput"] = dict(out["latency"], volumes=1)
    # the old single-figure fields, so that readers of earlier rounds' lines still find them
    out["value"] = out["latency"]["value"]
    out["sample"] = out["latency"]["sample"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
