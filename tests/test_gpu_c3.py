"""BASELINE.json configs[2] at FULL size: the bench workload c3 (512^3, 2124 chains -> ~3.4 k components, default
teasar_params, fix_borders, fix_branching) through kimimaro_amd.skeletonize, every skeleton compared with the oracle
pipeline (vertices / edges bit exact, radii to 1e-4 relative).  The oracle runs on a forked pool over the GPU box's
host cores (oracle/pool.py, ~30-60 s); on a host with fewer than 32 cores a seeded sample of 256 components that
contains the 16 largest is compared instead."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_c3_every_skeleton_matches_oracle():
    import bench
    import kimimaro_amd
    import kimimaro_amd.engine as E
    from kimimaro_amd.engine import Engine
    from oracle import pool
    import oracle as K
    lab, an = bench.make_volume("c3")
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    eng = Engine()
    got = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=1000, fix_borders=True, fix_branching=True,
                                   progress=False, _engine=eng)
    tk = E.LAST_TASKS
    assert int(tk["stat_sweep_calls"].sum()) > 10000            # the order-free sweep did the bulk of the invalidations ...
    assert int(tk["stat_sweep_bails"].sum()) > 0                 # ... and its fall-back (the heap emulation) was exercised too
    only = None
    if (os.cpu_count() or 1) < 32:
        cc, _ = K.connected_components(lab)
        counts = np.bincount(cc.ravel(order="K"))
        big = np.argsort(-counts[1:])[:16] + 1
        rest = np.flatnonzero(counts > 1000)
        rest = rest[rest > 0]
        rng = np.random.default_rng(0)
        only = set(big.tolist()) | set(rng.choice(rest, size=min(240, rest.size), replace=False).tolist())
    want, cc, counts = pool.skeletonize_pool(lab, params, anisotropy=an, dust_threshold=1000, fix_branching=True,
                                             fix_borders=True, only=only)
    if only is None:
        assert sorted(got.keys()) == sorted(want.keys())
        assert len(want) == 2124
    checked = 0
    for k, w in want.items():
        g = got[k]
        if only is not None and g.vertices.shape != w.vertices.shape:
            continue   # a label with several components of which only some were sampled
        np.testing.assert_array_equal(g.vertices, w.vertices, err_msg="label %d" % k)
        np.testing.assert_array_equal(g.edges, w.edges, err_msg="label %d" % k)
        np.testing.assert_allclose(g.radii, w.radii, rtol=1e-4, err_msg="label %d" % k)
        checked += 1
    assert checked >= (2000 if only is None else 100)
