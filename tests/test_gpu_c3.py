"""BASELINE.json configs[2] and configs[3] at FULL size: the bench workload c3 (512^3, 2124 chains -> ~3.4 k components,
default teasar_params, fix_borders, fix_branching)

  * through kimimaro_amd.skeletonize, every skeleton compared with the oracle pipeline (vertices / edges bit exact, radii to
    1e-4 relative);
  * as the rank / world shard of configs[3] (skeletonize_cc(rank=r, world=2) for r = 0, 1 on this GPU, merged through the
    wire format of the all-gather-v), against the same oracle result.

The oracle runs once per module on a forked pool over the GPU box's host cores (oracle/pool.py, ~30-60 s); on a host with
fewer than 32 cores a seeded sample of LABELS (every component of a sampled label) that contains the 16 largest components
is compared instead."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def eng():
    from kimimaro_amd.engine import Engine
    return Engine()


@pytest.fixture(scope="module")
def c3():
    import bench
    import kimimaro_amd
    from oracle import pool
    import oracle as K
    lab, an = bench.make_volume("c3")
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    only = None
    if (os.cpu_count() or 1) < 32:
        # a sample of labels; all components of a sampled label are traced, so every skeleton of `want` is complete
        cc, _ = K.connected_components(lab)
        flat_cc, flat_lab = cc.ravel(order="K"), np.asfortranarray(lab).ravel(order="K")
        counts = np.bincount(flat_cc)
        ids, where = np.unique(flat_cc, return_index=True)
        label_of = np.zeros(counts.size, dtype=np.int64)
        label_of[ids] = flat_lab[where]
        big = np.argsort(-counts[1:], kind="stable")[:16] + 1
        rng = np.random.default_rng(0)
        labels = np.unique(label_of[1:])
        chosen = set(label_of[big].tolist()) | set(rng.choice(labels, size=min(200, labels.size), replace=False).tolist())
        only = set(np.flatnonzero(np.isin(label_of, list(chosen)) & (np.arange(counts.size) > 0)).tolist())
    from oracle.cpu_pool_baseline import usable_cores
    aff, quota = usable_cores()                      # (the pool's boxes show 256 CPUs and grant 16: that many workers)
    want, cc, counts = pool.skeletonize_pool(lab, params, anisotropy=an, dust_threshold=1000, fix_branching=True,
                                             fix_borders=True, only=only, workers=max(2, int(min(aff, quota)) if quota else aff))
    return lab, an, params, want, only


def _compare(got, want, full):
    if full:
        assert sorted(got.keys()) == sorted(want.keys())
        assert len(want) == 2124
    for k, w in want.items():
        g = got[k]
        np.testing.assert_array_equal(g.vertices, w.vertices, err_msg="label %d" % k)
        np.testing.assert_array_equal(g.edges, w.edges, err_msg="label %d" % k)
        np.testing.assert_allclose(g.radii, w.radii, rtol=1e-4, err_msg="label %d" % k)
    assert len(want) >= (2000 if full else 100)


def test_c3_every_skeleton_matches_oracle(eng, c3):
    import kimimaro_amd
    lab, an, params, want, only = c3
    got = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=1000, fix_borders=True, fix_branching=True,
                                   progress=False, _engine=eng)
    tk = eng.last_tasks
    assert int(tk["stat_sweep_calls"].sum()) > 10000            # the order-free sweep did the bulk of the invalidations ...
    assert int(tk["stat_sweep_bails"].sum()) > 0                 # ... and its fall-back (the heap emulation) was exercised too
    _compare(got, want, only is None)


def test_c3_sharded_over_two_ranks(eng, c3):
    """configs[3] on the bench volume itself: two ranks' shards of c3, one after the other on this GPU, merged like the
    all-gather-v merges them."""
    from test_gpu_configs import _sharded
    lab, an, params, want, only = c3
    got, sizes = _sharded(eng, lab, an, params, 1000, 2)
    assert all(s > 0 for s in sizes)
    _compare(got, want, only is None)
