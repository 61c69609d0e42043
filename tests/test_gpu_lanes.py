"""Several volumes in flight (kimimaro_amd.lanes): every lane's skeletons are those of the oracle pipeline, whatever
runs beside it.  Six different volumes go through three lanes at once."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _same(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        assert np.array_equal(a[k].vertices, b[k].vertices), k
        assert np.array_equal(a[k].edges, b[k].edges), k
        np.testing.assert_allclose(a[k].radii, b[k].radii, rtol=1e-4)


def test_lanes_match_oracle_on_different_volumes():
    import kimimaro_amd
    from kimimaro_amd.lanes import Lanes
    from oracle import pipeline as P
    from shapes import voronoi_labels
    an = (16, 16, 40)
    vols = [voronoi_labels((96, 80, 48), 12 + s, 100 + s, pts_per_label=5, anisotropy=an) for s in range(6)]
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    kw = dict(anisotropy=an, dust_threshold=200, fix_borders=True, fix_branching=True, progress=False)
    lanes = Lanes(3)
    got = dict(lanes.run(lambda eng, k: kimimaro_amd.skeletonize(vols[k], params, _engine=eng, **kw), len(vols)))
    assert sorted(got) == list(range(len(vols)))
    for k, lab in enumerate(vols):
        want = P.skeletonize(lab, params, anisotropy=an, dust_threshold=200, fix_borders=True, fix_branching=True)
        assert len(want) > 3
        _same(got[k], want)
    # the same lanes again, one at a time: same answers (no state is left behind in a lane)
    again = dict(lanes.run(lambda eng, k: kimimaro_amd.skeletonize(vols[k], params, _engine=eng, **kw), 2, width=1))
    for k in again:
        _same(again[k], got[k])


def test_skeletonize_many_in_order_with_loaders():
    """kimimaro_amd.skeletonize_many: arrays and loader callables mixed, results in order, equal to one volume at a time"""
    import kimimaro_amd
    from shapes import voronoi_labels
    an = (16, 16, 40)
    vols = [voronoi_labels((80, 64, 40), 9 + s, 200 + s, pts_per_label=4, anisotropy=an) for s in range(5)]
    kw = dict(anisotropy=an, dust_threshold=200, fix_borders=True, progress=False)
    mixed = [vols[0], (lambda: vols[1]), vols[2], (lambda: vols[3]), vols[4]]
    got = list(kimimaro_amd.skeletonize_many(mixed, width=3, **kw))
    assert [k for k, _ in got] == list(range(5))
    for k, sk in got:
        _same(sk, kimimaro_amd.skeletonize(vols[k], **kw))
    assert len(list(kimimaro_amd.skeletonize_many([], **kw))) == 0
