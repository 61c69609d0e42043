"""Engine.scratch_budget: a volume whose per-label scratch exceeds the budget is traced in several launches of the path
loop (largest labels first) with the same skeletons."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_small_budget_splits_the_labels_and_changes_nothing():
    import kimimaro_amd
    from kimimaro_amd.engine import Engine
    from oracle import pipeline as P
    from shapes import voronoi_labels
    an = (16, 16, 40)
    lab = voronoi_labels((128, 96, 48), 30, 11, pts_per_label=5, anisotropy=an)
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    kw = dict(anisotropy=an, dust_threshold=300, fix_borders=True, fix_branching=True, progress=False)
    eng = Engine()
    whole = kimimaro_amd.skeletonize(lab, params, _engine=eng, **kw)
    n_all = len(eng.last_tasks)
    eng2 = Engine()
    eng2.scratch_budget = 24 << 20           # a few labels per launch
    split = kimimaro_amd.skeletonize(lab, params, _engine=eng2, **kw)
    assert len(eng2.last_tasks) == n_all
    want = P.skeletonize(lab, params, anisotropy=an, dust_threshold=300, fix_borders=True, fix_branching=True)
    assert sorted(split) == sorted(want) == sorted(whole)
    for k in want:
        for got in (split[k], whole[k]):
            assert np.array_equal(got.vertices, want[k].vertices), k
            assert np.array_equal(got.edges, want[k].edges), k
            np.testing.assert_allclose(got.radii, want[k].radii, rtol=1e-4)
    # the budget really forced more than one launch
    counts = np.bincount(__import__("oracle").connected_components(lab)[0].ravel())[1:]
    from kimimaro_amd.engine import SCRATCH_BYTES_PER_VOXEL, SCRATCH_BYTES_PER_LABEL
    assert int((counts[counts > 300] * SCRATCH_BYTES_PER_VOXEL + SCRATCH_BYTES_PER_LABEL).sum()) > eng2.scratch_budget
