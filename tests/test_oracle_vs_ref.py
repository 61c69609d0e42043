"""Pins the oracle against the reference ITSELF (ext/skeletontricks compiled where it lies into
oracle/_ref by oracle/build_ref.py).  Skipped when neither /root/reference nor a prebuilt
oracle/_ref module is present."""
import numpy as np
import pytest

import oracle as K
from shapes import random_walk_tube

pytestmark = pytest.mark.ref


def test_ball_matches_reference_random(refmod):
    rng = np.random.default_rng(42)
    for t in range(40):
        shape = (int(rng.integers(12, 34)), int(rng.integers(12, 30)), int(rng.integers(8, 28)))
        an = [(1, 1, 1), (16, 16, 40), (4, 4, 40), (1, 2, 3)][t % 4]
        m = random_walk_tube(shape, 900 + t, steps=30, step=2.5, radius=(1.2, 4.5))
        dbf = K.edt(m, an)
        idx = np.flatnonzero(m.ravel(order="F"))
        sel = rng.choice(idx, int(rng.integers(1, 16)), replace=False)
        path = K.locs_to_pts(sel, shape)
        scale, const = [1.5, 4, 0.5][t % 3], [0, 3 * an[0], 30][t % 3]
        a, b = m.copy(order="F"), m.copy(order="F")
        c1, _ = refmod.roll_invalidation_ball_inside_component(a, dbf, scale, const, an, [tuple(int(v) for v in p) for p in path])
        c2, _ = K.roll_invalidation_ball_inside_component(b, dbf, scale, const, an, path)
        assert c1 == c2
        np.testing.assert_array_equal(a, b)


def test_ball_x_border_quirk(refmod):
    """sources on the x faces of the crop exercise the duplicated corner entries
    (dijkstra_invalidation.hpp:116-123)."""
    m = np.ones((6, 9, 9), np.uint8, order="F")
    dbf = K.edt(m, (1, 1, 1), black_border=True)
    path = [(0, 4, 4), (5, 4, 4), (0, 0, 0), (5, 8, 8)]
    a, b = m.copy(order="F"), m.copy(order="F")
    c1, _ = refmod.roll_invalidation_ball_inside_component(a, dbf, 2.0, 1.5, (1, 1, 1), path)
    c2, _ = K.roll_invalidation_ball_inside_component(b, dbf, 2.0, 1.5, (1, 1, 1), path)
    assert c1 == c2
    np.testing.assert_array_equal(a, b)


def test_cube_matches_reference_random(refmod):
    rng = np.random.default_rng(7)
    for t in range(60):
        shape = tuple(int(s) for s in rng.integers(6, 22, size=3))
        m = np.asfortranarray((rng.random(shape) < 0.85).astype(np.uint8))
        dbf = np.asfortranarray(rng.uniform(0, 3, shape).astype(np.float32))
        path = [tuple(int(rng.integers(0, s)) for s in shape) for _ in range(int(rng.integers(1, 5)))]
        an = tuple(float(rng.uniform(0.5, 4.0)) for _ in range(3))
        scale, const = float(rng.uniform(0, 2)), float(rng.uniform(0.2, 3))
        a, b = m.copy(order="F"), m.copy(order="F")
        c1, _ = refmod.roll_invalidation_cube(a, dbf, path, scale, const, anisotropy=an)
        c2, _ = K.roll_invalidation_cube(b, dbf, path, scale, const, an)
        assert c1 == c2
        np.testing.assert_array_equal(a, b)


def test_small_helpers(refmod):
    rng = np.random.default_rng(1)
    f = np.asfortranarray(rng.integers(0, 3, (7, 6, 5)).astype(np.float32))
    g = f.copy(order="F")
    np.testing.assert_array_equal(refmod.zero2inf(f), K.zero2inf(g))
    np.testing.assert_array_equal(refmod.inf2zero(f), K.inf2zero(g))
    m = np.zeros((5, 6, 7), np.uint8, order="F")
    assert refmod.first_label(m) is None and K.first_label(m) is None
    m[3, 2, 4] = m[1, 5, 4] = m[4, 4, 6] = 1
    assert tuple(refmod.first_label(m)) == K.first_label(m)


def test_find_border_targets_kat(refmod):
    """automated_test.py:104-114: a 257^2 plate -> (128,128); restated helper == reference."""
    from scipy import ndimage
    from kimimaro_amd.border import find_border_targets
    labels = np.ones((257, 257), dtype=np.uint32, order="F")
    dt = np.asfortranarray(ndimage.distance_transform_edt(np.pad(labels, 1))[1:-1, 1:-1].astype(np.float32))
    want = refmod.find_border_targets(dt, labels, 100, 100)
    got = find_border_targets(dt, labels, 100, 100, 1)
    assert {k: (int(v[0]), int(v[1])) for k, v in got.items()} == {1: (128, 128)}
    assert {k: (int(v[0]), int(v[1])) for k, v in want.items()} == {1: (128, 128)}


def test_find_border_targets_random_planes_with_ties(refmod):
    """the product's host helper (kh_host_find_border_targets: per-pixel tie-break tuples + a first-minimum reduction)
    against the reference's find_border_targets on random multi-component planes whose distance values are plateaus,
    so that every rule of the tie-break cascade decides somewhere: same pixel per component, same dict order."""
    from kimimaro_amd.border import find_border_targets
    rng = np.random.default_rng(591)
    decided_by_ties = 0
    for t in range(120):
        sx, sy = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        nlab = int(rng.integers(1, 6))
        # components as vertical / horizontal bands and blocks (ids 1..nlab, 0 = background)
        cc = np.zeros((sx, sy), np.uint32, order="F")
        for l in range(1, nlab + 1):
            x0, y0 = int(rng.integers(0, sx)), int(rng.integers(0, sy))
            x1, y1 = int(rng.integers(x0, sx)) + 1, int(rng.integers(y0, sy)) + 1
            cc[x0:x1, y0:y1] = l
        # plateau-valued "distance transform": few distinct values, zeros included
        levels = [1, 2, 3, 8][t % 4]
        dt = np.asfortranarray(np.floor(rng.random((sx, sy)) * (levels + 1)).astype(np.float32))
        if t % 5 == 0:
            dt[...] = 1.0                                  # one plateau: the tie-break decides everything
        wx, wy = [(1, 1), (16, 16), (16, 40), (40, 16), (3, 7)][t % 5]
        want = refmod.find_border_targets(dt, cc, wx, wy)
        got = find_border_targets(dt, cc, wx, wy, nlab)
        assert list(got.keys()) == list(want.keys()), t
        for k in want:
            assert (int(got[k][0]), int(got[k][1])) == (int(want[k][0]), int(want[k][1])), (t, k)
            decided_by_ties += int(np.count_nonzero((cc == k) & (dt == dt[cc == k].max())) > 1)
    assert decided_by_ties > 100
