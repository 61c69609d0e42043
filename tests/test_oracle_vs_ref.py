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
