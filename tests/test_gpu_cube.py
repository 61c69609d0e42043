"""GPU parity for roll_invalidation_cube (kh_invalidate_cube): golden vectors produced by the reference,
the reference's own exact counts and layout behaviour (automated_test.py:632-825)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.asfortranarray(np.unpackbits(bits)[:n].reshape(shape, order="F").astype(np.uint8))


def test_cube_golden_vectors():
    from kimimaro_amd.ops import roll_invalidation_cube
    z = np.load(os.path.join(G, "invalidation_cube.npz"))
    for i in range(int(z["n"])):
        shape = tuple(z["shape_%d" % i])
        m = unpack(z["mask_%d" % i], shape)
        path = z["path_%d" % i]
        dbf = np.zeros(shape, np.float32, order="F")
        dbf[path[:, 0], path[:, 1], path[:, 2]] = z["dbfpath_%d" % i]
        scale, const = z["sc_%d" % i]
        cnt, out = roll_invalidation_cube(m, dbf, [tuple(p) for p in path.tolist()], scale, const, tuple(z["an_%d" % i]))
        assert cnt == int(z["count_%d" % i]), i
        np.testing.assert_array_equal(out, unpack(z["after_%d" % i], shape), err_msg="case %d" % i)


def test_cube_reference_counts_and_identity():
    from kimimaro_amd.ops import roll_invalidation_cube
    L = np.ones((10, 10, 10), np.uint8)
    D = np.zeros((10, 10, 10), np.float32)
    cnt, out = roll_invalidation_cube(L, D, [(5, 5, 5)], 0.0, 2.0, anisotropy=(1.0, 1.0, 1.0))
    assert cnt == 125 and out is L
    L = np.ones((13, 17, 14), np.uint8)
    D = np.zeros((13, 17, 14), np.float32)
    assert roll_invalidation_cube(L, D, [(1, 16, 0)], 0.0, 0.965, anisotropy=(0.94, 0.93, 2.58))[0] == 9


@pytest.mark.parametrize("shape,path,an,scale,const", [
    ((8, 8, 8), [(4, 4, 4)], (1.0, 2.0, 4.0), 1.0, 0.0),
    ((10, 12, 14), [(3, 4, 5), (6, 7, 8)], (1.0, 1.0, 1.0), 1.0, 1.0),
    ((9, 11, 7), [(0, 0, 0), (8, 10, 6)], (2.0, 1.0, 3.0), 0.5, 2.0),
])
def test_cube_c_and_f_layouts_agree(shape, path, an, scale, const):
    """automated_test.py:757-778: C- and F-ordered inputs give the same voxels; DBF is not mutated."""
    from kimimaro_amd.ops import roll_invalidation_cube
    rng = np.random.default_rng(0)
    D = rng.uniform(0.8, 2.5, size=shape).astype(np.float32)
    Lc, Lf = np.ascontiguousarray(np.ones(shape, np.uint8)), np.asfortranarray(np.ones(shape, np.uint8))
    Dc = np.ascontiguousarray(D)
    D0 = Dc.copy()
    c1, o1 = roll_invalidation_cube(Lc, np.asfortranarray(D), path, scale, const, anisotropy=an)
    c2, o2 = roll_invalidation_cube(Lf, Dc, path, scale, const, anisotropy=an)
    assert c1 == c2 and c1 > 0
    np.testing.assert_array_equal(np.asarray(o1), np.asarray(o2))
    np.testing.assert_array_equal(Dc, D0)
    assert o1.flags.c_contiguous and o2.flags.f_contiguous


def test_cube_rejects_non_contiguous():
    from kimimaro_amd.ops import roll_invalidation_cube
    L = np.ones((8, 8, 8), np.uint8)[::2]
    with pytest.raises(ValueError):
        roll_invalidation_cube(L, np.zeros(L.shape, np.float32), [(1, 1, 1)], 1.0, 1.0)


def test_cube_reference_random_recipe_on_gpu():
    """automated_test.py:710-747 verbatim recipe against the geometric reference, through kh_invalidate_cube."""
    from kimimaro_amd.ops import roll_invalidation_cube
    from test_oracle_golden import _expected_corner_cube
    rng = np.random.default_rng(seed=0xDECAFBAD)
    for trial in range(100):
        shape = tuple(int(s) for s in rng.integers(8, 24, size=3))
        labels = np.ones(shape, dtype=np.uint8)
        dbf = np.zeros(shape, dtype=np.float32)
        n_path = int(rng.integers(1, 4))
        path = [tuple(int(rng.integers(0, s)) for s in shape) for _ in range(n_path)]
        radius = float(rng.uniform(0.5, 3.0))
        anisotropy = tuple(float(rng.uniform(0.5, 4.0)) for _ in range(3))
        count, out = roll_invalidation_cube(labels.copy(), dbf, path, 0.0, radius, anisotropy=anisotropy)
        expected = set()
        for coord in path:
            expected |= _expected_corner_cube(coord, radius, shape, anisotropy)
        assert set(map(tuple, np.argwhere(out == 0).tolist())) == expected and count == len(expected), trial


def test_cube_singleton_and_dbf_layout_untouched():
    """automated_test.py:797-825: singleton volume; a C-ordered DBF with F-ordered labels is normalised
    without mutating the caller's array."""
    from kimimaro_amd.ops import roll_invalidation_cube
    L = np.ones((1, 1, 1), np.uint8)
    assert roll_invalidation_cube(L, np.zeros((1, 1, 1), np.float32), [(0, 0, 0)], 1.0, 1.0)[0] == 1
    rng = np.random.default_rng(2)
    Lf = np.asfortranarray(np.ones((9, 8, 7), np.uint8))
    Dc = np.ascontiguousarray(rng.uniform(0.8, 2.5, (9, 8, 7)).astype(np.float32))
    keep = Dc.copy()
    c, out = roll_invalidation_cube(Lf, Dc, [(4, 4, 3)], 1.0, 0.5)
    assert out is Lf and c > 0 and Dc.flags.c_contiguous
    np.testing.assert_array_equal(Dc, keep)
