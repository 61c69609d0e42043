"""GPU parity for row f1: kh_ccl26 (26-connected multi-label connected components, numbered by first
appearance) vs the oracle's union-find (ko_ccl26) and the host helper, bit exact."""
import numpy as np
import pytest

from shapes import voronoi_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from kimimaro_amd.engine import Engine
    return Engine()


@pytest.mark.parametrize("shape,dtype,seed", [
    ((64, 48, 40), np.uint32, 0), ((70, 33, 21), np.uint16, 1), ((33, 65, 17), np.uint8, 2),
    ((300, 5, 3), np.uint64, 3), ((40, 40, 1), np.uint32, 4), ((17, 1, 1), np.uint32, 5),
])
def test_ccl_matches_oracle(eng, shape, dtype, seed):
    import oracle
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 4, size=shape).astype(dtype)          # salt-and-pepper: thousands of tiny components
    lab[rng.random(shape) < 0.3] = 0
    lab = np.asfortranarray(lab)
    want, n_want = oracle.connected_components(lab)
    d_cc, n, rep = eng.ccl(lab)
    got = eng.to_host_volume(d_cc, shape)
    assert n == n_want
    np.testing.assert_array_equal(got, want)
    # representative = smallest linear index of each component
    flat = want.reshape(-1, order="F")
    first = np.full(n + 1, flat.size, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(flat.size))
    np.testing.assert_array_equal(rep[1:], first[1:])


def test_ccl_blobs_and_host_helper(eng):
    from kimimaro_amd import intake
    lab = voronoi_labels((96, 80, 40), 30, seed=9, pts_per_label=4, step=9.0)
    lab[40:44] = 0  # a background slab splits labels into several components
    d_cc, n, rep = eng.ccl(np.asfortranarray(lab))
    cc_host, n_host, remap_host = intake.compute_cc_labels(np.asfortranarray(lab))
    assert n == n_host
    np.testing.assert_array_equal(eng.to_host_volume(d_cc, lab.shape), cc_host)
    _, _, remap = intake.compute_cc_labels_device(eng, np.asfortranarray(lab))
    assert remap == remap_host


def test_faces_and_point_lookup(eng):
    from kimimaro_amd.intake import LazyVolume
    rng = np.random.default_rng(3)
    vol = np.asfortranarray(rng.integers(0, 1000, (13, 11, 7)).astype(np.uint32))
    lv = LazyVolume(eng, eng.to_device(vol), vol.shape)
    f = lv.faces()
    for a, b in zip(f, (vol[:, :, 0], vol[:, :, -1], vol[:, 0, :], vol[:, -1, :], vol[0, :, :], vol[-1, :, :])):
        np.testing.assert_array_equal(a, b)
    assert lv[(5, 6, 3)] == vol[5, 6, 3]
    np.testing.assert_array_equal(lv.host(), vol)


@pytest.mark.parametrize("shape,seed", [((40, 36, 30), 0), ((65, 20, 9), 1), ((33, 47, 1), 2), ((1, 50, 40), 3), ((70, 70, 70), 4)])
def test_fill_voids_matches_binary_fill_holes(eng, shape, seed):
    """kh_fill_voids (row f3) against scipy.ndimage.binary_fill_holes: closed voids of every size are filled, voids
    open to a face of the array and diagonal-only leaks (6-connected background) are handled alike."""
    import scipy.ndimage
    import torch
    rng = np.random.default_rng(seed)
    m = rng.random(shape) < 0.62                      # porous: many small voids, many leaks
    c = [s // 2 for s in shape]
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), axis=-1)
    r = np.sqrt(((g - c) ** 2).sum(-1))
    R = max(2, min(shape) // 2 - 2)
    m |= (r <= R) & (r >= R - 1.5)                    # a shell with a big closed void (if the array is thick enough)
    m = np.asfortranarray(m)
    want = scipy.ndimage.binary_fill_holes(m)
    d = torch.from_numpy(m.reshape(-1, order="F").astype(np.uint8)).cuda()
    d_out, n = eng.fill_voids(d, shape)
    got = d_out.cpu().numpy().reshape(shape, order="F").astype(bool)
    np.testing.assert_array_equal(got, want)
    assert n == int(want.sum() - m.sum())


def test_ccl_one_huge_component_repeatable(eng):
    """a single 10^6-voxel component (the reference's test_square plate): every voxel gets the component's id, run
    after run (a path-halving store landing after the flatten pass once left a few voxels with label 0)."""
    lab = np.ones((1000, 1000, 1), dtype=np.uint8, order="F")
    lab[-1, 0] = 0
    lab[0, -1] = 0
    for _ in range(4):
        d_cc, n, _ = eng.ccl(lab)
        cc = d_cc.cpu().numpy().view(np.uint32)
        assert n == 1
        assert int(np.count_nonzero(cc == 1)) == 999998 and int(np.count_nonzero(cc == 0)) == 2
