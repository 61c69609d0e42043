"""GPU parity: kh_edt (HIP) vs the oracle restatement (bit exact), through the C ABI."""
import numpy as np
import pytest

from shapes import random_walk_tube, voronoi_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from kimimaro_amd.engine import Engine
    return Engine()


def gpu_edt(eng, labels, anisotropy, black_border):
    lab = np.asfortranarray(labels)
    d = eng.to_device(lab)
    out = eng.edt(d, lab.dtype.itemsize, lab.shape, anisotropy, black_border)
    return out.cpu().numpy().reshape(lab.shape, order="F")


CASES = [
    ((64, 64, 64), 8, (1, 1, 1), np.uint32),
    ((70, 33, 21), 12, (16, 16, 40), np.uint32),
    ((130, 40, 17), 30, (4, 4, 40), np.uint16),
    ((65, 64, 5), 5, (40, 32, 20), np.uint8),
    ((200, 3, 2), 4, (3.7, 1.3, 2.2), np.uint32),     # non-integer anisotropy: still bit exact
    ((31, 1, 1), 3, (2, 2, 2), np.uint32),            # 1-D
    ((50, 60, 1), 6, (1, 2, 1), np.uint32),           # 2-D
    ((90, 120, 40), 20, (2, 9, 3), np.uint32),        # coarse y: the small-halo kernel on a non-final pass
    ((96, 150, 140), 2, (1, 1, 1), np.uint32),        # runs far longer than any halo: several bands per tile
    ((70, 210, 150), 1, (1, 2, 1), np.uint16),        # one label: bands three and more deep, partial tiles on every axis
    ((1100, 7, 3), 9, (2, 3, 5), np.uint16),          # rows longer than the x pass keeps in registers (8 x 64 voxels)
]


@pytest.mark.parametrize("shape,nlab,an,dtype", CASES)
@pytest.mark.parametrize("black_border", [False, True])
def test_edt_multilabel_bit_exact(eng, shape, nlab, an, dtype, black_border):
    import oracle
    lab = voronoi_labels(shape, nlab, seed=sum(shape), anisotropy=(1, 1, 1), dtype=np.uint32)
    lab = (lab % 250 + 1).astype(dtype) if dtype == np.uint8 else lab.astype(dtype)
    # punch background holes
    rng = np.random.default_rng(1)
    lab[rng.random(shape) < 0.05] = 0
    lab = np.asfortranarray(lab)
    got = gpu_edt(eng, lab, an, black_border)
    want = oracle.edt(lab, an, black_border)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("knobs", [{"KH_EDT_H": "8"}, {"KH_EDT_H": "16"}, {"KH_EDT_H": "24"}, {"KH_EDT_H": "32"}, {"KH_EDT_H": "40"}, {"KH_EDT_CHUNK": "1"}, {"KH_EDT_CHUNK": "16"},
                                   {"KH_EDT_H": "8", "KH_EDT_CHUNK": "3"}])
def test_edt_developer_knobs_do_not_change_the_result(eng, knobs, monkeypatch):
    """the other halo instances of edt_axis_kernel (8 and 40 rows: more / fewer bands per tile) and other numbers of tiles per block
    (the knobs of csrc/edt.hip's edt_impl, read at every call) give the same bits as the oracle"""
    import oracle
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for shape, nlab, an, bb in (((96, 150, 140), 2, (1, 1, 1), False), ((70, 133, 90), 9, (16, 16, 40), True)):
        lab = voronoi_labels(shape, nlab, seed=sum(shape), anisotropy=(1, 1, 1), dtype=np.uint32).astype(np.uint16)
        lab[np.random.default_rng(3).random(shape) < 0.03] = 0
        lab = np.asfortranarray(lab)
        np.testing.assert_array_equal(gpu_edt(eng, lab, an, bb), oracle.edt(lab, an, bb))


def test_edt_solid_black_border(eng):
    import oracle
    lab = np.ones((40, 50, 30), dtype=np.uint32, order="F")
    got = gpu_edt(eng, lab, (16, 16, 40), True)
    want = oracle.edt(lab, (16, 16, 40), True)
    np.testing.assert_array_equal(got, want)
    assert got.max() > 0 and np.isfinite(got).all()


def test_edt_solid_without_border_is_infinite(eng):
    """no label change and no border anywhere: every band of every tile runs off the volume, the result is +inf (oracle)"""
    import oracle
    lab = np.full((70, 130, 90), 3, dtype=np.uint16, order="F")
    got = gpu_edt(eng, lab, (16, 16, 40), False)
    want = oracle.edt(lab, (16, 16, 40), False)
    np.testing.assert_array_equal(got, want)
    assert np.isinf(got).all()
    lab[10, 100, 80] = 0      # one hole: finite everywhere, windows as wide as the volume
    got = gpu_edt(eng, lab, (16, 16, 40), False)
    np.testing.assert_array_equal(got, oracle.edt(lab, (16, 16, 40), False))
    assert np.isfinite(got).all()


def test_edt_single_label_matches_scipy(eng):
    from scipy import ndimage
    m = random_walk_tube((48, 40, 36), 3)
    got = gpu_edt(eng, m.astype(np.uint32), (16, 16, 40), False)
    ref = ndimage.distance_transform_edt(m, sampling=(16, 16, 40))
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-4)


def test_edt_full_size_properties(eng):
    """BASELINE config-2 size (512x512x100): size independent properties + sampled brute force."""
    shape = (512, 512, 100)
    an = (16, 16, 40)
    lab = voronoi_labels(shape, 333, seed=2, pts_per_label=12, step=24.0, anisotropy=an)
    got = gpu_edt(eng, lab, an, False)
    assert np.isfinite(got).all() and (got > 0).all()       # every voxel is foreground
    assert got.min() >= 16.0                                 # at least one voxel from a boundary
    # 1-Lipschitz along x inside a label: |d(x+1)-d(x)| <= wx where labels agree
    same = lab[1:, :, :] == lab[:-1, :, :]
    assert (np.abs(got[1:, :, :] - got[:-1, :, :])[same] <= 16.0 + 1e-3).all()
    # sampled exact check against brute force over a local window
    rng = np.random.default_rng(0)
    pts = np.stack([rng.integers(24, s - 24, 64) for s in shape[:2]] + [rng.integers(10, 90, 64)], axis=1)
    for x, y, z in pts:
        L = lab[x, y, z]
        r = int(np.ceil(got[x, y, z] / 16.0)) + 1
        rz = int(np.ceil(got[x, y, z] / 40.0)) + 1
        x0, x1, y0, y1 = max(0, x - r), min(512, x + r + 1), max(0, y - r), min(512, y + r + 1)
        z0, z1 = max(0, z - rz), min(100, z + rz + 1)
        sub = lab[x0:x1, y0:y1, z0:z1] != L
        gx, gy, gz = np.nonzero(sub)
        d2 = ((gx + x0 - x) * 16.0) ** 2 + ((gy + y0 - y) * 16.0) ** 2 + ((gz + z0 - z) * 40.0) ** 2
        assert np.isclose(np.sqrt(d2.min()), got[x, y, z], rtol=1e-6)


@pytest.mark.parametrize("an", [(1, 1), (40, 32), (16, 40)])
def test_edt_2d_black_border(eng, an):
    """kh_edt_nd with ndim = 2: a 2-D transform, bit-exact vs the oracle (which is pinned to scipy's EDT of the padded
    plane in tests/test_oracle_golden.py); a z pass over the missing axis would cap every value at wz."""
    import oracle
    rng = np.random.default_rng(11)
    lab = np.zeros((70, 45), np.uint32, order="F")
    lab[3:60, 5:40] = 1
    lab[20:30, :] = 2
    lab[rng.random(lab.shape) < 0.02] = 0
    d = eng.to_device(lab[..., np.newaxis])
    out = eng.edt(d, 4, (lab.shape[0], lab.shape[1], 1), (an[0], an[1], 1.0), True, ndim=2)
    got = out.cpu().numpy().reshape(lab.shape, order="F")
    want = oracle.edt(lab, an, True)
    np.testing.assert_array_equal(got, want)
    assert got.max() > 3 * min(an)


def test_border_targets_match_oracle_on_nonconvex_faces(eng):
    """kimimaro_amd.border.compute_border_targets (2-D CCL + GPU 2-D EDT + host arg-max) against the oracle's
    independent restatement (oracle/border.py) on faces with L- and ring-shaped components."""
    import oracle
    from kimimaro_amd.border import compute_border_targets
    from oracle import border as B
    lab = np.zeros((40, 36, 12), np.uint32, order="F")
    lab[1:36, 2:9, :] = 7
    lab[25:36, 2:30, :] = 7
    lab[3:20, 14:34, 2:12] = 9
    lab[8:15, 19:29, 2:12] = 0      # ring on the z = -1 face
    an = (16, 16, 40)
    cc, _ = oracle.connected_components(lab)
    got = compute_border_targets(cc, an, eng=eng)
    want = B.compute_border_targets(cc, an, oracle.edt, oracle.connected_components)
    assert sorted(got.keys()) == sorted(want.keys()) and len(want) >= 2
    for k in want:
        assert [tuple(p) for p in got[k].tolist()] == [tuple(p) for p in want[k].tolist()], k
