"""The reference's end-to-end known-answer tests (automated_test.py) through the HIP product path, at the
reference's own sizes (test_square 1000 x 1000, test_cube 128^3, test_fix_borders_{z,x,y} 256^3, test_joinability
256 x 256 x 20) with its own assertions; where the oracle pipeline is fast enough the result is additionally
compared with it bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TP = {"scale": 1.5, "const": 300, "pdrf_scale": 100000, "pdrf_exponent": 4, "soma_acceptance_threshold": 3500,
      "soma_detection_threshold": 750, "soma_invalidation_const": 300, "soma_invalidation_scale": 2}


@pytest.fixture(scope="module")
def eng():
    from kimimaro_amd.engine import Engine
    return Engine()


def same(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        np.testing.assert_array_equal(a[k].vertices, b[k].vertices)
        np.testing.assert_array_equal(a[k].edges, b[k].edges)
        np.testing.assert_allclose(a[k].radii, b[k].radii, rtol=1e-4)


def test_empty_and_sparse(eng):  # automated_test.py:17-31
    import kimimaro_amd
    assert kimimaro_amd.skeletonize(np.zeros((64, 64, 64), dtype=bool), fix_borders=True, _engine=eng) == {}
    labels = np.zeros((64, 64, 64), dtype=bool)
    labels[5, 5, 5] = labels[6, 5, 5] = labels[20, 20, 20] = True
    skels = kimimaro_amd.skeletonize(labels, dust_threshold=0, _engine=eng)
    assert len(skels) == 1  # single voxels don't get skeletonized


@pytest.mark.parametrize("corners", ["anti", "main"])
def test_square(eng, corners):  # :48-87
    import kimimaro_amd
    from oracle import pipeline as P
    n = 1000
    labels = np.ones((n, n), dtype=np.uint8)
    if corners == "anti":
        labels[-1, 0] = 0
        labels[0, -1] = 0
    else:
        labels[0, 0] = 0
        labels[-1, -1] = 0
    skels = kimimaro_amd.skeletonize(labels, teasar_params=TP, fix_borders=False, _engine=eng)
    assert len(skels) == 1
    skel = skels[1]
    assert skel.vertices.shape[0] == n
    assert skel.edges.shape[0] == n - 1
    assert abs(skel.cable_length() - (n - 1) * np.sqrt(2)) < 0.001
    assert skel.space == "physical"
    same(skels, P.skeletonize(labels, teasar_params=TP, fix_borders=False))


def test_cube(eng):  # :89-102
    import kimimaro_amd
    from oracle import pipeline as P
    n = 128
    labels = np.ones((n, n, n), dtype=np.uint8)
    labels[0, 0, 0] = 0
    labels[-1, -1, -1] = 0
    skels = kimimaro_amd.skeletonize(labels, fix_borders=False, _engine=eng)
    skel = skels[1]
    assert skel.vertices.shape[0] == n
    assert skel.edges.shape[0] == n - 1
    assert abs(skel.cable_length() - (n - 1) * np.sqrt(3)) < 0.001
    same(skels, P.skeletonize(labels, fix_borders=False))


def test_solid_image_fix_borders(eng):  # :33-37 at 40^3 (black_border EDT, border targets on all faces)
    import kimimaro_amd
    from oracle import pipeline as P
    labels = np.ones((40, 40, 40), dtype=bool)
    skels = kimimaro_amd.skeletonize(labels, fix_borders=True, _engine=eng)
    assert len(skels) == 1
    same(skels, P.skeletonize(labels, fix_borders=True))


FB = dict(teasar_params={"const": 250, "scale": 10, "pdrf_exponent": 4, "pdrf_scale": 100000},
          dust_threshold=1000, fix_branching=True, fix_borders=True)


def test_fix_borders_z(eng):  # :116-143, the reference's exact expectation: a straight line through (129, 129, *)
    import kimimaro_amd
    labels = np.zeros((256, 256, 256), dtype=np.uint8)
    labels[64:196, 64:196, :] = 128
    skels = kimimaro_amd.skeletonize(labels, anisotropy=(40, 32, 20), _engine=eng, **FB)
    skel = skels[128]
    assert skel.space == "physical"
    skel = skel.voxel_space()
    assert np.all(skel.vertices[:, 0] == 129)
    assert np.all(skel.vertices[:, 1] == 129)
    assert np.all(skel.vertices[:, 2] == np.arange(256))
    assert skel.space == "voxel"


@pytest.mark.parametrize("axis", [0, 1])
def test_fix_borders_x_y(eng, axis):  # :145-199
    import kimimaro_amd
    labels = np.zeros((256, 256, 256), dtype=np.uint8)
    if axis == 0:
        labels[:, 64:196, 64:196] = 128
    else:
        labels[64:196, :, 64:196] = 128
    skel = kimimaro_amd.skeletonize(labels, anisotropy=(1, 1, 1), _engine=eng, **FB)[128]
    for a in range(3):
        want = np.arange(256) if a == axis else 129
        assert np.all(skel.vertices[:, a] == want), a


def test_fix_borders_z_matches_oracle(eng):  # the same shape at 96^3, bit exact against the oracle pipeline
    import kimimaro_amd
    from oracle import pipeline as P
    labels = np.zeros((96, 96, 96), dtype=np.uint8)
    labels[24:74, 24:74, :] = 128
    skels = kimimaro_amd.skeletonize(labels, anisotropy=(40, 32, 20), _engine=eng, **FB)
    skel = skels[128].voxel_space()
    assert np.all(skel.vertices[:, 0] == skel.vertices[0, 0])
    assert np.all(skel.vertices[:, 1] == skel.vertices[0, 1])
    assert np.all(skel.vertices[:, 2] == np.arange(96))
    same(skels, P.skeletonize(labels, anisotropy=(40, 32, 20), **FB))


def test_parallel_quadrants(eng):  # :234-259 (4 labels; `parallel` has no meaning on one GPU)
    import kimimaro_amd
    labels = np.zeros((64, 64, 32), dtype=np.uint8)
    labels[0:32, 0:32, :] = 1
    labels[32:64, 0:32, :] = 2
    labels[0:32, 32:64, :] = 3
    labels[32:64, 32:64, :] = 4
    skels = kimimaro_amd.skeletonize(labels, TP, dust_threshold=100, parallel=2, _engine=eng)
    assert len(skels) == 4


def test_dimensions_and_object_ids(eng):  # :261-279
    import kimimaro_amd
    kimimaro_amd.skeletonize(np.zeros((10,), dtype=bool), _engine=eng)
    kimimaro_amd.skeletonize(np.zeros((10, 10), dtype=bool), _engine=eng)
    kimimaro_amd.skeletonize(np.zeros((10, 10, 10, 1), dtype=bool), _engine=eng)
    with pytest.raises(kimimaro_amd.DimensionError):
        kimimaro_amd.skeletonize(np.ones((10, 10, 10, 2), dtype=bool), dust_threshold=0, _engine=eng)
    labels = np.zeros((40, 40, 20), dtype=np.uint32)
    labels[:20] = 7
    labels[20:] = 9
    skels = kimimaro_amd.skeletonize(labels, TP, dust_threshold=10, object_ids=[9], _engine=eng)
    assert list(skels.keys()) == [9]


@pytest.mark.parametrize("axis", ["x", "y"])
def test_joinability(eng, axis):  # :281-333: two chunks with a 1-voxel overlap meet at the same face voxel iff fix_borders
    import kimimaro_amd
    from kimimaro_amd.skeleton import Skeleton

    def run(labels, fix_borders):
        return kimimaro_amd.skeletonize(labels, {"const": 10, "scale": 10, "pdrf_exponent": 4, "pdrf_scale": 100000},
                                        anisotropy=(1, 1, 1), dust_threshold=0, fix_branching=True, fix_borders=fix_borders,
                                        _engine=eng)

    labels = np.zeros((256, 256, 20), dtype=np.uint8)
    if axis == "x":
        labels[32:160, :, :] = 1
    else:
        labels[:, 32:160, :] = 1

    def both(fix_borders):
        s1 = run(labels[:, :, :10], fix_borders)[1]
        s2 = run(labels[:, :, 9:], fix_borders)[1]
        s2.vertices[:, 2] += 9
        return Skeleton.simple_merge([s1, s2]).consolidate()

    fb = both(True)
    assert len(fb.components()) == 1
    plain = both(False)
    assert not (fb.vertices.shape == plain.vertices.shape and np.array_equal(fb.vertices, plain.vertices)
                and np.array_equal(fb.edges, plain.edges))


# (BASELINE config 2 at full size: tests/test_gpu_configs.py compares every c2 skeleton with the pooled oracle)
