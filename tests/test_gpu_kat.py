"""The reference's end-to-end known-answer tests (automated_test.py) through the HIP product path.
Sizes are reduced where one huge single label would make the (sequential per label) invalidation take
minutes; every case is additionally compared with the oracle pipeline (bit exact)."""
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
def test_square(eng, corners):  # :48-87 at 320x320
    import kimimaro_amd
    from oracle import pipeline as P
    n = 320
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


def test_cube(eng):  # :89-102 at 48^3
    import kimimaro_amd
    from oracle import pipeline as P
    n = 48
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


def test_fix_borders_z(eng):  # :116-143 at 96^3
    import kimimaro_amd
    from oracle import pipeline as P
    labels = np.zeros((96, 96, 96), dtype=np.uint8)
    labels[24:74, 24:74, :] = 128
    kw = dict(teasar_params={"const": 250, "scale": 10, "pdrf_exponent": 4, "pdrf_scale": 100000},
              anisotropy=(40, 32, 20), dust_threshold=1000, fix_branching=True, fix_borders=True)
    skels = kimimaro_amd.skeletonize(labels, _engine=eng, **kw)
    skel = skels[128].voxel_space()
    assert np.all(skel.vertices[:, 0] == skel.vertices[0, 0])
    assert np.all(skel.vertices[:, 1] == skel.vertices[0, 1])
    assert np.all(skel.vertices[:, 2] == np.arange(96))
    same(skels, P.skeletonize(labels, **kw))


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


def test_joinability(eng):  # :281-333: two chunks with a 1-voxel overlap meet at the same face voxel iff fix_borders
    import kimimaro_amd
    from shapes import random_walk_tube
    vol = random_walk_tube((64, 48, 48), 314, steps=80, step=3.0, radius=(2.5, 5.0)).astype(np.uint32)
    a, b = vol[:33], vol[32:]
    params = dict(TP)
    params["const"] = 4
    sa = kimimaro_amd.skeletonize(a, params, dust_threshold=50, fix_borders=True, _engine=eng)
    sb = kimimaro_amd.skeletonize(b, params, dust_threshold=50, fix_borders=True, _engine=eng)
    if 1 in sa and 1 in sb:
        va = sa[1].vertices[sa[1].vertices[:, 0] == 32][:, 1:]
        vb = sb[1].vertices[sb[1].vertices[:, 0] == 0][:, 1:]
        if len(va) and len(vb):
            assert {tuple(v) for v in va.tolist()} & {tuple(v) for v in vb.tolist()}


def test_full_size_c2_properties(eng):
    """BASELINE config 2 (512x512x100, 333 chains, anisotropy (16,16,40)) at full size: size independent
    properties of the whole product path (the oracle would need minutes here)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import kimimaro_amd
    from kimimaro_amd import intake
    lab, an = bench.make_volume("c2")
    skels = kimimaro_amd.skeletonize(lab, anisotropy=an, dust_threshold=1000, fix_borders=True, progress=False, _engine=eng)
    ids, counts = np.unique(lab, return_counts=True)
    assert set(skels.keys()) <= set(ids.tolist())
    assert len(skels) >= 0.95 * np.count_nonzero(counts > 1000)
    d_cc, n, _ = eng.ccl(np.asfortranarray(lab))
    dbf = eng.edt(d_cc, 4, lab.shape, an, False).cpu().numpy().reshape(lab.shape, order="F")
    anf = np.asarray(an, dtype=np.float32)
    total = 0
    for k, s in skels.items():
        v = np.rint(s.vertices / anf).astype(np.int64)
        assert s.space == "physical" and s.id == k
        np.testing.assert_array_equal(np.multiply(v.astype(np.float32), anf, dtype=np.float32), s.vertices)  # lattice points
        assert (lab[v[:, 0], v[:, 1], v[:, 2]] == k).all()                   # every vertex lies inside its label
        np.testing.assert_array_equal(s.radii, dbf[v[:, 0], v[:, 1], v[:, 2]])  # radii are the DBF there
        e = s.edges.astype(np.int64)
        assert (np.abs(v[e[:, 0]] - v[e[:, 1]]).max(axis=1) == 1).all()      # edges join 26-neighbours
        assert len(np.unique(e, axis=0)) == len(e) and (e[:, 0] < e[:, 1]).all()
        total += len(v)
    assert total > 10000
