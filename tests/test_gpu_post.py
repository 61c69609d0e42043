"""Row f4 under GPU evidence: the output side of the path -- kimimaro_amd.post.postprocess (kimimaro/post.py:49-87) and the
Skeleton wire formats (to_swc, to_precomputed / from_precomputed; kimimaro_cli/__init__.py:104-107) -- fed with what the HIP
path itself produces for a 128^3 dense volume, and compared with the same steps applied to the oracle pipeline's skeletons.
The reference-made expectations of post.py (tests/golden/post.npz, made by tests/golden/make_golden.py from the reference's
own kimimaro/post.py) are replayed in this tier too, so the row is pinned in the same driver run as the kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AN = (16, 16, 40)
TP = {"scale": 1.5, "const": 300, "pdrf_scale": 100000, "pdrf_exponent": 4, "soma_acceptance_threshold": 3500,
      "soma_detection_threshold": 1100, "soma_invalidation_const": 300, "soma_invalidation_scale": 2}


@pytest.fixture(scope="module")
def eng():
    from kimimaro_amd.engine import Engine
    return Engine()


@pytest.fixture(scope="module")
def both(eng):
    """skeletonize() of the same 128^3 tessellation by the HIP path and by the oracle pipeline"""
    import kimimaro_amd
    from oracle import pipeline as P
    from shapes import voronoi_labels
    lab = voronoi_labels((128, 128, 128), 60, 11, pts_per_label=6, step=10.0, anisotropy=AN)
    kw = dict(teasar_params=TP, anisotropy=AN, dust_threshold=500, fix_borders=True, fix_branching=True)
    hip = kimimaro_amd.skeletonize(lab, _engine=eng, **kw)
    ora = P.skeletonize(lab, **kw)
    return hip, ora


def canonical(skel):
    v = np.asarray(skel.vertices, np.float32).reshape(-1, 3)
    e = np.asarray(skel.edges, np.int64).reshape(-1, 2)
    order = np.lexsort((v[:, 2], v[:, 1], v[:, 0]))
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    e = np.unique(np.sort(rank[e], axis=1), axis=0) if len(e) else e
    return v[order], np.asarray(skel.radii, np.float32)[order], e


def test_hip_output_equals_oracle_output(both):
    hip, ora = both
    assert len(hip) >= 40 and sorted(hip) == sorted(ora)
    for k in hip:
        np.testing.assert_array_equal(hip[k].vertices, ora[k].vertices)
        np.testing.assert_array_equal(hip[k].edges, ora[k].edges)
        np.testing.assert_allclose(hip[k].radii, ora[k].radii, rtol=1e-4)


def test_postprocess_of_hip_skeletons(both):
    """post.postprocess on the HIP skeletons == on the oracle's (the same graphs go in, the same forests must come out),
    every result is a forest, and the rules really fire on this volume (ticks culled, dust dropped)"""
    from kimimaro_amd import post
    from kimimaro_amd.skeleton import Skeleton
    hip, ora = both
    changed = 0
    for k in sorted(hip):
        as_product = lambda s: Skeleton(np.array(s.vertices), np.array(s.edges), np.array(s.radii), segid=k)
        a = post.postprocess(as_product(hip[k]), dust_threshold=1000, tick_threshold=1500)
        b = post.postprocess(as_product(ora[k]), dust_threshold=1000, tick_threshold=1500)
        for x, y in zip(canonical(a), canonical(b)):
            np.testing.assert_array_equal(x, y)
        assert a.id == k
        for comp in a.components():
            assert comp.edges.shape[0] == comp.vertices.shape[0] - 1 and post.find_cycle(comp.edges) == []
        changed += int(a.vertices.shape[0] != hip[k].vertices.shape[0])
    assert changed > 0


def test_wire_formats_round_trip(both):
    """to_precomputed -> from_precomputed returns the same arrays; to_swc lists every vertex once with its parent"""
    from kimimaro_amd.skeleton import Skeleton
    hip, _ = both
    for k in sorted(hip):
        s = hip[k]
        back = Skeleton.from_precomputed(s.to_precomputed(), segid=k)
        np.testing.assert_array_equal(back.vertices, np.asarray(s.vertices, np.float32))
        np.testing.assert_array_equal(back.edges, np.asarray(s.edges, np.uint32))
        np.testing.assert_array_equal(back.radii, np.asarray(s.radii, np.float32))
        rows = [l.split() for l in s.to_swc().splitlines() if l and not l.startswith("#")]
        assert len(rows) == s.vertices.shape[0]
        ids = [int(r[0]) for r in rows]
        assert sorted(ids) == list(range(1, len(rows) + 1))
        roots = sum(1 for r in rows if int(r[6]) == -1)
        assert roots == len(s.components())
        # one parent link per edge: a forest of len(components) trees
        assert len(rows) - roots == s.edges.shape[0]
        xyz = np.array([[float(r[2]), float(r[3]), float(r[4])] for r in rows], np.float32)
        got = xyz[np.lexsort((xyz[:, 2], xyz[:, 1], xyz[:, 0]))]
        v = np.asarray(s.vertices, np.float32)
        np.testing.assert_allclose(got, v[np.lexsort((v[:, 2], v[:, 1], v[:, 0]))], rtol=1e-6)


def test_reference_made_post_vectors_in_this_tier():
    """the 166 vectors made by the reference's own kimimaro/post.py (tests/golden/post.npz), replayed where the driver's
    GPU run records them (the checks themselves are tests/test_post.py's)"""
    import test_post as T
    for i in range(T.N):
        T.test_post_matches_reference_vector(i)
