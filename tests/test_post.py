"""Row f4: kimimaro_amd.post against tests/golden/post.npz = outputs of the reference's own kimimaro/post.py (run in the
build container by tests/golden/make_golden.py `post`, with stand-ins for the packages it imports that this image
lacks).  Skeletons are compared as graphs on coordinates: the same vertex rows (with their radii) and the same set of
edges between them, whatever order the implementation lists them in."""
import ast
import os

import numpy as np
import pytest

from kimimaro_amd import post
from kimimaro_amd.skeleton import Skeleton

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post.npz"))
N = int(GOLD["n"])


def canonical(vertices, edges, radii):
    vertices = np.asarray(vertices, np.float32).reshape(-1, 3)
    edges = np.asarray(edges, np.int64).reshape(-1, 2)
    order = np.lexsort((vertices[:, 2], vertices[:, 1], vertices[:, 0]))
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    e = np.sort(rank[edges], axis=1) if len(edges) else edges
    e = np.unique(e, axis=0) if len(e) else e
    return vertices[order], np.asarray(radii, np.float32)[order], e


def run(i):
    fn = str(GOLD["fn_%d" % i])
    args = ast.literal_eval(str(GOLD["args_%d" % i]))
    skel = Skeleton(GOLD["vin_%d" % i].copy(), GOLD["ein_%d" % i].copy(), GOLD["rin_%d" % i].copy(), segid=7)
    if fn == "postprocess":
        return post.postprocess(skel, dust_threshold=args[0], tick_threshold=args[1])
    if fn == "remove_dust":
        return post.remove_dust(skel.consolidate(remove_disconnected_vertices=True), args[0])
    if fn == "remove_loops":
        return post.remove_loops(skel.consolidate(remove_disconnected_vertices=True))
    if fn == "remove_ticks":
        return post.remove_ticks(skel.consolidate(remove_disconnected_vertices=True), args[0])
    if fn == "join_close_components":
        return post.join_close_components(skel, radius=args[0], restrict_by_radius=args[1])
    raise AssertionError(fn)


@pytest.mark.parametrize("i", range(N))
def test_post_matches_reference_vector(i):
    got = run(i)
    gv, gr, ge = canonical(got.vertices, got.edges, got.radii)
    wv, wr, we = canonical(GOLD["vout_%d" % i], GOLD["eout_%d" % i], GOLD["rout_%d" % i])
    fn = str(GOLD["fn_%d" % i])
    assert gv.shape == wv.shape and np.array_equal(gv, wv), fn
    assert np.array_equal(gr, wr), fn
    assert ge.shape == we.shape and np.array_equal(ge, we), fn


def test_vectors_cover_every_rule():
    fns = [str(GOLD["fn_%d" % i]) for i in range(N)]
    assert {"postprocess", "remove_dust", "remove_loops", "remove_ticks", "join_close_components"} <= set(fns)
    # the vectors do change things (loops removed, components joined, ticks culled)
    changed = sum(int(GOLD["ein_%d" % i].shape[0] != GOLD["eout_%d" % i].shape[0]) for i in range(N))
    assert changed > N // 3


def test_postprocess_keeps_the_label_and_yields_a_forest():
    rng = np.random.default_rng(3)
    v = rng.uniform(0, 500, (40, 3)).astype(np.float32)
    e = np.stack([np.arange(39), np.arange(1, 40)], axis=1)
    e = np.concatenate([e, [[0, 20], [5, 30]]])           # two loops
    out = post.postprocess(Skeleton(v, e, np.full(40, 10, np.float32), segid=42), dust_threshold=0, tick_threshold=0)
    assert out.id == 42
    for comp in out.components():
        assert comp.edges.shape[0] == comp.vertices.shape[0] - 1
        assert post.find_cycle(comp.edges) == []


def test_join_rejects_bad_radius():
    with pytest.raises(ValueError):
        post.join_close_components(Skeleton(), radius=0)


# ---- the reference's own known-answer tests for this row (automated_test.py:335-456, 611-632), on the product
def test_reference_kat_find_cycle():
    cyc = post.find_cycle(np.array([[0, 1], [1, 2], [2, 0], [2, 3], [2, 4]], dtype=np.int32))
    assert cyc == [0, 2, 1, 0]
    cyc = post.find_cycle(np.array([[0, 1], [1, 2], [2, 3], [3, 4], [4, 10], [10, 11], [11, 12], [12, 2], [4, 5], [5, 6], [6, 7]],
                                   dtype=np.int32))
    assert cyc == [2, 12, 11, 10, 4, 3, 2]
    cyc = post.find_cycle(np.array([[0, 1], [0, 20], [20, 21], [21, 22], [22, 23], [23, 21], [1, 2], [2, 3], [3, 4], [4, 5], [5, 6],
                                    [6, 7], [7, 10], [10, 11], [11, 6]], dtype=np.int32))
    assert cyc in ([21, 23, 22, 21], [6, 11, 10, 7, 6])
    assert post.find_cycle(np.zeros((0, 2), np.int32)) == []


def test_reference_kat_join_close_components_simple():
    skel = Skeleton([(0, 0, 0), (1, 0, 0), (10, 0, 0), (11, 0, 0)], edges=[(0, 1), (2, 3)], radii=[0, 1, 2, 3],
                    vertex_types=[0, 1, 2, 3], segid=1337)
    assert len(skel.components()) == 2
    assert len(post.join_close_components(skel, radius=np.inf).components()) == 1
    res = post.join_close_components(skel, radius=9)
    assert len(res.components()) == 1
    assert np.all(res.edges == [[0, 1], [1, 2], [2, 3]])
    assert len(post.join_close_components(skel, radius=8.5).components()) == 2


def test_reference_kat_join_close_components_complex():
    skel = Skeleton([(0, 0, 0), (1, 0, 0), (4, 0, 0), (6, 0, 0), (20, 0, 0), (21, 0, 0), (0, 0, 5), (0, 0, 10)],
                    edges=[(0, 1), (2, 3), (4, 5), (6, 7)])
    assert len(skel.components()) == 4
    res = post.join_close_components(skel, radius=np.inf)
    assert len(res.components()) == 1
    assert np.all(res.edges == [[0, 1], [0, 3], [1, 2], [3, 4], [4, 5], [5, 6], [6, 7]])


def test_reference_kat_join_close_components_by_radius():
    skel = Skeleton([(0, 0, 0), (1, 0, 0), (5, 0, 0), (11, 0, 0)], edges=[(0, 1), (2, 3)], radii=[100, 100, 100, 100],
                    vertex_types=[0, 1, 2, 3], segid=1337)
    for restrict in (False, True):
        res = post.join_close_components(skel, restrict_by_radius=restrict)
        assert len(res.components()) == 1
        assert np.all(res.edges == [[0, 1], [1, 2], [2, 3]])
    for radii, ncomp, edges in (([1, 1, 1, 1], 2, [[0, 1], [2, 3]]), ([1, 0.9, 3, 1], 2, [[0, 1], [2, 3]]),
                                ([1, 1, 3, 1], 1, [[0, 1], [1, 2], [2, 3]])):
        skel.radii = np.array(radii, dtype=np.float32)
        res = post.join_close_components(skel, restrict_by_radius=True)
        assert len(res.components()) == ncomp
        assert np.all(res.edges == edges)


def test_reference_kat_postprocess():
    skel = Skeleton([(0, 0, 0), (1, 0, 0), (4, 0, 0), (6, 0, 0), (20, 0, 0), (21, 0, 0), (0, 0, 5), (0, 0, 10)],
                    edges=[(0, 1), (2, 3), (4, 5), (6, 7), (0, 7), (1, 6)])
    res = post.postprocess(skel, dust_threshold=0, tick_threshold=0)
    v, r, e = canonical(res.vertices, res.edges, res.radii)
    wv, wr, we = canonical([(4, 0, 0), (6, 0, 0), (20, 0, 0), (21, 0, 0)], [(0, 1), (2, 3)], [-1] * 4)
    assert np.array_equal(v, wv) and np.array_equal(e, we)


def test_reference_kat_remove_row():
    """automated_test.py:566-586 (post.remove_row): every row equal to a doomed row goes, in either orientation."""
    arr = np.array([[0, 1], [1, 2], [2, 1], [2, 2], [2, 3], [3, 4]])
    assert np.array_equal(post._drop_edges(arr, np.array([[1, 2]])), [[0, 1], [2, 2], [2, 3], [3, 4]])
    assert post._drop_edges(np.zeros((0, 2), np.int64), np.array([[1, 2]])).size == 0
