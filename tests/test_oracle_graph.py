"""CPU checks of the oracle's voxel-graph restatements (oracle.color_connectivity_graph, oracle.edt_graph; cc3d and edt are absent
from the reference tree: PARITY UNPINNED) -- the properties any reading of those packages must have."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _graph_of_labels(lab):
    import oracle as K
    g = np.zeros(lab.shape, dtype=np.uint32, order="F")
    for k, d in enumerate(K._DIRS):
        src = tuple(slice(max(0, -c), n - max(0, c)) for c, n in zip(d, lab.shape))
        dst = tuple(slice(max(0, c), n - max(0, -c)) for c, n in zip(d, lab.shape))
        g[src] |= (lab[src] == lab[dst]).astype(np.uint32) << np.uint32(K._GRAPH_BIT[k])
    return g


def test_graph_of_the_labels_gives_the_labels_own_components():
    import oracle as K
    from shapes import voronoi_labels
    lab = np.asfortranarray(voronoi_labels((30, 26, 20), 9, seed=2))
    lab[12:14] = 0
    a, na = K.color_connectivity_graph(lab, _graph_of_labels(lab))
    b, nb = K.connected_components(lab)
    assert na == nb
    np.testing.assert_array_equal(a, b)


def test_a_wall_splits_a_label_and_sits_half_a_pitch_away():
    import oracle as K
    lab = np.ones((16, 6, 6), dtype=np.uint32, order="F")
    g = _graph_of_labels(lab)
    for k, d in enumerate(K._DIRS):
        if d[0] > 0:
            g[7] &= np.uint32(~(1 << K._GRAPH_BIT[k]) & 0xFFFFFFFF)
        if d[0] < 0:
            g[8] &= np.uint32(~(1 << K._GRAPH_BIT[k]) & 0xFFFFFFFF)
    cc, n = K.color_connectivity_graph(lab, g)
    assert n == 2 and (cc[:8] == 1).all() and (cc[8:] == 2).all()
    d = K.edt_graph(lab, g, (4, 4, 4), black_border=False)
    assert d[7, 3, 3] == 2.0 and d[8, 3, 3] == 2.0 and d[5, 3, 3] == 10.0
    # with a black border the array's faces are walls half a pitch away as well
    d = K.edt_graph(lab, g, (4, 4, 4), black_border=True)
    assert d[0, 3, 3] == 2.0 and d[15, 3, 3] == 2.0 and d[3, 0, 3] == 2.0 and d[3, 5, 3] == 2.0


def test_oracle_skeletonize_takes_a_graph():
    from oracle import pipeline as P
    from shapes import random_walk_tube
    a = random_walk_tube((40, 36, 30), 11, steps=22, step=2.6, radius=(2.5, 4.0))
    lab = np.asfortranarray(a.astype(np.uint32) * 7)
    g = _graph_of_labels(lab)
    params = dict(scale=3.0, const=2.0, pdrf_scale=5000, pdrf_exponent=4, soma_detection_threshold=1e9,
                  soma_acceptance_threshold=1e9, soma_invalidation_scale=1.0, soma_invalidation_const=0.0)
    out = P.skeletonize(lab, teasar_params=params, dust_threshold=20, fix_borders=False, voxel_graph=g)
    assert set(out) == {7} and len(out[7].vertices) > 5
