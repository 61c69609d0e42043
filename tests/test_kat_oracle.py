"""The reference's own end-to-end known-answer tests (automated_test.py) run through the ORACLE
pipeline on CPU.  They pin the parts of the oracle whose arithmetic lives in third-party packages that
are absent from the reference tree (edt, dijkstra3d): exact vertex counts, cable lengths to 1e-3 and
exact straight-line vertex sets."""
import numpy as np
import pytest

from oracle import pipeline as P

TP = {  # automated_test.py:54-63
    "scale": 1.5, "const": 300, "pdrf_scale": 100000, "pdrf_exponent": 4,
    "soma_acceptance_threshold": 3500, "soma_detection_threshold": 750,
    "soma_invalidation_const": 300, "soma_invalidation_scale": 2,
}


def test_empty_image():  # automated_test.py:17-21
    assert len(P.skeletonize(np.zeros((64, 64, 64), dtype=bool), fix_borders=True)) == 0


def test_very_sparse_image():  # :23-31
    labels = np.zeros((64, 64, 64), dtype=bool)
    labels[5, 5, 5] = labels[6, 5, 5] = labels[20, 20, 20] = True
    skels = P.skeletonize(labels, dust_threshold=0)
    assert len(skels) == 1


def test_binary_image():  # :39-46 (256x256x3 in the reference)
    labels = np.ones((96, 96, 3), dtype=bool)
    labels[-1, 0] = 0
    labels[0, -1] = 0
    assert len(P.skeletonize(labels, fix_borders=False)) == 1


@pytest.mark.parametrize("corners", ["anti", "main"])
def test_square(corners):  # :48-87, full size
    labels = np.ones((1000, 1000), dtype=np.uint8)
    if corners == "anti":
        labels[-1, 0] = 0
        labels[0, -1] = 0
    else:
        labels[0, 0] = 0
        labels[-1, -1] = 0
    skels = P.skeletonize(labels, teasar_params=TP, fix_borders=False)
    assert len(skels) == 1
    skel = skels[1]
    assert skel.vertices.shape[0] == 1000
    assert skel.edges.shape[0] == 999
    assert abs(skel.cable_length() - 999 * np.sqrt(2)) < 0.001
    assert skel.space == "physical"


def test_cube():  # :89-102, full size
    labels = np.ones((128, 128, 128), dtype=np.uint8)
    labels[0, 0, 0] = 0
    labels[-1, -1, -1] = 0
    skels = P.skeletonize(labels, fix_borders=False)
    assert len(skels) == 1
    skel = skels[1]
    assert skel.vertices.shape[0] == 128
    assert skel.edges.shape[0] == 127
    assert abs(skel.cable_length() - 127 * np.sqrt(3)) < 0.001
    assert skel.space == "physical"


def test_fix_borders_z():  # :116-143 at half linear size (the full size runs on the GPU tier)
    labels = np.zeros((128, 128, 128), dtype=np.uint8)
    labels[32:98, 32:98, :] = 128
    skels = P.skeletonize(labels, teasar_params={"const": 250, "scale": 10, "pdrf_exponent": 4, "pdrf_scale": 100000},
                          anisotropy=(40, 32, 20), dust_threshold=1000, fix_branching=True, fix_borders=True)
    skel = skels[128].voxel_space()
    assert np.all(skel.vertices[:, 0] == skel.vertices[0, 0])
    assert np.all(skel.vertices[:, 1] == skel.vertices[0, 1])
    assert 63 <= skel.vertices[0, 0] <= 66 and 63 <= skel.vertices[0, 1] <= 66
    assert np.all(skel.vertices[:, 2] == np.arange(128))


def test_dimensions():  # :261-279
    labels = np.zeros((10,), dtype=bool)
    P.skeletonize(labels)
    labels = np.zeros((10, 10), dtype=bool)
    P.skeletonize(labels)
    labels = np.zeros((10, 10, 10, 1), dtype=bool)
    P.skeletonize(labels)
    with pytest.raises(P.DimensionError):
        P.skeletonize(np.ones((10, 10, 10, 2), dtype=bool), dust_threshold=0)


def test_extra_targets():  # :201-231 (smaller plate)
    labels = np.zeros((100, 100, 1), dtype=np.uint8)
    labels[10:90, 10:90, 0] = 1
    base = P.skeletonize(labels, TP, dust_threshold=0, fix_borders=False)[1]
    more = P.skeletonize(labels, TP, dust_threshold=0, fix_borders=False, extra_targets_after=[(10, 89, 0)])[1]
    assert more.vertices.shape[0] >= base.vertices.shape[0]
    pre = P.skeletonize(labels, TP, dust_threshold=0, fix_borders=False, extra_targets_before=[(10, 89, 0)])[1]
    assert pre.vertices.shape[0] >= base.vertices.shape[0]


def test_solid_image():  # automated_test.py:33-37 at 48^3 (single label: black_border EDT, targets on all six faces)
    labels = np.ones((48, 48, 48), dtype=bool)
    assert len(P.skeletonize(labels, fix_borders=True)) == 1


@pytest.mark.parametrize("axis", [0, 1])
def test_fix_borders_x_y(axis):  # :145-199 at 96^3: a bar through the volume along x / y
    labels = np.zeros((96, 96, 96), dtype=np.uint8)
    sl = [slice(24, 74)] * 3
    sl[axis] = slice(None)
    labels[tuple(sl)] = 128
    skels = P.skeletonize(labels, teasar_params={"const": 250, "scale": 10, "pdrf_exponent": 4, "pdrf_scale": 100000},
                          anisotropy=(1, 1, 1), dust_threshold=1000, fix_branching=True, fix_borders=True)
    v = skels[128].voxel_space().vertices
    others = [i for i in range(3) if i != axis]
    assert np.all(v[:, axis] == np.arange(96))
    for o in others:
        assert np.all(v[:, o] == v[0, o]) and 47 <= v[0, o] <= 50


def test_parallel_quadrants():  # :234-259 (four labels; the reference runs them in two processes)
    labels = np.zeros((64, 64, 32), dtype=np.uint8)
    labels[0:32, 0:32, :] = 1
    labels[32:64, 0:32, :] = 2
    labels[0:32, 32:64, :] = 3
    labels[32:64, 32:64, :] = 4
    assert len(P.skeletonize(labels, TP, dust_threshold=100, parallel=2)) == 4


def test_joinability():  # :281-333: with fix_borders two overlapping chunks share a face vertex
    from shapes import random_walk_tube
    vol = random_walk_tube((64, 48, 48), 314, steps=80, step=3.0, radius=(2.5, 5.0)).astype(np.uint32)
    params = dict(TP)
    params["const"] = 4
    sa = P.skeletonize(vol[:33], params, dust_threshold=50, fix_borders=True)
    sb = P.skeletonize(vol[32:], params, dust_threshold=50, fix_borders=True)
    if 1 in sa and 1 in sb:
        va = {tuple(v[1:]) for v in sa[1].vertices[sa[1].vertices[:, 0] == 32].tolist()}
        vb = {tuple(v[1:]) for v in sb[1].vertices[sb[1].vertices[:, 0] == 0].tolist()}
        if va and vb:
            assert va & vb


def test_fill_all_holes_swallows_enclosed_components():
    """oracle restatement of kimimaro/intake.py:747-795: an enclosed component and a background void are filled with
    the enclosing label, a cavity open to the array face is not, later labels that were swallowed are skipped."""
    from oracle import pipeline as P
    cc = np.zeros((20, 20, 20), dtype=np.uint32, order="F")
    cc[2:18, 2:18, 2:18] = 1
    cc[6:10, 6:10, 6:10] = 2          # enclosed component
    cc[12:14, 12:14, 12:14] = 0       # enclosed background void
    cc[8:10, 8:10, 0:5] = 0           # tunnel to the z = 0 face (and beyond label 1's box: stays open)
    cc[0:2, 0:2, 0:2] = 3             # separate small component
    out = P.fill_all_holes(cc.copy(order="F"))
    assert set(np.unique(out)) == {0, 1, 3}
    assert (out[6:10, 6:10, 6:10] == 1).all() and (out[12:14, 12:14, 12:14] == 1).all()
    assert (out[8:10, 8:10, 0:2] == 0).all()
    assert (out[0:2, 0:2, 0:2] == 3).all()


def test_fix_avocados_merges_pits_into_their_fruit():
    """kimimaro/intake.py:187-193, 600-704 through the oracle restatement: a nucleus with a label of its own disappears into its
    cell (also when the wall of the volume cuts both, and when a nucleolus sits inside the nucleus); nothing else changes."""
    from shapes import avocado_volume
    from oracle import pipeline as P
    lab = avocado_volume()
    params = dict(P.DEFAULT_TEASAR_PARAMS)
    params.update(scale=1.5, const=20, soma_detection_threshold=10.0, soma_acceptance_threshold=1e9)
    plain = P.skeletonize(lab, params, anisotropy=(4, 4, 4), dust_threshold=50, fix_borders=False)
    fixed = P.skeletonize(lab, params, anisotropy=(4, 4, 4), dust_threshold=50, fix_borders=False, fix_avocados=True)
    assert {12, 22, 32, 33} <= set(plain)                # the pits are objects of their own without the option ...
    assert set(fixed) == {11, 21, 31, 41, 51}            # ... and gone with it
    for k in (41, 51):
        np.testing.assert_array_equal(fixed[k].vertices, plain[k].vertices)
