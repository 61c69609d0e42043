"""Host-side logic of the product (no GPU): Skeleton semantics, fast consolidation, format_labels."""
import numpy as np
import pytest


def test_consolidate_paths_equals_generic_consolidate():
    from kimimaro_amd.intake import consolidate_paths
    from kimimaro_amd.skeleton import Skeleton
    rng = np.random.default_rng(0)
    shape = (23, 17, 11)
    for trial in range(30):
        npaths = int(rng.integers(1, 6))
        lens = rng.integers(1, 12, npaths)
        locs = rng.integers(0, np.prod(shape), int(lens.sum())).astype(np.int64)
        if trial % 3 == 0 and locs.size > 3:
            locs[2] = locs[0]          # repeated vertices, self loops
            locs[1] = locs[0]
        radii = rng.random(locs.size).astype(np.float32)
        radii = radii[np.unique(locs, return_inverse=True)[1]]  # a vertex always has the same radius
        sx, sy = shape[0], shape[1]
        pts = np.stack([locs % sx, (locs // sx) % sy, locs // (sx * sy)], axis=1)
        parts, pos = [], 0
        for n in lens:
            sk = Skeleton.from_path(pts[pos:pos + n])
            sk.radii = radii[pos:pos + n]
            parts.append(sk)
            pos += n
        want = Skeleton.simple_merge(parts).consolidate()
        v, e, r = consolidate_paths(locs, lens.astype(np.int64), radii, shape)
        if want.empty():
            assert e.shape[0] == 0
            continue
        np.testing.assert_array_equal(v, want.vertices)
        np.testing.assert_array_equal(e, want.edges)
        np.testing.assert_array_equal(r, want.radii)


def test_skeleton_semantics():
    from kimimaro_amd.skeleton import Skeleton
    s = Skeleton.from_path([[0, 0, 0], [1, 1, 1], [2, 2, 2]])
    assert s.edges.tolist() == [[0, 1], [1, 2]] and not s.empty()
    assert Skeleton.from_path(np.zeros((0, 3))).empty()
    assert Skeleton.from_path([[3, 3, 3]]).empty()          # one vertex, no edge
    m = Skeleton.simple_merge([s, Skeleton.from_path([[2, 2, 2], [3, 2, 2]])]).consolidate()
    assert m.vertices.shape[0] == 4 and m.edges.shape[0] == 3
    assert abs(m.cable_length() - (2 * np.sqrt(3) + 1)) < 1e-5
    p = Skeleton(m.vertices * np.float32(2), m.edges, transform=[[2, 0, 0, 0], [0, 2, 0, 0], [0, 0, 2, 0]], space="physical")
    np.testing.assert_array_equal(p.voxel_space().vertices, m.vertices)
    assert "1 0" in m.to_swc().splitlines()[2]
    assert len(m.components()) == 1


def test_format_labels_and_dimension_error():
    from kimimaro_amd import intake
    assert intake.format_labels(np.zeros((5,), bool)).shape == (5, 1, 1)
    assert intake.format_labels(np.zeros((5, 4), np.uint8)).flags.f_contiguous
    assert intake.format_labels(np.zeros((5, 4, 3, 1), np.uint8)).shape == (5, 4, 3)
    with pytest.raises(intake.DimensionError):
        intake.format_labels(np.zeros((5, 4, 3, 2), np.uint8))
    lab = np.arange(24, dtype=np.uint32).reshape(2, 3, 4)
    out = intake.apply_object_mask(lab.copy(), [3, 5])
    assert sorted(np.unique(out).tolist()) == [0, 3, 5]


def test_defaults_match_reference():
    """kimimaro/intake.py:47-56 and kimimaro/trace.py:38-43."""
    import kimimaro_amd
    from kimimaro_amd.intake import TRACE_DEFAULTS
    assert kimimaro_amd.DEFAULT_TEASAR_PARAMS == {
        "scale": 1.5, "const": 300, "pdrf_scale": 100000, "pdrf_exponent": 4, "soma_acceptance_threshold": 3500,
        "soma_detection_threshold": 750, "soma_invalidation_const": 300, "soma_invalidation_scale": 2}
    assert TRACE_DEFAULTS["scale"] == 10 and TRACE_DEFAULTS["pdrf_exponent"] == 16 and TRACE_DEFAULTS["max_paths"] is None


def test_assembler_merge_equals_simple_merge_consolidate():
    """Assembler.finish merges the (disjoint) components of a label on integer keys; it must equal
    Skeleton.simple_merge(components).consolidate() (kimimaro/intake.py:587-593)."""
    from kimimaro_amd.intake import Assembler, consolidate_paths
    from kimimaro_amd.skeleton import Skeleton
    rng = np.random.default_rng(3)
    shape = (40, 30, 20)
    an = (16.0, 16.0, 40.0)
    asm = Assembler(shape, an, {1: 77, 2: 77, 3: 77, 4: 5})
    ref = {77: [], 5: []}
    used = set()
    for comp, orig in ((3, 77), (1, 77), (2, 77), (4, 5)):          # arrival order != component order
        # a random walk of distinct voxels (one path) + a second path starting on the first (shared vertex)
        locs = []
        while len(locs) < 30:
            l = int(rng.integers(0, np.prod(shape)))
            if l not in used:
                used.add(l)
                locs.append(l)
        path2 = [locs[7]] + locs[20:]
        allv = np.array(locs[:20] + path2, dtype=np.int64)
        lens = np.array([20, len(path2)], dtype=np.int64)
        radii = rng.random(allv.size).astype(np.float32)
        radii[20] = radii[7]
        verts, edges, r = consolidate_paths(allv, lens, radii, shape)
        asm.skeletons[orig].append((comp, verts, edges, r))
        ref[orig].append((comp, Skeleton(np.multiply(verts, np.float32(an), dtype=np.float32), edges, radii=r, segid=orig)))
    got = asm.finish()
    for orig, parts in ref.items():
        want = Skeleton.simple_merge([s for _, s in sorted(parts, key=lambda p: p[0])]).consolidate()
        np.testing.assert_array_equal(got[orig].vertices, want.vertices)
        np.testing.assert_array_equal(got[orig].edges, want.edges)
        np.testing.assert_array_equal(got[orig].radii, want.radii)
        assert got[orig].id == orig and got[orig].space == "physical"


def test_oracle_pool_equals_serial_oracle():
    """oracle/pool.py (per-component work on a forked pool, DBF on the component's box grown by one voxel and cut back
    to the reference's crop) gives exactly oracle.pipeline.skeletonize's result."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from shapes import voronoi_labels
    from oracle import pool, pipeline as P
    an = (16, 16, 40)
    lab = voronoi_labels((48, 48, 24), 6, seed=3, pts_per_label=3, step=8.0, anisotropy=an)
    params = dict(P.DEFAULT_TEASAR_PARAMS)
    params["const"] = 60
    a, _, _ = pool.skeletonize_pool(lab, params, anisotropy=an, dust_threshold=100, fix_borders=True, workers=2)
    b = P.skeletonize(lab, params, anisotropy=an, dust_threshold=100, fix_borders=True)
    assert sorted(a) == sorted(b) and len(a) >= 3
    for k in a:
        np.testing.assert_array_equal(a[k].vertices, b[k].vertices)
        np.testing.assert_array_equal(a[k].edges, b[k].edges)
        np.testing.assert_array_equal(a[k].radii, b[k].radii)


def test_batch_consolidation_equals_the_per_label_one():
    """intake.consolidate_paths_batch (all labels of a result group in one sort) against consolidate_paths label by label:
    random paths with shared vertices, empty slots, one-vertex paths (dropped: no edge refers to them), repeated vertices."""
    from kimimaro_amd.intake import consolidate_paths, consolidate_paths_batch
    rng = np.random.default_rng(3)
    shape = (40, 36, 20)
    verts, lens, voff, loff = [], [], [0], [0]
    for s in range(60):
        npaths = 0 if s % 11 == 0 else int(rng.integers(1, 5))
        tot = 0
        base = rng.integers(5, 15, 3)
        for p in range(npaths):
            n = 1 if (s + p) % 7 == 0 else int(rng.integers(2, 40))
            pts = np.clip(base + np.cumsum(rng.integers(-1, 2, (n, 3)), axis=0), 0, np.array(shape) - 1)
            verts.append((pts[:, 0] + shape[0] * (pts[:, 1] + shape[1] * pts[:, 2])).astype(np.uint32))
            lens.append(n)
            tot += n
        voff.append(voff[-1] + tot)
        loff.append(loff[-1] + npaths)
    res = {"verts": np.concatenate(verts), "radii": rng.random(voff[-1]).astype(np.float32), "lens": np.asarray(lens, dtype=np.uint32),
           "voff": np.asarray(voff), "loff": np.asarray(loff)}
    got = {s: (v, e, r) for s, v, e, r in consolidate_paths_batch(res, shape)}
    seen = 0
    for s in range(60):
        v0, v1 = res["voff"][s], res["voff"][s + 1]
        if v1 == v0:
            assert s not in got
            continue
        wv, we, wr = consolidate_paths(res["verts"][v0:v1].astype(np.int64), res["lens"][res["loff"][s]:res["loff"][s + 1]].astype(np.int64),
                                       res["radii"][v0:v1], shape)
        gv, ge, gr = got[s]
        assert gv.dtype == wv.dtype and ge.dtype == we.dtype
        np.testing.assert_array_equal(gv, wv)
        np.testing.assert_array_equal(ge, we)
        np.testing.assert_array_equal(gr, wr)
        seen += 1
    assert seen > 40
    assert list(consolidate_paths_batch({"verts": np.zeros(0, np.uint32), "radii": np.zeros(0, np.float32), "lens": np.zeros(0, np.uint32),
                                         "voff": np.array([0, 0]), "loff": np.array([0, 0])}, shape)) == []


def test_precomputed_roundtrip():
    """Skeleton.to_precomputed / from_precomputed (the wire format of row f4): byte layout and round trip."""
    from kimimaro_amd.skeleton import Skeleton
    a = Skeleton([[0, 0, 0], [16, 16, 40], [32, 16, 80]], [[0, 1], [1, 2]], radii=[1.5, 2.5, 3.5], vertex_types=[0, 1, 2], segid=7,
                 space="physical")
    blob = a.to_precomputed()
    assert len(blob) == 8 + 3 * 12 + 2 * 8 + 3 * 4 + 3
    assert np.frombuffer(blob, "<u4", 2).tolist() == [3, 2]
    b = Skeleton.from_precomputed(blob, segid=7)
    assert a == b and b.vertex_types.tolist() == [0, 1, 2]
    assert Skeleton.from_precomputed(Skeleton().to_precomputed()).empty()
    with pytest.raises(ValueError):
        Skeleton.from_precomputed(blob + b"x")


def test_precomputed_bytes_from_the_format_specification():
    """to_precomputed against a byte string assembled by hand from the Neuroglancer precomputed skeleton specification
    (neuroglancer/src/datasource/precomputed/skeletons.md, "Encoded skeleton file format"): num_vertices u32le,
    num_edges u32le, vertex_positions float32le [num_vertices][3] (C order), edges uint32le [num_edges][2], then every
    vertex attribute of the info file in order -- here radius float32le [num_vertices], vertex_types uint8 [num_vertices]
    (the attribute set kimimaro / cloud-volume declare).  Independent of from_precomputed."""
    import struct
    from kimimaro_amd.skeleton import Skeleton
    verts = [(0.0, 0.0, 0.0), (16.0, 32.0, 40.0), (-1.5, 2.25, 1e6)]
    edges = [(0, 1), (1, 2)]
    radii = [1.5, 2.5, 0.0]
    vtypes = [0, 3, 255]
    want = struct.pack("<II", 3, 2)
    want += b"".join(struct.pack("<fff", *v) for v in verts)
    want += b"".join(struct.pack("<II", *e) for e in edges)
    want += struct.pack("<fff", *radii)
    want += bytes(vtypes)
    # spot values of the spec's byte order: 1.5f = 00 00 C0 3F little endian, vertex 1's x = 16.0f = 00 00 80 41
    assert want[8 + 12:8 + 16] == bytes([0x00, 0x00, 0x80, 0x41]) and want[8 + 36 + 16:8 + 36 + 20] == bytes([0x00, 0x00, 0xC0, 0x3F])
    s = Skeleton(verts, edges, radii=radii, vertex_types=vtypes, segid=1, space="physical")
    assert s.to_precomputed() == want
    assert [a["id"] for a in s.extra_attributes] == ["radius", "vertex_types"]
    assert [(a["data_type"], a["num_components"]) for a in s.extra_attributes] == [("float32", 1), ("uint8", 1)]
    back = Skeleton.from_precomputed(want, segid=1)
    assert back == s and back.vertex_types.tolist() == vtypes


def test_plan_launches_groups_labels_under_the_scratch_budget():
    """Engine.scratch_budget: labels sorted by size, groups closed at the budget, an oversized label alone."""
    from kimimaro_amd.engine import plan_launches, SCRATCH_BYTES_PER_VOXEL, SCRATCH_BYTES_PER_LABEL
    counts = np.array([10, 5000, 200, 90000, 3000, 70000]) * 3
    need = counts * SCRATCH_BYTES_PER_VOXEL + SCRATCH_BYTES_PER_LABEL
    assert plan_launches(counts, int(need.sum())) == [list(range(6))]            # everything fits: one launch, caller order
    groups = plan_launches(counts, 30 << 20)
    flat = [i for g in groups for i in g]
    assert sorted(flat) == list(range(6))
    assert flat == [3, 5, 1, 4, 2, 0]                                             # largest first
    for g in groups:
        assert len(g) == 1 or int(need[g].sum()) <= (30 << 20)
    assert groups[0] == [3]                                                       # 270000 voxels: 30 MB + 2 MB, alone
    assert plan_launches(counts, 1) == [[3], [5], [1], [4], [2], [0]]             # nothing fits: one label per launch


def test_pow2_exponent_mirrors_the_reference_test():
    """kimimaro/trace.py:310-313,343: integers that are powers of two below 2**16 take the squaring branch; an integral float
    passes the reference's `int(num) != num` line and then fails in `num & (num - 1)` with TypeError."""
    from kimimaro_amd._abi import is_pow2_exponent
    assert [e for e in (1, 2, 4, 16, 2 ** 15, np.int64(8)) if is_pow2_exponent(e)] == [1, 2, 4, 16, 2 ** 15, np.int64(8)]
    assert not any(is_pow2_exponent(e) for e in (0, 3, 5, 6, 2 ** 16, 2 ** 17, 2.5, np.float32(1.5), -4))
    for e in (4.0, np.float64(16), np.float32(2)):
        with pytest.raises(TypeError):
            is_pow2_exponent(e)


def _key_table(an, rmax):
    """the flood's keys of all offsets up to rmax, with its float operation order (dijkstra_invalidation.hpp:310-316), and their
    ranks -- what Engine.level_table gets from kh_level_keys + torch.unique"""
    w = [np.float32(a) for a in an]
    dims = [int(rmax / float(w[i])) + 2 for i in range(3)]
    a = (np.arange(dims[0], dtype=np.float32) * w[0]) ** 2
    b = (np.arange(dims[1], dtype=np.float32) * w[1]) ** 2
    c = (np.arange(dims[2], dtype=np.float32) * w[2]) ** 2
    s = ((a[:, None, None] + b[None, :, None]).astype(np.float32) + c[None, None, :]).astype(np.float32)
    keys3 = np.sqrt(s).astype(np.float32)
    uniq, inv = np.unique(keys3, return_inverse=True)
    return uniq, inv.reshape(keys3.shape), dims


@pytest.mark.parametrize("an,rmax", [((16, 16, 40), 700.0), ((1, 1, 1), 40.0), ((8, 8, 40), 500.0), ((4, 3, 2), 90.0), ((40, 32, 20), 900.0)])
def test_level_window_bounds_every_neighbour_step(an, rmax):
    """kh_label_t.lev_window (engine.level_windows): an event goes from a voxel processed at a level >= rank(v, c) to a neighbour q at
    rank(q, c); the window must exceed rank(q, c) - rank(v, c) for EVERY offset of v inside the radius and every one of the 26 steps --
    checked here by brute force on the key table itself (the kernel checks every push as well and falls back to the heap when a bound is
    ever exceeded: this test is why that never happens)."""
    from kimimaro_amd.engine import level_windows
    uniq, rank, dims = _key_table(an, rmax)
    nlev = int(np.searchsorted(uniq, np.float32(rmax), side="left"))
    win = int(level_windows(uniq, an, [nlev], 1 << 20)[0])
    assert win >= 64 and win & (win - 1) == 0
    inside = rank < nlev                                         # offsets whose key is below the radius
    worst = 0
    A, B, C = dims
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                # neighbour offset (|a + dx|, |b + dy|, |c + dz|): the table holds absolute offsets
                ia = np.abs(np.arange(A)[:, None, None] + dx)
                ib = np.abs(np.arange(B)[None, :, None] + dy)
                ic = np.abs(np.arange(C)[None, None, :] + dz)
                ok = inside & (ia < A) & (ib < B) & (ic < C)
                rq = rank[np.minimum(ia, A - 1), np.minimum(ib, B - 1), np.minimum(ic, C - 1)]
                ok &= rq < nlev                                  # the neighbour is inside the radius too (else no event)
                if ok.any():
                    worst = max(worst, int((rq - rank)[ok].max()))
    assert 0 < worst < win, (worst, win)
    assert win <= 4 * (worst + 2)                                # and the bound is not wildly loose (LDS is what it costs)


def test_plan_arena_shapes():
    """engine.plan_arena: filtered labels get small chunks and an arena for what is pending (window + a few events per voxel), labels
    with more levels than the filter can pack keep the unfiltered budget, nothing exceeds the 22-bit chunk ids."""
    from kimimaro_amd.engine import plan_arena, SCHED_LEVELS
    cnt = np.array([1200, 40000, 177129, 10 ** 6, 5 * 10 ** 7])
    nlev = np.array([6000, 9000, 8142, SCHED_LEVELS + 5, 9000])
    win = np.array([1024, 1024, 1024, 1024, 0])
    shift, chunks = plan_arena(cnt, nlev, True, win)
    assert shift.tolist() == [5, 5, 6, 7, 6] and (chunks <= (1 << 22) - 2).all() and (chunks > 0).all()
    s0, c0 = plan_arena(cnt, nlev, False, win)
    assert (s0 >= shift).all() and ((c0 << s0) >= (chunks << shift)).all()          # the unfiltered arena is never smaller
    assert int(chunks[0]) < 2500 and int(chunks[2]) < 12000                              # bounded by the window, not by nlev


def test_host_ccl_equals_the_oracle_components():
    """kh_host_ccl26 (the faces of the volume in compute_border_targets; 3-D on request) skips the in-plane links that earlier
    pixels have made already: same components, same first-appearance numbering as the oracle's cc3d restatement"""
    from kimimaro_amd import border, intake
    from oracle import pipeline as P
    rng = np.random.default_rng(5)
    for t in range(40):
        shp = tuple(int(v) for v in rng.integers(1, 14, 3))
        lab = np.asfortranarray(rng.integers(0, rng.integers(2, 5), shp).astype([np.uint8, np.uint16, np.uint32, np.uint64][t % 4]))
        got, n, remap = intake.compute_cc_labels(lab)
        want, remap_o = P.compute_cc_labels(lab)
        np.testing.assert_array_equal(got, want)
        assert remap == remap_o and n == len(remap_o)
        plane = np.asfortranarray(lab[:, :, 0])
        cc, m = border._ccl2d(plane)
        want2, _ = P.compute_cc_labels(plane[:, :, None])
        np.testing.assert_array_equal(cc, want2[:, :, 0])


def test_native_component_merge_equals_simple_merge_consolidate_on_many_labels():
    """kh_host_merge_components (Assembler.finish: every label of a volume in one native call -- runs of sorted vertices merged,
    edge rows counting-sorted) against Skeleton.simple_merge(components).consolidate() (kimimaro/intake.py:587-593) on 300
    labels with one to six disjoint components each, random forests as edges"""
    from kimimaro_amd.intake import Assembler
    from kimimaro_amd.skeleton import Skeleton
    rng = np.random.default_rng(11)
    shape = (64, 48, 40)
    an = np.float32([16, 16, 40])
    ncomp = 900
    remap = {i + 1: int(rng.integers(1, 301)) for i in range(ncomp)}
    asm = Assembler(shape, an, remap)
    pool, pos = rng.permutation(int(np.prod(shape))), 0
    for segid in rng.permutation(np.arange(1, ncomp + 1)):          # arrival order != component order
        n = int(rng.integers(2, 120))
        key = np.sort(pool[pos:pos + n])
        pos += n
        verts = np.stack([key // (48 * 40), (key // 40) % 48, key % 40], 1).astype(np.float32)
        par = np.array([rng.integers(0, i) for i in range(1, n)])
        e = np.stack([par, np.arange(1, n)], 1)
        e = e[np.lexsort((e[:, 1], e[:, 0]))].astype(np.uint32)
        asm.skeletons[remap[int(segid)]].append((int(segid), verts, e, rng.random(n).astype(np.float32)))
    out = asm.finish()
    assert list(out) == list(asm.skeletons)
    for orig, parts in asm.skeletons.items():
        sk = [Skeleton(np.multiply(v, an, dtype=np.float32), e, radii=r, segid=orig) for _, v, e, r in sorted(parts, key=lambda p: p[0])]
        want = Skeleton.simple_merge(sk).consolidate()
        np.testing.assert_array_equal(out[orig].vertices, want.vertices)
        np.testing.assert_array_equal(out[orig].edges, want.edges)
        np.testing.assert_array_equal(out[orig].radii, want.radii)
        assert out[orig].id == orig and out[orig].vertices.flags.owndata and out[orig].edges.dtype == np.uint32


def test_native_consolidate_paths_equals_the_numpy_statement():
    """kh_host_consolidate_paths (one native call per result group, outside the interpreter) against
    intake.consolidate_paths_flat_numpy on random paths with repeated vertices, self loops, one-vertex paths and empty slots"""
    import numpy as np
    from kimimaro_amd import intake
    for seed, shape in ((0, (40, 37, 29)), (1, (7, 5, 3)), (2, (512, 512, 100))):
        rng = np.random.default_rng(seed)
        nvox = int(np.prod(shape))
        voff, loff, locs, lens = [0], [0], [], []
        for s in range(150):
            for p in range(int(rng.integers(0, 5))):
                n = int(rng.integers(1, 30))
                pts = rng.integers(0, max(nvox // 50, 2), n) * rng.integers(1, 3)
                locs.extend(int(v) % nvox for v in pts)
                lens.append(n)
            voff.append(len(locs))
            loff.append(len(lens))
        res = {"voff": np.array(voff), "loff": np.array(loff), "verts": np.array(locs, dtype=np.uint32),
               "lens": np.array(lens, dtype=np.uint32), "radii": rng.random(len(locs)).astype(np.float32)}
        a = intake.consolidate_paths_flat(res, shape)
        b = intake.consolidate_paths_flat_numpy(res, shape)
        for k in ("verts", "radii", "edges", "vstart", "estart"):
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)
    empty = {"voff": np.zeros(4, np.int64), "loff": np.zeros(4, np.int64), "verts": np.zeros(0, np.uint32),
             "lens": np.zeros(0, np.uint32), "radii": np.zeros(0, np.float32)}
    assert intake.consolidate_paths_flat(empty, (4, 4, 4)) is None
