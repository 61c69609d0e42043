"""BASELINE.json configs that round 2 left without a parity test on the HIP path:

  configs[1]  c2 (512x512x100, 333 chains, anisotropy (16,16,40)) at FULL size, every skeleton against the pooled oracle
              (round 2 checked size-independent properties only);
  configs[3]  the rank / world shard of ONE volume on the HIP path: skeletonize_cc(rank=r, world=2) for r = 0, 1 on one
              GPU, merged with distributed.merge_rank_results (what the all-gather-v hands every rank), equal to the
              oracle -- labels whose components land on different ranks included (small Voronoi volume and c2);
  configs[4]  c5 (1024^3, 8192 chains, anisotropy (8,8,40), fix_branching) through kimimaro_amd.skeletonize on ONE GPU
              (several launches of the path loop under Engine.scratch_budget), a seeded sample -- the 32 largest
              components + random ones -- compared bit exact with oracle.pool.skeletonize_pool(only=...).

Bars: vertices / edges bit exact, radii within 1e-4 relative (north_star).  /root/reference is not touched.
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


def _same(got, want, k):
    np.testing.assert_array_equal(got.vertices, want.vertices, err_msg="label %d" % k)
    np.testing.assert_array_equal(got.edges, want.edges, err_msg="label %d" % k)
    np.testing.assert_allclose(got.radii, want.radii, rtol=1e-4, err_msg="label %d" % k)


def _sharded(eng, lab, an, params, dust, world):
    """what `world` ranks produce for ONE volume, run one after the other on this GPU, and the merge every rank ends up with"""
    from collections import defaultdict
    from kimimaro_amd import intake
    from kimimaro_amd.distributed import merge_rank_results, pack_skeletons, unpack_skeletons
    lab = intake.format_labels(lab, in_place=False)
    empty = defaultdict(list)
    per_rank, sizes = [], []
    for r in range(world):
        d_cc, n, remap = intake.compute_cc_labels_device(eng, lab)
        cc = intake.LazyVolume(eng, d_cc, lab.shape)
        local = intake.skeletonize_cc(eng, cc, n, remap, params, np.asarray(an, dtype=np.float32), dust, True, True,
                                      empty, empty, black_border=False, rank=r, world=world, d_cc=d_cc)
        sizes.append(len(local))
        per_rank.append(unpack_skeletons(pack_skeletons(local)))     # through the wire format of the all-gather-v
    return merge_rank_results(per_rank), sizes


def test_shard_small_volume_two_and_three_ranks(eng):
    import kimimaro_amd
    from oracle import pipeline as P
    from shapes import voronoi_labels
    an = (16, 16, 40)
    lab = voronoi_labels((128, 128, 128), 60, seed=33, pts_per_label=6, step=14.0, anisotropy=an)
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    want = P.skeletonize(lab, params, anisotropy=an, dust_threshold=500, fix_borders=True, fix_branching=True)
    assert len(want) > 20
    for world in (2, 3):
        got, sizes = _sharded(eng, lab, an, params, 500, world)
        assert all(s > 0 for s in sizes) and sum(sizes) >= len(want)      # every rank traced something
        assert sorted(got) == sorted(want)
        for k in want:
            _same(got[k], want[k], k)


def test_c1_eight_balls_every_skeleton(eng):
    """BASELINE configs[0] as SURVEY 8d writes it: 64^3 u32, eight balls of radius 9-14 on background, anisotropy (1, 1, 1) --
    bench.make_volume("c1") -- through kimimaro_amd.skeletonize against the oracle pipeline, every skeleton."""
    import bench
    import kimimaro_amd
    from oracle import pipeline as P
    lab, an = bench.make_volume("c1")
    assert lab.shape == (64, 64, 64) and lab.dtype == np.uint32 and (lab == 0).any() and 2 <= len(np.unique(lab)) - 1 <= 8
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    for dust in (1000, 100):
        want = P.skeletonize(lab, params, anisotropy=an, dust_threshold=dust, fix_borders=True, fix_branching=True)
        got = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=dust, fix_borders=True, fix_branching=True,
                                       progress=False, _engine=eng)
        assert sorted(got) == sorted(want) and len(want) >= 2
        for k in want:
            _same(got[k], want[k], k)


@pytest.fixture(scope="module")
def c2():
    import bench
    import kimimaro_amd
    from oracle import pool
    lab, an = bench.make_volume("c2")
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    from oracle.cpu_pool_baseline import usable_cores
    aff, quota = usable_cores()
    want, cc, counts = pool.skeletonize_pool(lab, params, anisotropy=an, dust_threshold=1000, fix_branching=True, fix_borders=True,
                                             workers=max(2, int(min(aff, quota)) if quota else aff))
    return lab, an, params, want, cc, counts


def test_c2_full_size_every_skeleton(eng, c2):
    import kimimaro_amd
    lab, an, params, want, _, _ = c2
    got = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=1000, fix_borders=True, fix_branching=True,
                                   progress=False, _engine=eng)
    assert sorted(got) == sorted(want) and len(want) >= 300
    for k in want:
        _same(got[k], want[k], k)


@pytest.mark.parametrize("mode", ["ghosts", "paranoid", "off"])
def test_c2_ghost_modes_every_skeleton(c2, mode):
    """DESIGN.md 3.4.6 at full size: c2 traced with ghosts (the default: a call of the sweep that leaves voxels undecided goes on
    with them as ghosts, the label rolls back only when a ghost would matter), with every ghost call rolled back at once (the
    roll-back path itself: journal revival, rail weights restored, the call redone by the heap emulation) and with ghosts off
    (rounds 2-4) -- all three equal to the pooled oracle, skeleton by skeleton.  The counters prove the paths ran."""
    import kimimaro_amd
    from kimimaro_amd.engine import Engine
    lab, an, params, want, _, _ = c2
    e = Engine()
    e.ghosts = mode != "off"
    e.ghost_paranoid = mode == "paranoid"
    got = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=1000, fix_borders=True, fix_branching=True,
                                   progress=False, _engine=e)
    assert sorted(got) == sorted(want)
    for k in want:
        _same(got[k], want[k], k)
    tk = e.last_tasks
    ghost_calls, rollbacks = int(tk["stat_ghost_calls"].sum()), int(tk["stat_rollbacks"].sum())
    bails = int(tk["stat_sweep_bails"].sum())
    if mode == "off":
        assert ghost_calls == 0 and rollbacks == 0 and bails > 0
    elif mode == "paranoid":
        assert ghost_calls > 0 and 0 < rollbacks <= ghost_calls
    else:
        assert ghost_calls > 0 and rollbacks <= ghost_calls   # (most ghosts are killed for certain by a later ball)
    print("c2 %s: ghost calls %d, roll-backs %d, calls redone by the heap %d" % (mode, ghost_calls, rollbacks, bails))


def test_c2_sharded_over_two_ranks(eng, c2):
    from kimimaro_amd.intake import shard_components
    lab, an, params, want, cc, counts = c2
    got, sizes = _sharded(eng, lab, an, params, 1000, 2)
    assert all(s > 0 for s in sizes)
    assert sorted(got) == sorted(want)
    for k in want:
        _same(got[k], want[k], k)
    # the case the merge exists for: a label whose components were traced by different ranks
    segids = [i for i in range(1, counts.size) if counts[i] > 1000]
    owner = {}
    for r in range(2):
        for s in shard_components(segids, counts, r, 2):
            owner[s] = r
    flat_cc, flat_lab = cc.ravel(order="K"), np.asfortranarray(lab).ravel(order="K")
    ids, where = np.unique(flat_cc, return_index=True)
    first = {int(i): int(flat_lab[w]) for i, w in zip(ids, where)}
    ranks_of_label = {}
    for s in segids:
        ranks_of_label.setdefault(first[s], set()).add(owner[s])
    split = [l for l, rs in ranks_of_label.items() if len(rs) > 1 and l in want]
    assert len(split) > 0, "no label of c2 was split over the two ranks: the merge path was not exercised"


def test_c5_sample_matches_oracle(eng):
    """1024^3 / (8,8,40): the sweep's certificate at the third anisotropy, the multi-launch path at full size.
    The sample is drawn by LABEL -- every component of a sampled label is traced by the oracle -- so every skeleton of the
    sample is complete and every one is compared: a wrong invalidation changes the number of paths, i.e. the shape."""
    if os.environ.get("KIMI_SKIP_C5") == "1":
        pytest.skip("KIMI_SKIP_C5=1")
    import bench
    import kimimaro_amd
    from oracle import pool
    lab, an = bench.make_volume("c5")
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    got = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=1000, fix_borders=True, fix_branching=True,
                                   progress=False, _engine=eng)
    tk = eng.last_tasks
    assert len(tk) > 8000 and int(tk["stat_sweep_calls"].sum()) > 20000
    # component -> label from the device's own numbering (kh_ccl26 numbers the components like the oracle's CCL: by first
    # appearance in the F-order raster, tests/test_gpu_ccl.py): the label at the component's smallest linear index
    _, ncomp, rep = eng.ccl(lab)
    label_of = np.zeros(ncomp + 1, dtype=np.int64)
    label_of[1:] = np.asfortranarray(lab).reshape(-1, order="F")[rep[1:ncomp + 1].astype(np.int64)]
    segs, cnts = tk["segid"].astype(np.int64), tk["count"].astype(np.int64)
    big = segs[np.argsort(-cnts, kind="stable")[:32]]
    rng = np.random.default_rng(5)
    from oracle.cpu_pool_baseline import usable_cores
    aff, quota = usable_cores()
    cores = int(min(aff, quota)) if quota else aff            # the pool's boxes show 256 CPUs and grant 16 (cgroup quota)
    nrand = 480 if cores >= 64 else 96
    labels = np.unique(label_of[segs])
    chosen = set(label_of[big].tolist()) | set(rng.choice(labels, size=min(nrand, labels.size), replace=False).tolist())
    only = set(np.flatnonzero(np.isin(label_of, list(chosen)) & (np.arange(ncomp + 1) > 0)).tolist())
    want, _, _ = pool.skeletonize_pool(lab, params, anisotropy=an, dust_threshold=1000, fix_branching=True, fix_borders=True,
                                       only=only, workers=max(2, cores))
    assert sorted(want) == sorted(k for k in chosen if k in got) and len(want) >= 64
    for k, w in want.items():
        _same(got[k], w, k)


def test_all_gather_v_over_rccl_on_the_device(eng):
    """configs[3]'s only collective on the backend an 8-GPU node uses: distributed.gather_skeletons over "nccl" (= RCCL on
    ROCm) with device tensors, world_size 1 on this GPU -- the branch the gloo tests cannot reach."""
    import socket
    import torch
    import torch.distributed as dist
    import kimimaro_amd
    from kimimaro_amd import distributed as D
    from shapes import voronoi_labels
    if dist.is_initialized():
        pytest.skip("a process group exists already in this process")
    an = (16, 16, 40)
    lab = voronoi_labels((64, 64, 48), 10, seed=5, pts_per_label=5, step=10.0, anisotropy=an)
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    params["const"] = 64
    local = kimimaro_amd.skeletonize(lab, params, anisotropy=an, dust_threshold=200, fix_borders=True, progress=False, _engine=eng)
    assert len(local) > 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        blobs = D.all_gather_v(D.pack_skeletons(local), device=dev)
        assert len(blobs) == 1 and blobs[0] == D.pack_skeletons(local)
        merged = D.gather_skeletons(local, device=dev)
    finally:
        dist.destroy_process_group()
    assert sorted(merged) == sorted(local)
    for k in local:
        _same(merged[k], local[k], k)


def test_soma_of_the_c2soma_workload_matches_oracle(eng):
    """row f3 at a size that matters: the 1.1e6-voxel ellipsoid of bench.py's `c2soma` workload (DBF max 1400 nm above
    soma_detection_threshold, a 5 x 5 x 3 void inside) takes the soma branch of kimimaro/trace.py:108-134 -- fill_voids, EDT of
    the filled crop, soma root, the one-off soma invalidation, free_space_radius -- and must give the oracle's skeleton."""
    import bench
    import kimimaro_amd
    from oracle import pipeline as P
    lab, an = bench.make_volume("c2soma")
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    kw = dict(anisotropy=an, dust_threshold=1000, fix_borders=True, fix_branching=True, object_ids=[999999])
    got = kimimaro_amd.skeletonize(lab, params, progress=False, _engine=eng, **kw)
    want = P.skeletonize(lab, params, **kw)
    assert sorted(got) == sorted(want) == [999999] and want[999999].vertices.shape[0] > 50
    _same(got[999999], want[999999], 999999)
