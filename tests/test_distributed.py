"""N > 1 path on CPU: world_size 2 over gloo.  The component shard of the product (intake.shard_components: the
connected components of one volume, largest first to the least loaded rank -- the split skeletonize_cc(rank, world)
uses) and the skeleton all-gather-v are exercised with skeletons produced by the ORACLE on each rank's shard (the HIP
path needs a GPU); the merged result must equal the single-process result, including labels whose components land
on different ranks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kimimaro_amd import distributed as D
    from kimimaro_amd.intake import shard_components
    import oracle as K
    from oracle import pipeline as P, pool
    from shapes import voronoi_labels
    lab = voronoi_labels((48, 48, 32), 9, seed=21, pts_per_label=4, step=8.0)
    params = dict(P.DEFAULT_TEASAR_PARAMS)
    params["const"] = 4
    # each rank traces only its shard of the connected components, exactly the split of skeletonize_cc(rank, world)
    cc, _ = K.connected_components(lab)
    counts = np.bincount(cc.ravel(order="K"))
    segids = [i for i in range(1, counts.size) if counts[i] > 50]
    mine = shard_components(segids, counts, rank, world)
    assert 0 < len(mine) < len(segids)
    local, _, _ = pool.skeletonize_pool(lab, params, dust_threshold=50, fix_borders=False, only=mine, workers=1)
    # pool.skeletonize_pool hands back the oracle's Skeleton; the exchange packs the product's attribute set
    from kimimaro_amd.skeleton import Skeleton
    local = {k: Skeleton(v.vertices, v.edges, v.radii, segid=k, transform=v.transform, space=v.space) for k, v in local.items()}
    merged = D.gather_skeletons(local)
    q.put((rank, sorted(merged.keys()), {k: (v.vertices.copy(), v.edges.copy(), v.radii.copy()) for k, v in merged.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_two_ranks_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pipeline as P
    from shapes import voronoi_labels
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lab = voronoi_labels((48, 48, 32), 9, seed=21, pts_per_label=4, step=8.0)
    params = dict(P.DEFAULT_TEASAR_PARAMS)
    params["const"] = 4
    want = P.skeletonize(lab, params, dust_threshold=50, fix_borders=False)
    assert len(want) > 2
    for rank, keys, skels in results:
        assert keys == sorted(want.keys())
        for k in keys:
            np.testing.assert_array_equal(skels[k][0], want[k].vertices)
            np.testing.assert_array_equal(skels[k][1], want[k].edges)
            np.testing.assert_array_equal(skels[k][2], want[k].radii)


def test_pack_roundtrip_and_shard():
    from kimimaro_amd import distributed as D
    from kimimaro_amd.skeleton import Skeleton
    a = Skeleton([[0, 0, 0], [1, 1, 1], [2, 2, 3]], [[0, 1], [1, 2]], radii=[1, 2, 3], segid=7, space="physical")
    b = Skeleton([[5, 5, 5], [6, 5, 5]], [[0, 1]], radii=[4, 5], segid=9, space="physical")
    out = D.unpack_skeletons(D.pack_skeletons({7: a, 9: b}))
    assert sorted(out) == [7, 9]
    np.testing.assert_array_equal(out[7].vertices, a.vertices)
    np.testing.assert_array_equal(out[9].edges, b.edges)
    assert D.unpack_skeletons(D.pack_skeletons({})) == {}
    assert D.shard(list(range(10)), 1, 4) == [1, 5, 9]
    assert D.shard([10, 11, 12, 13], 0, 2, weights=[5, 9, 1, 3]) == [11] and D.shard([10, 11, 12, 13], 1, 2, weights=[5, 9, 1, 3]) == [10, 12, 13]
    big = D.unpack_skeletons(D.pack_skeletons({2 ** 63 + 5: a}))
    assert list(big) == [2 ** 63 + 5]
    # a label split over two ranks merges like intake.py:587-593
    m = D.merge_rank_results([{7: a}, {7: b, 9: b}])
    assert m[7].vertices.shape[0] == 5 and m[9].vertices.shape[0] == 2
