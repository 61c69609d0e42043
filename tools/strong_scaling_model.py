#!/usr/bin/env python
"""What ONE volume's components cost per rank when they are dealt over N GPUs (BASELINE.json configs[3], strong scaling), from the
per-label cycle counters of a measured single-GPU run -- no multi-GPU box needed, and no efficiency is claimed: the driver measures
the curve; this prints what the design predicts so that the prediction is a checked-in number.

    python tools/strong_scaling_model.py profiles/r05_c3_tasks.npz [--clock-ghz 2.4] [--preamble-s 0.12] [--host-s 0.04]

Input: the npz `KIMI_BENCH_DUMP_TASKS=... python bench.py --inflight 1` writes (Engine.last_tasks of the instrumented pass: voxels,
shader kilo-cycles per phase, heap pushes, roll-backs per connected component).
Model: a rank's step = the whole-volume preamble, which every rank repeats (CCL, EDT, statistics, border targets)
  + max(  the longest CHAIN among its components -- one workgroup per component, so a component is a sequential chain,
          the total workgroup time of its components / the workgroup slots of the GPU  )
  + host assembly of its skeletons.
The split is kimimaro_amd.intake.shard_components (largest first to the least loaded rank, by voxels).
The volume's step on N GPUs is the slowest rank; "chain bound" says which term decides."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tasks")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--preamble-s", type=float, default=0.12, help="whole-volume preamble every rank repeats")
    ap.add_argument("--host-s", type=float, default=0.04, help="host assembly of ALL skeletons of the volume (a rank does its share)")
    ap.add_argument("--slots", type=int, default=768, help="workgroup slots of one GPU at 256 threads per component (3 per CU)")
    args = ap.parse_args()
    from kimimaro_amd.intake import shard_components
    z = np.load(args.tasks)
    counts = z["count"].astype(np.int64)
    cyc = (z["cyc_target"].astype(np.int64) + z["cyc_rail"].astype(np.int64) + z["cyc_inval"].astype(np.int64)) * 1024
    sec = cyc / (args.clock_ghz * 1e9)
    n = len(counts)
    ids = list(range(n))
    cmap = {i: int(counts[i]) for i in ids}
    print("components %d, voxels %d, total workgroup time %.1f s, longest chain %.3f s (%d voxels, %d heap pushes)" % (
        n, int(counts.sum()), float(sec.sum()), float(sec.max()), int(counts[int(np.argmax(sec))]),
        int(z["stat_heap_pushes"][int(np.argmax(sec))])))
    base = None
    for world in (1, 2, 4, 8):
        steps = []
        for rank in range(world):
            mine = np.asarray(shard_components(ids, cmap, rank, world), dtype=np.int64)
            chain = float(sec[mine].max()) if mine.size else 0.0
            fill = float(sec[mine].sum()) / args.slots
            steps.append((args.preamble_s + max(chain, fill) + args.host_s * (mine.size / max(n, 1)), chain, fill, mine.size, int(counts[mine].sum())))
        worst = max(steps)
        base = base or worst[0]
        print("N=%d: step %.3f s (x%.2f vs N=1)  slowest rank: chain %.3f s, fill %.3f s, %d components, %d voxels -> %s bound; per-rank steps %s" % (
            world, worst[0], base / worst[0], worst[1], worst[2], worst[3], worst[4], "chain" if worst[1] >= worst[2] else "fill",
            " ".join("%.3f" % s[0] for s in steps)))
    print("(strong scaling of ONE volume is bounded by its longest chain; throughput over many volumes scales with the GPUs: "
          "bench.py --scaling weak)")


if __name__ == "__main__":
    main()
