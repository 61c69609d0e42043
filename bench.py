#!/usr/bin/env python
"""bench.py -- labels/s of the MI355X TEASAR hot path on BASELINE.json's workload.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" = one pass of kimimaro.skeletonize's work over the synthetic label volume, which is ALREADY
RESIDENT IN HBM: connected components (kh_ccl26) -> whole-volume EDT -> per-label statistics -> border
targets -> find_root -> DAF -> PDRF -> TEASAR path loop for every component -> D2H of the paths ->
Skeleton assembly -> (N > 1) all-gather-v of the skeletons.  Only format_labels and the H2D copy of the
input are outside the timed region (`preamble_s`).

N > 1 (default --scaling strong = BASELINE.json configs[3]): ONE volume, its connected components dealt over the ranks
(largest first to the least loaded rank, kimimaro_amd.intake.shard_components), every rank recomputes the whole-volume
preamble, the skeletons are all-gathered; value = components of the volume / max-over-ranks step time.  The same run
then measures the weak mode as well (every rank its own volume of the workload's size: rank 0's volume mirrored along
the axes given by the bits of r, label ids offset) and prints it as the extra object "weak_scaling".

Workload (config.workload): "c3" = 512x512x512, 2124 chains, anisotropy (16,16,40), default
teasar_params, dust_threshold=1000, fix_borders=True, fix_branching=True  (BASELINE.json configs[2],
the configuration the metric is quoted on).  "c2" = 512x512x100 / 333 labels (configs[1]).  "c5" = 1024^3, 8192
chains, anisotropy (8,8,40) (configs[4]; one volume at a time, its labels in as many launches of the path loop as
Engine.scratch_budget asks for).
The reference's own volume (benchmarks/connectomics.npy.ckl.gz) cannot be decoded here,
so the volume is synthetic: data = "synthetic".

Steps in flight (--inflight F; default: as many as 85 % of the free HBM pays for -- 10.8 GB per 512^3 volume since round 6 -- 24 at most, and no
more than fill the rounds of the run evenly): the wall clock of ONE volume is the chain of its largest component -- a handful of waves
for seconds while the rest of the GPU idles.  The K timed steps are therefore issued from F lanes (host threads by
default, --lanes process for processes), each with a HIP stream, an Engine and scratch of its own and one wave per label in its path
loop, so that the tail of one volume overlaps the next ones; every step still does all of its work inside the timed region and
ms_per_step = wall / K.  The latency of a single volume on an otherwise idle GPU is measured in the same run (untimed, on the default
engine: what kimimaro_amd.skeletonize() does) and printed as single_volume_ms; --inflight 1 times the steps one after the other.

Extra objects on the JSON line:
  roofline      the DOMINANT kernel, trace_paths_kernel (97 % of the GPU time): SURVEY 8d per-label algorithmic bytes of one
                volume / the longest of the volume's (overlapped) path-loop launches, HIP events on each launch's own stream;
                `traffic` from the newest committed rocprofv3 --pmc pass (marked STALE when it predates this round's kernels).
  roofline_edt  EDT pass kernels (the kernel BASELINE.json's metric names): algorithmic bytes / the pass durations measured
                with HIP events on the launch stream (kh_edt_timed); edt_total_GBps = (3L + 20) B/voxel over the three passes.
  value_single_volume  components / single_volume_ms: the rate of ONE volume alone on the GPU (`value` is the pipelined rate).
  cpu_baseline  the oracle (CPU restatement, 1 core) on a bounded sample of the same labels.
  cpu_baseline_all_cores  the same work on a process pool over every USABLE host core (affinity and cgroup quota are
                printed: the pool's 256-core boxes grant 16), components largest first, in two legs -- `latency`: one
                volume's components; `throughput`: several volumes' components at once (as many as the GPU run has in
                flight, four at most: the pool is throughput bound from one volume on).
  chains / chains_under_load  the labels whose chains set the wall clock of the path kernel (cycles per phase, heap pushes,
                why their call left the sweep), alone and -- thread lanes -- in the lanes' last volumes.
  speedup_latency / speedup_throughput  the like-for-like ratios: one volume alone on the GPU vs the latency leg, the
                pipelined `value` vs the throughput leg.  Never mixed.
  rank_times    (N > 1) every rank's own seconds per step and the seconds it spent in the skeleton gather.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# one hardware queue per lane stream (up to 12 lanes + RCCL): with the default of 4, a fifth stream shares a queue and its kernels wait for the
# seconds-long path kernel queued before them (profiles/r02b_inflight_timeline.txt)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (shape, chains, points per chain, seed, anisotropy)
    "c1": ((64, 64, 64), 8, 0, 1, (1, 1, 1)),         # eight balls on background (make_balls), not a tessellation
    "c2": ((512, 512, 100), 333, 12, 2, (16, 16, 40)),
    "c3": ((512, 512, 512), 2124, 16, 3, (16, 16, 40)),
    "c5": ((1024, 1024, 1024), 8192, 24, 5, (8, 8, 40)),
    # c2 with a soma in its middle: an ellipsoid of 1400 nm radius (~1.1e6 voxels, DBF max above soma_detection_threshold)
    # with a small internal void, so that the label takes the soma branch of kimimaro/trace.py:108-134 (row f3)
    "c2soma": ((512, 512, 100), 333, 12, 2, (16, 16, 40)),
    # the same with TWO such somas (256 voxels apart in x): they are traced side by side (Engine.soma_lanes)
    "c2soma2": ((512, 512, 100), 333, 12, 2, (16, 16, 40)),
    "mini": ((128, 128, 64), 40, 8, 4, (16, 16, 40)),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_volume(name):
    """SURVEY.md 8d recipe: dense 'neurite' tessellation = nearest chain point under the anisotropic
    metric, label ids 1000 + perm(i).  Deterministic (np.random.default_rng(seed))."""
    from scipy.spatial import cKDTree
    shape, nchains, npts, seed, an = WORKLOADS[name]
    cache = os.environ.get("KIMI_VOLUME_CACHE")     # developer knob: keep the generated volume between runs on one box
    if cache:
        path = os.path.join(cache, "%s_seed%d.npy" % (name, seed))
        if os.path.exists(path):
            return np.asfortranarray(np.load(path)), an
    if name == "c1":
        return make_balls(shape, nchains, seed), an
    rng = np.random.default_rng(seed)
    anf = np.asarray(an, dtype=np.float64)
    shp = np.asarray(shape, dtype=np.float64)
    step = np.array([24.0, 24.0, 24.0 * anf[0] / anf[2]])
    pts, owner = [], []
    for l in range(nchains):
        p = rng.uniform(0, 1, 3) * shp
        for _ in range(npts):
            pts.append(p.copy())
            owner.append(l)
            d = rng.normal(size=3)
            d /= np.linalg.norm(d) + 1e-9
            p = np.clip(p + step * d, 0, shp - 1)
    tree = cKDTree(np.asarray(pts) * anf)
    ids = (1000 + rng.permutation(nchains)).astype(np.uint32)
    owner = np.asarray(owner)
    lab = np.empty(shape, dtype=np.uint32, order="F")
    gx, gy = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    base = np.stack([gx.ravel(order="F") * anf[0], gy.ravel(order="F") * anf[1]], axis=1)
    for z in range(shape[2]):  # per slab: bounded host memory
        q = np.concatenate([base, np.full((base.shape[0], 1), z * anf[2])], axis=1)
        _, idx = tree.query(q, workers=-1)
        lab[:, :, z] = ids[owner[idx]].reshape(shape[0], shape[1], order="F")
    if name in ("c2soma", "c2soma2"):
        centres = [(shp - 1) / 2.0] if name == "c2soma" else [(shp - 1) / 2.0 - np.array([128.0, 0, 0]), (shp - 1) / 2.0 + np.array([128.0, 0, 0])]
        gz = np.arange(shape[2])
        for k, c in enumerate(centres):
            d2 = (((gx - c[0]) * anf[0]) ** 2 + ((gy - c[1]) * anf[1]) ** 2)[:, :, None] + (((gz - c[2]) * anf[2]) ** 2)[None, None, :]
            lab[d2 <= 1400.0 ** 2] = np.uint32(999999 - k)
            v = c + np.array([20.0, -12.0, 4.0])       # the void: a 5 x 5 x 3 box of background off the centre
            lab[int(v[0]) - 2:int(v[0]) + 3, int(v[1]) - 2:int(v[1]) + 3, int(v[2]) - 1:int(v[2]) + 2] = 0
    if cache:
        os.makedirs(cache, exist_ok=True)
        np.save(path, lab)
    return lab, an


def make_balls(shape, nballs, seed):
    """`nballs` balls of radius U(9, 14) voxels on background 0, centres rejection-sampled at least 2 voxels
    apart (a later ball overwrites an earlier one where they overlap), label ids 1000 + perm(i); u32, F order."""
    rng = np.random.default_rng(seed)
    shp = np.asarray(shape, dtype=np.float64)
    lab = np.zeros(shape, dtype=np.uint32, order="F")
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float64)
    ids = (1000 + rng.permutation(nballs)).astype(np.uint32)
    centres = []
    for i in range(nballs):
        r = rng.uniform(9.0, 14.0)
        while True:
            c = rng.uniform(0, 1, 3) * (shp - 1)
            if all(np.linalg.norm(c - o) >= 2.0 for o in centres):
                break
        centres.append(c)
        lab[((g - c) ** 2).sum(-1) <= r * r] = ids[i]
    return lab


def cpu_baseline(cc_labels, remapping, an, params, dust_threshold, budget_s=15.0):
    """The oracle (oracle/kimi_oracle.c via oracle.pipeline.trace) on one core, label by label on the
    label's bounding box grown by one voxel (the EDT of the label's own voxels is exact there), until
    `budget_s` seconds of CPU work are spent.  Reported baseline only -- never the thing measured."""
    import scipy.ndimage
    import oracle
    from oracle import pipeline as P
    counts = np.bincount(cc_labels.ravel(order="K"))
    segids = [i for i in range(1, counts.size) if counts[i] > dust_threshold]
    rng = np.random.default_rng(0)
    rng.shuffle(segids)
    slices = scipy.ndimage.find_objects(cc_labels.T)
    t0 = time.perf_counter()
    done = 0
    vox = 0
    for sid in segids:
        slc = slices[sid - 1][::-1]
        glo = [max(0, s.start - 1) for s in slc]
        grown = tuple(slice(g, min(n, s.stop + 1)) for g, s, n in zip(glo, slc, cc_labels.shape))
        inner = tuple(slice(s.start - g, s.stop - g) for s, g in zip(slc, glo))
        crop = np.asfortranarray(cc_labels[grown])
        dbf = oracle.edt(crop, an, black_border=False)[inner]   # exact for the label's own voxels; traced on the reference's crop
        mask = np.asfortranarray(crop[inner] == sid)
        dbf = np.asfortranarray(np.where(mask, dbf, 0.0).astype(np.float32))
        P.trace(mask, dbf, anisotropy=an, fix_branching=True, **params)
        done += 1
        vox += int(counts[sid])
        if time.perf_counter() - t0 > budget_s and done >= 4:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "labels/s", "cores": 1, "kind": "port",
            "sample": "%d of %d labels (%d voxels), EDT on bbox+1 and full trace per label, %.1f s of CPU, "
                      "seeded shuffle" % (done, len(segids), vox, dt)}


def cpu_baseline_all_cores(cc_labels, an, params, dust_threshold, one_core_rate, volumes_in_flight=1, budget_s=25.0,
                           timeout_s=300.0):
    """the same per-label work as cpu_baseline on every usable host core: oracle/cpu_pool_baseline.py in a child process
    (forked worker pool over a read-only map of the component volume, components largest first), under a hard timeout.
    Two legs: ONE volume through the pool (latency) and `volumes_in_flight` volumes' components through the same pool at
    once (throughput) -- the like-for-like partners of single_volume_ms and of the pipelined `value`."""
    import subprocess
    import tempfile
    if (os.cpu_count() or 1) < 2:
        return None
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(tmpdir, "kimi_bench_cc_%d.npy" % os.getpid())
    try:
        np.save(path, cc_labels)
        args = {"anisotropy": [float(a) for a in an], "params": params, "dust_threshold": int(dust_threshold),
                "one_core_rate": float(one_core_rate), "budget_s": float(budget_s), "volumes_in_flight": int(volumes_in_flight)}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_pool_baseline.py"), path, json.dumps(args)],
                             capture_output=True, text=True, timeout=timeout_s, check=True)
        return json.loads(out.stdout.strip().splitlines()[-1])
    finally:
        if os.path.exists(path):
            os.remove(path)


def _daf_tie_note(workload):
    """the one tie the reference leaves to numpy's unstable argsort (CachedTargetFinder, skeletontricks.pyx:1004): how many skeletons
    of this workload depend on it (tools/daf_tie_exposure.py --flip, committed under profiles/)"""
    path = os.path.join(ROOT, "profiles", "r06_%s_daf_tie_flip.json" % workload)
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    o = d.get("opposite_tie_rule", {})
    return ("%d of %d target selections had an equal-DAF rival; with the opposite tie rule %d of %d skeletons differ (oracle, %s)"
            % (d["selections_with_an_equal_daf_rival"], d["target_selections"], o.get("skeletons_that_differ", -1),
               o.get("skeletons", -1), os.path.basename(path)))


def volume_step(e, st):
    """the whole step of one rank's share of the volume in `st` on engine e (current stream of the calling thread):
    connected components -> EDT -> statistics -> border targets -> searches -> path loop -> host skeletons."""
    from collections import defaultdict
    from kimimaro_amd import intake
    lab = st["lab"]
    log = st.get("phase_log")          # developer knob KIMI_BENCH_LANE_PHASES=1: where a volume's time goes UNDER LOAD
    tm = [("begin", time.perf_counter())] if log is not None else None
    d_cc, nlabels, rep_ = e.ccl_device(st["d_lab"], lab.dtype.itemsize, lab.shape)
    orig = st["flat"][rep_[1:].astype(np.int64)]
    remapping = {i + 1: orig[i].item() for i in range(nlabels)}
    cc = intake.LazyVolume(e, d_cc, lab.shape)
    del d_cc                           # (reached through `cc`, which lets go of the u32 ids as soon as the u16 copy serves)
    empty = defaultdict(list)
    out = intake.skeletonize_cc(e, cc, nlabels, remapping, st["params"], st["an"], st["dust"], True, st["fix_borders"],
                                empty, empty, black_border=False, rank=st["shard"][0], world=st["shard"][1], timings=tm)
    if log is not None:
        tm.append(("end", time.perf_counter()))
        log.append(tm)
    return out


def _lane_setup(eng, index, path, an, params, dust, fix_borders, shard):
    """a lane process of kimimaro_amd.lanes.ProcessLanes: its own copy of the label volume, resident in ITS HBM allocation"""
    lab = np.asfortranarray(np.load(path, mmap_mode="r"))
    st = {"lab": lab, "d_lab": eng.to_device(lab), "flat": lab.reshape(-1, order="F"), "an": np.asarray(an, dtype=np.float32),
          "params": dict(params), "dust": dust, "fix_borders": fix_borders, "shard": tuple(shard)}
    eng.sync()
    return st


def _lane_step(st, eng, payload):
    return volume_step(eng, st)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("KIMI_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("KIMI_BENCH_INFLIGHT", "0")),
                    help="volumes in flight per GPU: consecutive steps are issued from that many host threads, each on a HIP "
                         "stream and with scratch of its own, so the tail of one volume (a handful of workgroups tracing its "
                         "largest components) overlaps the next volumes.  0 (default) = 4, and 7 / 10 / 12 for the strong mode on "
                         "2 / 4 / 8 GPUs (a rank's share of a volume is smaller there, the chain of its largest component "
                         "is not).  1 = one step after the other (the latency "
                         "of a single volume, which is reported either way as single_volume_ms).")
    ap.add_argument("--lanes", choices=["thread", "process"], default=os.environ.get("KIMI_BENCH_LANES", "thread"),
                    help="what a lane is: a host thread of this process (kimimaro_amd.lanes.Lanes) or a process of its own "
                         "(kimimaro_amd.lanes.ProcessLanes: no shared interpreter lock)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fix-borders", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=os.environ.get("KIMI_BENCH_SCALING", "strong"),
                    help="N > 1: strong (default, BASELINE.json configs[3]) = ONE volume, its components dealt over the GPUs "
                         "(largest first to the least loaded rank); weak = every GPU skeletonizes its own volume of the "
                         "workload's size (a mirrored copy with its own label ids).  The mode that is not selected is "
                         "measured as well (fewer steps) and printed as an extra object.")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    # KIMI_BENCH_BACKEND=gloo is a dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks then
    # share devices and the skeleton exchange goes through host tensors); the measured configuration is nccl (= RCCL)
    backend = os.environ.get("KIMI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import kimimaro_amd
    from kimimaro_amd import _abi, intake
    from kimimaro_amd.engine import Engine
    from kimimaro_amd.distributed import gather_skeletons

    eng = Engine()
    if world > 1:
        # rank 0 builds the volume once and broadcasts it over RCCL (8 ranks running the KD-tree recipe at the
        # same time on one host would take minutes); any failure falls back to local generation (deterministic).
        shape0, _, _, _, an = WORKLOADS[args.workload]
        try:
            if rank == 0:
                base_lab, an = make_volume(args.workload)
                d_vol = eng.to_device(base_lab)
            else:
                d_vol = torch.empty(int(np.prod(shape0)), dtype=torch.int32, device=eng.device)
            dist.broadcast(d_vol, src=0)
            if rank != 0:
                base_lab = d_vol.cpu().numpy().view(np.uint32).reshape(shape0, order="F")
            del d_vol
        except Exception as e:  # pragma: no cover
            print("bench: broadcast of the volume failed (%r); generating locally" % (e,), file=sys.stderr)
            base_lab, an = make_volume(args.workload)
    else:
        base_lab, an = make_volume(args.workload)
    an = np.asarray(an, dtype=np.float32)
    params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
    dust = 1000
    fix_borders = not args.no_fix_borders
    from collections import defaultdict
    empty = defaultdict(list)
    result = {}
    state = {}

    def prepare(mode):
        """the label volume of this rank for `mode`, resident in HBM (outside the timed region)."""
        lab = base_lab
        if mode == "weak" and world > 1:
            # one volume per GPU: rank r works on the copy mirrored along the axes given by the bits of r (same object
            # statistics, same largest component, different geometry) with label ids of its own, so that the gathered
            # result holds the skeletons of all N volumes.  No component is shared between ranks.
            flips = [a for a in range(3) if (rank >> a) & 1]
            lab = np.asfortranarray(np.flip(lab, axis=flips)) if flips else lab
            lab = np.where(lab != 0, lab + np.uint32(1000000 * rank), 0).astype(np.uint32, order="F")
        t = time.perf_counter()
        lab = intake.format_labels(lab, in_place=True)
        state["lab"] = lab
        state["d_lab"] = eng.to_device(lab)  # the input volume is resident in HBM before the timed region
        eng.sync()
        state["flat"] = lab.reshape(-1, order="F")
        state["shard"] = (rank, world) if mode == "strong" else (0, 1)
        state.update(an=an, params=params, dust=dust, fix_borders=fix_borders)
        dt = time.perf_counter() - t
        if args.lanes == "process" and widths.get(mode, 1) > 1:
            # the lane processes load the prepared volume from shared memory and keep their own copy in HBM
            path = "/dev/shm/kimi_bench_%d_%d_%s.npy" % (os.getpid(), rank, mode)
            np.save(path, lab)
            state["shm"] = path
        return dt

    def components():
        lab = state["lab"]
        d_cc, n, rep = eng.ccl_device(state["d_lab"], lab.dtype.itemsize, lab.shape)  # kimimaro/utility.py:58-83 on the GPU
        orig = state["flat"][rep[1:].astype(np.int64)]
        return d_cc, n, {i + 1: orig[i].item() for i in range(n)}

    def local_step(e):
        return volume_step(e, state)

    def finish(local):
        if world > 1:
            tg = time.perf_counter()
            local = gather_skeletons(local, device=eng.device if backend == "nccl" else None)
            state["gather_s"] = state.get("gather_s", 0.0) + (time.perf_counter() - tg)
        result["skels"] = local
        return local

    def step():
        return finish(local_step(eng))

    # volumes in flight (kimimaro_amd/lanes.py): one host thread + HIP stream + Engine per lane.  The collective of a step
    # is issued by this thread, in step order, so every rank issues the same sequence.
    from kimimaro_amd.lanes import Lanes
    def width_of(mode):
        if args.inflight > 0:
            return args.inflight
        # A lane costs HBM: the whole-volume fields of its volume (~7.5 GB at 512^3) + the per-label scratch of the components
        # it traces (~11 GB for all components of c3; a rank of the strong mode holds 1 / N of them).  As many lanes as 80 %
        # of the memory that is free now pays for, 24 at most (round 6: twenty fit): the step time keeps falling up to there (one volume's chain
        # of ~3 s is overlapped by the others' GPU-filling phases; measured at c3: 4 / 8 lanes = 1220 / 784 ms per step).
        from kimimaro_amd.lanes import lanes_for
        share = 1.0 / world if (mode == "strong" and world > 1) else 1.0
        state["hbm_free_gb"] = round(torch.cuda.mem_get_info()[0] / 1e9, 1)
        most = lanes_for(WORKLOADS[args.workload][0], torch.cuda.mem_get_info()[0], most=24, share=share)
        # K equal volumes started together stay in lock step, so a run of K steps is ceil(K / lanes) rounds: the fewest
        # rounds the memory allows, and no more lanes than fill them evenly (K = 20: 10 + 10 rather than 12 + 8)
        rounds = -(-max(args.steps, 1) // most)
        return int(-(-max(args.steps, 1) // rounds))

    widths = {m: width_of(m) for m in (("weak", "strong") if world > 1 else (args.scaling,))}
    if dist:
        # every rank must run the same number of fill steps (each ends in the gather collective): the ranks agree on the
        # smallest lane count any of them came up with from its own free memory
        dev_w = eng.device if backend == "nccl" else "cpu"
        for m_ in sorted(widths):
            wt = torch.tensor([widths[m_]], dtype=torch.int64, device=dev_w)
            dist.all_reduce(wt, op=dist.ReduceOp.MIN)
            widths[m_] = int(wt.item())
    lanes = Lanes(max(widths.values()), device=eng.device) if max(widths.values()) > 1 and args.lanes == "thread" else None
    plane = {"obj": None, "peak": 0}      # process lanes of the mode being measured (kimimaro_amd.lanes.ProcessLanes)

    def open_process_lanes(width):
        from kimimaro_amd.lanes import ProcessLanes
        if args.lanes == "process" and width > 1:
            plane["obj"] = ProcessLanes(width, setup=_lane_setup, device=eng.device.index,
                                        setup_args=(state["shm"], [float(a) for a in an], params, dust, fix_borders, state["shard"]))

    def close_process_lanes():
        if plane["obj"] is not None:
            plane["obj"].close()
            plane["peak"] = max(plane["peak"], sum(s_["hbm_reserved_peak"] for s_ in plane["obj"].stats))
            plane["obj"] = None
        if state.get("shm"):
            if os.path.exists(state["shm"]):
                os.remove(state["shm"])
            state["shm"] = None

    def run_steps(n, width):
        if plane["obj"] is not None and width > 1:
            want = os.environ.get("KIMI_BENCH_STAGGER")
            on = (n >= 4 * width) if want is None else (want == "1" and n > width)
            stagger = state.get("single_ms", 0.0) / 1e3 / width if on else 0.0
            for _, local in plane["obj"].run(_lane_step, [None] * n, width=width, stagger=stagger):
                finish(local)
            return
        if lanes is None or width <= 1:
            for _ in range(n):
                finish(local_step(lanes.engines[0] if lanes is not None else eng))
            return
        # lanes offset by latency / width, so that the GPU-filling part of one volume meets the tails of the others instead
        # of the other lanes' GPU-filling parts: measured +2.3 % over 20 steps, -3 % over 8 (the ramp costs more than the
        # interleaving wins), hence only for long runs.  KIMI_BENCH_STAGGER=0 / 1 forces it off / on.
        want = os.environ.get("KIMI_BENCH_STAGGER")
        on = (n >= 4 * width) if want is None else (want == "1" and n > width)
        stagger = state.get("single_ms", 0.0) / 1e3 / width if on else 0.0
        if os.environ.get("KIMI_BENCH_STAGGER_S"):           # developer knob: seconds between the lanes' first jobs
            stagger = float(os.environ["KIMI_BENCH_STAGGER_S"]) if n > width else 0.0
        for _, local in lanes.run(lambda e, k: local_step(e), n, width=width, stagger=stagger):
            finish(local)

    def measure(mode, warmup, steps, latency=True):
        width = widths[mode]
        torch.cuda.empty_cache()           # scratch of the other mode / of the preparation goes back to the device
        if latency:
            # one volume alone on an otherwise idle GPU, the way kimimaro_amd.skeletonize() runs it (default engine: 256 threads
            # per label, the largest labels on a second stream): its latency, untimed; twice, the first call fills the pool
            eng.time_kernels = True        # HIP events around the path-loop launches of these PRODUCTION calls (-> roofline)
            for _ in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                state["single_ms"] = (time.perf_counter() - t1) * 1e3
                state["production_kernel_ms"] = list(getattr(eng, "last_path_kernel_ms", None) or [])
                state["production_span_ms"] = float(getattr(eng, "last_path_span_ms", 0.0) or 0.0)
            eng.time_kernels = False
            torch.cuda.empty_cache()
        open_process_lanes(width)
        if width > 1:
            run_steps(width, width)        # every lane once: fills the scratch pool of its stream (not a warm-up step)
        run_steps(warmup, width)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        state["gather_s"] = 0.0
        if os.environ.get("KIMI_BENCH_LANE_PHASES") == "1" and width > 1 and plane["obj"] is None:
            state["phase_log"] = []        # (every mark of the log is a stream synchronisation of its lane: a diagnostic run)
        t0 = time.perf_counter()
        run_steps(steps, width)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0      # this rank's own steps (before it waits for the slowest rank)
        if state.get("phase_log"):
            acc, first = {}, min(tm[0][1] for tm in state["phase_log"])
            for tm in state["phase_log"]:
                for (_, a), (name, b) in zip(tm[:-1], tm[1:]):
                    acc.setdefault(name, []).append(b - a)
            state["phases_under_load"] = {"mean_s": {k: round(float(np.mean(v)), 4) for k, v in acc.items()},
                                          "max_s": {k: round(float(np.max(v)), 4) for k, v in acc.items()},
                                          "volume_s": [round(tm[-1][1] - tm[0][1], 3) for tm in state["phase_log"]],
                                          "start_s": [round(tm[0][1] - first, 3) for tm in state["phase_log"]]}
        state["phase_log"] = None
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            dev = eng.device if backend == "nccl" else "cpu"
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            mine = torch.tensor([own, state["gather_s"]], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            state["rank_times"] = {"own_s_per_step": [round(float(e[0]) / max(steps, 1), 4) for e in every],
                                   "gather_s_per_step": [round(float(e[1]) / max(steps, 1), 4) for e in every]}
        return elapsed

    # the other scaling mode first (short), the selected one last so that everything below describes the selected one
    other = None
    if world > 1:
        omode = "weak" if args.scaling == "strong" else "strong"
        prepare(omode)
        osteps = max(1, min(args.steps, 4))
        try:
            oel = measure(omode, 1, osteps, latency=False)
        finally:
            close_process_lanes()          # (also removes the /dev/shm copy of the volume when the measurement raised)
        other = {"mode": omode, "elapsed": oel, "steps": osteps, "skeletons": len(result["skels"])}
    preamble_s = prepare(args.scaling)
    try:
        elapsed = measure(args.scaling, args.warmup, args.steps)
    finally:
        close_process_lanes()              # (their HBM goes back before the instrumented pass of this process)
    # what the lanes' last volumes looked like UNDER LOAD (thread lanes: the engines are still there): the longest chain of
    # each and the cycle sums, to be read against the solo pass below (`chains`)
    loaded = None
    if lanes is not None:
        loaded = []
        for e_ in lanes.engines:
            lt = e_.last_tasks
            if lt is None or not len(lt):
                continue
            tot_ = lt["cyc_target"].astype(np.int64) + lt["cyc_rail"].astype(np.int64) + lt["cyc_inval"].astype(np.int64)
            j = int(np.argmax(tot_))
            loaded.append({"longest_voxels": int(lt["count"][j]), "longest_Mcyc": round(int(tot_[j]) * 1024 / 1e6, 1),
                           "longest_heap_pushes": int(lt["stat_heap_pushes"][j]),
                           "sum_Mcyc_inval": round(float(lt["cyc_inval"].astype(np.int64).sum()) * 1024 / 1e6, 0),
                           "sum_Mcyc_rail": round(float(lt["cyc_rail"].astype(np.int64).sum()) * 1024 / 1e6, 0)})
    inflight = widths[args.scaling]
    # ---- instrumented pass (phase times, sweep statistics): one volume alone, on lane 0 while its scratch pool is warm
    import contextlib
    import kimimaro_amd.engine as E
    ieng = lanes.engines[0] if lanes is not None else eng
    timings = []
    tk = None
    if rank == 0:
        with (lanes._scopes[0] if lanes is not None else contextlib.nullcontext()):
            torch.cuda.synchronize()
            t_ccl = time.perf_counter()
            lab0 = state["lab"]
            i_cc, i_n, i_rep = ieng.ccl_device(state["d_lab"], lab0.dtype.itemsize, lab0.shape)
            ieng.sync()
            t_ccl = time.perf_counter() - t_ccl
            i_orig = state["flat"][i_rep[1:].astype(np.int64)]
            intake.skeletonize_cc(ieng, intake.LazyVolume(ieng, i_cc, lab0.shape), i_n, {i + 1: i_orig[i].item() for i in range(i_n)},
                                  params, an, dust, True, fix_borders, empty, empty, black_border=False, d_cc=i_cc, timings=timings)
            tk = ieng.last_tasks
            dump = os.environ.get("KIMI_BENCH_DUMP_TASKS")      # tools/strong_scaling_model.py reads this (per-label voxels and cycles)
            if dump and tk is not None:
                np.savez_compressed(dump, **{k: tk[k] for k in tk.dtype.names})      # the whole task records (kh_label_t)
            state["retries"] = getattr(ieng, "last_retries", 0)
            state["path_kernel_ms"] = list(getattr(ieng, "last_path_kernel_ms", []))
            del i_cc
    lanes = None
    torch.cuda.empty_cache()
    lab = state["lab"]
    shape = lab.shape
    nskel = len(result["skels"])
    d_cc, nlabels, remapping = components()
    cc_labels = eng.to_host_volume(d_cc, shape)
    counts = np.bincount(cc_labels.ravel(order="K"))
    ncomp = int((counts[1:] > dust).sum())
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    units = ncomp * (world if args.scaling == "weak" else 1)   # weak: every rank has a volume with the same component count
    value = units / (ms_per_step / 1e3)
    if other is not None:
        oms = other["elapsed"] / other["steps"] * 1e3
        ounits = ncomp * (world if other["mode"] == "weak" else 1)
        other = {"scaling": other["mode"], "value": round(ounits / (oms / 1e3), 3), "unit": "labels/s", "ms_per_step": round(oms, 3),
                 "steps": other["steps"], "warmup": 1, "skeletons": other["skeletons"], "volumes_in_flight": widths[other["mode"]],
                 "note": ("every GPU its own volume of the workload's size (mirrored copies, own label ids)" if other["mode"] == "weak"
                          else "components of ONE volume dealt over the GPUs")}

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the EDT pass kernels: HIP events on the launch stream (kh_edt_timed)
    import ctypes as C
    nvox = int(np.prod(shape))
    out = eng.empty(nvox, torch.float32)
    ws = eng.empty(2 * nvox, torch.float32)
    d_cc, _, _ = components()
    d_edt_lab, L = eng.narrow(d_cc)     # u16 component ids when there are < 65536 of them, as the step itself uses
    ms3 = (C.c_float * 3)()
    acc = np.zeros(3)
    reps = 10
    for i in range(reps + 2):
        _abi.check(eng.lib.kh_edt_timed(eng.ptr(d_edt_lab), L, shape[0], shape[1], shape[2], float(an[0]), float(an[1]),
                                        float(an[2]), 0, eng.ptr(ws), eng.ptr(out), eng.stream(), ms3))
        if i >= 2:
            acc += np.array(list(ms3))
    pass_ms = acc / reps
    pass_bytes = np.array([(L + 4) * nvox, (L + 8) * nvox, (L + 8) * nvox], dtype=np.float64)
    k = int(np.argmax(pass_ms))
    lt = {2: "uint16", 4: "uint32"}[L]
    names = ["edt_x_rows_kernel<%s> (x pass)" % lt, "edt_axis_kernel<%s> (y pass)" % lt, "edt_axis_kernel<%s> (z pass)" % lt]
    achieved = pass_bytes[k] / (pass_ms[k] * 1e-3) / 1e9
    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    # runs, gfx950 x2 FETCH correction calibrated on the x pass): the newest profiles/rNN_c3_edt_pmc.json.  Only valid for c3.
    traffic = None
    pmc = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r06c_c3_edt_pmc.json", "r06_c3_edt_pmc.json", "r05_c3_edt_pmc.json", "r04_c3_edt_pmc.json", "r02_c3_edt_pmc.json"))
                if os.path.exists(q)), "")
    if args.workload == "c3" and os.path.exists(pmc):
        kern = json.load(open(pmc))["kernels"]
        lts = {2: "unsigned short", 4: "unsigned int"}[L]
        tag = ["edt_x_", "edt_axis_kernel<%s, false" % lts, "edt_axis_kernel<%s, true" % lts][k]
        hit = [v for name, v in kern.items() if tag in name]
        if hit:
            traffic = hit[0]["hbm_bytes_corrected"]
    roofline_edt = {"bound": "hbm", "kernel": names[k], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": "profiles/%s (rocprofv3 --pmc passes of tools/edt_only.py on the same volume%s; not live)"
                                  % (os.path.basename(pmc), "" if ("r06c_" in pmc) else "; STALE: measured on the round-5 y / z kernels" if ("r05_" in pmc or "r06_" in pmc) else "; STALE: measured before round 5's x pass") if traffic else None,
                "bytes_per_launch": int(pass_bytes[k]), "ms_per_launch": round(float(pass_ms[k]), 4),
                "edt_pass_ms": [round(float(x), 4) for x in pass_ms],
                "edt_total_GBps": round(float((3 * L + 20) * nvox / (pass_ms.sum() * 1e-3) / 1e9), 1),
                # SURVEY 8d's EDT figure: (3L + 20) B per voxel over the three passes together, as a fraction of the HBM peak
                "edt_total_frac": round(float((3 * L + 20) * nvox / (pass_ms.sum() * 1e-3) / 1e9) / HBM_PEAK_GBS, 4)}

    # ---- the path kernel: per-label algorithmic bytes of SURVEY 8d with Vc := Nf (no crops here)
    phases = {"ccl": round(t_ccl, 4)}
    prev = None
    for name, ts in timings:
        if prev is not None:   # (a name repeats when the labels ran in several launches: accumulated)
            key = "host_setup" if name == "setup" else name
            phases[key] = round(phases.get(key, 0.0) + ts - prev, 4)
        prev = ts
    nf = tk["count"].astype(np.float64).sum()
    settled = tk["stat_settled"].astype(np.float64).sum()
    trace_bytes = (4 + 9) * nf + 10 * nf + 12 * nf + 12 * nf + 12 * settled + 2 * nf
    tr_s = phases.get("paths", float("nan"))
    calls = int(tk["stat_sweep_calls"].astype(np.int64).sum())
    bails = int(tk["stat_sweep_bails"].astype(np.int64).sum())
    ghost_calls = int(tk["stat_ghost_calls"].astype(np.int64).sum())
    rollbacks = int(tk["stat_rollbacks"].astype(np.int64).sum())
    sweep = {"invalidation_calls": calls, "certified": calls - bails, "fell_back_to_heap": bails,
             "calls_that_went_on_with_ghosts": ghost_calls, "rollbacks": rollbacks,
             "labels_with_ghosts": int(np.count_nonzero(tk["stat_ghost_calls"])),
             "levels": int(tk["stat_sweep_levels"].astype(np.int64).sum()), "events": int(tk["stat_sweep_events"].astype(np.int64).sum()),
             "labels_with_fallback": int(np.count_nonzero(tk["stat_sweep_bails"])),
             "voxels_of_labels_with_fallback": int(tk["count"][tk["stat_sweep_bails"] > 0].astype(np.int64).sum()),
             "voxels": int(tk["count"].astype(np.int64).sum()),
             "note": "order-free level sweep (csrc/sweep.h); a call it cannot certify is redone by the exact heap emulation"}
    # HBM traffic of that kernel from the committed counter passes (tools/pmc_trace_r3.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs over one c3 volume, FETCH x 2 on gfx950): only valid for c3
    tr_traffic, tr_src = None, None
    tpmc = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r06c_c3_trace_pmc.json", "r06_c3_trace_pmc.json", "r05_c3_trace_pmc.json",
                                                                         "r04_c3_trace_pmc.json")) if os.path.exists(q)), "")
    if args.workload == "c3" and os.path.exists(tpmc):
        for name, v in json.load(open(tpmc))["kernels"].items():
            if "trace_paths_kernel" in name and v.get("hbm_bytes_corrected_per_volume"):
                # (the volume's launches -- the largest labels as a kernel variant of their own -- added up)
                tr_traffic = (tr_traffic or 0.0) + v["hbm_bytes_corrected_per_volume"]
                stale = "" if ("r06_" in os.path.basename(tpmc) or "r06c_" in os.path.basename(tpmc)) else "; STALE: measured on an earlier round's kernels"
                tr_src = ("profiles/%s (rocprofv3 --pmc passes over one volume, all path-kernel launches of the volume; not live%s)"
                          % (os.path.basename(tpmc), stale))
    # the dominant kernel (97 % of the GPU time): its launches of ONE volume overlap on two streams (the largest labels on the
    # second one), so the duration that counts is the span of the path phase; the HIP events of each launch are listed beside it
    # (production_kernel_ms: the launches of the single_volume_ms call -- the shipped kernels; path_kernel_ms: the one launch of
    # the instrumented pass below, the profile build, which is where the cycle counters of `chains` come from)
    kms = state.get("production_kernel_ms") or state.get("path_kernel_ms") or []
    span_s = max([m for _, m in kms], default=float("nan")) / 1e3 if kms else tr_s
    if state.get("production_kernel_ms") and state.get("production_span_ms", 0.0) > 0.0:
        span_s = state["production_span_ms"] / 1e3          # first start to last end of the volume's (overlapped) launches
    roofline = {"bound": "hbm", "kernel": "trace_paths_kernel", "achieved": round(trace_bytes / span_s / 1e9, 3),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(trace_bytes / span_s / 1e9 / HBM_PEAK_GBS, 6),
                "traffic": tr_traffic, "traffic_source": tr_src, "bytes_per_launch": int(trace_bytes),
                "ms_per_launch": round(span_s * 1e3, 3),
                "launches": [{"labels": int(c), "ms": round(float(m), 3)} for c, m in kms],
                "launch_note": "algorithmic bytes of ONE volume (SURVEY 8d, all its labels) / the span (first start to last end) of the volume's overlapped "
                               "path-loop launches (the 256 largest labels run as trace_paths_kernel<false, 2> on a second stream, the "
                               "rest as <false, 1>), HIP events on each launch's own stream, taken during the single_volume_ms call "
                               "(one volume alone on the GPU, production kernels)",
                "instrumented_launch_ms": [round(float(m), 3) for _, m in (state.get("path_kernel_ms") or [])],
                "phase_seconds": tr_s, "heap_pushes": int(tk["stat_heap_pushes"].astype(np.int64).sum()),
                "note": "latency bound: level-synchronous sweep per label; the wall clock is the largest label whose call "
                        "needed the exact heap emulation"}

    # the labels whose chains set the wall clock of the path kernel (clock64 ticks / 1024 per phase, solo pass)
    tot = tk["cyc_target"].astype(np.int64) + tk["cyc_rail"].astype(np.int64) + tk["cyc_inval"].astype(np.int64)
    chains = []
    for i in np.argsort(-tot)[:4]:
        chains.append({"voxels": int(tk["count"][i]), "paths": int(tk["n_paths"][i]), "Mcyc_target": round(int(tk["cyc_target"][i]) * 1024 / 1e6, 1),
                       "Mcyc_rail": round(int(tk["cyc_rail"][i]) * 1024 / 1e6, 1), "Mcyc_inval": round(int(tk["cyc_inval"][i]) * 1024 / 1e6, 1),
                       "heap_pushes": int(tk["stat_heap_pushes"][i]), "sweep_calls": int(tk["stat_sweep_calls"][i]),
                       "sweep_bails": int(tk["stat_sweep_bails"][i]), "sweep_levels": int(tk["stat_sweep_levels"][i]),
                       "sweep_events": int(tk["stat_sweep_events"][i]), "bail_why": int(tk["stat_sweep_why"][i])})
    nofb = tk["stat_sweep_bails"] == 0
    chain_info = {"longest": chains,
                  "sum_Mcyc": {"target": round(float(tk["cyc_target"].astype(np.int64).sum()) * 1024 / 1e6, 0),
                               "rail": round(float(tk["cyc_rail"].astype(np.int64).sum()) * 1024 / 1e6, 0),
                               "inval": round(float(tk["cyc_inval"].astype(np.int64).sum()) * 1024 / 1e6, 0),
                               "inval_of_labels_without_fallback": round(float(tk["cyc_inval"][nofb].astype(np.int64).sum()) * 1024 / 1e6, 0)},
                  "bail_reasons_or": int(np.bitwise_or.reduce(tk["stat_sweep_why"].astype(np.int64))) if len(tk) else 0,
                  "labels_bailing_for_arena": int(np.count_nonzero(tk["stat_sweep_why"] & 4)),
                  "labels_bailing_for_lists_or_levels": int(np.count_nonzero(tk["stat_sweep_why"] & (8 | 16))),
                  "labels_retraced_for_scratch": int(state.get("retries", 0))}

    cpu = None
    cpu_all = None
    if not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(cc_labels, remapping, an, params, dust)
            try:
                # the pool is throughput bound (245 components/s for one volume, 248 for four, 247 for eight at once on the
                # 16-CPU cgroup of the pool's boxes): four volumes' worth is the bounded sample of the throughput leg
                cpu_all = cpu_baseline_all_cores(cc_labels, an, params, dust, cpu["value"], volumes_in_flight=min(inflight, 4))
                if isinstance(cpu_all, dict):
                    cpu_all["volumes"] = min(inflight, 4)      # the throughput leg's sample (the pool is saturated from one volume on)
            except Exception as e:
                cpu_all = {"value": None, "unit": "labels/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        except Exception as e:  # the baseline must never take the bench line down
            cpu = {"value": None, "unit": "labels/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}

    # like-for-like ratios: one volume vs one volume, pipelined vs pipelined (never mixed)
    speedup_latency = speedup_throughput = None
    single_ms = state.get("single_ms", float("nan"))
    if cpu_all and cpu_all.get("latency") and cpu_all["latency"].get("value") and single_ms == single_ms:
        speedup_latency = round((ncomp / (single_ms / 1e3)) / cpu_all["latency"]["value"], 2)
    if cpu_all and cpu_all.get("throughput") and cpu_all["throughput"].get("value") and world == 1:
        speedup_throughput = round(value / cpu_all["throughput"]["value"], 2)

    line = {
        "metric": "labels/sec on a dense connectomics-shaped volume (skeletonize hot path, labels resident in HBM)",
        "value": round(value, 3), "unit": "labels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling,
        "volumes_in_flight": inflight, "single_volume_ms": round(state.get("single_ms", float("nan")), 3),
        "value_single_volume": round(ncomp / (single_ms / 1e3), 3) if single_ms == single_ms else None,
        "hbm_reserved_peak_gb": round((torch.cuda.max_memory_reserved() + plane["peak"]) / 1e9, 1),
        "hbm_free_when_lanes_were_chosen_gb": state.get("hbm_free_gb"),
        "daf_tie_note": _daf_tie_note(args.workload),
        "lanes": args.lanes if inflight > 1 else "none",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d uint32, %d chains -> %d components > dust, anisotropy=%s, "
                               "default teasar_params, fix_branching=True, fix_borders=%s, dust_threshold=%d"
                               % (args.workload, shape[0], shape[1], shape[2], WORKLOADS[args.workload][1], ncomp,
                                  tuple(float(a) for a in an), fix_borders, dust),
                   "parallelism": ("one such volume per GPU (mirrored copies, own label ids) on %d GPU(s), no data-path "
                                   "collective, skeleton all-gather-v at the end" % world) if args.scaling == "weak" else
                                  ("components of ONE volume dealt over %d GPU(s) (largest first to the least loaded rank), "
                                   "skeleton all-gather-v" % world),
                   "volumes_in_flight": inflight},
        "skeletons": nskel, "labels_per_s_by_label_count": round(nskel * (world if args.scaling == "weak" else 1) / (ms_per_step / 1e3), 3),
        "preamble_s": round(preamble_s, 3), "phases_s": phases, "sweep": sweep, "chains": chain_info, "chains_under_load": loaded,
        "phases_under_load": state.get("phases_under_load"),
        "roofline": roofline, "roofline_edt": roofline_edt, "cpu_baseline": cpu,
        "cpu_baseline_all_cores": cpu_all,
        "speedup_latency": speedup_latency, "speedup_throughput": speedup_throughput,
        "speedup_note": "latency: components / single_volume_ms vs ONE volume on the all-cores pool; throughput: value (%d volumes "
                        "in flight) vs %d volumes' components through the same pool at once (cpu_baseline_all_cores.volumes: the "
                        "pool is saturated from one volume on, 245 / 248 / 247 components/s for 1 / 4 / 8 volumes)"
                        % (inflight, min(inflight, 4)),
    }
    if state.get("rank_times"):
        line["rank_times"] = state["rank_times"]
    if other is not None:
        line["weak_scaling" if other["scaling"] == "weak" else "strong_scaling"] = other
    print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
