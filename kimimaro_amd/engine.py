"""Host orchestration of the HIP kernels (one process = one GPU).

PyTorch is used ONLY as plumbing: device allocations (caching allocator), H2D/D2H copies and the
current HIP stream.  Every computation is a hand-written kernel in libkimi_hip.so reached through the
C ABI of include/kimi_hip.h.

The engine works on whole-volume, Fortran-ordered 1-D device arrays.  Because connected components
are disjoint, all labels share one DAF / PDRF / dist / alive volume and every label is processed by
its own workgroup (kimimaro/intake.py:434-517 runs them one after the other on cropped copies).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import _abi

NONE32 = 0xFFFFFFFF


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _abi.HipUnavailableError("kimimaro_amd: torch sees no GPU; the product path has no CPU fallback.")
    return torch


# per-label scratch of the path loop as plan_launches books it: work lists 16 B per voxel, heap 24, event arena ~24, voxel list and DAF
# 8, ghost journal 8, path buffers and saved rail weights ~6 = 86 B, booked as 110; + the fixed parts of a small label (a 32 768-node
# heap, the arena's level chunks, 64 Ki path slots).  Round 2 booked 300 B per voxel (28 GB of event arena per c3 volume then):
# c5 ran as three launches one after the other, 12.4 s of paths; as one launch it is 91 -> ~200 GB of HBM and half the time.
SCRATCH_BYTES_PER_VOXEL = 60        # round 6: voxel list + DAF 8, work lists 16, the pool's share of heap and journal ~5, path buffers ~1,
SCRATCH_BYTES_PER_LABEL = 3 << 19    # booked with slack; per label: the event arena (1.25 x the level window + 320 chunks: 0.6-1.4 MB)
# Round 6: heap and ghost journal come out of one pool per launch, on demand (KH_TRACE_SCRATCH_POOL): this fraction of what all
# labels together could ask for (c3: 141 of 3 402 labels ever run the heap emulation -- 6 % of the nodes --, 360 ever hold a ghost);
# a label the pool cannot serve is traced again with scratch of its own, like every other overflow
SCRATCH_POOL_FRACTION = float(os.environ.get("KH_SCRATCH_POOL_FRACTION", "0.15"))


def plan_launches(counts, budget):
    """Groups of label positions for successive launches of the path loop: largest labels first, a group is closed when the
    next label's scratch (SCRATCH_BYTES_PER_VOXEL per voxel + SCRATCH_BYTES_PER_LABEL) would take it over `budget` bytes; a label
    larger than the budget gets a launch of its own.  One group = everything fits."""
    counts = np.asarray(counts, dtype=np.int64)
    need = counts * SCRATCH_BYTES_PER_VOXEL + SCRATCH_BYTES_PER_LABEL
    if int(need.sum()) <= budget:
        return [list(range(len(counts)))]
    groups, cur, acc = [], [], 0
    for i in np.argsort(-counts, kind="stable").tolist():
        if cur and acc + int(need[i]) > budget:
            groups.append(cur)
            cur, acc = [], 0
        cur.append(i)
        acc += int(need[i])
    groups.append(cur)
    return groups


SCHED_LEVELS = (1 << 17) - 1     # SW_SCHED_LEVELS (csrc/sweep.h): labels with more levels run the sweep unfiltered


def plan_arena(cnt, nlev, filtered, window=None):
    """(log2 slots per chunk, chunks) of the sweep's event arena per label (numpy arrays; csrc/sweep.h: fixed-size chunks of
    8-byte events chained per level; the chunks of a level go back to a free stack when the level has been processed).
    Unfiltered a voxel is handed ~13 events per call; with the pending-deadline filter ~2 (measured at c3: 1.72e9 -> 2.8e8
    events per volume), and a level of a large call holds tens of events instead of hundreds -- smaller chunks, a fraction of
    the event budget.  What has to fit is what is PENDING at one time: one partly filled chunk per level that has events
    (at most `window` levels when the label has a level window, never more levels than the label can get events) plus the
    pending events themselves.  An arena that runs out makes the call fall back to the heap emulation (SW_BAIL_ARENA in
    stat_sweep_why): a matter of speed, not of results."""
    cnt = np.asarray(cnt, dtype=np.int64)
    nlev = np.asarray(nlev, dtype=np.int64)
    filt = np.asarray(filtered, dtype=bool) & (nlev <= SCHED_LEVELS)
    shift = np.where(filt, np.where(cnt >= 65536, 6, 5), np.where(cnt >= 32768, 7, 6)).astype(np.int64)
    per_voxel = np.where(filt, 3, 14)
    levels = np.where(filt, np.minimum(nlev, 3 * cnt + 64), nlev)
    if window is not None:
        window = np.asarray(window, dtype=np.int64)
        levels = np.where(window > 0, np.minimum(levels, window), levels)
    chunks = levels + levels // 2 + ((per_voxel * cnt) >> shift) + 320
    if window is not None:
        # Round 6, measured on c3 (`cyc_push` of the task records = most chunks a label ever had in use): 0.41 x the window on average,
        # 0.94 x at the 99th percentile, 1.21 x at most -- the chunks of a level go back to the label's free stack when the level is
        # done, so what is in use is what is PENDING, and that is bounded by the window, not by the label's size.  The arena was
        # 6.9 GB per c3 volume for 0.55 GB of use, and the volumes in flight are bounded by memory.
        chunks = np.where(window > 0, np.minimum(chunks, window + window // 4 + 320), chunks)
    chunks = np.minimum(chunks, (1 << 20) - 2)    # 20-bit chunk ids (SW_NOCHUNK)
    return shift, chunks


def plan_spill(cnt):
    """entries (a power of two) of the sweep's candidate-spill table per label (csrc/sweep.h: the fifth to eighth possible
    owner of a voxel; 12 bytes per entry at the front of the label's arena).  Few voxels ever need one -- five in the label
    of c3 whose longest call used to be abandoned for them -- so the table is small: Nf / 64, between 256 and 16384 entries."""
    cnt = np.maximum(np.asarray(cnt, dtype=np.int64), 1)
    return 2 ** np.clip(np.ceil(np.log2(cnt / 64.0)), 8, 14).astype(np.int64)


def arena_units(chunks, shift, spill):
    """256-byte units of a label's arena: [spill table: 12 B per entry][free stack: 4 B per chunk][chunks of 8-byte slots]"""
    return (spill * 12 + 255) // 256 + (chunks * 4 + 255) // 256 + ((chunks * 8) << shift) // 256


def level_windows(keys, anisotropy, nlev, lds_levels):
    """kh_label_t.lev_window per label (numpy u32): a power of two above the number of levels an event can lie ahead of the
    level being processed, or 0 when that does not fit `lds_levels` words.  An event's key is the distance of a 26-neighbour
    of the processed voxel from a source whose key of that voxel is not above the current one, so it exceeds the current key
    by one step (the longest neighbour offset) at most: the bound is the largest number of distinct keys in such an interval
    over the label's levels, taken from the sorted key table itself (+ slack for the keys' own rounding)."""
    keys = np.asarray(keys, dtype=np.float64)
    nlev = np.asarray(nlev, dtype=np.int64)
    if keys.size == 0:
        return np.zeros(nlev.shape, dtype=np.uint32)
    step = float(np.sqrt(sum(float(np.float32(a)) ** 2 for a in anisotropy)))
    ahead = np.searchsorted(keys, (keys + step) * (1.0 + 1e-6) + 1e-6, side="right") - 1 - np.arange(keys.size)
    worst = np.maximum.accumulate(ahead)                         # worst[i] = most levels ahead over levels 0..i
    w = worst[np.clip(nlev - 1, 0, keys.size - 1)] + 2
    win = np.maximum(64, 2 ** np.ceil(np.log2(np.maximum(w, 1))).astype(np.int64))
    return np.where((nlev > 0) & (win <= int(lds_levels)), win, 0).astype(np.uint32)


def int_key_mode(anisotropy, rmax):
    """(gq, gx, gy, gz) of the sweep's INTEGER levels (csrc/sweep.h) for balls up to `rmax`, or None when the table of ranks
    has to serve.  With an integral anisotropy the flood's key of an offset (a, b, c) is sqrtf of the exact integer
    T = (wx a)^2 + (wy b)^2 + (wz c)^2 as long as T < 2^24 (every product and partial sum is an integer below 2^24: no rounding
    before the square root), and T = gq * S with gq = gcd(wx^2, wy^2, wz^2), S = gx a^2 + gy b^2 + gz c^2.  sqrtf is monotone; two
    values of T that differ lie at least gq apart, i.e. their roots gq / (2 sqrt T) apart, which exceeds an ulp of sqrt T
    (<= sqrt T * 2^-23) while T < gq * 2^22 -- taken with a factor two of margin.  Then S orders the keys and tells equal ones exactly as
    the floats do, for every offset the sweep evaluates: the voxels of a ball and their neighbours (one step further out)."""
    w = [float(np.float32(a)) for a in anisotropy]
    if any(v < 1.0 or v != int(v) or v > 4096.0 for v in w) or not np.isfinite(rmax) or rmax <= 0:
        return None
    q = [int(v) * int(v) for v in w]
    gq = int(np.gcd.reduce(q))
    step = float(np.sqrt(sum(q)))
    tmax = (float(rmax) + step) ** 2 * (1.0 + 1e-6)
    if tmax >= 2.0 ** 24 or tmax >= gq * 2.0 ** 21:
        return None
    return gq, q[0] // gq, q[1] // gq, q[2] // gq


def int_levels(anisotropy, gq, rmax, lds_levels):
    """per label (numpy arrays over `rmax`): the number of integer levels a ball of that radius can touch and the level window
    (a power of two above the number of levels an event can lie ahead of the level being processed, 0 when that does not fit
    `lds_levels` words).  An event's level is S of a 26-neighbour of the processed voxel from a source whose key of that voxel is
    not above the current one: at most ((d + step)^2 - d^2) / gq levels ahead for d up to the radius."""
    rmax = np.asarray(rmax, dtype=np.float64)
    w = [float(np.float32(a)) for a in anisotropy]
    step = float(np.sqrt(sum(v * v for v in w)))
    ok = np.isfinite(rmax) & (rmax > 0)
    r = np.where(ok, rmax, 0.0)
    nlev = np.where(ok, np.floor(r * r * (1.0 + 1e-6) / gq) + 2, 0).astype(np.int64)
    ahead = np.ceil((2.0 * r * step + step * step) * (1.0 + 1e-6) / gq) + 2
    win = np.maximum(64, 2 ** np.ceil(np.log2(np.maximum(ahead, 1))).astype(np.int64))
    return nlev, np.where(ok & (win <= int(lds_levels)), win, 0).astype(np.int64)


class Engine:
    def __init__(self, device=None):
        self.lib = _abi.require_gpu()
        self.torch = _torch()
        if device is None:
            device = self.torch.cuda.current_device()
        self.device = self.torch.device("cuda", device) if isinstance(device, int) else self.torch.device(device)
        # one process drives one GPU: the library launches on the CURRENT device, so make this engine's device current
        self.torch.cuda.set_device(self.device)
        self.last_tasks = None   # task records (statistics, status) of this engine's most recent run_labels call
        self.last_path_kernel_ms = []   # (labels, milliseconds) of its path-loop launches when `timings` was asked for (HIP events)
        self.last_path_span_ms = 0.0    # first start to last end of those launches (they overlap on two streams)
        self.profile = False  # True: kh_trace_paths also fills the pop / push / fire cycle split (slower)
        self._side = None     # second stream: the biggest labels run there while the others are collected
        self.split_slots = int(os.environ.get("KH_SPLIT_SLOTS", "256"))   # labels that go to the second stream (one big-LDS workgroup per CU) when results are consumed incrementally
        self.split_min_voxels = 16384       # ... if they have at least this many voxels
        self.sweep = os.environ.get("KH_SWEEP", "1") != "0"   # False: every invalidation runs as the heap emulation (tests, comparisons)
        self.sweep_filter = True            # (the pending-deadline filter of the sweep is no longer optional: its words also say "dead")
        # integer levels (csrc/sweep.h) whenever the anisotropy allows them; False: always the table of ranks (tests, A/B runs)
        self.int_keys = os.environ.get("KH_SWEEP_INT_KEYS", "1") != "0"
        # heap and ghost journal of a launch's labels from one pool, on demand (False: a slice per label, rounds 1-5)
        self.scratch_pool = os.environ.get("KH_SCRATCH_POOL", "1") != "0"
        self.scratch_pool_fraction = SCRATCH_POOL_FRACTION      # tests: a pool too small for the labels that ask (they are traced again)
        # find_root, the DAF search and compute_pdrf inside the path kernel, by each label's own workgroup (KH_TRACE_FUSED_EDF);
        # False: as launches of their own over all labels in front of it (rounds 1-5; kept for return_fields and other exponents)
        self.fuse_edf = os.environ.get("KH_FUSE_EDF", "1") != "0"
        # volumes in flight: this many of the largest labels of a call are fused and launched FIRST, on a second stream (0: off)
        self.early_labels = int(os.environ.get("KH_EARLY_LABELS", "0"))
        # volumes in flight: called (once per volume) between the searches and the path loop, with this lane's stream drained --
        # kimimaro_amd.lanes._CohortGate holds the lane there until the searches of every volume of its cohort are through
        self.path_gate = None
        self.heap_prio = os.environ.get("KH_HEAP_PRIO", "0") == "1"         # s_setprio 3 for the heap-emulation wave (A/B knob)
        self.sweep_window = os.environ.get("KH_SWEEP_WINDOW", "1") != "0"   # level words for a window of levels only (A/B knob)
        # ghosts: a call of the sweep that leaves voxels undecided goes on with them as ghosts instead of running
        # the heap emulation at once; "paranoid" rolls every such call back at once (a test of the roll-back; results identical)
        self.ghosts = os.environ.get("KH_GHOSTS", "1") != "0"
        self.ghost_paranoid = os.environ.get("KH_GHOSTS", "1") == "paranoid"
        # the launch of the largest labels (second stream, single-volume mode) keeps two chunks of the invalidation heap in LDS
        # (128 KiB per workgroup, one workgroup per CU): every pop of a heap emulation saves one of its two L2 round trips
        self.big_lds_heap = os.environ.get("KH_BIG_LDS_HEAP", "1") != "0"
        # how many of the largest labels get such a workgroup: None = all of the second-stream launch in single-volume mode
        # (256 threads per label), none with 64 / 128 threads; a number = that many, whatever the thread count (the lanes)
        self.big_lds_labels = int(os.environ["KH_BIG_LDS_LABELS"]) if os.environ.get("KH_BIG_LDS_LABELS") else None
        self.big_threads = int(os.environ.get("KH_BIG_THREADS", "0"))     # threads per label of that launch (0: trace_threads)
        # threads per label in the path loop (64, 128 or 256).  256 serves one volume best (its searches are 4 x as wide);
        # with volumes in flight 64 does: a label whose call runs on the heap emulation -- one wave for seconds -- then holds a
        # twelfth of a CU instead of a third, and the sweep's levels hold tens of events, not hundreds (kimimaro_amd.lanes
        # sets it for its engines; measured at c3, 8 lanes: 784 vs 1134 ms per step)
        self.trace_threads = int(os.environ.get("KH_TRACE_THREADS", "256"))
        # threads per label of the two distance-field searches before the path loop (kh_edf_batch; 512 serves one volume best)
        self.edf_threads = int(os.environ.get("KH_EDF_THREADS", "512"))
        # soma labels of one volume traced side by side (kimimaro_amd.intake._trace_soma_labels): host threads with an Engine
        # and a stream each; 1 = one after the other.  The lanes set 1 for their engines (the other volumes fill the GPU).
        self.soma_lanes = int(os.environ.get("KH_SOMA_LANES", "4"))
        # HIP events around every launch of the path loop, on the launch's own stream, also without the phase marks
        # (bench.py: the production launches of one volume -> last_path_kernel_ms)
        self.time_kernels = False
        self._soma_pool = None
        self.sweep_table_limit = 1 << 24    # largest level table (entries)
        # Level words of a label stay in LDS up to this many levels (4 B each), beyond in HBM.  This sizes the LDS of the
        # path kernel's workgroups: 8192 -> 39 KiB, which leaves the registers (3 workgroups per CU) as the occupancy limit
        # instead of LDS (2 per CU at 16384) -- +20 % labels/s with several volumes in flight, nothing for a single one.
        self.sweep_lds_levels = min(int(os.environ.get("KH_SWEEP_LDS_LEVELS", 8192)), _abi.SWEEP_LDS_LEVELS)
        self._level_tables = {}
        self.scratch_divisor = 1            # tests: shrink the heap / path scratch to exercise the overflow retry
        self.arena_divisor = 1              # tests: shrink the sweep's event arena (a call that runs out falls back to the heap)
        # Cap of the level window (words of LDS per label's workgroup; an event beyond it abandons the call to the heap emulation).
        # The launch gives EVERY workgroup the LDS of its neediest label: at c3 one label of 3 402 wants 4 096 words (20 KB), which
        # holds a CU to 7 one-wave workgroups where the registers allow 12; with 2 048 (11.5 KB) all 12 fit.  The lanes set 2 048
        # for their 64-thread engines (kimimaro_amd.lanes); 0 = no cap (one volume alone: three 256-thread workgroups per CU).
        # The cap applies only when at most 0.5 % of the launch's labels want more (window_cap_always: whatever the share; tests).
        self.window_cap = int(os.environ.get("KH_WINDOW_CAP", "0"))
        self.window_cap_always = "KH_WINDOW_CAP" in os.environ
        # Per-label scratch (heap, work lists, event arena, path buffers: SCRATCH_BYTES_PER_VOXEL per voxel of a label) of ONE path-loop
        # launch.  Labels beyond it go to further launches of the same call, largest labels first (callers that consume
        # results incrementally only); the whole-volume fields (~40 B per voxel of the volume) are not counted.
        self.scratch_budget = int(float(os.environ.get("KH_SCRATCH_BUDGET_GB", "150")) * 1e9)

    # -- plumbing -----------------------------------------------------------
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, n, dtype):
        return self.torch.empty(int(n), dtype=dtype, device=self.device)

    def to_device(self, arr):
        """numpy (any shape, F order expected) -> 1-D device tensor in memory order."""
        flat = np.ascontiguousarray(arr.reshape(-1, order="F"))
        if flat.dtype == np.uint16:
            flat = flat.view(np.int16)
        elif flat.dtype == np.uint32:
            flat = flat.view(np.int32)
        elif flat.dtype == np.uint64:
            flat = flat.view(np.int64)
        return self.torch.from_numpy(flat).to(self.device)

    @staticmethod
    def ptr(t):
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def optr(t):
        """pointer of an optional tensor (None -> NULL)"""
        return C.c_void_p(t.data_ptr() if t is not None else 0)

    SCHED_PAD = 4      # words in front of (and behind) the volume's filter words: the sweep reads rows of three around a voxel

    def sched_volume(self, nvox):
        """the filter words of the invalidation sweep (csrc/sweep.h): one u32 per voxel, all ones ("alive, no deadline pending"),
        with SCHED_PAD readable words on either side; hand `sched_ptr(t)` to the library."""
        return self.torch.full((int(nvox) + 2 * self.SCHED_PAD,), -1, dtype=self.torch.int32, device=self.device)

    def sched_ptr(self, t):
        return C.c_void_p(t.data_ptr() + 4 * self.SCHED_PAD if t is not None else 0)

    def sweep_levels(self, shape, anisotropy, rmax_t, cnt):
        """The sweep's level arguments for labels with largest ball radii `rmax_t` (f32 array) and voxel counts `cnt`:
        dict(rank_ptr, rdims, nlev, win, ok) -- integer mode (rank_ptr NULL, rdims = (gx, gy, gz)) when the anisotropy allows it
        (int_key_mode), else the table of ranks (level_table); None when no label can use the sweep."""
        rmax_t = np.asarray(rmax_t, dtype=np.float32)
        finite = rmax_t[np.isfinite(rmax_t)]
        if not finite.size or float(finite.max()) <= 0:
            return None
        top = float(finite.max())
        mode = int_key_mode(anisotropy, top) if self.int_keys else None
        if mode is not None:
            gq, gx, gy, gz = mode
            nlev, win = int_levels(anisotropy, gq, rmax_t.astype(np.float64), self.sweep_lds_levels)
            ok = np.isfinite(rmax_t) & (rmax_t > 0) & (nlev <= _abi.SWEEP_MAX_LEVELS)
            if not self.sweep_window:
                win = np.zeros_like(win)
            return {"d_rank": None, "rdims": (gx, gy, gz), "nlev": np.where(ok, nlev, 0), "win": np.where(ok, win, 0), "ok": ok}
        d_rank, rdims, keys, covered = self.level_table(shape, anisotropy, top)
        nlev = np.searchsorted(keys, rmax_t, side="left").astype(np.int64)   # keys below the radius: levels 0..nlev-1
        ok = np.isfinite(rmax_t) & (rmax_t <= covered) & (nlev <= _abi.SWEEP_MAX_LEVELS) & (nlev > 0)
        nlev = np.where(ok, nlev, 0)
        win = level_windows(keys, anisotropy, nlev, self.sweep_lds_levels).astype(np.int64) if self.sweep_window \
            else np.zeros(nlev.shape, dtype=np.int64)
        return {"d_rank": d_rank, "rdims": rdims, "nlev": nlev, "win": win, "ok": ok}

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def soma_lane_pool(self, width):
        """the lanes of the soma labels (made once per Engine): engines of the single-volume kind on streams of their own"""
        from .lanes import Lanes, _StreamScope
        if self._soma_pool is None or self._soma_pool.width < width:
            if self._soma_pool is not None:
                self._soma_pool.close()

            def make():
                e = Engine(self.device)
                e.soma_lanes = 1
                # the lane engines trace with the parent's settings (a test that switches the sweep or the ghosts off means the somas too)
                for knob in ("sweep", "ghosts", "ghost_paranoid", "int_keys", "scratch_divisor", "arena_divisor", "window_cap", "window_cap_always", "profile",
                             "trace_threads", "edf_threads", "sweep_window", "sweep_lds_levels", "big_lds_heap", "scratch_pool",
                             "scratch_pool_fraction", "heap_prio"):
                    setattr(e, knob, getattr(self, knob))
                return e
            self._soma_pool = Lanes(width, device=self.device, engine_factory=make,
                                    stream_factory=lambda e: _StreamScope(self.torch, e))
        return self._soma_pool

    def sync_stream(self):
        """wait for the calling thread's current stream only (the phase marks: another lane's kernels are not this volume's)"""
        self.torch.cuda.current_stream(self.device).synchronize()

    # -- f1: connected components on the device ---------------------------------
    def ccl(self, labels, d_graph=None):
        """26-connected multi-label CCL (kimimaro/utility.py:58-83).  labels: host ndarray (F order, integer).
        Returns (d_cc u32 device tensor, N, representative[N+1] host array of smallest linear indices)."""
        t = self.torch
        lab = labels
        if lab.dtype == bool:
            lab = lab.view(np.uint8)
        if lab.dtype.kind not in "ui":
            lab = lab.astype(np.uint64)
        lab = np.asfortranarray(lab)
        return self.ccl_device(self.to_device(lab), lab.dtype.itemsize, lab.shape, d_graph)

    def ccl_device(self, d_lab, itemsize, shape, d_graph=None):
        """kh_ccl26 on a label volume that is already resident in HBM; d_graph (u32 per voxel, cc3d's layout): kh_ccl26_graph,
        the components of the voxel connectivity graph (kimimaro/utility.py:73-75)."""
        t = self.torch
        n = int(shape[0]) * int(shape[1]) * int(shape[2])
        d_parent = self.empty(n, t.int32)
        d_counts = self.empty((n + 1023) // 1024, t.int32)
        d_cc = self.empty(n, t.int32)
        d_rep = self.empty(n + 1, t.int32)
        d_total = self.empty(1, t.int32)
        d_cc16 = self.empty(n, t.int16)
        if d_graph is not None:
            _abi.check(self.lib.kh_ccl26_graph(self.ptr(d_lab), itemsize, self.ptr(d_graph), shape[0], shape[1], shape[2],
                                               self.ptr(d_parent), self.ptr(d_counts), self.ptr(d_cc), self.ptr(d_rep), self.ptr(d_total),
                                               self.ptr(d_cc16), self.stream()))
        else:
            _abi.check(self.lib.kh_ccl26(self.ptr(d_lab), itemsize, shape[0], shape[1], shape[2], self.ptr(d_parent),
                                         self.ptr(d_counts), self.ptr(d_cc), self.ptr(d_rep), self.ptr(d_total), self.ptr(d_cc16),
                                         self.stream()))
        ncomp = int(d_total.cpu().numpy().view(np.uint32)[0])
        # fewer than 65536 components: the u16 copy of the ids serves every later sweep (narrow())
        # (keyed by the tensor OBJECT, which the record keeps alive: the caching allocator hands a freed volume's address
        # to the next volume of the same size, so an address is no identity)
        self._narrow = (d_cc, d_cc16) if ncomp < 65536 else None
        rep = d_rep[: ncomp + 1].cpu().numpy().view(np.uint32)
        return d_cc, ncomp, rep

    def narrow(self, d_cc):
        """(device label volume, bytes per label) to sweep over: the u16 copy kh_ccl26 made of `d_cc` when there is one."""
        nr = getattr(self, "_narrow", None)
        if nr is not None and nr[0] is d_cc and d_cc is not None:
            return nr[1], 2
        return d_cc, 4

    def fill_voids(self, d_mask, shape, ndim=3):
        """kh_fill_voids (fill_voids.fill, kimimaro/trace.py:109) on a u8 mask resident in HBM; ndim < 3: a 2-D / 1-D image given as
        shape (nx, ny, 1) / (nx, 1, 1), whose border is its outline.  Returns (filled u8 mask on the device, number of voxels that changed)."""
        t = self.torch
        shape = tuple(int(v) for v in shape) + (1,) * (3 - len(shape))
        n = int(shape[0]) * int(shape[1]) * int(shape[2])
        d_parent = self.empty(n, t.int32)
        d_open = self.empty(n, t.uint8)
        d_out = self.empty(n, t.uint8)
        d_cnt = self.empty(1, t.int64)
        _abi.check(self.lib.kh_fill_voids_nd(self.ptr(d_mask), int(ndim), shape[0], shape[1], shape[2], self.ptr(d_parent),
                                             self.ptr(d_open), self.ptr(d_out), self.ptr(d_cnt), self.stream()))
        return d_out, int(d_cnt.cpu().numpy()[0])

    def to_host_volume(self, d, shape, dtype=np.uint32):
        return d.cpu().numpy().view(dtype).reshape(shape, order="F")

    def face(self, d, shape, axis, index):
        """one face of a device volume as a host array (Fortran order semantics: d is [sx, sy, sz], x fastest)."""
        v = d.view(shape[2], shape[1], shape[0])  # torch C order (z, y, x) == F order (x, y, z)
        if axis == 2:
            f = v[index, :, :]      # (y, x)
        elif axis == 1:
            f = v[:, index, :]      # (z, x)
        else:
            f = v[:, :, index]      # (z, y)
        return np.asfortranarray(f.contiguous().cpu().numpy().view(np.uint32).T)  # -> (x,y) / (x,z) / (y,z)

    # -- a1 -------------------------------------------------------------------
    def edt(self, d_labels, label_bytes, shape, anisotropy, black_border, out=None, workspace=None, ndim=3):
        """ndim: dimensionality of the caller's array (trailing axes of extent 1 beyond it are not axes)."""
        sx, sy, sz = shape
        n = sx * sy * sz
        t = self.torch
        if out is None:
            out = self.empty(n, t.float32)
        if workspace is None:
            workspace = self.empty(n, t.float32)  # ping-pong buffer of the passes (include/kimi_hip.h)
        _abi.check(self.lib.kh_edt_nd(self.ptr(d_labels), label_bytes, int(ndim), sx, sy, sz, float(anisotropy[0]),
                                      float(anisotropy[1]), float(anisotropy[2]), int(bool(black_border)),
                                      self.ptr(workspace), self.ptr(out), self.stream()))
        return out

    def edt_graph(self, d_labels, label_bytes, d_graph, shape, anisotropy, black_border):
        """edt.edt(labels, voxel_graph=) (kimimaro/intake.py:174-183; PARITY UNPINNED, include/kimi_hip.h): the doubled image of the
        graph's walls, a binary transform with half the pitch, sampled at the voxels."""
        sx, sy, sz = (int(v) for v in shape)
        n = sx * sy * sz
        t = self.torch
        cells = self.empty(8 * n, t.uint8)
        _abi.check(self.lib.kh_edt_graph_cells(self.ptr(d_labels), label_bytes, self.ptr(d_graph), sx, sy, sz, int(bool(black_border)),
                                               self.ptr(cells), self.stream()))
        half = [np.float32(a) / np.float32(2) for a in anisotropy]
        fine = self.edt(cells, 1, (2 * sx, 2 * sy, 2 * sz), half, black_border)
        out = self.empty(n, t.float32)
        _abi.check(self.lib.kh_edt_graph_sample(self.ptr(fine), sx, sy, sz, self.ptr(out), self.stream()))
        return out

    def label_stats(self, d_labels, label_bytes, d_dbf, shape, nlabels):
        t = self.torch
        n1 = nlabels + 1
        counts = self.empty(n1, t.int32)
        dmax = self.empty(n1, t.float32)
        first = self.empty(n1, t.int32)
        xmin = self.empty(n1, t.int32)
        xmax = self.empty(n1, t.int32)
        yz = self.empty(4 * n1, t.int32)
        nvox = shape[0] * shape[1] * shape[2]
        _abi.check(self.lib.kh_label_stats(self.ptr(d_labels), label_bytes, self.ptr(d_dbf), nvox, shape[0], shape[1],
                                           nlabels, self.ptr(counts), self.ptr(dmax), self.ptr(first), self.ptr(xmin),
                                           self.ptr(xmax), self.ptr(yz), self.stream()))
        u32 = lambda x: x.cpu().numpy().view(np.uint32)
        self.last_yz_extent = u32(yz).reshape(n1, 4)  # [ymin, ymax, zmin, zmax] per label (bounding boxes)
        return u32(counts), dmax.cpu().numpy(), u32(first), u32(xmin), u32(xmax)

    def crop(self, d, shape, lo, hi, dtype=np.uint32):
        """host copy of the box [lo, hi) of a device volume, as an (x, y, z) Fortran-ordered array."""
        v = d.view(shape[2], shape[1], shape[0])[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
        return np.asfortranarray(v.contiguous().cpu().numpy().view(dtype).transpose(2, 1, 0))

    # -- level table of the order-free invalidation sweep (csrc/sweep.h) -------------
    def level_table(self, shape, anisotropy, rmax):
        """(d_rank int32 [ra*rb*rc], (ra, rb, rc), keys host f32 sorted, radius covered) for balls up to `rmax`.
        rank[a + ra*(b + rb*c)] = index of the flood's key of offset (a, b, c) among the distinct keys; the keys
        are computed by the device with the flood's own float operations (kh_level_keys), sorting and ranking is
        plumbing (torch.unique).  The table is clipped to `sweep_table_limit` entries."""
        t = self.torch
        w = [float(np.float32(a)) for a in anisotropy]
        dims = [int(min(int(rmax / w[i]) + 2, int(shape[i]))) for i in range(3)]
        while dims[0] * dims[1] * dims[2] > self.sweep_table_limit:
            rmax *= 0.8
            dims = [int(min(int(rmax / w[i]) + 2, int(shape[i]))) for i in range(3)]
        # every offset with a key below `covered` is inside the table (one row of slack for the rounding of the key)
        covered = min([w[i] * (dims[i] - 1) for i in range(3) if dims[i] < shape[i]] + [float("inf")])
        key = (tuple(w), tuple(dims))
        hit = self._level_tables.get(key)
        if hit is None:
            n = dims[0] * dims[1] * dims[2]
            d_keys = self.empty(n, t.float32)
            _abi.check(self.lib.kh_level_keys(dims[0], dims[1], dims[2], w[0], w[1], w[2], self.ptr(d_keys), self.stream()))
            uniq, inverse = t.unique(d_keys, sorted=True, return_inverse=True)
            hit = (inverse.to(t.int32).contiguous(), tuple(dims), uniq.cpu().numpy())
            if len(self._level_tables) > 8:
                self._level_tables.clear()
            self._level_tables[key] = hit
        return hit[0], hit[1], hit[2], covered

    # -- one binary object on its own (the function-level mirrors of ops.py) ---------------
    def single_object(self, mask, anisotropy, rmax=0.0, dbf=None, voxel_graph=None):
        """Device context of ONE binary object given as a host mask (x, y, z; Fortran order): component volume (0 / 1),
        voxel list, neighbour masks, per-label scratch and -- for ball radii up to `rmax` -- the level table and the
        event arena of the invalidation sweep.  Returns a dict of device tensors + the kh_label_t record."""
        t = self.torch
        lib = self.lib
        P = self.ptr
        m = np.asarray(mask)
        while m.ndim < 3:
            m = m[..., np.newaxis]
        shape = tuple(int(v) for v in m.shape)
        nvox = shape[0] * shape[1] * shape[2]
        cc = np.asfortranarray((m != 0).astype(np.uint32))
        d_cc = self.to_device(cc)
        d_dbf = self.to_device(np.asfortranarray(dbf, dtype=np.float32).reshape(shape, order="F")) if dbf is not None \
            else t.zeros(nvox, dtype=t.float32, device=self.device)
        counts, dbf_max, first, xmin, xmax = self.label_stats(d_cc, 4, d_dbf, shape, 1)
        cnt = int(counts[1])
        task = np.zeros(1, dtype=_abi.LABEL_T)
        task["segid"] = 1
        task["count"] = cnt
        # the reference runs on the array it is given: the x faces of THAT array matter to the heap order
        # (dijkstra_invalidation.hpp:116-123), not the object's own extent
        task["xmin"], task["xmax"] = 0, shape[0] - 1
        task["source"] = first[1]
        task["root"] = NONE32
        task["q_capacity"] = cnt + 64
        hcap = 27 * cnt + 4096          # every push of the flood fits: a voxel is pushed by at most 26 neighbours (and the sweep's lists)
        task["heap_capacity"] = hcap
        task["path_capacity"] = 4 * cnt + 1024
        ctx = {"shape": shape, "nvox": nvox, "count": cnt, "d_cc": d_cc, "d_dbf": d_dbf, "dbf_max": float(dbf_max[1])}
        d_slot = t.from_numpy(np.array([-1, 0], dtype=np.int32)).to(self.device)
        d_off = t.zeros(1, dtype=t.int32, device=self.device)
        d_cur = self.empty(1, t.int32)
        d_lists = self.empty(max(cnt, 1), t.int32)
        d_nbr = self.empty(nvox, t.int32)
        st = self.stream()
        _abi.check(lib.kh_scatter_lists(P(d_cc), 4, nvox, P(d_slot), 1, P(d_off), P(d_cur), P(d_lists), st))
        _abi.check(lib.kh_neighbor_mask(P(d_cc), 4, shape[0], shape[1], shape[2], P(d_nbr), st))
        d_gate = None
        if voxel_graph is not None:
            # voxel_graph= / voxel_connectivity_graph= of the reference's calls: directions the caller's words do not allow
            # leave the neighbour masks every search and the invalidation work from (kh_apply_voxel_graph)
            vg = np.asarray(voxel_graph)
            while vg.ndim < 3:
                vg = vg[..., np.newaxis]
            if tuple(int(v) for v in vg.shape) != shape:
                raise ValueError("voxel_graph must have the shape of the labels")
            d_graph = self.to_device(np.asfortranarray(vg.astype(np.uint32)))
            d_gate = t.zeros(nvox + 4, dtype=t.uint8, device=self.device)
            _abi.check(lib.kh_apply_voxel_graph(P(d_nbr), P(d_graph), nvox, P(d_gate), st))
        ctx["d_gate"] = d_gate
        ctx.update(d_slot=d_slot, d_lists=d_lists, d_nbr=d_nbr, d_queues=self.empty(4 * (cnt + 64), t.int32),
                   d_heap=self.empty(2 * hcap, t.int64), d_qstate=t.zeros(nvox + 4, dtype=t.uint8, device=self.device))
        d_rank, rdims, max_nlev, ev_units = None, (0, 0, 0), 0, 0
        rmax = float(np.float32(rmax))
        sweep_on = False
        if self.sweep and cnt > 0 and np.isfinite(rmax) and rmax > 0:
            lv = self.sweep_levels(shape, anisotropy, [rmax], [cnt])
            if lv is not None and int(lv["nlev"][0]) > 0:
                nlev, win = int(lv["nlev"][0]), int(lv["win"][0])
                shift, chunks = (int(v) for v in plan_arena(cnt, nlev, True, win))
                in_lds = win > 0 or nlev <= self.sweep_lds_levels       # else: heap emulation only (the kernel decides the same)
                spill = int(plan_spill(cnt))
                ev_units = int(arena_units(chunks, shift, spill))   # spill table, (unused stack), chunks
                task["nlev"], task["sweep_rmax"], task["ev_chunks"], task["ev_shift"] = nlev, np.float32(rmax), chunks, shift
                task["ev_spill"] = spill
                task["lev_window"] = win
                max_nlev = win if win > 0 else (nlev if in_lds else 0)
                d_rank, rdims = lv["d_rank"], lv["rdims"]
                sweep_on = True
        d_arena = self.empty(max(ev_units, 1) * 32 + 32, t.int64)
        ctx.update(d_rank=d_rank, rdims=rdims, max_nlev=max_nlev, d_arena=d_arena,
                   arena_ptr=C.c_void_p((d_arena.data_ptr() + 255) & ~255),
                   sweep_on=sweep_on,
                   d_cstate=t.zeros(nvox if sweep_on else 1, dtype=t.int64, device=self.device),
                   d_sched=self.sched_volume(nvox) if sweep_on else None,
                   d_task=t.from_numpy(task.view(np.uint8).reshape(-1).copy()).to(self.device), task=task)
        return ctx

    def invalidate_ball(self, ctx, d_alive, path_locs, scale, const, anisotropy):
        """kh_invalidate_ball on the object of `ctx` (Engine.single_object): d_alive (u8, device) is mutated.
        Returns (voxels invalidated, task record after the call)."""
        t = self.torch
        P = self.ptr
        shape = ctx["shape"]
        d_path = t.from_numpy(np.asarray(path_locs, dtype=np.uint32).view(np.int32).copy()).to(self.device)
        d_cnt = t.zeros(1, dtype=t.int64, device=self.device)
        rank_ptr = P(ctx["d_rank"]) if ctx["d_rank"] is not None else C.c_void_p(0)
        rd = ctx["rdims"] if ctx["sweep_on"] else (0, 0, 0)       # (the kernel derives the filter words from the caller's mask)
        _abi.check(self.lib.kh_invalidate_ball(P(ctx["d_task"]), P(ctx["d_lists"]), P(ctx["d_nbr"]), shape[0], shape[1], shape[2],
                                               float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]), P(ctx["d_dbf"]),
                                               P(d_alive), P(ctx["d_queues"]), P(ctx["d_heap"]), P(d_path), int(d_path.numel()),
                                               np.float32(scale), np.float32(const), rank_ptr, rd[0], rd[1], rd[2], ctx["max_nlev"],
                                               P(ctx["d_cstate"]), self.sched_ptr(ctx["d_sched"]), ctx["arena_ptr"],
                                               self.optr(ctx.get("d_gate")), P(d_cnt), self.stream()))
        task = ctx["d_task"].cpu().numpy().view(_abi.LABEL_T).copy()
        if int(task["status"][0]):
            raise _abi.KimiHipError("kh_invalidate_ball: %s" % _abi.describe_status(int(task["status"][0])))
        return int(d_cnt.item()), task

    # -- the per-label pipeline -------------------------------------------------
    def run_labels(self, d_cc, label_bytes, d_dbf, shape, anisotropy, nlabels, segids, counts, dbf_max, first_index,
                   xmin, xmax, roots, targets_before, targets_after, params, fix_branching=True, max_paths=None,
                   return_fields=False, timings=None, soma=None, consume=None, scratch_scale=1, voxel_graph=None):
        """Run find_root -> DAF -> PDRF -> path loop for the connected components `segids`.

        segids/counts/...: host arrays indexed by position (same order).  roots: array of linear indices or
        NONE32.  targets_before/after: list (per label) of lists of linear indices (LIFO stacks as in
        kimimaro/trace.py:225-228).  Returns a dict with per-label path arrays -- or, when `consume` is given,
        hands such dicts (one per group of labels, as the groups finish) to `consume` and returns None.
        """
        t = self.torch
        lib = self.lib
        st = self.stream()
        sx, sy, sz = shape
        nvox = sx * sy * sz
        nl = len(segids)
        if nl == 0:
            return {"order": np.zeros(0, np.int64), "tasks": np.zeros(0, _abi.LABEL_T), "paths": []}
        segids = np.asarray(segids, dtype=np.int64)
        counts = np.asarray(counts, dtype=np.int64)
        if consume is not None and nl > 1:
            # more per-label scratch than the budget: several launches, each over a group of labels that fits
            groups = plan_launches(counts, self.scratch_budget)
            if len(groups) > 1:
                done, retried = [], 0
                pick_list = lambda a, g: [a[i] for i in g] if a is not None else None
                for g in groups:
                    g = np.asarray(g, dtype=np.int64)
                    sub_soma = None if soma is None else {k: np.asarray(v)[g] for k, v in soma.items()}
                    self.run_labels(d_cc, label_bytes, d_dbf, shape, anisotropy, nlabels, segids[g], counts[g],
                                    np.asarray(dbf_max)[g], np.asarray(first_index)[g], np.asarray(xmin)[g],
                                    np.asarray(xmax)[g], np.asarray(roots, dtype=np.uint32)[g], pick_list(targets_before, g),
                                    pick_list(targets_after, g), params, fix_branching=fix_branching, max_paths=max_paths,
                                    timings=timings, soma=sub_soma, consume=consume,
                                    scratch_scale=scratch_scale, voxel_graph=voxel_graph)
                    done.append(self.last_tasks)
                    retried += self.last_retries
                self.last_tasks = np.concatenate(done)
                self.last_retries = retried           # (each nested call resets it: accumulate over the groups)
                return None
        order = np.argsort(-counts, kind="stable")  # big labels first: their workgroups start first
        slot_of_label = -np.ones(nlabels + 1, dtype=np.int32)
        slot_of_label[segids[order]] = np.arange(nl, dtype=np.int32)

        cnt = counts[order]
        list_off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
        total = int(cnt.sum())
        qcap = cnt + 64
        q_off = np.concatenate([[0], np.cumsum(qcap)[:-1]]).astype(np.int64)
        # heap / path scratch are sized for the common case; a label that overflows them is traced again on its own
        # with `scratch_scale` times as much (below) -- the reference has no such limits
        # heap: 3 nodes per voxel for small labels, 1.5 per voxel + 4096 for the others (the deepest heap of c3's largest
        # label holds 0.7 nodes per voxel); never less than the sweep's lists need (11 / 8 nodes per voxel + 1536)
        hbase = np.maximum((3 * cnt) // 2 + 4096, np.minimum(3 * cnt + 2048, 32768))
        hcap = np.maximum(hbase * scratch_scale // self.scratch_divisor, 64)
        h_off = np.concatenate([[0], np.cumsum(hcap)[:-1]]).astype(np.int64)
        # path buffers: c3's labels write 527 vertices at most, 3 % of their voxels at most (round 5 kept 64 Ki entries for every label
        # above 16 Ki voxels: 2 GB per volume); a label that needs more is traced again with `scratch_scale` x 8
        pcap = np.maximum((cnt // 16 + 2048) * scratch_scale // self.scratch_divisor, 8)
        # first attempt: heap and journal from a pool (engine.SCRATCH_POOL_FRACTION); retries and test runs with shrunk scratch: slices
        use_pool = self.scratch_pool and scratch_scale == 1 and self.scratch_divisor == 1 and nl >= 4
        p_off = np.concatenate([[0], np.cumsum(pcap)[:-1]]).astype(np.int64)
        if max(total, int(qcap.sum()), int(hcap.sum()), int(pcap.sum())) >= 2 ** 32:
            raise ValueError("kimimaro_amd: scratch offsets exceed 32 bits; shard the labels")

        tasks = np.zeros(nl, dtype=_abi.LABEL_T)
        tasks["segid"] = segids[order]
        tasks["list_offset"] = list_off
        tasks["count"] = cnt
        tasks["xmin"] = np.asarray(xmin)[order]
        tasks["xmax"] = np.asarray(xmax)[order]
        tasks["source"] = np.asarray(first_index)[order]
        tasks["root"] = np.asarray(roots, dtype=np.uint32)[order]
        f = np.float32
        dm = np.asarray(dbf_max, dtype=np.float32)[order]
        # M = f32(1 / dbf_max ** 1.01) with numpy scalar semantics, kimimaro/trace.py:335-336
        tasks["M"] = np.array([f(1 / (f(v) ** 1.01)) if v > 0 else f(0) for v in dm], dtype=np.float32)
        tasks["q_offset"] = q_off
        tasks["q_capacity"] = qcap
        tasks["heap_offset"] = h_off
        tasks["heap_capacity"] = hcap
        tasks["path_offset"] = p_off
        tasks["path_capacity"] = pcap
        tasks["max_paths"] = 0 if max_paths is None else int(max_paths)
        if _abi.is_pow2_exponent(params["pdrf_exponent"]):     # KH_TRACE_FUSED_EDF: compute_pdrf's parameters travel with the task
            tasks["pdrf_log2e"] = int(params["pdrf_exponent"]).bit_length() - 1
            tasks["pdrf_scale"] = np.float32(params["pdrf_scale"])
        if soma is not None:  # per label (caller order): soma_mode, fsr, soma_radius, soma_scale, soma_const
            for key in ("soma_mode", "fsr", "soma_radius", "soma_scale", "soma_const"):
                tasks[key] = np.asarray(soma[key])[order]
        # order-free invalidation sweep: level table + per-label event arenas
        d_rank, rdims, max_nlev = None, (0, 0, 0), 0
        ev_total = 0
        sweep_on = False
        if self.sweep and nl > 0:
            rmax_t = (np.float32(params["scale"]) * dm + np.float32(params["const"])).astype(np.float32)   # f32 ops as pyx:393-395
            lv = self.sweep_levels(shape, anisotropy, rmax_t, cnt)
            if lv is not None and int(lv["nlev"].max()) > 0:
                nlev, win, ok = lv["nlev"], lv["win"], lv["ok"]
                if self.window_cap and (self.window_cap_always or
                                        np.count_nonzero(win > int(self.window_cap)) <= max(1, int(0.005 * win.size))):
                    # (only when few labels pay for it: a capped label's widest calls are redone by the heap emulation)
                    win = np.where(win > 0, np.minimum(win, int(self.window_cap)), win)
                # fixed-size event chunks, chained per level (csrc/sweep.h): what is pending at one time
                shift, chunks = plan_arena(cnt, nlev, True, win)
                chunks = np.maximum(chunks // int(self.arena_divisor), 8)
                in_lds = (win > 0) | (nlev <= self.sweep_lds_levels)   # the others: heap emulation only (the kernel decides the same)
                # [spill table, 12 B per entry][(the free stack of rounds 4-5, 4 B per chunk: unused since round 6)][chunks]
                spill = plan_spill(cnt)
                units = np.where((nlev > 0) & in_lds, arena_units(chunks, shift, spill), 0)
                ev_off = np.concatenate([[0], np.cumsum(units)[:-1]]).astype(np.int64)
                ev_total = int(units.sum())
                self.last_arena_bytes = ev_total * 256
                if ev_total >= 2 ** 32:
                    raise ValueError("kimimaro_amd: event arena offsets exceed 32 bits; shard the labels")
                tasks["nlev"] = nlev
                tasks["sweep_rmax"] = np.where(ok, rmax_t, 0).astype(np.float32)
                tasks["ev_offset"] = ev_off
                tasks["ev_chunks"] = np.where(nlev > 0, chunks, 0)
                tasks["ev_shift"] = shift
                tasks["ev_spill"] = np.where((nlev > 0) & in_lds, spill, 0)
                tasks["lev_window"] = win
                in_words = np.where(win > 0, win, np.where(in_lds, nlev, 0))     # LDS words each label wants
                max_nlev = int(in_words.max()) if in_words.size else 0
                d_rank, rdims = lv["d_rank"], lv["rdims"]
                sweep_on = True
        tgt = []
        tgt_off = np.zeros(nl, dtype=np.int64)
        for s, o in enumerate(order):
            tgt_off[s] = len(tgt)
            b = list(targets_before[o]) if targets_before is not None else []
            a = list(targets_after[o]) if targets_after is not None else []
            tasks["n_before"][s] = len(b)
            tasks["n_after"][s] = len(a)
            tgt.extend(b)
            tgt.extend(a)
        tasks["tgt_offset"] = tgt_off
        tgt_arr = np.asarray(tgt + [0], dtype=np.uint32)

        d_tasks = t.from_numpy(tasks.view(np.uint8).reshape(-1)).to(self.device)
        d_slot = t.from_numpy(slot_of_label).to(self.device)
        d_off = t.from_numpy(list_off.astype(np.uint32).view(np.int32)).to(self.device)
        d_cur = self.empty(nl, t.int32)
        d_lists = self.empty(max(total, 1), t.int32)
        d_tgt = t.from_numpy(tgt_arr.view(np.int32)).to(self.device)
        d_nbr = self.empty(nvox, t.int32)
        d_queues = self.empty(4 * int(qcap.sum()), t.int32)
        d_qstate = self.torch.zeros(nvox + 4, dtype=t.uint8, device=self.device)
        P = self.ptr
        wx, wy, wz = (float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]))

        def mark(name):
            if timings is not None:
                self.sync_stream()
                import time
                timings.append((name, time.perf_counter()))

        mark("setup")
        _abi.check(lib.kh_scatter_lists(P(d_cc), label_bytes, nvox, P(d_slot), nl, P(d_off), P(d_cur), P(d_lists), st))
        _abi.check(lib.kh_neighbor_mask(P(d_cc), label_bytes, sx, sy, sz, P(d_nbr), st))
        d_gate = None
        if voxel_graph is not None:
            # voxel_graph= of kimimaro.trace.trace (a u32 device volume, cc3d's bit layout): the directions a voxel's word does not
            # allow leave the masks every search and the invalidation work from; d_gate = the corner entries at the x faces
            d_gate = t.zeros(nvox + 4, dtype=t.uint8, device=self.device)
            _abi.check(lib.kh_apply_voxel_graph(P(d_nbr), P(voxel_graph), nvox, P(d_gate), st))
        mark("lists+nbrmask")
        expo = params["pdrf_exponent"]
        # The searches (find_root, DAF) and the PDRF either run inside the path kernel, by each label's own workgroup
        # (KH_TRACE_FUSED_EDF), or as launches over the labels in front of it (`searches` below).  With volumes in flight
        # (`early_labels`, set by kimimaro_amd.lanes) the LARGEST labels are fused and go first, on a second stream: their chains --
        # the heap emulation of a volume's large labels, seconds long -- then start at once instead of behind the searches of all
        # labels; the rest keeps the batch kernels (70 VGPRs: twice the waves per CU of the path kernel) on the lane's own stream.
        can_fuse = _abi.is_pow2_exponent(expo) and not return_fields
        early = int(min(self.early_labels, nl - 1)) if (can_fuse and consume is not None and self.early_labels > 0 and nl > 1) else 0
        fuse_rest = can_fuse and self.fuse_edf and early == 0
        d_ldaf = self.empty(max(total, 1), t.float32)
        d_pdrf = self.empty(nvox + 4, t.float32)
        fields = {"d_field": None}

        def searches(first, count):
            """find_root + DAF + PDRF of the task slots [first, first + count) as batch launches on the current stream"""
            if count <= 0:
                return
            if fields["d_field"] is None:
                fields["d_field"] = self.empty(nvox + 4, t.float32)
            d_field = fields["d_field"]
            _searches_launch(first, count, d_field)
            if not return_fields:
                fields["d_field"] = None     # the DAF lives on in list order (d_ldaf); its volume goes back to the pool

        def _searches_launch(first, count, d_field):
            tasks_ptr = C.c_void_p(d_tasks.data_ptr() + first * _abi.LABEL_T.itemsize)
            lo = int(list_off[first])
            n_list = int(cnt[first:first + count].sum())
            lists_ptr = C.c_void_p(d_lists.data_ptr() + 4 * lo)
            ldaf_ptr = C.c_void_p(d_ldaf.data_ptr() + 4 * lo)
            d_slot_use, keep = d_slot, 0
            if first > 0 or first + count < nl:
                # only these labels: the others' voxels are left alone (they are another launch's business)
                sl = slot_of_label.copy()
                sl[segids[order][:first]] = -1
                sl[segids[order][first + count:]] = -1
                d_slot_use, keep = t.from_numpy(sl).to(self.device), _abi.PDRF_KEEP_OTHERS
            # find_root (trace.py:291-308) then DAF (trace.py:139-145)
            _abi.check(lib.kh_edf_batch(tasks_ptr, count, 1 | (self.edf_threads << 8), P(d_lists), P(d_nbr), sx, sy, sz, wx, wy, wz, P(d_field), P(d_qstate), P(d_queues), st))
            mark("edf_root")
            _abi.check(lib.kh_edf_batch(tasks_ptr, count, 2 | (self.edf_threads << 8), P(d_lists), P(d_nbr), sx, sy, sz, wx, wy, wz, P(d_field), P(d_qstate), P(d_queues), st))
            mark("edf_daf")
            _abi.check(lib.kh_gather_f32(P(d_field), lists_ptr, n_list, ldaf_ptr, st))
            # PDRF (trace.py:148)
            pdrf_call = lambda stage: _abi.check(lib.kh_pdrf(P(d_cc), label_bytes, nvox, P(d_slot_use), P(d_tasks), P(d_dbf), P(d_field),
                                                             stage | keep if stage >= 0 else stage, np.float32(params["pdrf_scale"]),
                                                             P(d_pdrf), st))
            if _abi.is_pow2_exponent(expo):
                pdrf_call(int(expo).bit_length() - 1)          # repeated squaring, trace.py:343-345
            else:
                # trace.py:346-347: np.power.  Its rounding is the host numpy's (libm / SVML powf), which no device powf can
                # promise, so exactly that function is applied -- by numpy itself -- between the two device halves.
                # Only the selected labels' voxels make the trip (their list is on the device already): gathered into a compact
                # array, raised on the host, scattered back -- 8 B per foreground voxel instead of 8 B per voxel of the volume.
                pdrf_call(_abi.PDRF_BASE)
                d_base = self.empty(max(total, 1), t.float32)
                _abi.check(lib.kh_gather_f32(P(d_pdrf), P(d_lists), total, P(d_base), st))
                base = d_base[:total].cpu().numpy()
                with np.errstate(all="ignore"):
                    np.power(base, expo, out=base)
                idx = d_lists[:total].to(t.int64) & 0xFFFFFFFF       # u32 linear indices kept in an int32 tensor
                d_pdrf.index_copy_(0, idx, t.from_numpy(base).to(self.device))
                pdrf_call(_abi.PDRF_FINISH)
            mark("pdrf")

        searched = False
        if early == 0 and not fuse_rest:
            searches(0, nl)          # (before the path loop's own volumes are allocated: the DAF volume's block serves them afterwards)
            searched = True
        d_dist = self.empty(nvox + 4, t.float32)     # (+ padding: the searches read rows of three words)
        _abi.check(lib.kh_fill_f32(P(d_dist), nvox + 4, float("inf"), st))
        d_alive = self.empty(nvox, t.uint8)
        _abi.check(lib.kh_init_alive(P(d_cc), label_bytes, nvox, P(d_slot), P(d_alive), st))
        jnodes = (2 * qcap + 3) // 4                        # a label's ghost journal in 16-byte nodes
        if use_pool:
            frac = float(self.scratch_pool_fraction)
            pool_nodes = int(max(frac * float((hcap + jnodes).sum()), (2 if frac >= 0.05 else 0) * float((hcap + jnodes).max()))) + 1
            pool_nodes = min(pool_nodes, 2 ** 32 - 2)
            d_heap = self.empty(2 * pool_nodes, t.int64)   # 16-byte nodes; node 0 = {handed out, capacity}
            d_heap[:2] = t.from_numpy(np.array([1, pool_nodes, 0, 0], dtype=np.uint32).view(np.int64)).to(self.device)
        else:
            d_heap = self.empty(2 * int(hcap.sum()), t.int64)  # 16-byte nodes
        d_cstate = t.zeros(nvox if sweep_on else 1, dtype=t.int64, device=self.device)
        d_sched = self.sched_volume(nvox) if sweep_on else None
        d_arena = self.empty(max(ev_total, 1) * 32 + 32, t.int64)   # units of 256 bytes, 256-byte aligned start
        arena_ptr = C.c_void_p((d_arena.data_ptr() + 255) & ~255)
        d_pverts = self.empty(int(pcap.sum()), t.int32)
        d_plens = self.empty(int(pcap.sum()), t.int32)
        # ghosts: the journal of the voxels that changed since the call that made the first ghost (2 entries per voxel of a
        # label at most: made a ghost, killed) and the weights the path vertices had before they became rails
        use_ghosts = self.ghosts and sweep_on
        d_journal = self.empty(2 * int(qcap.sum()), t.int32) if use_ghosts and not use_pool else None
        d_psave = self.empty(int(pcap.sum()), t.float32) if use_ghosts and fix_branching else None
        if use_ghosts and not fix_branching:
            d_psave = None
        # tasks are sorted by size, so the biggest labels (the tail of the run) are dispatched first; when the results
        # are consumed incrementally they go to a second stream and the others are collected while they still run
        n_large = int(min(self.split_slots, np.count_nonzero(cnt >= self.split_min_voxels)))
        if self.big_lds_labels is not None:     # (also in a lane, whose split_slots is 0: the lane then uses a second stream)
            n_large = int(min(self.big_lds_labels, np.count_nonzero(cnt >= self.split_min_voxels)))
        if early > 0:
            n_large = early
        # KH_TRACE_PROFILE | KH_TRACE_HEAP_PRIO | KH_TRACE_THREADS_64 / _128
        # ... | KH_TRACE_NO_GHOSTS | KH_TRACE_GHOST_PARANOID
        prof = (1 if self.profile else 0) | (2 if self.heap_prio else 0) | {64: 4, 128: 8}.get(self.trace_threads, 0) | \
            (0 if use_ghosts else 16) | (32 if use_ghosts and self.ghost_paranoid else 0)
        rank_ptr = P(d_rank) if d_rank is not None else C.c_void_p(0)
        if not sweep_on:
            rdims = (0, 0, 0)

        if os.environ.get("KH_DEBUG_ALLOC") == "1":      # developer knob: where every array of this call lives (to place a fault address)
            for name, tt in (("tasks", d_tasks), ("lists", d_lists), ("nbr", d_nbr), ("queues", d_queues), ("qstate", d_qstate),
                             ("ldaf", d_ldaf), ("pdrf", d_pdrf), ("dist", d_dist), ("alive", d_alive), ("heap", d_heap),
                             ("cstate", d_cstate), ("sched", d_sched), ("arena", d_arena), ("pverts", d_pverts), ("plens", d_plens),
                             ("journal", d_journal), ("psave", d_psave), ("dbf", d_dbf), ("cc", d_cc), ("tgt", d_tgt)):
                if tt is not None:
                    print("KHALLOC %-8s %#x .. %#x (%d B)" % (name, tt.data_ptr(), tt.data_ptr() + tt.numel() * tt.element_size(),
                                                             tt.numel() * tt.element_size()), file=sys.stderr, flush=True)

        kernel_events = []     # (first, count, start, end): HIP events on the stream each path-loop launch went to (timings only)

        def launch(first, count, stream, tstream=None, big=False, fused=False):
            tasks_ptr = C.c_void_p(d_tasks.data_ptr() + first * _abi.LABEL_T.itemsize)
            big_ok = self.trace_threads == 256 if self.big_lds_labels is None else True
            flags = prof | (64 if big and self.big_lds_heap and big_ok else 0) | (256 if use_pool else 0) | (512 if fused else 0)
            # (KH_TRACE_BIG_LDS_HEAP, KH_TRACE_SCRATCH_POOL, KH_TRACE_FUSED_EDF)
            if big and self.big_threads:          # the second-stream launch with a thread count of its own
                flags = (flags & ~12) | {64: 4, 128: 8}.get(self.big_threads, 0)
            if timings is not None or self.time_kernels:
                tstream = tstream if tstream is not None else t.cuda.current_stream(self.device)
                ev0, ev1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                ev0.record(tstream)
                kernel_events.append((first, count, ev0, ev1, tstream))
            _abi.check(lib.kh_trace_paths(tasks_ptr, count, P(d_lists), P(d_ldaf), P(d_nbr), sx, sy, sz, wx, wy, wz,
                                          P(d_dbf), P(d_pdrf), P(d_dist), P(d_alive), P(d_qstate), P(d_tgt),
                                          np.float32(params["scale"]), np.float32(params["const"]), P(d_queues), P(d_heap),
                                          P(d_pverts), P(d_plens), rank_ptr, rdims[0], rdims[1], rdims[2], max_nlev,
                                          P(d_cstate), self.sched_ptr(d_sched), arena_ptr, self.optr(d_journal), self.optr(d_psave),
                                          self.optr(d_gate), flags | (128 if d_gate is not None else 0), int(bool(fix_branching)), stream))
            if timings is not None or self.time_kernels:
                kernel_events[-1][3].record(kernel_events[-1][4])

        def kernel_times():
            """milliseconds of each path-loop launch of this call, from HIP events on the launch's own stream"""
            self.last_path_kernel_ms = [(c, e0.elapsed_time(e1)) for _, c, e0, e1, _ in kernel_events]
            # the span of the path phase: first start to last end over the (overlapped) launches of this call
            first = min(kernel_events, key=lambda k: -min(k[2].elapsed_time(o[2]) for o in kernel_events))[2] if kernel_events else None
            self.last_path_span_ms = max((first.elapsed_time(e1) for _, _, _, e1, _ in kernel_events), default=0.0) if first else 0.0

        def collect(lo, hi):
            """results of task slots [lo, hi) (device -> host on the current stream)."""
            isz = _abi.LABEL_T.itemsize
            part = d_tasks[lo * isz:hi * isz].cpu().numpy().view(_abi.LABEL_T).copy()
            overflow = (part["status"] & 7) != 0          # work list / heap / path buffer too small: traced again below
            bad = np.flatnonzero((part["status"] & ~np.uint32(7)) != 0)
            if bad.size or (overflow.any() and scratch_scale >= 64):
                s = int(bad[0]) if bad.size else int(np.flatnonzero(overflow)[0])
                raise _abi.KimiHipError("label %d (cc id): %s" % (int(part["segid"][s]),
                                                                   _abi.describe_status(int(part["status"][s]))))
            for s in np.flatnonzero(overflow):
                retry.append(int(order[lo + s]))
                part["n_vertices"][s] = 0
                part["n_paths"][s] = 0
            # gather the used part of the path buffers: build a flat index on the host (small), gather on device
            nverts = part["n_vertices"].astype(np.int64)
            npaths = part["n_paths"].astype(np.int64)
            def ranges(starts, counts):
                """concatenation of starts[s] + arange(counts[s]) over s, without a Python loop"""
                total = int(counts.sum())
                before = np.cumsum(counts) - counts
                return np.repeat(starts - before, counts) + np.arange(total, dtype=np.int64)
            vidx = ranges(p_off[lo:hi].astype(np.int64), nverts)
            lidx = ranges(p_off[lo:hi].astype(np.int64), npaths)
            d_vidx = t.from_numpy(vidx).to(self.device)
            d_lidx = t.from_numpy(lidx).to(self.device)
            verts_dev = d_pverts[d_vidx]
            verts = verts_dev.cpu().numpy().view(np.uint32)
            lens = d_plens[d_lidx].cpu().numpy().view(np.uint32)
            d_radii = self.empty(max(verts.size, 1), t.float32)
            if verts.size:
                _abi.check(lib.kh_gather_f32(P(d_dbf), P(verts_dev.contiguous()), verts.size, P(d_radii), self.stream()))
            radii = d_radii.cpu().numpy()[: verts.size]
            return {"order": order[lo:hi], "tasks": part, "verts": verts, "radii": radii, "lens": lens,
                    "voff": np.concatenate([[0], np.cumsum(nverts)]), "loff": np.concatenate([[0], np.cumsum(npaths)])}

        retry = []   # positions (caller order) of the labels whose scratch overflowed
        self.last_retries = 0

        def run_retry(sink):
            """re-trace the overflowed labels with 8x the scratch.  sink = None: returns the nested call's own (already
            spliced) result, whose `order` indexes `retry`; else the groups go to `sink` like every other result."""
            if not retry:
                return None
            nretry = len(retry)
            pick = np.asarray(retry, dtype=np.int64)
            sub = lambda a: [a[i] for i in pick] if a is not None else None
            subsoma = None if soma is None else {k: np.asarray(v)[pick] for k, v in soma.items()}
            got = self.run_labels(d_cc, label_bytes, d_dbf, shape, anisotropy, nlabels, segids[pick], counts[pick],
                                  np.asarray(dbf_max)[pick], np.asarray(first_index)[pick], np.asarray(xmin)[pick],
                                  np.asarray(xmax)[pick], np.asarray(roots, dtype=np.uint32)[pick], sub(targets_before),
                                  sub(targets_after), params, fix_branching=fix_branching, max_paths=max_paths, soma=subsoma,
                                  consume=sink, scratch_scale=scratch_scale * 8, voxel_graph=voxel_graph)
            self.last_retries = nretry + self.last_retries      # (the nested call counted its own)
            return got

        def splice_retried(records):
            """consume paths: re-trace the overflowed labels (their groups go to `consume`) and put the nested call's records
            -- the good attempt's status bits and statistics -- in the place of the failed attempt's."""
            if not retry:
                return records
            run_retry(consume)
            pos = {int(sid): i for i, sid in enumerate(records["segid"])}
            for rec in self.last_tasks:
                records[pos[int(rec["segid"])]] = rec
            return records

        if consume is not None and 0 < n_large < nl:
            # The largest labels are the tail of the run.  They go to a second stream (as large-LDS workgroups); the
            # rest runs on the caller's stream and its results are copied back and handed to `consume` (the Skeleton
            # assembly on the host) while the big labels are still being traced.
            cur = t.cuda.current_stream(self.device)
            if self._side is None:
                self._side = t.cuda.Stream(device=self.device)
            if not (fuse_rest or early > 0 or searched):
                searches(0, nl)                 # (no fusion at all: every label's searches before either launch)
            self._side.wait_stream(cur)
            launch(0, n_large, C.c_void_p(self._side.cuda_stream), self._side, big=(early == 0), fused=(fuse_rest or early > 0))
            try:
                if early > 0:
                    searches(n_large, nl - n_large)
                launch(n_large, nl - n_large, st, fused=fuse_rest)
                small = collect(n_large, nl)
                consume(small)                  # overlaps the big labels' kernel: no device-wide sync in here
            finally:
                cur.wait_stream(self._side)     # the scratch of this call must outlive the side stream's kernel
                if sys.exc_info()[0] is not None:
                    self._side.synchronize()
            big = collect(0, n_large)
            mark("paths")
            if timings is not None or self.time_kernels:
                kernel_times()
            consume(big)
            mark("d2h")
            self.last_tasks = splice_retried(np.concatenate([big["tasks"], small["tasks"]]))
            return None
        if not fuse_rest and not searched:
            searches(0, nl)
        if self.path_gate is not None and consume is not None:
            self.sync_stream()
            self.path_gate()
        launch(0, nl, st, fused=fuse_rest)
        mark("paths")
        res = collect(0, nl)                  # (its device -> host copies wait for the launch)
        if timings is not None or self.time_kernels:
            kernel_times()
        mark("d2h")
        if consume is not None:
            consume(res)
            self.last_tasks = splice_retried(res["tasks"])
            return None
        if retry:
            # splice the re-traced labels into the result (callers without a sink: single labels, tests)
            # (the nested call splices its own further retries before it returns: one dict, `order` indexing `retry`)
            redo = [run_retry(None)]
            pos_of = {int(o): s for s, o in enumerate(res["order"])}
            per_v = [res["verts"][res["voff"][s]:res["voff"][s + 1]] for s in range(nl)]
            per_r = [res["radii"][res["voff"][s]:res["voff"][s + 1]] for s in range(nl)]
            per_l = [res["lens"][res["loff"][s]:res["loff"][s + 1]] for s in range(nl)]
            for sub in redo:
                for s2, o2 in enumerate(sub["order"]):
                    s = pos_of[retry[int(o2)]]
                    per_v[s] = sub["verts"][sub["voff"][s2]:sub["voff"][s2 + 1]]
                    per_r[s] = sub["radii"][sub["voff"][s2]:sub["voff"][s2 + 1]]
                    per_l[s] = sub["lens"][sub["loff"][s2]:sub["loff"][s2 + 1]]
                    for f in ("n_paths", "n_vertices", "status"):
                        res["tasks"][f][s] = sub["tasks"][f][s2]
            res["verts"] = np.concatenate(per_v) if per_v else res["verts"]
            res["radii"] = np.concatenate(per_r) if per_r else res["radii"]
            res["lens"] = np.concatenate(per_l) if per_l else res["lens"]
            res["voff"] = np.concatenate([[0], np.cumsum([len(v) for v in per_v])])
            res["loff"] = np.concatenate([[0], np.cumsum([len(v) for v in per_l])])
        self.last_tasks = res["tasks"]
        if return_fields:
            res["daf"] = fields["d_field"][:nvox].cpu().numpy()
            res["pdrf"] = d_pdrf[:nvox].cpu().numpy()
            res["alive"] = d_alive.cpu().numpy()
        return res
