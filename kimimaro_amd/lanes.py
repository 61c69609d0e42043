"""Several volumes in flight on one GPU.

The wall clock of ONE volume is the chain of its largest connected component: a handful of workgroups trace for
seconds while the rest of the GPU idles.  A deployment skeletonizes thousands of chunks
(kimimaro's callers cut a dataset into 512^3 tasks, README.md "Scaling"), so the GPU is kept busy by overlapping
consecutive volumes: every lane is a host thread with a HIP stream and an Engine of its own (scratch comes from
that stream's pool of the caching allocator), jobs are taken in order, results are handed back in order by the
calling thread -- which is therefore the only thread that issues collectives.

    lanes = Lanes(4)
    for k, skeletons in lanes.run(lambda eng, k: kimimaro_amd.skeletonize(volumes[k], _engine=eng, ...), len(volumes)):
        ...

No CPU fallback: a lane is an Engine, and an Engine needs the GPU.
"""
from __future__ import annotations

import os
import threading
import time


LANE_THREADS = 64     # threads per label of a lane's path loop (Engine.trace_threads): one wave, twelve labels per CU
LANE_EDF_THREADS = 128   # threads per label of a lane's distance-field searches (Engine.edf_threads)
LANE_SWITCH_INTERVAL = 0.0005   # seconds: sys.setswitchinterval while several lanes run (Python's default is 0.005)
LANE_WINDOW_CAP = 2048   # level words per label in LDS (Engine.window_cap): 11.5 KB per workgroup, so that twelve fit a CU's 160 KB


def ensure_hw_queues(width):
    """Every lane stream needs a hardware queue of its own: with fewer queues than streams a lane's kernels wait behind
    another lane's seconds-long path kernel (profiles/r02b_inflight_timeline.txt).  The HIP runtime reads
    GPU_MAX_HW_QUEUES once, when it initialises (default 4): set it here while that is still possible, otherwise verify it
    and fail loudly rather than run with the silently serialised lanes that were measured as broken."""
    want = max(8, 2 * int(width))
    have = os.environ.get("GPU_MAX_HW_QUEUES")
    if have is not None and not have.strip().isdigit():
        have = None           # not a number: treat as unset
    try:
        import torch
        started = torch.cuda.is_initialized()
    except ImportError:      # host-logic tests without torch
        started = False
    if not started:
        if have is None or int(have) < width + 1:
            if have is not None:
                import warnings
                warnings.warn("kimimaro_amd.Lanes(%d): GPU_MAX_HW_QUEUES=%s is too small for %d lanes; raised to %d"
                              % (width, have, width, want))
            os.environ["GPU_MAX_HW_QUEUES"] = str(want)
        return
    if (have is None and width > 3) or (have is not None and int(have) < width + 1):
        if os.environ.get("KIMI_LANES_ALLOW_SHARED_QUEUES") == "1":
            return
        raise RuntimeError(
            "kimimaro_amd.Lanes(%d): the HIP runtime is already initialised with GPU_MAX_HW_QUEUES=%s; %d lanes need at "
            "least %d hardware queues.  Export GPU_MAX_HW_QUEUES=%d before the first GPU call (importing kimimaro_amd "
            "before touching the GPU does it), or set KIMI_LANES_ALLOW_SHARED_QUEUES=1 to accept lanes that wait for "
            "each other." % (width, have or "unset (4)", width, width + 1, want))


class _CohortGate:
    """Phase alignment of the volumes in flight.  Jobs k = c * width .. (c + 1) * width - 1 form cohort c; a lane that has issued
    the searches of its volume (find_root, DAF, PDRF: short launches that fill the GPU) waits here, with its stream drained, until
    every job of the cohort has done so, and only then launches its path loop (seconds of a few long chains).  Without the gate
    the path workgroups of the lanes that got through their preamble first hold every wave slot while the other lanes' search
    workgroups (eight waves each, all on one CU) wait for slots: in a round of twenty volumes the forty search launches took
    3.4 s for 1.3 s of work (profiles/r06_lanes20_timeline.txt).  A job that ends without reaching the gate (an exception, a
    volume without labels) is counted by done()."""

    def __init__(self, n, width):
        self.n, self.width = int(n), int(width)
        self.cv = threading.Condition()
        self.arrived = {}          # cohort -> set of jobs that reached the gate or ended
        self.broken = False

    def _size(self, c):
        return min(self.width, self.n - c * self.width)

    def _arrive(self, k):
        c = k // self.width
        self.arrived.setdefault(c, set()).add(k)
        self.cv.notify_all()
        return c

    def wait(self, k):
        with self.cv:
            c = self._arrive(k)
            while len(self.arrived[c]) < self._size(c) and not self.broken:
                self.cv.wait(timeout=0.25)

    def done(self, k):
        with self.cv:
            self._arrive(k)

    def abort(self):
        with self.cv:
            self.broken = True
            self.cv.notify_all()


class Lanes:
    def __init__(self, width, device=None, engine_factory=None, stream_factory=None):
        """width lanes on `device`.  The factories exist for the host-logic tests (no GPU): engine_factory() -> object
        handed to the job, stream_factory(engine) -> context manager entered by the lane's thread (or None)."""
        if width < 1:
            raise ValueError("Lanes: width must be >= 1")
        self.width = int(width)
        if engine_factory is None:
            ensure_hw_queues(self.width)
            from .engine import Engine
            import torch

            def engine_factory():
                e = Engine(device)
                if "KH_TRACE_THREADS" not in os.environ:
                    e.trace_threads = LANE_THREADS
                if "KH_FUSE_EDF" not in os.environ:
                    # the searches stay batch launches of their own in a lane (70 VGPRs: twice the waves per CU of the path kernel);
                    # fused into the path kernel (the single-volume default) twenty volumes took 523 ms per step against 428
                    e.fuse_edf = False
                if "KH_EDF_THREADS" not in os.environ:
                    # two waves per label in the searches: eight labels per CU instead of two (the kernel holds four waves per SIMD);
                    # behind the cohort gate the forty search launches of a round are one phase: 373 / 363 / 368 ms per step with
                    # 512 / 128 / 64 threads
                    e.edf_threads = LANE_EDF_THREADS
                if "KH_WINDOW_CAP" not in os.environ and e.trace_threads == LANE_THREADS:
                    e.window_cap = LANE_WINDOW_CAP
                e.soma_lanes = 1        # (no lanes inside a lane: the other volumes are what fills the GPU)
                return e

            def stream_factory(eng):
                return _StreamScope(torch, eng)
        self.engines = [engine_factory() for _ in range(self.width)]
        self._scopes = [stream_factory(e) if stream_factory is not None else None for e in self.engines]

    def close(self):
        """drop the lanes' engines AND their stream scopes (each scope holds its engine and HIP stream: clearing `engines`
        alone frees nothing), so that their scratch goes back to the allocator"""
        self.engines = []
        self._scopes = []

    def run(self, job, n, width=None, stagger=0.0, cohorts=None):
        """Generator over (k, job(engine, k)) for k = 0..n-1, in order, with at most `width` jobs in flight.  An exception
        of job k is raised when k is reached (later jobs may have run).
        stagger: lane i takes its first job i * stagger seconds after lane 0.  Jobs of equal length started together stay
        in lock step -- their GPU-filling phases collide and their tails leave the GPU idle together; offset by
        (duration of one job) / width they interleave.
        cohorts: the lanes' path loops start together, cohort by cohort (_CohortGate); default on (KH_COHORT_GATE=0: off).
        When the consumer stops early (an exception of job k, or the generator is closed) no NEW job is started, but the
        jobs already in flight are waited for before this returns."""
        width = self.width if width is None else max(1, min(int(width), self.width))
        if n <= 0:
            return
        if cohorts is None:
            cohorts = os.environ.get("KH_COHORT_GATE", "1") != "0"
        gate = _CohortGate(n, min(width, n)) if (cohorts and min(width, n) > 1 and stagger <= 0) else None
        lock = threading.Lock()
        nxt = [0]
        out = [None] * n
        ready = [threading.Event() for _ in range(n)]
        stop = [False]

        alive = [0]
        failed = [None]   # the first failure of a lane outside its jobs

        def worker(eng, scope, delay):
            def loop():
                if delay > 0:
                    time.sleep(delay)
                while True:
                    with lock:
                        k = nxt[0]
                        nxt[0] += 1
                    if k >= n or stop[0]:
                        return
                    try:
                        if gate is not None:
                            _set_gate(eng, gate, k)
                        out[k] = (True, job(eng, k))
                        if scope is not None and hasattr(scope, "synchronize"):
                            scope.synchronize()
                    except BaseException as ex:  # handed to the caller at position k
                        out[k] = (False, ex)
                    finally:
                        if gate is not None:
                            _set_gate(eng, None, k)
                            gate.done(k)
                    ready[k].set()
            failure = None
            try:
                if scope is None:
                    loop()
                else:
                    with scope:
                        loop()
            except BaseException as ex:
                failure = ex   # the lane itself failed (its stream scope, not a job); the other lanes take over its jobs
            # Leaving and "was I the last one" are ONE critical section: two lanes failing at the same moment (a lost GPU
            # makes every scope fail together) must not both conclude that somebody else is still there.  The last lane
            # to leave after a lane failure hands that failure to every job not yet taken, so that the consumer raises
            # instead of waiting forever on events nobody would set.
            with lock:
                alive[0] -= 1
                if failure is not None and failed[0] is None:
                    failed[0] = failure
                last = alive[0] == 0 and failed[0] is not None
                first = nxt[0]
                if last:
                    nxt[0] = n
            if last:
                for k in range(first, n):
                    out[k] = (False, RuntimeError("kimimaro_amd.Lanes: every lane failed outside its job: %r" % (failed[0],)))
                    ready[k].set()

        threads = [threading.Thread(target=worker, args=(self.engines[i], self._scopes[i], i * float(stagger)), daemon=True)
                   for i in range(min(width, n))]
        alive[0] = len(threads)
        # A lane's thread gives the GIL up at every GPU wait and needs it back afterwards; with CPython's default switch interval
        # (5 ms) it then stands behind whichever lane is running Python, and a preamble with some twenty such waits per volume
        # turns into a convoy.  A short interval hands the GIL over quickly (KH_SWITCH_INTERVAL seconds; 0 keeps Python's value).
        import sys
        old_interval = sys.getswitchinterval()
        want_interval = float(os.environ.get("KH_SWITCH_INTERVAL", LANE_SWITCH_INTERVAL))
        if want_interval > 0 and len(threads) > 1:
            sys.setswitchinterval(want_interval)
        for th in threads:
            th.start()
        try:
            for k in range(n):
                ready[k].wait()
                ok, val = out[k]
                out[k] = None
                if not ok:
                    raise val
                yield k, val
        finally:
            stop[0] = True
            if gate is not None:
                gate.abort()
            for th in threads:
                th.join()
            sys.setswitchinterval(old_interval)


def _set_gate(eng, gate, k):
    """Engine.path_gate: called once, by run_labels, between the searches of a volume and its path loop (kimimaro_amd/engine.py)"""
    if gate is None:
        fn = None
    else:
        state = {"open": False}

        def fn():
            if not state["open"]:
                state["open"] = True
                gate.wait(k)
    try:
        eng.path_gate = fn
    except AttributeError:      # (the host-logic tests hand over engines that are not Engine objects)
        pass


def lanes_for(shape, free_bytes, most=24, share=1.0):
    """how many volumes of `shape` to keep in flight on a GPU with `free_bytes` of HBM free: a lane holds the whole-volume
    fields of its volume (34 B per voxel at 512^3 scale: u16 ids, DBF, neighbour masks, PDRF, search scratch, the sweep's per-voxel
    words) and the per-label scratch of the components it traces (50 B per voxel when the volume is all foreground: voxel lists,
    work lists, event arenas, the pool heap and journal are taken from, path buffers; `share` = the fraction of them this process
    traces).  Measured in round 6: 10.8 GB per lane at 512^3 (216.7 GB reserved with twenty lanes; round 5: 19.7 GB, ten lanes);
    85 % of the free memory at most, `most` lanes at most."""
    nvox = 1
    for v in shape:
        nvox *= int(v)
    per_lane = (34.0 + 50.0 * float(share)) * nvox
    return int(max(1, min(int(most), (0.85 * float(free_bytes)) // per_lane)))


def skeletonize_many(volumes, teasar_params=None, lanes=None, width=None, **kwargs):
    """kimimaro.skeletonize over MANY label volumes with several of them in flight on this GPU -- the form in which a dataset
    cut into chunks (kimimaro's own deployment, README "Scaling"; `parallel=` of kimimaro/intake.py:344-408 has no meaning on
    one GPU) reaches the throughput of bench.py.  `volumes`: a sequence of label arrays, or of zero-argument callables that load
    one (called by the lane that takes the job, so loading overlaps tracing).  Yields (index, {label: Skeleton}) in order.
    `lanes`: a Lanes object to reuse; else `width` lanes are made (default: lanes_for() on the first volume's shape).
    The other keyword arguments are those of kimimaro_amd.skeletonize."""
    from . import intake
    n = len(volumes)
    if n == 0:
        return
    params = intake.DEFAULT_TEASAR_PARAMS if teasar_params is None else teasar_params
    cache = {}

    def load(k):
        v = volumes[k]
        return v() if callable(v) else v

    own = lanes is None
    if own:
        if width is None:
            import torch
            cache[0] = load(0)
            width = min(n, lanes_for(cache[0].shape, torch.cuda.mem_get_info()[0]))
        lanes = Lanes(max(1, int(width)))

    def job(eng, k):
        lab = cache.pop(k) if k in cache else load(k)
        return intake.skeletonize(lab, params, _engine=eng, **kwargs)

    try:
        yield from lanes.run(job, n)
    finally:
        if own:
            lanes.close()


class _StreamScope:
    """Makes the lane's device and a non-blocking stream of its own current in the lane's thread."""

    def __init__(self, torch, eng):
        self.torch = torch
        self.eng = eng
        self.stream = torch.cuda.Stream(device=eng.device)
        self._ctx = None
        # the other lanes are what overlaps the tail of this lane's largest components: no second stream per lane
        eng.split_slots = 0

    def __enter__(self):
        self.torch.cuda.set_device(self.eng.device)
        self._ctx = self.torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)

    def synchronize(self):
        self.stream.synchronize()


# ---------------------------------------------------------------------------------------------------------------------
# Lanes as PROCESSES.  The thread lanes above share one Python interpreter: a volume costs ~0.2 s of Python (border targets,
# result collection, Skeleton assembly) and every kernel launch re-acquires the interpreter lock, so with four volumes in
# flight the lanes wait for each other's Python as much as for the GPU.  The reference parallelises the same way
# (kimimaro/intake.py:344-408: a pool of processes over the components); here a process owns a whole volume at a time.

def _lane_main(conn, device, setup, setup_args, index, engine_factory):
    """worker process: one Engine (own HIP context, own caching allocator), jobs one at a time over the pipe."""
    import traceback
    try:
        if engine_factory is None:
            from .engine import Engine
            eng = Engine(device)
            if "KH_TRACE_THREADS" not in os.environ:
                eng.trace_threads = LANE_THREADS
            if "KH_FUSE_EDF" not in os.environ:
                eng.fuse_edf = False
            eng.split_slots = 0       # the other lanes are what overlaps the tail of this lane's largest components
        else:
            eng = engine_factory()
        ctx = setup(eng, index, *setup_args) if setup is not None else None
        conn.send(("ready", None))
    except BaseException:
        conn.send(("failed", traceback.format_exc()))
        return
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            return
        if msg is None:
            try:
                conn.send(("stats", {"hbm_reserved_peak": int(eng.torch.cuda.max_memory_reserved()) if hasattr(eng, "torch") else 0}))
            except Exception:
                pass
            return
        k, work, payload = msg
        try:
            res = work(ctx, eng, payload)
            if hasattr(eng, "sync"):
                eng.sync()
            conn.send((k, True, res))
        except BaseException:
            conn.send((k, False, traceback.format_exc()))


class ProcessLanes:
    """`width` worker processes on one GPU, each with an Engine of its own.

        lanes = ProcessLanes(4, setup=load_my_state, setup_args=(...))      # setup(engine, lane_index, *setup_args) -> ctx
        for k, result in lanes.run(work, payloads):                          # work(ctx, engine, payload) -> picklable result
            ...
        lanes.close()

    `setup` and `work` are module-level functions (they travel by name to processes started with the "spawn" method: a
    forked child could not use the HIP runtime its parent has initialised).  Results come back in order; an exception of
    job k is raised when k is reached."""

    def __init__(self, width, setup=None, setup_args=(), device=None, start_timeout=900.0, engine_factory=None):
        """engine_factory (module-level callable, host-logic tests only): what a lane builds instead of Engine(device)."""
        import multiprocessing as mp
        if width < 1:
            raise ValueError("ProcessLanes: width must be >= 1")
        self.width = int(width)
        self.stats = []
        ctx = mp.get_context("spawn")
        self._conns, self._procs = [], []
        for i in range(self.width):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_lane_main, args=(b, device, setup, tuple(setup_args), i, engine_factory), daemon=True)
            p.start()
            b.close()
            self._conns.append(a)
            self._procs.append(p)
        t_end = time.time() + start_timeout
        for i, c in enumerate(self._conns):
            if not c.poll(max(0.0, t_end - time.time())):
                self.close()
                raise RuntimeError("kimimaro_amd.ProcessLanes: lane %d did not come up within %.0f s" % (i, start_timeout))
            try:
                tag, info = c.recv()
            except EOFError:
                tag, info = "failed", "the process died while starting"
            if tag != "ready":
                self.close()
                raise RuntimeError("kimimaro_amd.ProcessLanes: lane %d failed to start:\n%s" % (i, info))

    def run(self, work, payloads, width=None, stagger=0.0):
        """Generator over (k, work(ctx, engine, payloads[k])), in order, at most one job per lane in flight (`width` lanes).
        stagger: lane i gets its first job i * stagger seconds after lane 0 (see Lanes.run)."""
        from multiprocessing.connection import wait
        payloads = list(payloads)
        n = len(payloads)
        width = self.width if width is None else max(1, min(int(width), self.width))
        if n <= 0:
            return
        out = [None] * n
        ready = [threading.Event() for _ in range(n)]
        stop = [False]

        def pump():
            conns = self._conns[:width]
            busy = {}                      # connection -> job index
            t0 = time.time()
            first_at = {c: t0 + i * float(stagger) for i, c in enumerate(conns)}
            nxt = 0
            dead = set()
            while True:
                now = time.time()
                for c in conns:
                    if c not in busy and c not in dead and nxt < n and not stop[0] and now >= first_at[c]:
                        try:
                            c.send((nxt, work, payloads[nxt]))
                            busy[c] = nxt
                            nxt += 1
                        except (BrokenPipeError, OSError):
                            dead.add(c)
                if not busy:
                    if nxt >= n or stop[0]:
                        return
                    if len(dead) == len(conns):
                        for k in range(nxt, n):
                            out[k] = (False, "every lane process died")
                            ready[k].set()
                        return
                    time.sleep(0.002)      # (only while lanes wait for their staggered start)
                    continue
                pending = [first_at[c] - now for c in conns if c not in busy and c not in dead and first_at[c] > now]
                for c in wait(list(busy), timeout=min(pending) if pending and nxt < n else None):
                    k = busy.pop(c)
                    try:
                        kk, ok, val = c.recv()
                        out[kk] = (ok, val)
                    except (EOFError, OSError):
                        dead.add(c)
                        out[k] = (False, "the lane process died during job %d" % k)
                    ready[k].set()

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        try:
            for k in range(n):
                ready[k].wait()
                ok, val = out[k]
                out[k] = None
                if not ok:
                    raise RuntimeError("kimimaro_amd.ProcessLanes: job %d failed in its lane:\n%s" % (k, val))
                yield k, val
        finally:
            stop[0] = True
            th.join()

    def close(self):
        for c in self._conns:
            try:
                c.send(None)
            except Exception:
                pass
        for c, p in zip(self._conns, self._procs):
            try:
                if c.poll(30.0):
                    tag, info = c.recv()
                    if tag == "stats":
                        self.stats.append(info)
            except Exception:
                pass
            p.join(30.0)
            if p.is_alive():
                p.terminate()
            try:
                c.close()
            except Exception:
                pass
        self._conns, self._procs = [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
