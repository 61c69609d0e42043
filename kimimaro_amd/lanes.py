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
                return Engine(device)

            def stream_factory(eng):
                return _StreamScope(torch, eng)
        self.engines = [engine_factory() for _ in range(self.width)]
        self._scopes = [stream_factory(e) if stream_factory is not None else None for e in self.engines]

    def run(self, job, n, width=None, stagger=0.0):
        """Generator over (k, job(engine, k)) for k = 0..n-1, in order, with at most `width` jobs in flight.  An exception
        of job k is raised when k is reached (later jobs may have run).
        stagger: lane i takes its first job i * stagger seconds after lane 0.  Jobs of equal length started together stay
        in lock step -- their GPU-filling phases collide and their tails leave the GPU idle together; offset by
        (duration of one job) / width they interleave.
        When the consumer stops early (an exception of job k, or the generator is closed) no NEW job is started, but the
        jobs already in flight are waited for before this returns."""
        width = self.width if width is None else max(1, min(int(width), self.width))
        if n <= 0:
            return
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
                        out[k] = (True, job(eng, k))
                        if scope is not None and hasattr(scope, "synchronize"):
                            scope.synchronize()
                    except BaseException as ex:  # handed to the caller at position k
                        out[k] = (False, ex)
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
            for th in threads:
                th.join()


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
