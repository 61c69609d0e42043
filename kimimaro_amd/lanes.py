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

import threading
import time


class Lanes:
    def __init__(self, width, device=None, engine_factory=None, stream_factory=None):
        """width lanes on `device`.  The factories exist for the host-logic tests (no GPU): engine_factory() -> object
        handed to the job, stream_factory(engine) -> context manager entered by the lane's thread (or None)."""
        if width < 1:
            raise ValueError("Lanes: width must be >= 1")
        self.width = int(width)
        if engine_factory is None:
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
        (duration of one job) / width they interleave."""
        width = self.width if width is None else max(1, min(int(width), self.width))
        if n <= 0:
            return
        lock = threading.Lock()
        nxt = [0]
        out = [None] * n
        ready = [threading.Event() for _ in range(n)]
        stop = [False]

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
            if scope is None:
                loop()
            else:
                with scope:
                    loop()

        threads = [threading.Thread(target=worker, args=(self.engines[i], self._scopes[i], i * float(stagger)), daemon=True)
                   for i in range(min(width, n))]
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
