"""Host logic of kimimaro_amd.lanes (several volumes in flight): order of the results, bounded width, error hand-over.
No GPU: the lanes get stand-in engines."""
import os
import threading
import time

import pytest

from kimimaro_amd.lanes import Lanes


class _Eng:
    pass


def _lanes(width):
    return Lanes(width, engine_factory=_Eng, stream_factory=None)


def test_results_come_back_in_order_and_lanes_overlap():
    lanes = _lanes(3)
    live = [0]
    peak = [0]
    lock = threading.Lock()
    seen_engines = set()

    def job(eng, k):
        with lock:
            live[0] += 1
            peak[0] = max(peak[0], live[0])
            seen_engines.add(id(eng))
        time.sleep(0.05 if k % 2 == 0 else 0.01)     # later jobs finish before earlier ones
        with lock:
            live[0] -= 1
        return k * k

    got = list(lanes.run(job, 9))
    assert got == [(k, k * k) for k in range(9)]
    assert 2 <= peak[0] <= 3
    assert len(seen_engines) <= 3


def test_width_one_and_empty():
    lanes = _lanes(2)
    assert list(lanes.run(lambda e, k: k, 0)) == []
    order = []
    assert [v for _, v in lanes.run(lambda e, k: order.append(k) or k, 4, width=1)] == [0, 1, 2, 3]
    assert order == [0, 1, 2, 3]


def test_exception_is_raised_at_its_position():
    lanes = _lanes(2)

    def job(eng, k):
        if k == 2:
            raise ValueError("job 2")
        return k

    out = []
    with pytest.raises(ValueError, match="job 2"):
        for k, v in lanes.run(job, 6):
            out.append(v)
    assert out == [0, 1]
    # the lanes are usable again
    assert [v for _, v in lanes.run(lambda e, k: k + 1, 3)] == [1, 2, 3]


def test_bad_width():
    with pytest.raises(ValueError):
        _lanes(0)


def test_every_lane_failing_outside_its_job_raises_instead_of_hanging():
    """a lane whose stream scope cannot be entered never takes a job; when that happens to ALL lanes nobody would set the
    jobs' events -- the consumer must get an error, not wait forever (round-2 advisor finding)."""
    class BadScope:
        def __enter__(self):
            raise OSError("no stream")

        def __exit__(self, *a):
            return False

    lanes = Lanes(2, engine_factory=_Eng, stream_factory=lambda e: BadScope())
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="every lane failed"):
        list(lanes.run(lambda e, k: k, 5))
    assert time.perf_counter() - t0 < 5.0


def test_all_lanes_failing_at_the_same_moment_still_raise():
    """round-3 advisor finding: two lanes whose scopes fail together could both read "somebody else is still alive" and
    leave the consumer waiting forever.  The lanes meet at a barrier and fail together, with the interpreter switching
    threads as often as it can; `run` is driven from a watchdog thread so that a hang fails the test instead of the suite."""
    import sys
    import threading
    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-6)
    try:
        for _ in range(150):
            bar = threading.Barrier(4)

            class BadScope:
                def __enter__(self):
                    bar.wait(timeout=5)
                    raise OSError("no stream")

                def __exit__(self, *a):
                    return False

            lanes = Lanes(4, engine_factory=_Eng, stream_factory=lambda e: BadScope())
            res = []

            def drive():
                try:
                    list(lanes.run(lambda e, k: k, 6))
                    res.append("no error")
                except RuntimeError as ex:
                    res.append(str(ex))

            th = threading.Thread(target=drive, daemon=True)
            th.start()
            th.join(10)
            assert not th.is_alive(), "Lanes.run hangs when all lanes fail at once"
            assert res and "every lane failed" in res[0]
    finally:
        sys.setswitchinterval(old)


def test_one_failing_lane_is_covered_by_the_others():
    class Scope:
        def __init__(self, bad):
            self.bad = bad

        def __enter__(self):
            if self.bad:
                raise OSError("no stream")
            return self

        def __exit__(self, *a):
            return False

    made = []

    def factory(e):
        made.append(e)
        return Scope(bad=len(made) == 1)

    lanes = Lanes(3, engine_factory=_Eng, stream_factory=factory)
    assert [v for _, v in lanes.run(lambda e, k: k * 2, 7)] == [0, 2, 4, 6, 8, 10, 12]


def test_hw_queue_guard(monkeypatch):
    from kimimaro_amd import lanes as L
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    L.ensure_hw_queues(6)          # HIP not started in the CPU suite: the variable is set for the runtime to read
    import os
    assert int(os.environ["GPU_MAX_HW_QUEUES"]) >= 7


def test_process_lanes_order_overlap_and_errors():
    """ProcessLanes without a GPU (engine_factory): results in order from several processes, a failing job raised at its
    position, the lanes usable again afterwards, a failing setup reported at construction."""
    import lane_helpers as H
    from kimimaro_amd.lanes import ProcessLanes
    with ProcessLanes(3, setup=H.setup, setup_args=(100,), engine_factory=H.make_engine, start_timeout=120) as lanes:
        t0 = time.perf_counter()
        got = list(lanes.run(H.work, list(range(9))))
        dt = time.perf_counter() - t0
        assert [k for k, _ in got] == list(range(9))
        assert [v[0] for _, v in got] == [100 + i for i in range(9)]
        assert len({v[2] for _, v in got}) == 3 and os.getpid() not in {v[2] for _, v in got}   # three other processes
        assert dt < 9 * 0.05 * 0.9                                                               # ... working at the same time
        seen = []
        with pytest.raises(RuntimeError, match="job asked to fail"):
            for k, v in lanes.run(H.work, [1, 2, "boom", 4, 5]):
                seen.append(k)
        assert seen == [0, 1]
        assert [v[0] for _, v in lanes.run(H.work, [7, 8], width=1)] == [107, 108]              # still alive, one lane only
    with pytest.raises(RuntimeError, match="failed to start"):
        ProcessLanes(2, setup=H.bad_setup, engine_factory=H.make_engine, start_timeout=120)


def test_process_lanes_survive_a_dead_lane():
    import lane_helpers as H
    from kimimaro_amd.lanes import ProcessLanes
    with ProcessLanes(2, setup=H.setup, setup_args=(0,), engine_factory=H.make_engine, start_timeout=120) as lanes:
        res = {}
        with pytest.raises(RuntimeError, match="died"):
            for k, v in lanes.run(H.work, [1, "die", 3, 4]):
                res[k] = v
        assert 0 in res


def test_lanes_for_memory_rule():
    from kimimaro_amd.lanes import lanes_for
    assert lanes_for((512, 512, 512), 288e9) == 21                 # 11.3 GB per lane by the rule (round 6: 10.8 measured), 85 % of the memory
    assert lanes_for((512, 512, 512), 270e9) == 20
    assert lanes_for((512, 512, 512), 400e9) == 24                 # capped at 24
    assert lanes_for((512, 512, 512), 100e9) == 7
    assert lanes_for((1024, 1024, 1024), 288e9) == 2               # c5: two volumes in flight (round 5: one)
    assert lanes_for((512, 512, 512), 288e9, share=0.125) >= 24    # a rank of eight holds an eighth of the per-label scratch
    assert lanes_for((64, 64, 64), 1e9, most=5) == 5


def test_cohort_gate_aligns_the_path_loops_and_survives_jobs_that_never_reach_it():
    """Engine.path_gate (kimimaro_amd.lanes._CohortGate): no job of a cohort passes the gate before every job of the cohort has
    reached it or ended; a job that raises, or never calls the gate, does not hold the others; the last, smaller cohort works."""
    lanes = _lanes(3)
    lock = threading.Lock()
    reached, passed_when = {}, {}

    def job(eng, k):
        time.sleep(0.01 * (k % 3))
        if k == 1:
            raise ValueError("job 1 fails before its gate")
        if k == 4:
            return k                      # (a volume without labels: the gate is never called)
        with lock:
            reached[k] = time.perf_counter()
        eng.path_gate()
        eng.path_gate()                   # (one-shot: a second call -- the overflow retry -- returns at once)
        with lock:
            passed_when[k] = time.perf_counter()
        return k

    got = []
    with pytest.raises(ValueError):
        for k, v in lanes.run(job, 8, cohorts=True):
            got.append(v)
    assert got == [0]
    got = [v for _, v in lanes.run(lambda e, k: job(e, k + 2), 6, cohorts=True)]     # jobs 2..7: cohorts {2,3,4}, {5,6,7}
    assert got == [2, 3, 4, 5, 6, 7]
    for cohort in ((2, 3), (5, 6, 7)):
        last_reach = max(reached[k] for k in cohort)
        assert all(passed_when[k] >= last_reach for k in cohort)
    # without the gate the engines carry no hook
    assert [v for _, v in lanes.run(lambda e, k: getattr(e, "path_gate", None) is None, 3, cohorts=False)] == [True] * 3
