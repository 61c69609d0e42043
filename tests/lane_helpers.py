"""module-level helpers of tests/test_lanes.py: functions that travel by name to the spawned lane processes."""
import os
import time


class FakeEngine:
    pass


def make_engine():
    return FakeEngine()


def setup(eng, index, base):
    return {"index": index, "base": base, "pid": os.getpid()}


def work(ctx, eng, payload):
    if payload == "boom":
        raise ValueError("job asked to fail")
    if payload == "die":
        os._exit(3)
    time.sleep(0.05)
    return (ctx["base"] + payload, ctx["index"], ctx["pid"])


def bad_setup(eng, index):
    raise OSError("no state for lane %d" % index)
