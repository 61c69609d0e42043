import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts: kimimaro_amd.lanes needs a queue per stream
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def refmod():
    """The reference's own ext/skeletontricks compiled into oracle/_ref (or skip)."""
    from oracle import build_ref
    mod = build_ref.load()
    if mod is None:
        pytest.skip("oracle/_ref not available (reference absent and no prebuilt module)")
    return mod
