"""Build recipe for oracle/_ref: the reference's own ext/skeletontricks, compiled
from the sources WHERE THEY LIE under /root/reference (nothing is copied into this
repo; only the compiled module lands in oracle/_ref/, which is git-ignored).

TEST INFRASTRUCTURE ONLY.  The product path (kimimaro_amd/*) never imports this.

What it gives us: a real oracle for
  roll_invalidation_ball_inside_component, roll_invalidation_cube,
  CachedTargetFinder, find_target, first_label, zero2inf, inf2zero,
  find_border_targets, get_mapping
which we use to (1) pin oracle/*.c (tests/test_oracle_vs_ref.py, runs only in the
build container where /root/reference exists) and (2) generate tests/golden/*.npz.

Recipe = cython (pyx -> cpp, written to oracle/_ref/build) + g++ -std=c++17 -O3,
i.e. what the reference's setup.py:27-36 does, driven by hand instead of by its
build system.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("KIMI_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "ext", "skeletontricks")


def build(force=False):
    import numpy as np

    os.makedirs(os.path.join(OUT, "build"), exist_ok=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    target = os.path.join(OUT, "skeletontricks" + ext)
    if os.path.exists(target) and not force:
        return target
    if not os.path.isdir(SRC):
        return None  # GPU box: reference absent, use prebuilt or skip
    cpp = os.path.join(OUT, "build", "skeletontricks.cpp")
    subprocess.check_call(
        [sys.executable, "-m", "cython", "-3", "--cplus", "-o", cpp,
         os.path.join(SRC, "skeletontricks.pyx")]
    )
    inc = sysconfig.get_paths()["include"]
    subprocess.check_call(
        ["g++", "-std=c++17", "-O3", "-shared", "-fPIC", "-w",
         "-I", SRC, "-I", inc, "-I", np.get_include(),
         cpp, "-o", target]
    )
    os.remove(cpp)  # keep only the binary: no derived reference source stays in the tree
    return target


def load():
    """Import the compiled reference module (None if unavailable)."""
    target = build()
    if target is None or not os.path.exists(target):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("skeletontricks", target)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
