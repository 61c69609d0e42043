"""Build recipe for the product library: hipcc --offload-arch=gfx950 -> kimimaro_amd/libkimi_hip.so.

In-tree build (the .so travels to the GPU box with the snapshot; it is git-ignored).
-ffp-contract=off: no FMA contraction anywhere -- squared distances, PDRF and Dijkstra sums must be
rounded op by op exactly like numpy / the reference's g++ build / the oracle.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkimi_hip.so")
SOURCES = ["common.hip", "edt.hip", "prep.hip", "trace.hip", "ccl.hip"]
DEPS = SOURCES + ["common.h", "sweep.h", os.path.join("..", "..", "include", "kimi_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, out=None):
    """out: another file name for a developer build (KH_HIPCC_DEFINES=-DKH_SWEEP_PROBE ...; load it with KIMI_HIP_LIB)"""
    if out is not None:
        force = True
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
           "-o", out or LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    extra = os.environ.get("KH_HIPCC_DEFINES", "").split()   # developer probes, e.g. -DKH_SWEEP_PROBE (csrc/sweep.h)
    cmd[1:1] = extra
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose="-v" in sys.argv))
