"""Build recipe for the oracle (test infrastructure): gcc -> oracle/libkimi_oracle.so.

-ffp-contract=off: every float op is individually rounded, like numpy and like the
reference's g++ -O3 build on baseline x86-64 (no FMA), and like the HIP kernels.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kimi_oracle.c")
LIB = os.path.join(HERE, "libkimi_oracle.so")


def build(force=False):
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    subprocess.check_call(
        ["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-shared", "-fPIC", SRC, "-o", LIB, "-lm"]
    )
    return LIB


if __name__ == "__main__":
    print(build(force=True))
