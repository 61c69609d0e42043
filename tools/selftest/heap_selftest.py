"""Developer self-test (GPU): random scripts of pushes / batched fires / pops through the wave code of the invalidation heap
(tools/selftest/heap_selftest.hip includes kimimaro_amd/csrc/trace.hip) against bits/stl_heap.h restated in Python.
Build here:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -o tools/selftest/heap_selftest.so \
             tools/selftest/heap_selftest.hip kimimaro_amd/csrc/common.hip
Run on the GPU box: python tools/selftest/heap_selftest.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "heap_selftest.so"))
lib.heap_selftest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]


class Ref:
    def __init__(self):
        self.a = []

    def push(self, k, i):
        a = self.a
        a.append(None)
        hole = len(a) - 1
        while hole > 0:
            p = (hole - 1) // 2
            if not a[p][0] >= k:
                break
            a[hole] = a[p]
            hole = p
        a[hole] = (k, i)

    def pop(self):
        a = self.a
        top = a[0]
        ln = len(a)
        if ln > 1:
            ln -= 1
            value = a[ln]
            hole = child = 0
            while child < (ln - 1) // 2:
                child = 2 * (child + 1)
                if a[child][0] >= a[child - 1][0]:
                    child -= 1
                a[hole] = a[child]
                hole = child
            if (ln & 1) == 0 and child == (ln - 2) // 2:
                child = 2 * (child + 1)
                a[hole] = a[child - 1]
                hole = child - 1
            while hole > 0:
                p = (hole - 1) // 2
                if not a[p][0] >= value[0]:
                    break
                a[hole] = a[p]
                hole = p
            a[hole] = value
        a.pop()
        return top


def one(seed, steps, nkeys, wlds, live):
    rng = np.random.default_rng(seed)
    ref = Ref()
    script, want = [], []
    nid = 0
    for _ in range(int(rng.integers(1, 40))):
        script += [1, 0, nid]
        ref.push(0, nid)
        nid += 1
    base = 0
    for _ in range(steps):
        if not ref.a:
            break
        script.append(0)
        want.append(ref.pop()[1])
        if rng.random() < live and len(ref.a) < 200000:
            keys = [int(base + rng.integers(0, nkeys)) for _ in range(26)]
            m = 0
            for l in range(26):
                if rng.random() < 0.45:
                    m |= 1 << l
            script += [2, m & 0xFFFFFFFF, m >> 32] + keys + [nid]
            for l in range(26):
                if (m >> l) & 1:
                    ref.push(keys[l], nid)
                    nid += 1
        if rng.random() < 0.03:
            base += 1
    cap = 1 << 19
    dev = torch.device("cuda", 0)
    d_nodes = torch.zeros(4 * (cap + 300000), dtype=torch.int32, device=dev)
    d_script = torch.from_numpy(np.asarray(script, dtype=np.uint32).view(np.int32)).to(dev)
    d_pop = torch.zeros(max(len(want), 1) + 8, dtype=torch.int32, device=dev)
    d_n = torch.zeros(2, dtype=torch.int32, device=dev)
    rc = lib.heap_selftest(d_nodes.data_ptr(), cap, d_script.data_ptr(), len(script), d_pop.data_ptr(), d_n.data_ptr(), wlds)
    assert rc == 0, rc
    n, npop = (int(v) for v in d_n.cpu().numpy())
    got = d_pop.cpu().numpy().view(np.uint32)[:npop]
    ok = npop == len(want) and n == len(ref.a) and np.array_equal(got, np.asarray(want, dtype=np.uint32))
    if ok and n:
        nodes = d_nodes.cpu().numpy().view(np.uint32).reshape(-1, 4)[:n]
        ok = np.array_equal(nodes[:, 0], np.asarray([k for k, _ in ref.a], dtype=np.uint32)) and \
            np.array_equal(nodes[:, 1], np.asarray([i for _, i in ref.a], dtype=np.uint32))
    if not ok:
        first = next((i for i in range(min(npop, len(want))) if got[i] != want[i]), None)
        print("MISMATCH seed %d wlds %d: pops %d vs %d, left %d vs %d, first differing pop %s" % (seed, wlds, npop, len(want), n, len(ref.a), first))
    return ok, len(want), len(ref.a)


if __name__ == "__main__":
    bad = 0
    total = 0
    for seed in range(24):
        steps = 400 if seed < 8 else (4000 if seed < 20 else 40000)
        ok, npop, left = one(seed, steps, [3, 50, 7, 200][seed % 4], 65 if seed % 2 == 0 else 4161, 0.3 if seed < 20 else 0.55)
        total += npop
        bad += not ok
        print("seed %d: %s (%d pops, %d left in the heap)" % (seed, "ok" if ok else "FAILED", npop, left), flush=True)
    print("heap_selftest: %d scripts failed of 24, %d pops" % (bad, total))
    sys.exit(1 if bad else 0)
