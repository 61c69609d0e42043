"""Times the three EDT passes on bench.py's volume (default c3) with HIP events on the launch stream (kh_edt_timed), the way
bench.py's roofline_edt does: mean of 10 runs after 2 warm-ups."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from kimimaro_amd import _abi, intake
from kimimaro_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
eng = Engine()
lab, an = bench.make_volume(name)
lab = intake.format_labels(lab, in_place=True)
d_cc, n, _ = eng.ccl_device(eng.to_device(lab), lab.dtype.itemsize, lab.shape)
d, L = eng.narrow(d_cc)
nvox = lab.size
out = eng.empty(nvox, torch.float32)
ws = eng.empty(2 * nvox, torch.float32)
ms3 = (C.c_float * 3)()
acc = np.zeros(3)
for i in range(12):
    _abi.check(eng.lib.kh_edt_timed(eng.ptr(d), L, lab.shape[0], lab.shape[1], lab.shape[2], float(an[0]), float(an[1]), float(an[2]), 0,
                                    eng.ptr(ws), eng.ptr(out), eng.stream(), ms3))
    if i >= 2:
        acc += np.array(list(ms3))
ms = acc / 10
print("EDTTIME %s L=%d: x %.4f y %.4f z %.4f ms, total %.4f ms -> %.1f GB/s of (3L+20) B/voxel = %.4f of 8 TB/s" % (
    name, L, ms[0], ms[1], ms[2], ms.sum(), (3 * L + 20) * nvox / (ms.sum() * 1e-3) / 1e9, (3 * L + 20) * nvox / (ms.sum() * 1e-3) / 1e9 / 8000))
