"""Runs only the whole-volume EDT on bench.py's c3 volume (for rocprofv3 --pmc passes), on the u16 component ids the
step itself uses when there are fewer than 65536 components."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kimimaro_amd.engine import Engine
from kimimaro_amd import intake
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
eng = Engine()
lab, an = bench.make_volume(name)
lab = intake.format_labels(lab, in_place=True)
d_cc, n, _ = eng.ccl_device(eng.to_device(lab), lab.dtype.itemsize, lab.shape)
d, L = eng.narrow(d_cc)
nvox = lab.size
out = eng.empty(nvox, torch.float32); ws = eng.empty(nvox, torch.float32)
for _ in range(3):
    eng.edt(d, L, lab.shape, an, False, out, ws)
eng.sync()
print("done", n, "components, label bytes", L, float(out.max().item()))
