"""Runs only the whole-volume EDT on bench.py's c3 volume (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kimimaro_amd.engine import Engine
from kimimaro_amd import intake
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
eng = Engine()
lab, an = bench.make_volume(name)
cc, n, _ = intake.compute_cc_labels(intake.format_labels(lab, in_place=True))
d = eng.to_device(cc)
out = eng.empty(cc.size, torch.float32); ws = eng.empty(2 * cc.size, torch.float32)
for _ in range(3):
    eng.edt(d, 4, cc.shape, an, False, out, ws)
eng.sync()
print("done", float(out.max().item()))
