"""Developer probe: where does the HOST time of one volume go?  cProfile around skeletonize_cc on one engine (the GPU waits
show up under the .cpu() / synchronize calls; everything else holds the GIL of a lane's thread).
usage: python tools/host_profile.py [c3]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import kimimaro_amd  # noqa: E402
from kimimaro_amd import intake  # noqa: E402
from kimimaro_amd.engine import Engine  # noqa: E402
from collections import defaultdict  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
lab, an = bench.make_volume(which)
eng = Engine()
eng.split_slots = 0
lab = intake.format_labels(lab, in_place=True)
d_lab = eng.to_device(lab)
flat = lab.reshape(-1, order="F")
params = dict(kimimaro_amd.DEFAULT_TEASAR_PARAMS)
empty = defaultdict(list)


def step():
    d_cc, n, rep = eng.ccl_device(d_lab, lab.dtype.itemsize, lab.shape)
    orig = flat[rep[1:].astype(np.int64)]
    remap = {i + 1: orig[i].item() for i in range(n)}
    return intake.skeletonize_cc(eng, intake.LazyVolume(eng, d_cc, lab.shape), n, remap, params, np.asarray(an, dtype=np.float32), 1000,
                                 True, True, empty, empty, black_border=False, d_cc=d_cc)


step()
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
