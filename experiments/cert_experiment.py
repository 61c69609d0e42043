"""Developer experiment: how often does the order-free certificate of experiments/cert_ball.c hold on
the invalidation calls of realistic labels, and is the certified result equal to the exact (libstdc++ order) one?
Build: gcc -O2 -ffp-contract=off -shared -fPIC experiments/cert_ball.c -o experiments/cert_ball3.so -lm
Usage: python experiments/cert_experiment.py mini|c2 [max_labels]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle as K
from oracle import pipeline as P
import bench
cert = C.CDLL(os.path.join(ROOT, "tests", "experiments", "cert_ball3.so"))
cert.cert_ball3.restype = C.c_int64
cert.cert_ball3.argtypes = [C.c_void_p] + [C.c_int64] * 3 + [C.c_float] * 3 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
orig = K.roll_invalidation_ball_inside_component
stat = dict(calls=0, vox=0, ok=0, okvox=0, wrong=0, bail1=0, bail2=0, levels=0, nodes=0, ops=0, amb=0, t_exact=0.0, t_cert=0.0)
log = []
def hooked(labels, DBF, scale, const, anisotropy, path, return_stats=False):
    lab = labels.view(np.uint8)
    sx, sy, sz = lab.shape
    p = np.asarray(path, dtype=np.int64).reshape(-1, 3)
    locs = (p[:, 0] + sx * (p[:, 1] + sy * p[:, 2])).astype(np.uint64)
    radii = np.empty(locs.size, dtype=np.float32)
    K.lib().ko_ball_radii(DBF.ctypes.data_as(C.c_void_p), locs.ctypes.data_as(C.c_void_p), locs.size, np.float32(scale), np.float32(const), radii.ctypes.data_as(C.c_void_p))
    m = lab.copy(order='F')
    st = np.zeros(8, dtype=np.int64)
    t0 = time.perf_counter()
    c = cert.cert_ball3(m.ctypes.data_as(C.c_void_p), sx, sy, sz, float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]),
                       locs.ctypes.data_as(C.c_void_p), radii.ctypes.data_as(C.c_void_p), locs.size, st.ctypes.data_as(C.c_void_p))
    t1 = time.perf_counter()
    out = orig(labels, DBF, scale, const, anisotropy, path, return_stats=True)
    t2 = time.perf_counter()
    stat['t_cert'] += t1 - t0; stat['t_exact'] += t2 - t1
    stat['calls'] += 1; stat['vox'] += out[0]; stat['ops'] += out[2]
    stat['levels'] += int(st[0]); stat['nodes'] += int(st[1])
    if c >= 0:
        stat['ok'] += 1; stat['okvox'] += out[0]; stat['amb'] += int(st[5])
        if c != out[0] or not np.array_equal(m, lab):
            stat['wrong'] += 1
    else:
        stat['bail%d' % min(int(st[2]),2)] += 1
    log.append((int(out[0]), int(locs.size), int(c), int(st[0]), int(st[3]), int(st[2]), int(st[1]), int(st[6]), int(st[7]), int(st[5]), int(out[2])))
    return out if return_stats else out[:2]
K.roll_invalidation_ball_inside_component = hooked

name = sys.argv[1] if len(sys.argv) > 1 else "mini"
maxl = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lab, an = bench.make_volume(name)
cc, n = K.connected_components(lab)
counts = np.bincount(cc.ravel())
order = np.argsort(-counts[1:]) + 1
import scipy.ndimage
slices = scipy.ndimage.find_objects(cc.T)
only = os.environ.get('ONLY')
for sid in ([int(only)] if only else order[:maxl]):
    slc = slices[sid - 1][::-1]
    grown = tuple(slice(max(0, s.start - 1), min(nn, s.stop + 1)) for s, nn in zip(slc, cc.shape))
    crop = np.asfortranarray(cc[grown])
    dbf = K.edt(crop, an, black_border=False)
    mask = crop == sid
    dbf = np.where(mask, dbf, 0.0).astype(np.float32)
    n0 = len(log)
    paths = P.trace(mask, dbf, anisotropy=an, fix_branching=True, return_paths=True, **P.DEFAULT_TEASAR_PARAMS)
    print(sid, int(counts[sid]), len(paths), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in stat.items()}, flush=True)
    for row in sorted(log[n0:], key=lambda r: -r[0])[:8]:
        print("    call vox=%d nsrc=%d cert=%d levels=%d dead_at_end=%d bail=%d events=%d emitted=%d peak_pending=%d max_per_level=%d heap_pushes=%d" % row)
