"""Developer experiment: the order-free sweep with GHOSTS (experiments/cert_ball4.c) over whole labels.
A voxel the sweep cannot decide stays a ghost for the following calls.  Counts, per label, the calls, the ghosts, and
whether a ghost ever interferes with the control flow of compute_paths (target choice / termination): only then would
the exact heap emulation be needed.  Also checks soundness against the exact result after every call.
Build: gcc -O2 -ffp-contract=off -shared -fPIC experiments/cert_ball4.c -o experiments/cert_ball4.so -lm
Usage: python experiments/ghost_experiment.py mini|c2|c3 [max_labels] [first]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle as K
from oracle import pipeline as P
import bench
CERT = os.environ.get("CERT_LIB", "cert_ball4")     # cert_ball5 = + candidate spill
cert = C.CDLL(os.path.join(ROOT, "tests", "experiments", CERT + ".so"))
cert_fn = getattr(cert, CERT)
cert_fn.restype = C.c_int64
cert_fn.argtypes = [C.c_void_p] + [C.c_int64] * 3 + [C.c_float] * 3 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
orig = K.roll_invalidation_ball_inside_component
state = dict(amask=None, ahead=0, need_exact=0, calls=0, ghost_calls=0, ghosts_made=0, unsound=0, hard_bail=0)
def hooked(labels, DBF, scale, const, anisotropy, path, return_stats=False, **kw):
    lab = labels.view(np.uint8)
    if state['amask'] is None:
        state['amask'] = lab.copy(order='F'); state['ahead'] = 0
    am = state['amask']
    sx, sy, sz = lab.shape
    p = np.asarray(path, dtype=np.int64).reshape(-1, 3)
    locs = (p[:, 0] + sx * (p[:, 1] + sy * p[:, 2])).astype(np.uint64)
    radii = np.empty(locs.size, dtype=np.float32)
    K.lib().ko_ball_radii(DBF.ctypes.data_as(C.c_void_p), locs.ctypes.data_as(C.c_void_p), locs.size, np.float32(scale), np.float32(const), radii.ctypes.data_as(C.c_void_p))
    st = np.zeros(8, dtype=np.int64)
    c = cert_fn(am.ctypes.data_as(C.c_void_p), sx, sy, sz, float(anisotropy[0]), float(anisotropy[1]), float(anisotropy[2]),
                        locs.ctypes.data_as(C.c_void_p), radii.ctypes.data_as(C.c_void_p), locs.size, st.ctypes.data_as(C.c_void_p))
    out = orig(labels, DBF, scale, const, anisotropy, path, return_stats=True)
    state['calls'] += 1
    if c < 0:
        state['hard_bail'] += 1; state['need_exact'] = 1
        am[...] = lab     # resynchronise
    else:
        if st[4] > 0: state['ghost_calls'] += 1; state['ghosts_made'] += int(st[4])
        if CERT != 'cert_ball4': state['spills'] = state.get('spills', 0) + int(st[6]); state['ovf'] = state.get('ovf', 0) + int(st[7])
        if np.any((am == 0) & (lab != 0)) or np.any((am == 1) & (lab == 0)): state['unsound'] += 1
    return out if return_stats else out[:2]
K.roll_invalidation_ball_inside_component = hooked
orig_find = P._TargetFinder.find_target
def find_target(self, labels):
    t = orig_find(self, labels)
    am = state['amask']
    if am is not None and t is not None:
        aflat = am.ravel(order='F'); o = self.order; h = state['ahead']
        while h < o.size and aflat[o[h]] == 0: h += 1
        state['ahead'] = h
        if h < o.size and aflat[o[h]] == 3: state['need_exact'] = 1   # a ghost could be the target
    return t
P._TargetFinder.find_target = find_target

name = sys.argv[1] if len(sys.argv) > 1 else "mini"
maxl = int(sys.argv[2]) if len(sys.argv) > 2 else 10
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lab, an = bench.make_volume(name)
cc, n = K.connected_components(lab)
counts = np.bincount(cc.ravel())
order = np.argsort(-counts[1:]) + 1
import scipy.ndimage
slices = scipy.ndimage.find_objects(cc.T)
tot = dict(labels=0, vox=0, calls=0, ghost_labels=0, ghost_vox=0, exact_labels=0, exact_vox=0, ghosts=0, unsound=0, hard=0, end_ghost_labels=0)
for sid in order[first:first + maxl]:
    if counts[sid] <= 1000: break
    slc = slices[sid - 1][::-1]
    grown = tuple(slice(max(0, s.start - 1), min(nn, s.stop + 1)) for s, nn in zip(slc, cc.shape))
    crop = np.asfortranarray(cc[grown])
    dbf = K.edt(crop, an, black_border=False)
    mask = crop == sid
    dbf = np.where(mask, dbf, 0.0).astype(np.float32)
    state.update(amask=None, ahead=0, need_exact=0, calls=0, ghost_calls=0, ghosts_made=0, unsound=0, hard_bail=0, spills=0, ovf=0)
    paths = P.trace(mask, dbf, anisotropy=an, fix_branching=True, return_paths=True, **P.DEFAULT_TEASAR_PARAMS)
    endg = int(np.count_nonzero(state['amask'] == 3)) if state['amask'] is not None else 0
    if endg: state['need_exact'] = 1; tot['end_ghost_labels'] += 1
    tot['labels'] += 1; tot['vox'] += int(counts[sid]); tot['calls'] += state['calls']
    tot['ghosts'] += state['ghosts_made']; tot['unsound'] += state['unsound']; tot['hard'] += state['hard_bail']
    if state['ghosts_made']: tot['ghost_labels'] += 1; tot['ghost_vox'] += int(counts[sid])
    if state['need_exact']: tot['exact_labels'] += 1; tot['exact_vox'] += int(counts[sid])
    if state['ghosts_made'] or state['need_exact'] or state.get('spills'):
        print("label", sid, "vox", int(counts[sid]), "calls", state['calls'], "ghost calls", state['ghost_calls'], "ghosts", state['ghosts_made'], "end ghosts", endg, "need exact", state['need_exact'], "spills", state.get('spills', 0), "ovf voxels", state.get('ovf', 0), flush=True)
    if tot['labels'] % 20 == 0: print(tot, flush=True)
print(tot)
