"""Developer experiment: does the tie order of the invalidation heap matter?
Replays every roll_invalidation_ball_inside_component call of the oracle pipeline with three canonical total
orders (experiments/canon_heap.c) and counts the calls whose final mask differs from the libstdc++ order.
Build first: gcc -O2 -ffp-contract=off -shared -fPIC experiments/canon_heap.c -o experiments/canon_heap.so -lm"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle as K
from oracle import pipeline as P
from shapes import random_walk_tube, voronoi_labels
canon = C.CDLL(os.path.join(ROOT, "tests", "experiments", "canon_heap.so"))
canon.canon_ball.restype = C.c_int64
canon.canon_ball.argtypes=[C.c_void_p]+[C.c_int64]*3+[C.c_float]*3+[C.c_void_p,C.c_void_p,C.c_int64,C.c_int]
orig = K.roll_invalidation_ball_inside_component
stat = {'calls':0,'vox':0, 'mis':[0,0,0], 'misvox':[0,0,0]}
def hooked(labels, DBF, scale, const, anisotropy, path, return_stats=False):
    lab = labels.view(np.uint8)
    sx,sy,sz = lab.shape
    p = np.asarray(path,dtype=np.int64).reshape(-1,3)
    locs=(p[:,0]+sx*(p[:,1]+sy*p[:,2])).astype(np.uint64)
    radii=np.empty(locs.size,dtype=np.float32)
    K.lib().ko_ball_radii(DBF.ctypes.data_as(C.c_void_p), locs.ctypes.data_as(C.c_void_p), locs.size, np.float32(scale), np.float32(const), radii.ctypes.data_as(C.c_void_p))
    res=[]
    for mode in range(3):
        m = lab.copy(order='F')
        c = canon.canon_ball(m.ctypes.data_as(C.c_void_p), sx,sy,sz, float(anisotropy[0]),float(anisotropy[1]),float(anisotropy[2]), locs.ctypes.data_as(C.c_void_p), radii.ctypes.data_as(C.c_void_p), locs.size, mode)
        res.append(m)
    out = orig(labels, DBF, scale, const, anisotropy, path, return_stats=True)
    stat['calls']+=1; stat['vox']+=out[0]
    for mode in range(3):
        if not np.array_equal(res[mode], lab):
            stat['mis'][mode]+=1; stat['misvox'][mode]+= int(np.count_nonzero(res[mode]!=lab))
    return out if return_stats else out[:2]
K.roll_invalidation_ball_inside_component = hooked
for seed in range(40):
    an = [(1,1,1),(16,16,40),(4,4,40)][seed%3]
    shape=(48,48,40)
    m = random_walk_tube(shape, 1000+seed, steps=60, step=3.0, radius=(1.2,5.0))
    cc,n = K.connected_components(m)
    big = np.argmax(np.bincount(cc.ravel())[1:])+1
    m = (cc==big).astype(np.uint8)
    dbf = K.edt(m, an)
    params = dict(scale=[1.5,4,0.5][seed%3], const=[an[0]*2, an[0]*0.5, an[0]*6][(seed//3)%3], pdrf_scale=100000, pdrf_exponent=4)
    paths = P.trace(m, dbf, anisotropy=an, return_paths=True, **params)
    print(seed, an, int(m.sum()), len(paths), stat)
