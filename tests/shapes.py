"""Deterministic synthetic shapes shared by the tests (numpy only)."""
import numpy as np


def random_walk_tube(shape, seed, steps=40, step=3.0, radius=(1.5, 4.0)):
    """A blobby random-walk tube: union of balls along a random walk (uint8 mask, F order)."""
    rng = np.random.default_rng(seed)
    sx, sy, sz = shape
    mask = np.zeros(shape, dtype=np.uint8, order="F")
    p = np.array([sx / 2, sy / 2, sz / 2], dtype=np.float64)
    gx, gy, gz = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
    d = rng.normal(size=3)
    for _ in range(steps):
        d = d + 0.7 * rng.normal(size=3)
        d /= np.linalg.norm(d) + 1e-9
        p = np.clip(p + step * d, 1, np.array(shape) - 2)
        r = rng.uniform(*radius)
        mask[(gx - p[0]) ** 2 + (gy - p[1]) ** 2 + (gz - p[2]) ** 2 <= r * r] = 1
    return mask


def voronoi_labels(shape, nlabels, seed, pts_per_label=4, step=6.0, anisotropy=(1, 1, 1), dtype=np.uint32):
    """Dense 'neurite' tessellation (SURVEY.md 8d recipe, small): every voxel gets the label of
    the nearest point of `nlabels` random-walk chains under the anisotropic metric."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    an = np.asarray(anisotropy, dtype=np.float64)
    shp = np.asarray(shape, dtype=np.float64)
    pts, owner = [], []
    for l in range(nlabels):
        p = rng.uniform(0, 1, 3) * shp
        for _ in range(pts_per_label):
            pts.append(p.copy())
            owner.append(l)
            stepv = rng.normal(size=3)
            stepv /= np.linalg.norm(stepv) + 1e-9
            p = np.clip(p + step * stepv * (an.min() / an), 0, shp - 1)
    pts = np.asarray(pts) * an
    tree = cKDTree(pts)
    gx, gy, gz = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    q = np.stack([gx.ravel(order="F"), gy.ravel(order="F"), gz.ravel(order="F")], axis=1) * an
    _, idx = tree.query(q)
    ids = 1000 + rng.permutation(nlabels)
    lab = ids[np.asarray(owner)[idx]].astype(dtype)
    return lab.reshape(shape, order="F")


def soma_shape(hole=False, shape=(64, 64, 64)):
    """A ball with four dendrites (hub and spoke), optionally with an internal void."""
    m = np.zeros(shape, np.uint8, order="F")
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float64)
    c = np.array([30, 32, 31.0])
    m[((g - c) ** 2).sum(-1) <= 14 ** 2] = 1
    for d in [np.array([1, 0.2, 0.1]), np.array([-0.3, 1, 0.2]), np.array([0.2, -0.4, 1]), np.array([-1, -0.6, -0.3])]:
        d = d / np.linalg.norm(d)
        for t in np.arange(10, 30, 0.5):
            p = c + d * t
            if (p < 1).any() or (p > np.array(shape) - 2).any():
                break
            m[((g - p) ** 2).sum(-1) <= 2.2 ** 2] = 1
    if hole:
        m[28:32, 30:34, 29:33] = 0
    return m


def lollipop(length=1500, radius=14):
    """A 1-voxel-thick stick ending in a ball: with pdrf_scale 1e5 the accumulated PDRF along the stick
    exceeds 2^24 times the tiny PDRF near the ball centre -> float-absorption plateau in the railroad."""
    n = 2 * radius + 5
    shape = (length + n, n, n)
    m = np.zeros(shape, np.uint8, order="F")
    c = np.array([length + radius + 2, n // 2, n // 2])
    g = np.stack(np.meshgrid(np.arange(length, shape[0]), np.arange(n), np.arange(n), indexing="ij"), -1)
    ball = ((g - c) ** 2).sum(-1) <= radius ** 2
    m[length:][ball] = 1
    m[1:length + 3, n // 2, n // 2] = 1
    return m, tuple(int(v) for v in c)


def avocado_volume(shape=(72, 64, 48), seed=3):
    """cells whose nucleus carries a label of its own ("pit" inside "fruit", kimimaro/intake.py:600-704): a free-standing avocado,
    one cut by a wall of the volume, a nested one (nucleolus in nucleus in cell), a plain blob and a thin process, on background."""
    lab = np.zeros(shape, dtype=np.uint32, order="F")
    gx, gy, gz = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")

    def ball(c, r):
        return ((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0

    lab[ball((20, 20, 22), (14, 13, 12))] = 11          # fruit
    lab[ball((21, 20, 22), (7, 6, 6))] = 12             # its pit
    lab[ball((4, 48, 24), (12, 11, 10))] = 21           # fruit cut by the x = 0 wall
    lab[ball((3, 48, 24), (6, 6, 5))] = 22              # pit on the wall
    lab[ball((52, 40, 24), (16, 15, 14))] = 31          # nested: cell
    lab[ball((52, 40, 24), (10, 9, 9))] = 32            # nucleus
    lab[ball((52, 40, 25), (4, 4, 4))] = 33             # nucleolus
    lab[ball((50, 10, 12), (8, 7, 7))] = 41             # plain blob
    lab[30:66, 58:61, 40:43] = 51                       # thin process
    return lab
