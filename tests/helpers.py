"""Shared test helpers: instantiate one SceneDesc on the CPU oracle (checker) and on the GPU (product)."""
import numpy as np

import ag_pathtracer_amd as ag
from oracle import binding as ob


def oracle_scene(desc, max_depth=None):
    s = desc.instantiate(ob.OracleScene())
    if max_depth is not None:
        s.set_max_depth(max_depth)
    return s


_CTX = None


def gpu_context():
    global _CTX
    if _CTX is None:
        _CTX = ag.Context(0)
    return _CTX


def gpu_scene(desc):
    return desc.instantiate(ag.Scene(gpu_context()))


def scene_bounds(desc):
    lo = np.full(3, np.inf)
    hi = np.full(3, -np.inf)
    for op in desc.ops:
        if op[0] == "mesh":
            lo = np.minimum(lo, op[1].min(0))
            hi = np.maximum(hi, op[1].max(0))
        elif op[0] in ("sphere", "area_light"):
            lo = np.minimum(lo, op[1] - op[2])
            hi = np.maximum(hi, op[1] + op[2])
        elif op[0] == "plane":
            half = np.array([op[2][0] / 2, 0, op[2][1] / 2])
            lo = np.minimum(lo, op[1] - half)
            hi = np.maximum(hi, op[1] + half)
    return lo, hi


def random_rays(desc, n, seed=1, tmax=None):
    """A mix that exercises hits, misses, grazing and axis-parallel directions, rays starting inside the scene box
    and short (shadow-like) rays."""
    rng = np.random.RandomState(seed)
    lo, hi = scene_bounds(desc)
    ext = hi - lo
    c = 0.5 * (lo + hi)
    rays = np.zeros(n, ag.RAY_DTYPE)
    o = c + (rng.uniform(-1, 1, (n, 3)) * ext * 0.9)
    # half of the rays aim at a random vertex neighbourhood (guaranteed near-hits / edge hits)
    verts = np.concatenate([op[1] for op in desc.ops if op[0] == "mesh"])
    tgt = verts[rng.randint(len(verts), size=n)] + rng.normal(0, 0.002, (n, 3)) * np.linalg.norm(ext)
    d = rng.normal(size=(n, 3))
    aim = rng.uniform(size=n) < 0.6
    d[aim] = (tgt - o)[aim]
    # exact vertex aims (hits on shared vertices/edges -> tie handling)
    exact = rng.uniform(size=n) < 0.05
    d[exact] = (verts[rng.randint(len(verts), size=n)] - o)[exact]
    # axis-parallel directions (zero components: the reference's inf/NaN slab path)
    ax = rng.uniform(size=n) < 0.05
    axis = rng.randint(3, size=n)
    sign = rng.choice([-1.0, 1.0], size=n)
    dax = np.zeros((n, 3))
    dax[np.arange(n), axis] = sign
    d[ax] = dax[ax]
    # two-zero-component and one-zero-component mixes
    one0 = rng.uniform(size=n) < 0.03
    d[one0, rng.randint(3)] = 0.0
    d[np.all(d == 0, axis=1)] = [0, 0, 1]
    rays["o"] = o.astype(np.float32)
    rays["d"] = d.astype(np.float32)
    t = np.full(n, np.float32(3.402823466e+38), np.float32)
    short = rng.uniform(size=n) < 0.3
    t[short] = (rng.uniform(0.05, 1.5, n) * np.linalg.norm(ext)).astype(np.float32)[short]
    if tmax is not None:
        t[:] = tmax
    rays["tmax"] = t
    return rays


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
