"""Developer experiment: how much faster does the production closest-hit kernel trace incoherent bounce-like rays when they
arrive sorted by (origin cell, direction octant)?  Upper bound for a ray-sorting pass between shade and trace."""
import os
import sys
import time

os.environ["AGPT_INTERSECT_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
desc = ag.scenes.scene_c3()
rng = np.random.RandomState(1)
meshes = [op for op in desc.ops if op[0] == "mesh"]
# random points on random triangles (uniform over triangles), uniform directions
vs, tris = [], []
base = 0
for op in meshes:
    v, idx = op[1], op[4]
    vs.append(v)
    tris.append(idx[:, 0].reshape(-1, 3) + base)
    base += len(v)
V = np.concatenate(vs)
T = np.concatenate(tris)
t = rng.randint(len(T), size=n)
b = rng.dirichlet([1, 1, 1], n).astype(np.float32)
P = (V[T[t, 0]] * b[:, :1] + V[T[t, 1]] * b[:, 1:2] + V[T[t, 2]] * b[:, 2:]).astype(np.float32)
D = rng.normal(size=(n, 3)).astype(np.float32)
D /= np.linalg.norm(D, axis=1, keepdims=True)
rays = np.zeros(n, ag.RAY_DTYPE)
rays["o"] = P + np.float32(1e-3) * D
rays["d"] = D
rays["tmax"] = 3.402823466e+38

ctx = ag.Context(0)
scene = desc.instantiate(ag.Scene(ctx))


def run(r, label):
    best = 1e9
    for _ in range(3):
        hits, st = scene.Intersect(r, counters=True)
        best = min(best, st.trace_ms)
    print("%-28s %.2f ms  %.0f Mrays/s  hit %.1f%%" % (label, best, len(r) / best / 1e3, 100 * (hits["hit"] == 1).mean()))
    return best


run(rays, "random order")
lo, hi = P.min(0), P.max(0)
for bits in (2, 3, 4, 5):
    cell = np.minimum(((P - lo) / (hi - lo) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    key = (cell[:, 0] << (2 * bits)) | (cell[:, 1] << bits) | cell[:, 2]
    octant = (D[:, 0] < 0).astype(np.int64) | ((D[:, 1] < 0).astype(np.int64) << 1) | ((D[:, 2] < 0).astype(np.int64) << 2)
    run(rays[np.argsort(key, kind="stable")], "sorted by cell (%d bits/axis)" % bits)
    run(rays[np.argsort((key << 3) | octant, kind="stable")], "  + octant")
octant = (D[:, 0] < 0).astype(np.int64) | ((D[:, 1] < 0).astype(np.int64) << 1) | ((D[:, 2] < 0).astype(np.int64) << 2)
run(rays[np.argsort(octant, kind="stable")], "sorted by octant only")
