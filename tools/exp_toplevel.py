"""Closest-hit / any-hit throughput on long primitive lists (top-level structure): python tools/exp_toplevel.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ag_pathtracer_amd as ag
from test_gpu_long_lists import _many_prims


def main():
    ctx = ag.Context(0)
    cases = ((64, 9), (200, 9), (500, 9), (1000, 9), (200, 100), (500, 100), (1000, 100))
    if len(sys.argv) > 1:   # python tools/exp_toplevel.py 1000 [sphere_every]
        cases = ((int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 100),)
    for n_prims, sphere_every in cases:
        d = _many_prims(n_prims, 30 + n_prims, duplicates=False, sphere_every=sphere_every)
        scene = d.instantiate(ag.Scene(ctx))
        rng = np.random.RandomState(1)
        n = 1 << 22
        rays = np.zeros(n, ag.RAY_DTYPE)
        o = rng.uniform(-7, 7, (n, 3)).astype(np.float32)
        dd = rng.normal(size=(n, 3)).astype(np.float32)
        dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        rays["o"], rays["d"], rays["tmax"] = o, dd, np.float32(3.4e38)
        rd = torch.from_numpy(rays.view(np.uint8)).cuda()
        out = torch.zeros(n * ag.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        for any_hit in (0, 1):
            scene.intersect_device(rd.data_ptr(), n, out.data_ptr(), any_hit)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                scene.intersect_device(rd.data_ptr(), n, out.data_ptr(), any_hit)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print("%5d primitives (1 sphere in %3d)  %s  %8.1f Mrays/s  (%.2f ms per %d rays)" % (n_prims, sphere_every, "any-hit" if any_hit else "closest", n / dt / 1e6, dt * 1e3, n), flush=True)
        scene.close()


if __name__ == "__main__":
    main()
