#!/usr/bin/env python3
"""Render one of the BASELINE scenes on the GPU and write a PNG (Accumulator::CopyToSurface resolve: gamma 2.2,
0x00RRGGBB) -- the presentation end of SURVEY.md section 8(f) rank 4.

    python tools/render.py --scene c3 --width 960 --height 540 --spp 64 --out gpurun_out/c3.png
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="c3", choices=["c1", "c2", "c3", "c5", "simple", "heightfield"])
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--out", default="gpurun_out/render.png")
    ap.add_argument("--pfm", default=None, help="also write the linear float image")
    a = ap.parse_args()
    W, H = a.width, a.height
    aspect = W / float(H)
    desc = {"c1": lambda: ag.scenes.scene_c1(), "c2": lambda: ag.scenes.scene_c2(aspect=aspect),
            "c3": lambda: ag.scenes.scene_c3(aspect=aspect), "c5": lambda: ag.scenes.scene_c5(aspect=aspect),
            "simple": lambda: ag.scenes.scene_simple_test(),
            "heightfield": lambda: ag.scenes.scene_heightfield(361, True, W, H)}[a.scene]()
    ctx = ag.Context(0)
    t0 = time.time()
    scene = desc.instantiate(ag.Scene(ctx))
    print("scene build + BVH + upload: %.2f s" % (time.time() - t0))
    ptr = ctx.alloc(W * H * 16)
    ctx.memset(ptr, 0, W * H * 16)
    t0 = time.time()
    st = ag.PathTracer(5).render(scene, W, H, a.spp, ptr)
    dt = time.time() - t0
    packed = ctx.resolve(ptr, W * H, a.spp)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    ag.binding.write_png(a.out, packed, W, H)
    if a.pfm:
        ag.binding.write_pfm(a.pfm, ctx.download(ptr, (H, W, 4)), a.spp)
    print("%s: %dx%d @%d spp, %d tris, %.2f s, %.1f Mrays/s -> %s" % (desc.name, W, H, a.spp, desc.n_tris, dt, st.rays / dt / 1e6, a.out))


if __name__ == "__main__":
    main()
