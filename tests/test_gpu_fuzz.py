"""Seeded random scenes: every primitive kind, material kind and light kind the path supports, mixed at random (including
degenerate, tiny and huge triangles, overlapping spheres, multi-triangle leaves, sphere-only and light-less scenes, thin-lens
cameras and varying MaxDepth), rendered and ray-cast on the GPU and compared with the oracle bit for bit."""
import os

import numpy as np
import pytest

import ag_pathtracer_amd as ag
from helpers import bits, gpu_scene, oracle_scene, random_rays
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def random_scene(seed):
    rng = np.random.RandomState(seed)
    d = ag.SceneDesc("fuzz-%d" % seed)
    mats = []
    for _ in range(rng.randint(1, 6)):
        kind = rng.choice([ag.MAT_DISNEY, ag.MAT_DISNEY, ag.MAT_DISNEY, ag.MAT_MIRROR, ag.MAT_DIFFUSE_ONLY])
        mats.append(d.add_material(kind, rng.uniform(0.05, 1.0, 3), float(rng.choice([0.0, 0.02, 0.3, 0.7, 1.0, rng.uniform()])),
                                   float(rng.choice([0.0, 1.0, rng.uniform()]))))
    n_mesh = rng.randint(0, 5)
    for mi in range(n_mesh):
        style = rng.randint(4)
        if style == 0:      # smooth patch with normals and uvs
            n = rng.randint(2, 14)
            v, nn, t, idx = ag.scenes.heightfield(n, S=float(rng.uniform(0.5, 2.5)))
            v = v + rng.uniform(-1, 1, 3).astype(np.float32)
        elif style == 1:    # triangle soup without normals, sizes over six orders of magnitude, some degenerate
            k = rng.randint(1, 60)
            c = rng.uniform(-2, 2, (k, 1, 3))
            size = 10.0 ** rng.uniform(-4, 1, (k, 1, 1))
            v = (c + rng.normal(size=(k, 3, 3)) * size).astype(np.float32).reshape(-1, 3)
            deg = rng.uniform(size=k) < 0.1
            v.reshape(k, 3, 3)[deg, 2] = v.reshape(k, 3, 3)[deg, 1]     # two equal vertices
            nn, t = None, None
            idx = np.stack([np.arange(3 * k, dtype=np.int32)] * 3, 1)
        elif style == 2:    # blob with shared vertices
            v, nn, t, idx = ag.scenes.blob_mesh(rng.randint(4, 12), rng.randint(3, 10), center=tuple(rng.uniform(-1.5, 1.5, 3)),
                                                radius=float(rng.uniform(0.2, 1.2)), seed=int(rng.randint(1 << 30)))
        else:               # stacked triangles: one multi-triangle leaf
            k = rng.randint(2, 14)
            c = rng.randint(-8, 8, 3).astype(np.float32) / np.float32(4)
            v = []
            for _ in range(k):
                a, b, h = (rng.randint(1, 16, 3) / np.float32(8.0)).astype(np.float32)
                v += [c + np.float32([-a, -b, -h]), c + np.float32([a, -b, h]), c + np.float32([rng.choice([-a, a]), b, 0])]
            v = np.array(v, np.float32)
            nn, t = None, None
            idx = np.stack([np.arange(3 * k, dtype=np.int32)] * 3, 1)
        d.add_mesh(v, nn, t, idx, mats[rng.randint(len(mats))], int(rng.choice([1, 1, 2, 4])))
    for _ in range(rng.randint(0, 4) + (1 if n_mesh == 0 else 0)):
        d.add_sphere(rng.uniform(-2, 2, 3), float(rng.uniform(0.1, 1.2)), mats[rng.randint(len(mats))])
    if rng.uniform() < 0.12:   # a long primitive list: top-level tree + candidate words per chunk of 64
        for i in range(rng.randint(60, 150)):
            c = rng.uniform(-3, 3, 3)
            if rng.uniform() < 0.5:
                d.add_sphere(c, float(rng.uniform(0.05, 0.3)), mats[rng.randint(len(mats))])
            else:
                k = rng.randint(1, 5)
                v = (c + rng.normal(size=(k, 3, 3)) * 0.3).astype(np.float32).reshape(-1, 3)
                d.add_mesh(v, None, None, np.stack([np.arange(3 * k, dtype=np.int32)] * 3, 1), mats[rng.randint(len(mats))], 1)
    if rng.uniform() < 0.4:
        d.add_plane([float(rng.uniform(-1, 1)), float(rng.uniform(-2.5, -1)), float(rng.uniform(-1, 1))],
                    [float(rng.uniform(1, 6)), float(rng.uniform(1, 6))], mats[rng.randint(len(mats))])
    for _ in range(rng.randint(0, 3)):
        d.add_area_light(rng.uniform(-3, 3, 3) + np.array([0, 3, 0]), float(rng.uniform(0.1, 0.8)), rng.uniform(2, 40, 3))
    if rng.uniform() < 0.5:
        d.add_uniform_infinite_light(rng.uniform(0.05, 1.0, 3))
    if rng.uniform() < 0.3:
        d.add_infinite_area_light(ag.scenes.synthetic_hdr(16, 8, seed=int(rng.randint(1 << 30))))
    eye = rng.uniform(-1, 1, 3) * 2 + np.array([0, 1, -5])
    d.set_camera(eye, rng.uniform(-0.5, 0.5, 3), [0, 1, 0], float(rng.choice([1.0, 16 / 9])), float(rng.uniform(25, 70)),
                 float(rng.choice([0.0, 0.0, 0.15])))
    return d


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGPT_FUZZ_SEEDS", "24"))))
def test_random_scene_matches_oracle(seed):
    d = random_scene(1000 + seed)
    rng = np.random.RandomState(seed)
    W, H, spp = int(rng.choice([24, 40, 64])), int(rng.choice([24, 36])), int(rng.choice([1, 3]))
    if os.environ.get("AGPT_FUZZ_BIG"):
        W, H, spp = 8 * W, 8 * H, 4
    depth = int(rng.choice([0, 1, 2, 5, 5, 8]))
    g = gpu_scene(d)
    o = oracle_scene(d, depth)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        oacc, ost = o.render(W, H, spp, seed_base=seed, rng_mode=ob.RNG_PER_SAMPLE, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    pt = ag.PathTracer(depth)
    gacc, gst = pt.render_to_host(g, W, H, spp, seed_base=seed)
    gcnt, gcst = pt.render_to_host(g, W, H, spp, seed_base=seed, counters=True)
    assert gacc.tobytes() == gcnt.tobytes()
    assert np.array_equal(gacc[..., :3].view(np.uint32), oacc[..., :3].view(np.uint32)), d.name
    assert (gst.closest_rays, gst.anyhit_rays, gst.outliers) == (ost.closest_rays, ost.anyhit_rays, ost.outliers)
    assert (gcst.interior_visits, gcst.tri_tests) == (ost.interior_visits, ost.tri_tests)
    if any(op[0] == "mesh" for op in d.ops):
        rays = random_rays(d, 4000, seed=seed)
        gh, _ = g.Intersect(rays)
        oh, _ = o.intersect(rays, any_hit=False)
        m = oh["hit"] == 1
        assert np.array_equal(gh["hit"], oh["hit"]) and np.array_equal(gh["prim"], oh["prim"]) and np.array_equal(gh["tri"], oh["tri"])
        assert np.array_equal(bits(gh["t"][m]), bits(oh["t"][m]))
        gp, _ = g.IntersectP(rays)
        op_, _ = o.intersect(rays, any_hit=True)
        assert np.array_equal(gp["hit"], op_["hit"])
    g.close()
