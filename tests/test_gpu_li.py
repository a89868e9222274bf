"""Single-ray Integrator::Li (integrator.h:28-31, PathTracer::Li :120-191) through agpt_li_batch: the entry the reference's
split-screen compare (myapp.cpp:168) and mouse picking (myapp.cpp:197-201) call, against the oracle's oracle_li on the same
rays and the same RandomFloat() streams -- radiance bit for bit, the stream's state after the path, and the ray totals."""
import numpy as np
import pytest

import ag_pathtracer_amd as ag
from helpers import gpu_scene, oracle_scene
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def camera_rays(o, n, seed):
    """n camera rays (Camera::GetRay through the oracle) with one stream each; the stream state left by GetRay starts Li."""
    rng = np.random.RandomState(seed)
    rays = np.zeros(n, ag.RAY_DTYPE)
    states = np.zeros(n, np.uint32)
    for i in range(n):
        r, st = o.camera_ray(float(rng.uniform()), float(rng.uniform()), rng=int(rng.randint(1, 2 ** 31 - 1)))
        rays[i] = r
        states[i] = st
    return rays, states


@pytest.mark.parametrize("which,depth", [("c1", 5), ("c1", 0), ("c3", 5), ("lens_mirror", 3)])
def test_li_batch_matches_oracle(which, depth):
    if which == "c3":
        desc = ag.scenes.scene_c3(scale=0.05)
    else:
        desc = ag.scenes.scene_c1()
        if which == "lens_mirror":
            desc.add_material(ag.MAT_MIRROR, [.9, .9, .9])
            desc.add_sphere([2.2, 0.0, 0.5], 1.0, 2)
            desc.set_camera([-1.46, 1.16, -4.64], [0, 0, 0], [0, 1, 0], 1.0, 45.0, 0.1)
    o = oracle_scene(desc, depth)
    g = gpu_scene(desc)
    n = 3000
    rays, states = camera_rays(o, n, seed=7)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        want = np.zeros((n, 3), np.float32)
        after = np.zeros(n, np.uint32)
        total = 0
        for i in range(n):
            L, s_after, st = o.li(rays[i], int(states[i]))
            want[i] = L
            after[i] = s_after
            total += st.rays
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    got, got_after, st = ag.PathTracer(depth).Li(g, rays, states)
    g.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got_after, after)
    assert st.rays == total and st.samples == n


def test_li_batch_degenerate_calls():
    g = gpu_scene(ag.scenes.scene_c1())
    pt = ag.PathTracer(5)
    out, after, st = pt.Li(g, np.zeros(0, ag.RAY_DTYPE), np.zeros(0, np.uint32))
    assert out.shape == (0, 3) and st.rays == 0
    # a ray that leaves the scene: Li = the infinite lights' Le (integrator.h:144-145), no RandomFloat() drawn
    r = np.zeros(1, ag.RAY_DTYPE)
    r["o"] = [0, 50, 0]
    r["d"] = [0, 1, 0]
    r["tmax"] = 3.402823466e+38
    out, after, st = pt.Li(g, r, np.uint32([12345]))
    assert np.allclose(out[0], [.4, .45, .5]) and after[0] == 12345 and st.rays == 1
    g.close()


@pytest.mark.gpu
def test_dbg_integrator_uv_view_matches_the_oracle():
    """DbgIntegrator::Li (integrator.h:107-118): the hit's uv as a colour -- meshes with texture coordinates (the backdrop's own),
    without (the (0,0) (1,0) (1,1) default of trianglemesh.cpp:52-56), spheres (atan2 / acos), a plane, emitters, misses."""
    d = ag.scenes.scene_c1()
    v, n, t, idx = ag.scenes.blob_mesh(6, 5, center=(2.5, 0.5, 1.0), radius=0.8, seed=4)
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    d.add_mesh(v, n, None, idx, m, 1)            # no texture coordinates
    d.add_plane([0, -0.5, -3], [3, 2], m)
    from helpers import random_rays
    rays = random_rays(d, 30000, seed=12)
    g = gpu_scene(d)
    got = ag.PathTracer.DbgLi(g, rays)
    g.close()
    o = oracle_scene(d)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        want = o.dbg_li(rays)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    lit = got.any(axis=1)
    assert 3000 < lit.sum() < len(rays) - 1000            # hits and misses
    assert (got[lit][:, 1] > 0).sum() > 2000               # (u, v, 0) / 5 colours, not only the red marker
