"""Long primitive lists (Scene::primitives, scene.h:5-19): for 65 ... 2048 primitives k_candidates walks a
top-level tree over the meshes' root boxes and k_trace_fast<LIST> visits each ray's candidates in list order.  The tree only
selects CANDIDATES, so parity = the hit records of the reference's linear walk (oracle), ties between primitives resolved in
list order.  (The host-side tree builder is tested on the CPU in test_toplevel.py.)"""
import numpy as np
import pytest

import ag_pathtracer_amd as ag


def _many_prims(n_prims, seed, duplicates=True, sphere_every=9):
    rng = np.random.RandomState(seed)
    d = ag.SceneDesc("toplevel-%d" % n_prims)
    mats = [d.add_material(ag.MAT_DIFFUSE_ONLY, rng.uniform(.2, .9, 3)) for _ in range(3)]
    blobs = []
    for i in range(n_prims):
        c = rng.uniform(-6, 6, 3)
        if i % sphere_every == 4:
            d.add_sphere(c, float(rng.uniform(0.1, 0.4)), mats[i % 3])
        elif duplicates and i % 50 == 49 and blobs:
            # the same geometry again, hundreds of list positions (several chunks) later: every ray that hits it ties
            v, n, t, idx = blobs[rng.randint(len(blobs))]
            d.add_mesh(v, n, t, idx, mats[i % 3], 1)
        else:
            v, n, t, idx = ag.scenes.blob_mesh(5, 4, center=tuple(c), radius=float(rng.uniform(0.15, 0.5)), seed=i)
            blobs.append((v, n, t, idx))
            d.add_mesh(v, n, t, idx, mats[i % 3], 1)
    return d


@pytest.mark.gpu
def test_gpu_900_primitives_hit_records_match_the_list_walk():
    """Production kernel == instrumented reference-order kernel == oracle, bit for bit, closest hit and any hit, with
    coincident meshes hundreds of list positions apart."""
    from test_gpu_intersect import check_closest, random_rays
    d = _many_prims(900, 21)
    assert d.n_prims == 900
    hits = check_closest(d, random_rays(d, 120000, seed=5))
    assert hits > 15000


@pytest.mark.gpu
def test_gpu_2000_primitives_chunks_beyond_the_sixteenth_and_the_fallback_beyond_2048():
    """2,000 primitives = 32 chunks of 64 (the chunks-to-visit word of a ray is 32 bits wide): production == reference-order
    kernel == oracle, with coincident meshes over a thousand list positions apart.  2,100 primitives are past the top-level
    structure's limit: the reference-order kernel takes over, same answers."""
    from test_gpu_intersect import check_closest, random_rays
    d = _many_prims(2000, 77)
    assert d.n_prims == 2000
    assert check_closest(d, random_rays(d, 60000, seed=6)) > 10000
    d = _many_prims(2100, 78, duplicates=False)
    assert check_closest(d, random_rays(d, 20000, seed=7)) > 3000


@pytest.mark.gpu
def test_gpu_render_1500_primitives_against_oracle():
    from test_gpu_render import compare
    d = _many_prims(1498, 24)
    d.add_area_light([0.0, 8.0, 0.0], 0.8, [70, 65, 60])
    d.add_plane([0, -6.5, 0], [10, 10], 0)
    d.add_uniform_infinite_light([.25, .3, .35])
    d.set_camera([0, 3, -16], [0, 0, 0], [0, 1, 0], 1.5, 50.0, 0.0)
    assert d.n_prims == 1500
    compare(d, 48, 36, 2)


@pytest.mark.gpu
def test_gpu_render_700_primitives_against_oracle():
    from test_gpu_render import compare
    d = _many_prims(698, 22)
    d.add_area_light([0.0, 8.0, 0.0], 0.8, [70, 65, 60])
    d.add_plane([0, -6.5, 0], [10, 10], 0)
    d.add_uniform_infinite_light([.25, .3, .35])
    d.set_camera([0, 3, -16], [0, 0, 0], [0, 1, 0], 1.5, 50.0, 0.0)
    assert d.n_prims == 700
    compare(d, 64, 48, 2)


@pytest.mark.gpu
def test_gpu_ray_with_more_candidates_than_the_list_holds():
    """120 meshes around the same spot: most rays cross far more than the 24 root boxes a k_candidates thread records; such a ray
    gets every primitive as candidate (any superset is valid) and the answers stay those of the list walk."""
    from test_gpu_intersect import check_closest, random_rays
    rng = np.random.RandomState(3)
    d = ag.SceneDesc("crowded")
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    for i in range(120):
        v, n, t, idx = ag.scenes.blob_mesh(5, 4, center=tuple(rng.uniform(-0.3, 0.3, 3)), radius=float(rng.uniform(0.5, 2.0)), seed=i)
        d.add_mesh(v, n, t, idx, m, 1)
    for i in range(6):
        d.add_sphere(rng.uniform(-2, 2, 3), 0.3, m)
    assert check_closest(d, random_rays(d, 40000, seed=9)) > 20000


@pytest.mark.gpu
def test_gpu_axis_parallel_rays_on_a_long_list():
    """Direction components of exactly 0 (the reference relies on inf / NaN propagation in its slab test): the conservative form
    of the candidate walk does not hold for them, they take every primitive as candidate."""
    from test_gpu_intersect import check_closest
    d = _many_prims(300, 41)
    rng = np.random.RandomState(2)
    n = 30000
    rays = np.zeros(n, ag.RAY_DTYPE)
    rays["o"] = rng.uniform(-7, 7, (n, 3)).astype(np.float32)
    dirs = np.zeros((n, 3), np.float32)
    axis = rng.randint(0, 3, n)
    dirs[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
    two = rng.uniform(size=n) < 0.5   # half of them lie in a coordinate plane instead of on an axis
    dirs[two, (axis[two] + 1) % 3] = rng.normal(size=two.sum()).astype(np.float32)
    rays["d"] = dirs
    rays["tmax"] = np.float32(3.4e38)
    assert check_closest(d, rays) > 1000


@pytest.mark.gpu
def test_gpu_long_list_render_is_split_invariant():
    """300 primitives: the film is the same bit for bit whether it is rendered in one batch, in batches of one sample (the path
    pool and the candidate words are re-used), with the production kernel counting its own work, or as three ranks' interleaved
    row blocks reassembled -- and the work counters of the counting run are plausible."""
    from helpers import gpu_scene
    from ag_pathtracer_amd import tiles
    d = _many_prims(297, 51)
    d.add_area_light([0.0, 8.0, 0.0], 0.8, [70, 65, 60])
    d.add_area_light([4.0, 7.0, -3.0], 0.5, [30, 40, 60])
    d.add_plane([0, -6.5, 0], [10, 10], 0)
    d.add_uniform_infinite_light([.25, .3, .35])
    d.set_camera([0, 3, -16], [0, 0, 0], [0, 1, 0], 1.5, 50.0, 0.0)
    assert d.n_prims == 300
    W, H, spp = 96, 72, 3
    g = gpu_scene(d)
    pt = ag.PathTracer(5)
    whole, st = pt.render_to_host(g, W, H, spp)
    batched, _ = pt.render_to_host(g, W, H, spp, samples_per_batch=1)
    counted, cst = pt.render_to_host(g, W, H, spp, counters=2)
    assert whole.tobytes() == batched.tobytes() == counted.tobytes()
    assert cst.rays == st.rays and cst.interior_visits > 5 * cst.closest_rays and cst.root_tests >= cst.closest_rays // 4
    world = 3
    rows_max = tiles.max_local_rows(H, world)
    ctx = g.ctx
    gathered = []
    for rank in range(world):
        local = ctx.alloc(rows_max * W * 16)
        try:
            ctx.memset(local, 0, rows_max * W * 16)
            pt.render(g, W, H, spp, local, accum_pitch=W, interleave=(tiles.BLOCK_ROWS, world, rank))
            gathered.append(ctx.download(local, (rows_max, W, 4)))
        finally:
            ctx.free(local)
    full = tiles.deinterleave(gathered, W, H, world)
    g.close()
    assert np.array_equal(full.view(np.uint32), whole.view(np.uint32))


@pytest.mark.gpu
def test_gpu_li_batch_on_a_long_list():
    """agpt_li_batch (single-ray Integrator::Li) through the top-level-tree path: 300 primitives, area light + sky; radiance and the
    RNG state after each path equal the oracle's PathTracer::Li."""
    from helpers import gpu_scene, oracle_scene
    from oracle import binding as ob
    d = _many_prims(298, 23, duplicates=False)
    d.add_area_light([0.0, 8.0, 0.0], 0.8, [70, 65, 60])
    d.add_uniform_infinite_light([.25, .3, .35])
    d.set_camera([0, 3, -16], [0, 0, 0], [0, 1, 0], 1.5, 50.0, 0.0)
    assert d.n_prims == 299
    o = oracle_scene(d, 5)
    g = gpu_scene(d)
    rng = np.random.RandomState(11)
    n = 800
    rays = np.zeros(n, ag.RAY_DTYPE)
    states = np.zeros(n, np.uint32)
    for i in range(n):
        r, st = o.camera_ray(float(rng.uniform()), float(rng.uniform()), rng=int(rng.randint(1, 2 ** 31 - 1)))
        rays[i] = r
        states[i] = st
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        want = np.zeros((n, 3), np.float32)
        after = np.zeros(n, np.uint32)
        for i in range(n):
            L, s_after, _ = o.li(rays[i], int(states[i]))
            want[i] = L
            after[i] = s_after
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    got, got_after, _ = ag.PathTracer(5).Li(g, rays, states)
    g.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got_after, after)
