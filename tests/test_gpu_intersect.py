"""GPU parity, closest-hit / any-hit queries: agpt_intersect_batch (Scene::Intersect / IntersectP on the GPU)
against the CPU oracle on identical rays.  Bar: BIT-EXACT hit flag, primitive, triangle, t, b1, b2 -- the kernels do
only IEEE +,-,*,/ in the reference's operation order (no FMA contraction; Markstein-corrected division)."""
import numpy as np
import pytest

import ag_pathtracer_amd as ag
from helpers import bits, gpu_scene, oracle_scene, random_rays

pytestmark = pytest.mark.gpu


def check_closest(desc, rays):
    g = gpu_scene(desc)
    o = oracle_scene(desc)
    gh, gst = g.Intersect(rays, counters=True)   # instrumented generic kernel
    gf, _ = g.Intersect(rays)                     # production kernel
    assert gh.tobytes() == gf.tobytes()
    oh, ost = o.intersect(rays, any_hit=False)
    assert np.array_equal(gh["hit"], oh["hit"])
    assert np.array_equal(gh["prim"], oh["prim"])
    assert np.array_equal(gh["tri"], oh["tri"])
    m = oh["hit"] == 1
    assert np.array_equal(bits(gh["t"][m]), bits(oh["t"][m]))
    assert np.array_equal(bits(gh["b1"][m]), bits(oh["b1"][m]))
    assert np.array_equal(bits(gh["b2"][m]), bits(oh["b2"][m]))
    # the instrumented kernel walks exactly the nodes / triangles the reference's recursion does
    assert gst.interior_visits == ost.interior_visits
    assert gst.tri_tests == ost.tri_tests
    gp, _ = g.IntersectP(rays)
    gpc, _ = g.IntersectP(rays, counters=True)
    op_, _ = o.intersect(rays, any_hit=True)
    assert np.array_equal(gp["hit"], op_["hit"])
    assert np.array_equal(gpc["hit"], op_["hit"])
    g.close()
    return int(m.sum())


def test_c1_backdrop_sphere():
    desc = ag.scenes.scene_c1()
    hits = check_closest(desc, random_rays(desc, 200000, seed=11))
    assert hits > 20000


def test_heightfield_70k():
    desc = ag.scenes.scene_heightfield(187, disney=False)
    hits = check_closest(desc, random_rays(desc, 300000, seed=12))
    assert hits > 50000


def test_c2_blob_multi_mesh_and_light_spheres():
    desc = ag.scenes.scene_c2(n_seg=96)
    hits = check_closest(desc, random_rays(desc, 300000, seed=13))
    assert hits > 50000


def test_c3_atrium_many_meshes():
    desc = ag.scenes.scene_c3(scale=0.12)
    hits = check_closest(desc, random_rays(desc, 200000, seed=14))
    assert hits > 50000


def test_edge_meshes():
    """No normals / no uvs (default uvs (0,0),(1,0),(1,1)), a degenerate (zero-area) triangle that passes the
    Moeller-Trumbore test only by rounding, duplicate triangles (identical centroids -> 2-prim leaves), a
    single-triangle mesh (root is a leaf)."""
    d = ag.SceneDesc("edge")
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0], [2, 2, 2], [3, 3, 3], [2.5, 2.5, 2.5]], np.float32)
    tris = np.array([[0, 1, 2], [1, 3, 2], [0, 1, 2], [5, 6, 7], [0, 4, 3]], np.int32)
    idx = np.stack([tris.reshape(-1)] * 3, 1)
    d.add_mesh(v, None, None, idx, m, 1)
    d.add_mesh(v[:3] + np.float32([0, 0, -1]), None, None, np.stack([np.arange(3, dtype=np.int32)] * 3, 1), m, 1)
    d.add_sphere([0.5, 0.5, 1.5], 0.4, m)
    rays = random_rays(d, 100000, seed=15)
    # rays along the degenerate triangle's line and through shared edges
    extra = np.zeros(6, ag.RAY_DTYPE)
    extra["o"] = [[1, 1, 1], [0.5, 0.5, 3], [0.25, 0.25, 3], [4, 4, 4], [0, 0, 3], [1, 0, 3]]
    extra["d"] = [[1, 1, 1], [0, 0, -1], [0, 0, -1], [-1, -1, -1], [0, 0, -1], [0, 0, -1]]
    extra["tmax"] = 3.402823466e+38
    check_closest(d, np.concatenate([rays, extra]))


def test_empty_and_invalid_inputs():
    desc = ag.scenes.scene_c1()
    g = gpu_scene(desc)
    out, _ = g.Intersect(np.zeros(0, ag.RAY_DTYPE))
    assert out.shape == (0,)
    with pytest.raises(ag.AgptError):
        bad = ag.Scene(g.ctx)
        bad.add_mesh(np.zeros((3, 3), np.float32), None, None, np.zeros((0, 3), np.int32), -1)
    uncommitted = ag.Scene(g.ctx)
    with pytest.raises(ag.AgptError):
        uncommitted.Intersect(np.zeros(4, ag.RAY_DTYPE))
    g.close()


def test_more_than_64_primitives_go_through_the_top_level_tree():
    """> 64 primitives: k_candidates + k_trace_fast<LIST> (75 primitives = candidate words for 2 chunks of the list)."""
    rng = np.random.RandomState(5)
    d = ag.SceneDesc("many")
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    for i in range(70):
        c = rng.uniform(-4, 4, 3)
        v, n, t, idx = ag.scenes.blob_mesh(8, 7, center=tuple(c), radius=0.5, seed=i)
        d.add_mesh(v, n if i % 2 else None, t if i % 3 else None, idx, m, 1)
    for i in range(5):
        d.add_sphere(rng.uniform(-4, 4, 3), 0.4, m)
    hits = check_closest(d, random_rays(d, 150000, seed=16))
    assert hits > 20000


def test_generic_kernel_forced(monkeypatch):
    """The generic (reference-order, instrumentable) kernel and the production kernel give identical records."""
    desc = ag.scenes.scene_c3(scale=0.05)
    rays = random_rays(desc, 100000, seed=17)
    g = gpu_scene(desc)
    fast, _ = g.Intersect(rays)       # production kernel
    g.close()
    monkeypatch.setenv("AGPT_FORCE_GENERIC", "1")
    ctx = ag.Context(0)
    s = desc.instantiate(ag.Scene(ctx))
    gen, _ = s.Intersect(rays)
    s.close()
    ctx.close()
    assert fast.tobytes() == gen.tobytes()


def test_plane_primitive():
    """Plane (intersectable.h:119-157; unused by the reference's scenes): closest/any-hit records and a rendered image."""
    from test_gpu_render import compare
    d = ag.scenes.scene_c1()
    m = d.add_material(ag.MAT_DISNEY, [.8, .3, .2], .4, 0.)
    d.add_plane([0.5, -0.5, -1.0], [3.0, 2.0], m)
    rays = random_rays(d, 150000, seed=21)
    # rays parallel to the plane (D.y == 0) and rays grazing its edges
    extra = np.zeros(4, ag.RAY_DTYPE)
    extra["o"] = [[0.5, -0.5, -6], [0.5, 3, -1], [2.0, 3, -1], [0.5, -3, 0.0]]
    extra["d"] = [[0, 0, 1], [0, -1, 0], [0, -1, 0], [0, 1, 0]]
    extra["tmax"] = 3.402823466e+38
    check_closest(d, np.concatenate([rays, extra]))
    compare(d, 96, 96, 2)


def deep_mesh():
    """A pathological mesh whose binned-SAH tree (depth 39) is deeper than the 23 stack entries the production kernel keeps in LDS
    (the rest live in its HBM spill column): triangle sizes in geometric progression (x8) along one axis make every split peel
    off one triangle."""
    n = 40
    v, tris, scale = [], [], []
    x = 0.0
    for i in range(n):
        w = 8.0 ** (i - n // 2)
        v += [[x, 0, 0], [x + w, 0, 0], [x, w, 0]]
        tris.append([3 * i, 3 * i + 1, 3 * i + 2])
        scale.append(w)
        x += w * 1.5
    v = np.array(v, np.float32)
    idx = np.stack([np.array(tris, np.int32).reshape(-1)] * 3, 1)
    return v, idx, scale, x


def deep_mesh_scene():
    """deep_mesh() with a camera in front of its largest triangles, a sphere light and a sky."""
    v, idx, scale, x = deep_mesh()
    d = ag.SceneDesc("deep")
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    d.add_mesh(v, None, None, idx, m, 1)
    d.add_area_light([float(x) * 0.5, float(scale[-1]), float(scale[-1]) * 2], float(scale[-1]) * 0.2, [30., 30., 30.])
    d.add_uniform_infinite_light([.3, .3, .3])
    d.set_camera([float(x) * 0.6, float(scale[-1]) * 0.3, float(scale[-1]) * 2.5], [float(x) * 0.6, float(scale[-1]) * 0.3, 0.0], [0, 1, 0], 1.0, 50.0, 0.0)
    return d


def test_deep_bvh_spills_the_traversal_stack():
    """Closest-hit / any-hit records on deep_mesh() against the oracle, then a rendered image through the same spill path."""
    v, idx, scale, x = deep_mesh()
    n = len(scale)
    nodes, order, depth = ag.bvh_build(v, idx, 1)
    assert depth > 32
    d = ag.SceneDesc("deep")
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    d.add_mesh(v, None, None, idx, m, 1)
    rng = np.random.RandomState(3)
    rays = np.zeros(20000, ag.RAY_DTYPE)
    t = rng.randint(n, size=len(rays))
    w = np.float32(scale)[t][:, None]
    bary = rng.dirichlet([1, 1, 1], len(rays)).astype(np.float32) * np.float32(1.2) - np.float32(0.05)
    tgt = (v[3 * t] * bary[:, :1] + v[3 * t + 1] * bary[:, 1:2] + v[3 * t + 2] * bary[:, 2:]).astype(np.float32)
    rays["o"] = tgt + w * (np.float32([0, 0, 1.0]) + rng.normal(0, 0.3, (len(rays), 3)).astype(np.float32))
    rays["d"] = tgt - rays["o"]
    rays["tmax"] = 3.402823466e+38
    check_closest(d, rays)
    # the full path on the same mesh
    from test_gpu_render import compare
    compare(deep_mesh_scene(), 64, 64, 2)


def test_degenerate_scenes_and_calls():
    """Only analytic primitives (no mesh), no lights at all, zero samples, a one-triangle mesh (the BVH root is a leaf)."""
    from test_gpu_render import compare
    d = ag.SceneDesc("spheres-only")
    a = d.add_material(ag.MAT_DISNEY, [.8, .6, .2], .3, 1.)
    b = d.add_material(ag.MAT_DISNEY, [.2, .6, .8], .8, 0.)
    d.add_sphere([0, 0, 0], 1.0, a)
    d.add_sphere([0, -101, 0], 100.0, b)
    d.add_uniform_infinite_light([.7, .7, .7])
    d.set_camera([0, 1, -5], [0, 0, 0], [0, 1, 0], 1.0, 40.0, 0.0)
    compare(d, 64, 64, 2)
    nol = ag.SceneDesc("no-lights")
    m = nol.add_material(ag.MAT_DISNEY, [.8, .6, .2], .3, 0.)
    nol.add_mesh(np.float32([[-1, 0, 2], [1, 0, 2], [0, 1, 2]]), None, None, np.stack([np.arange(3, dtype=np.int32)] * 3, 1), m, 1)
    nol.set_camera([0, .3, -3], [0, .3, 0], [0, 1, 0], 1.0, 40.0, 0.0)
    compare(nol, 48, 48, 1)
    g = gpu_scene(nol)
    ctx = g.ctx
    ptr = ctx.alloc(48 * 48 * 16)
    ctx.memset(ptr, 0, 48 * 48 * 16)
    st = ag.PathTracer(5).render(g, 48, 48, 0, ptr)
    assert st.rays == 0 and not ctx.download(ptr, (48, 48, 4)).any()
    with pytest.raises(ag.AgptError):
        ag.PathTracer(5).render(g, 48, 48, 1, ptr, tile=(40, 40, 16, 16))   # tile outside the film
    ctx.free(ptr)
    g.close()


def stacked_triangles(n, centre, seed):
    """n triangles whose bounding boxes share one centre exactly: the builder cannot split them (degenerate centroid
    bounds, bvhtrimesh.h:236-247) and emits ONE leaf of n primitives whatever maxPrimsInNode says."""
    rng = np.random.RandomState(seed)
    c = np.float32(centre)
    v = []
    for i in range(n):
        a, b, h = (rng.randint(1, 16, 3) / np.float32(8.0)).astype(np.float32)
        v += [c + np.float32([-a, -b, -h]), c + np.float32([a, -b, h]), c + np.float32([rng.choice([-a, a]), b, 0])]
    v = np.array(v, np.float32)
    return v, np.stack([np.arange(3 * n, dtype=np.int32)] * 3, 1)


def test_multi_triangle_leaves_inline_and_big():
    """Leaves of 5 (inline count encoding), 20 (big-leaf side table) and a mesh whose ROOT is one 12-triangle leaf."""
    hv, hn, ht, hidx = ag.scenes.heightfield(15)
    v5, i5 = stacked_triangles(5, [4, 1, 0], 1)
    v20, i20 = stacked_triangles(20, [-4, 1, 1], 2)
    v = np.concatenate([hv, v5, v20])
    idx = np.concatenate([hidx, i5 + len(hv), i20 + len(hv) + len(v5)])
    idx[:, 1:] = 0
    nodes, order, depth = ag.bvh_build(v, idx, 1)
    counts = sorted(set(int(c) for c in nodes["count"]))
    assert 5 in counts and 20 in counts
    d = ag.SceneDesc("leaves")
    m = d.add_material(ag.MAT_DIFFUSE_ONLY, [.5, .5, .5])
    d.add_mesh(v, None, None, idx, m, 1)
    v12, i12 = stacked_triangles(12, [0, 3, 0], 3)
    d.add_mesh(v12, None, None, i12, m, 1)
    nodes12, _, _ = ag.bvh_build(v12, i12, 1)
    assert int(nodes12["count"][0]) == 12
    rays = random_rays(d, 40000, seed=9)
    # aim a third of the rays at the stacks so that the multi-triangle leaves are really entered
    tgt = np.float32([[4, 1, 0], [-4, 1, 1], [0, 3, 0]])[np.arange(len(rays) // 3) % 3]
    rays["d"][:len(tgt)] = tgt + np.random.RandomState(4).normal(0, 0.4, tgt.shape).astype(np.float32) - rays["o"][:len(tgt)]
    assert check_closest(d, rays) > 8000


def test_device_pointer_entry_point_matches_host_entry_point():
    """agpt_intersect_device (rays and hits resident in HBM) == agpt_intersect_batch (host arrays), closest and any-hit."""
    d = ag.scenes.scene_c1()
    rays = random_rays(d, 50000, seed=11)
    g = gpu_scene(d)
    ctx = g.ctx
    host_hits, _ = g.Intersect(rays)
    host_occ, _ = g.IntersectP(rays)
    rp = ctx.alloc(rays.nbytes)
    hp = ctx.alloc(len(rays) * ag.HIT_DTYPE.itemsize)
    ctx.upload(rp, rays)
    g.intersect_device(rp, len(rays), hp)
    dev_hits = ctx.download(hp, (len(rays),), ag.HIT_DTYPE)
    assert dev_hits.tobytes() == host_hits.tobytes()
    st = g.intersect_device(rp, len(rays), hp, any_hit=True, counters=True)
    dev_occ = ctx.download(hp, (len(rays),), ag.HIT_DTYPE)
    assert np.array_equal(dev_occ["hit"], host_occ["hit"]) and st.anyhit_rays == len(rays)
    with pytest.raises(ag.AgptError):
        g.intersect_device(0, 4, hp)
    ctx.free(rp)
    ctx.free(hp)
    g.close()


def test_render_with_200_primitives_lights_in_different_chunks():
    """Four chunks of the primitive list with meshes, spheres, two area lights (one in the first chunk, one in the last) and a
    sky: closest hits, any-hit and MIS queries whose candidates span chunk boundaries, against the oracle."""
    from test_gpu_render import compare
    rng = np.random.RandomState(8)
    d = ag.SceneDesc("200-prims")
    mats = [d.add_material(ag.MAT_DISNEY, rng.uniform(.2, .9, 3), float(rng.uniform(.1, 1)), float(rng.choice([0., 1.]))) for _ in range(6)]
    d.add_area_light([0.0, 6.0, 0.0], 0.7, [60, 55, 50])
    for i in range(197):
        c = rng.uniform(-5, 5, 3) * np.array([1, 0.5, 1])
        if i % 4 == 3:
            d.add_sphere(c, float(rng.uniform(0.15, 0.5)), mats[i % 6])
        else:
            v, n, t, idx = ag.scenes.blob_mesh(6, 5, center=tuple(c), radius=float(rng.uniform(0.2, 0.6)), seed=i)
            d.add_mesh(v, n if i % 2 else None, t, idx, mats[i % 6], 1)
    d.add_plane([0, -3.0, 0], [8, 8], mats[0])
    d.add_area_light([3.0, 5.0, -2.0], 0.5, [40, 45, 60])
    d.add_uniform_infinite_light([.3, .35, .4])
    d.set_camera([0, 2, -12], [0, 0, 0], [0, 1, 0], 1.5, 50.0, 0.0)
    assert d.n_prims == 200
    compare(d, 96, 64, 2)
    rays = random_rays(d, 60000, seed=3)
    assert check_closest(d, rays) > 10000
