"""GPU parity, full path: agpt_render (MyApp::Tick's per-pixel loop -> PathTracer::Li on the GPU) against the CPU
oracle on identical per-(pixel, sample) RNG streams.

Tolerance (fp32 radiance): the kernels evaluate the same IEEE expression tree as the oracle; the only
implementation-defined ingredient is sin/cos/acos, which both sides define as the correctly rounded fp32 value
(oracle trig mode 1; device: fp64 evaluation rounded once).  fp64 libm's differ by <1 ulp(fp64), so a rounded fp32
result can differ in ~1e-8 of the calls.  Bar used below, per pixel of the accumulated image:
    >= 99.9 % of pixels bit-identical, every pixel within rel 1e-4 (+abs 1e-6) except <= 1e-4 of the pixels
    (a 1-ulp direction change that flips a hit/miss edge decides a whole path), image mean within rel 1e-4,
and exactly equal ray counts (closest + any-hit) up to that same 1e-4 fraction.
"""
import numpy as np
import pytest

import ag_pathtracer_amd as ag
from helpers import gpu_scene, oracle_scene
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def compare(desc, W, H, spp, max_depth=5, tile=None, seed_base=0, spb=0):
    g = gpu_scene(desc)
    o = oracle_scene(desc, max_depth)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        oacc, ost = o.render(W, H, spp, tile=tile, seed_base=seed_base, rng_mode=ob.RNG_PER_SAMPLE, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    gacc, gst = ag.PathTracer(max_depth).render_to_host(g, W, H, spp, tile=tile, seed_base=seed_base,
                                                        samples_per_batch=spb, counters=True)   # instrumented kernels
    gfast, gfst = ag.PathTracer(max_depth).render_to_host(g, W, H, spp, tile=tile, seed_base=seed_base,
                                                          samples_per_batch=spb)                # production kernels
    g.close()
    assert gfast.tobytes() == gacc.tobytes()
    assert gfst.rays == gst.rays
    a, b = gacc[..., :3].reshape(-1, 3), oacc[..., :3].reshape(-1, 3)
    same = np.all(a.view(np.uint32) == b.view(np.uint32), axis=1)
    close = np.all(np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-6, axis=1)
    n = a.shape[0]
    info = dict(bit_exact=float(same.mean()), close=float(close.mean()), n=n,
                mean_rel=float(np.max(np.abs(a.mean(0, dtype=np.float64) - b.mean(0, dtype=np.float64)) /
                                      np.maximum(b.mean(0, dtype=np.float64), 1e-12))),
                rays=(gst.rays, ost.rays), closest=(gst.closest_rays, ost.closest_rays),
                anyhit=(gst.anyhit_rays, ost.anyhit_rays), outliers=(gst.outliers, ost.outliers),
                interior=(gst.interior_visits, ost.interior_visits), tris=(gst.tri_tests, ost.tri_tests))
    print(desc.name, info)
    assert info["bit_exact"] == 1.0, info
    assert info["close"] >= 1 - 1e-4, info
    assert info["mean_rel"] <= 1e-4, info
    assert gst.rays == ost.rays, info
    assert gst.outliers == ost.outliers
    return info


def test_c1_256x256_1spp():
    """BASELINE config 1 (Cornell-box-scale: backdrop + gold microfacet sphere + sphere light + sky)."""
    compare(ag.scenes.scene_c1(), 256, 256, 1)


def test_c1_multi_spp_batches_and_tile():
    d = ag.scenes.scene_c1()
    compare(d, 96, 64, 5, spb=2)                      # 3 batches (2+2+1): accumulation order across batches
    compare(d, 128, 96, 2, tile=(32, 16, 64, 48))     # a tile of a larger film: global pixel seeds


def test_c2_blob_diffuse_only():
    """BASELINE config 2 geometry class at reduced size (single DisneyDiffuse lobe, three sphere lights + sky)."""
    compare(ag.scenes.scene_c2(n_seg=64), 160, 90, 2)


def test_c3_atrium_disney_nee_mis():
    """BASELINE config 3 geometry class at reduced size (closed atrium, ~30 Disney materials incl. metals)."""
    compare(ag.scenes.scene_c3(scale=0.05), 160, 90, 2)


def test_heightfield_disney():
    compare(ag.scenes.scene_heightfield(64, disney=True), 160, 90, 2)


def test_lens_and_mirror_and_depth():
    """Thin-lens camera (rejection-sampled disk: variable RNG draws), a perfectly specular mirror sphere
    (specularBounce path, no NEE) and non-default MaxDepth."""
    d = ag.scenes.scene_c1()
    d.add_material(ag.MAT_MIRROR, [.9, .9, .9])
    d.add_sphere([2.2, 0.0, 0.5], 1.0, 2)
    d.set_camera([-1.46, 1.16, -4.64], [0, 0, 0], [0, 1, 0], 1.0, 45.0, 0.1)
    compare(d, 96, 96, 2, max_depth=3)
    compare(d, 64, 64, 1, max_depth=0)


def test_answered_rays_are_counted_not_traced():
    """The production path settles two kinds of the reference's Scene::Intersect calls without a traversal -- MIS queries whose ray
    misses the sampled sphere light (integrator.h:76-88) and the ray after a path's last bounce, whose hit nothing reads
    (integrator.h:139-150) -- and still counts them: image and ray totals equal the oracle's, which traces every one of them; the
    reference-order instrumented run traces them too (answered_rays == 0) and gives the same image."""
    desc = ag.scenes.scene_c1()
    W, H, spp = 96, 64, 2
    g = gpu_scene(desc)
    o = oracle_scene(desc, 5)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        oacc, ost = o.render(W, H, spp, rng_mode=ob.RNG_PER_SAMPLE, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    pt = ag.PathTracer(5)
    fast, fst = pt.render_to_host(g, W, H, spp)
    inst, ist = pt.render_to_host(g, W, H, spp, counters=True)
    own, cst = pt.render_to_host(g, W, H, spp, counters=2)
    g.close()
    assert fast.tobytes() == inst.tobytes() == own.tobytes()
    assert np.array_equal(fast[..., :3].view(np.uint32), oacc[..., :3].view(np.uint32))
    assert (fst.closest_rays, fst.anyhit_rays) == (ost.closest_rays, ost.anyhit_rays) == (ist.closest_rays, ist.anyhit_rays)
    assert ist.answered_rays == 0
    assert 0 < fst.answered_rays < fst.closest_rays and cst.answered_rays == fst.answered_rays
    # the work the production kernel counts for itself is less than the reference's recursion does: the difference is those rays
    assert cst.interior_visits < ist.interior_visits


def test_trace_all_rays_gives_the_same_image_and_totals():
    """agpt_render_params::trace_all_rays: every ray query of the reference goes through the BVH; image, ray totals and outliers
    equal the default run's, answered_rays is 0."""
    desc = ag.scenes.scene_c1()
    W, H, spp = 80, 60, 2
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    fast, fst = pt.render_to_host(g, W, H, spp)
    full, ast = pt.render_to_host(g, W, H, spp, trace_all_rays=True)
    g.close()
    assert fast.tobytes() == full.tobytes()
    assert (fst.closest_rays, fst.anyhit_rays, fst.outliers) == (ast.closest_rays, ast.anyhit_rays, ast.outliers)
    assert ast.answered_rays == 0 and fst.answered_rays > 0


# ---- the path order of the timed configuration ----------------------------------------------------------------------------
# A batch orders its paths as ((s / G) * NP + p) * G + s % G with G = sample_group(S) = the largest power of two <= 64 dividing
# the batch's sample count, and k_accumulate adds a pixel's samples line-wise (4 per 64-B line) when G % 4 == 0.  bench.py
# runs S = 64 -> G = 64: the cases below hold that order -- and the mixed ones a batch split produces -- against the oracle's
# loop (myapp.cpp:163-175 over samples, AddSample order myapp.h:17-19) on films that take the 8x8-block pixel order (64x64) and
# the row-major one (80x60).
@pytest.mark.parametrize("W,H", [(64, 64), (80, 60)])
@pytest.mark.parametrize("spp,spb", [(4, 0), (12, 0), (64, 0), (64, 48), (64, 16), (192, 64), (192, 128), (70, 0)])
def test_sample_bundles_match_oracle(W, H, spp, spb):
    """G = 4 (spp 4, 12), 64 (spp 64), 16 + 16 (48 + 16 split), 16 (x4), 64 (x3), 64 + 64 (128 + 64), 2 (spp 70): pixels
    bit-identical to the oracle and equal ray totals."""
    compare(ag.scenes.scene_c1(), W, H, spp, spb=spb)


def test_sample_bundle_splits_agree():
    """64 spp in one batch (one wave per pixel's 64 samples) == 64 x 1 spp accumulated progressively == a two-rank interleave
    of 8-row blocks at 64 spp, gathered -- bit for bit, on both pixel orders."""
    from ag_pathtracer_amd import tiles
    desc = ag.scenes.scene_c1()
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    ctx = g.ctx
    for (W, H) in ((64, 64), (80, 60)):
        full, st = pt.render_to_host(g, W, H, 64)
        ptr = ctx.alloc(W * H * 16)
        ctx.memset(ptr, 0, W * H * 16)
        rays = 0
        for s in range(64):
            rays += pt.render(g, W, H, 1, ptr, spp_begin=s).rays
        assert ctx.download(ptr, (H, W, 4)).tobytes() == full.tobytes() and rays == st.rays
        bufs = []
        for r in range(2):
            ctx.memset(ptr, 0, W * H * 16)
            pt.render(g, W, H, 64, ptr, interleave=(tiles.BLOCK_ROWS, 2, r))
            bufs.append(ctx.download(ptr, (H, W, 4))[:tiles.max_local_rows(H, 2)].copy())
        assert tiles.deinterleave(bufs, W, H, 2).tobytes() == full.tobytes()
        ctx.free(ptr)
    g.close()


def emitter_gauntlet():
    """C1's geometry with six emitter spheres of different sizes hovering just above the floor around the gold sphere, two of them
    overlapping: bounce rays cross one, two or three emitters (entering and leaving each is a re-cast of its own,
    integrator.h:152-161), some start inside one."""
    d = ag.scenes.scene_c1()
    d.name = "emitter-gauntlet"
    for c, r in (([1.6, -0.2, 0.4], 0.7), ([2.2, -0.1, 0.9], 0.6), ([-1.8, -0.3, 0.8], 0.6), ([0.2, -0.4, -1.9], 0.5),
                 ([0.0, 1.9, 0.3], 0.5), ([-0.6, -0.5, 2.4], 0.45)):
        d.add_area_light(c, r, ag.scenes.KEY_LIGHT * np.float32(3))
    return d


def test_emitter_pass_through_recast_in_the_trace_kernel(monkeypatch):
    """A ray whose closest hit is a primitive without a material is re-cast by k_trace_fast itself where the next vertex would do
    nothing else (ext_d.w = 1): same image, same ray counts as the oracle and as the flow that sends the path through k_shade
    for it (AGPT_NO_RECAST=1), in fewer wavefront iterations."""
    d = emitter_gauntlet()
    compare(d, 96, 96, 4)                       # instrumented kernels (k_shade re-casts) == production kernels == oracle
    g = gpu_scene(d)
    pt = ag.PathTracer(5)
    a, sa = pt.render_to_host(g, 160, 120, 8)
    monkeypatch.setenv("AGPT_NO_RECAST", "1")
    b, sb = pt.render_to_host(g, 160, 120, 8)
    monkeypatch.delenv("AGPT_NO_RECAST")
    g.close()
    assert a.tobytes() == b.tobytes()
    assert (sa.rays, sa.closest_rays, sa.anyhit_rays, sa.shaded_vertices) == (sb.rays, sb.closest_rays, sb.anyhit_rays, sb.shaded_vertices)
    print("iterations with the in-kernel re-cast %d, without %d" % (sa.iterations, sb.iterations))
    assert sa.iterations < sb.iterations
    o = oracle_scene(d, 5)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        oacc, ost = o.render(160, 120, 8, rng_mode=ob.RNG_PER_SAMPLE, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    assert np.array_equal(a[..., :3].view(np.uint32), oacc[..., :3].view(np.uint32))
    assert sa.rays == ost.rays


def test_side_streams_for_the_mis_and_shadow_launches(monkeypatch):
    """AGPT_MULTI_STREAM=2 (opt-in, read at agpt_init): the MIS and shadow launches of an iteration run on streams of their own
    beside the closest-hit launch.  Same image, same counts -- on a scene that qualifies (short list, shallow BVH) and on one
    that must fall back to one stream (the launches would share the HBM stack spill)."""
    from test_gpu_intersect import deep_mesh_scene
    for d, size in ((emitter_gauntlet(), (128, 96)), (ag.scenes.scene_c1(), (96, 96)), (deep_mesh_scene(), (64, 48))):
        g = gpu_scene(d)
        a, sa = ag.PathTracer(5).render_to_host(g, size[0], size[1], 4)
        g.close()
        monkeypatch.setenv("AGPT_MULTI_STREAM", "2")
        ctx = ag.Context(0)
        monkeypatch.delenv("AGPT_MULTI_STREAM")
        g2 = d.instantiate(ag.Scene(ctx))
        b, sb = ag.PathTracer(5).render_to_host(g2, size[0], size[1], 4)
        g2.close()
        ctx.close()
        assert a.tobytes() == b.tobytes(), d.name
        assert (sa.rays, sa.shaded_vertices, sa.iterations) == (sb.rays, sb.shaded_vertices, sb.iterations), d.name
