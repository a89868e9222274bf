"""BASELINE.json's full-size configurations on the GPU.  The oracle cannot render 1080p/64spp in test time, so at full
size parity is carried by (a) oracle comparison on a small tile of the full film and scene (the RNG streams are global
per (pixel, sample), so a tile of the big render is exactly what the oracle computes for that tile), and
(b) size-independent properties: tile-split, sample-split and rank-split invariance (all bit-exact), determinism."""
import numpy as np
import pytest

import ag_pathtracer_amd as ag
from ag_pathtracer_amd import tiles
from helpers import gpu_scene, oracle_scene
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def oracle_tile(desc, W, H, spp, tile):
    o = oracle_scene(desc)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        return o.render(W, H, spp, tile=tile, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)


def tile_rows(acc, H, tile):
    x0, y0, w, h = tile
    return acc[H - y0 - h:H - y0, x0:x0 + w, :3]


def test_c3_full_scene_1080p_tile_vs_oracle_and_invariances():
    W, H, spp = 1920, 1080, 2
    desc = ag.scenes.scene_c3()
    assert 254000 <= desc.n_tris <= 266000 and desc.n_prims == 37
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    full, st = pt.render_to_host(g, W, H, spp)
    assert st.samples == W * H * spp and st.outliers == 0
    # (a) a 96x54 tile of the full film against the oracle
    tile = (912, 500, 96, 54)
    oacc, ost = oracle_tile(desc, W, H, spp, tile)
    a, b = tile_rows(full, H, tile), tile_rows(oacc, H, tile)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # (b1) determinism
    again, st2 = pt.render_to_host(g, W, H, spp)
    assert again.tobytes() == full.tobytes() and st2.rays == st.rays
    # (b2) sample split: 1 spp + 1 spp accumulated progressively == 2 spp; small batches == large batches
    ctx = g.ctx
    ptr = ctx.alloc(W * H * 16)
    ctx.memset(ptr, 0, W * H * 16)
    pt.render(g, W, H, 1, ptr, spp_begin=0)
    pt.render(g, W, H, 1, ptr, spp_begin=1, samples_per_batch=1)
    split = ctx.download(ptr, (H, W, 4))
    assert split.tobytes() == full.tobytes()
    # (b3) tile split == rank split: the 8-row-block interleave of a 3-rank run, gathered, equals the full render
    world = 3
    bufs = []
    for r in range(world):
        rows = tiles.max_local_rows(H, world)
        ctx.memset(ptr, 0, W * H * 16)
        for (y0, h, off) in tiles.row_blocks(H, r, world):
            pt.render(g, W, H, spp, ptr, tile=(0, y0, W, h), accum_row0=tiles.block_accum_row0(H, y0, h, off), want_stats=False)
        bufs.append(ctx.download(ptr, (H, W, 4))[:rows].copy())
    assert tiles.deinterleave(bufs, W, H, world).tobytes() == full.tobytes()
    # (b4) the same split through the native row interleave of agpt_render (one call per rank, what bench.py uses)
    for world in (2, 8):
        bufs = []
        for r in range(world):
            rows = tiles.max_local_rows(H, world)
            ctx.memset(ptr, 0, W * H * 16)
            st_r = pt.render(g, W, H, spp, ptr, interleave=(tiles.BLOCK_ROWS, world, r))
            assert st_r.samples == W * tiles.local_rows(H, r, world) * spp
            bufs.append(ctx.download(ptr, (H, W, 4))[:rows].copy())
        assert tiles.deinterleave(bufs, W, H, world).tobytes() == full.tobytes()
    # (b5) the production path answers ~30 % of the reference's ray queries without tracing them: with every query
    #      traced the film and the ray totals are the same, bit for bit
    traced, st_all = pt.render_to_host(g, W, H, spp, trace_all_rays=True)
    assert traced.tobytes() == full.tobytes()
    assert (st_all.closest_rays, st_all.anyhit_rays) == (st.closest_rays, st.anyhit_rays)
    assert st_all.answered_rays == 0 and 0.2 * st.rays < st.answered_rays < 0.4 * st.rays
    ctx.free(ptr)
    g.close()


def test_c2_full_size_tile_vs_oracle():
    """BASELINE config 2: 70,010-triangle blob + backdrop, single DisneyDiffuse lobe, 1280x720 (16 spp in the config;
    2 spp here), three sphere lights + sky."""
    W, H, spp = 1280, 720, 2
    desc = ag.scenes.scene_c2()
    assert 69000 <= desc.n_tris <= 71000
    g = gpu_scene(desc)
    full, st = ag.PathTracer(5).render_to_host(g, W, H, spp)
    g.close()
    tile = (600, 330, 96, 54)
    oacc, ost = oracle_tile(desc, W, H, spp, tile)
    a, b = tile_rows(full, H, tile), tile_rows(oacc, H, tile)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert st.outliers == 0


def test_c5_class_all_metal_multi_mesh():
    """BASELINE config 5's material class (all-metal GGX + MIS, many meshes) at reduced geometry size."""
    W, H, spp = 192, 108, 2
    desc = ag.scenes.scene_c3(scale=0.08, all_metal=True)
    g = gpu_scene(desc)
    acc, st = ag.PathTracer(5).render_to_host(g, W, H, spp)
    g.close()
    oacc, ost = oracle_tile(desc, W, H, spp, (0, 0, W, H))
    same = np.all(acc[..., :3].view(np.uint32) == oacc[..., :3].view(np.uint32), axis=-1)
    assert same.all()
    assert st.rays == ost.rays


def gather_ranks(pt, g, W, H, spp, world):
    ctx = g.ctx
    ptr = ctx.alloc(W * H * 16)
    bufs = []
    for r in range(world):
        rows = tiles.max_local_rows(H, world)
        ctx.memset(ptr, 0, W * H * 16)
        pt.render(g, W, H, spp, ptr, interleave=(tiles.BLOCK_ROWS, world, r))
        bufs.append(ctx.download(ptr, (H, W, 4))[:rows].copy())
    ctx.free(ptr)
    return tiles.deinterleave(bufs, W, H, world)


def test_c4_same_scene_4k_split_over_8_ranks():
    """BASELINE config 4: the C3 scene at 3840x2160 (256 spp in the config; 1 spp here), framebuffer split over 8 ranks.
    The 8 rank shares, gathered, are the single-GPU film bit for bit; two oracle tiles pin the film itself."""
    W, H, spp = 3840, 2160, 1
    desc = ag.scenes.scene_c3()
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    full, st = pt.render_to_host(g, W, H, spp)
    assert st.samples == W * H * spp and st.outliers == 0
    assert gather_ranks(pt, g, W, H, spp, 8).tobytes() == full.tobytes()
    g.close()
    for tile in ((1800, 1000, 96, 54), (300, 1700, 64, 64)):
        oacc, _ = oracle_tile(desc, W, H, spp, tile)
        a, b = tile_rows(full, H, tile), tile_rows(oacc, H, tile)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_c5_five_million_triangles_17_meshes():
    """BASELINE config 5: ~5 M triangles in 17 meshes (16 perturbed colonnade copies + a terrain), all-metal GGX + MIS,
    3840x2160 (1024 spp on 8 GPUs in the config; 1 spp here).  Same BVHs as the oracle's builder, same hits and
    node / triangle visit counts on random rays, oracle tiles of the 4K film bit-exact, 8-rank split == full film."""
    W, H, spp = 3840, 2160, 1
    desc = ag.scenes.scene_c5()
    assert 4.9e6 <= desc.n_tris <= 5.1e6 and desc.n_prims == 21
    g = gpu_scene(desc)
    o = oracle_scene(desc)
    for prim in (0, 1, 16):
        gn, go = g.bvh(prim)
        on, oo = o.bvh(prim)
        assert gn.tobytes() == on.tobytes() and np.array_equal(go, oo)
    from helpers import random_rays, bits
    rays = random_rays(desc, 200000, seed=5)
    gh, gst = g.Intersect(rays, counters=True)
    gf, _ = g.Intersect(rays)
    oh, ost = o.intersect(rays, any_hit=False)
    assert gh.tobytes() == gf.tobytes()
    m = oh["hit"] == 1
    assert m.sum() > 20000
    assert np.array_equal(gh["hit"], oh["hit"]) and np.array_equal(gh["prim"], oh["prim"]) and np.array_equal(gh["tri"], oh["tri"])
    assert np.array_equal(bits(gh["t"][m]), bits(oh["t"][m]))
    assert gst.interior_visits == ost.interior_visits and gst.tri_tests == ost.tri_tests
    gp, _ = g.IntersectP(rays)
    op_, _ = o.intersect(rays, any_hit=True)
    assert np.array_equal(gp["hit"], op_["hit"])
    pt = ag.PathTracer(5)
    full, st = pt.render_to_host(g, W, H, spp)
    assert st.samples == W * H * spp and st.outliers == 0
    assert gather_ranks(pt, g, W, H, spp, 8).tobytes() == full.tobytes()
    g.close()
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        for tile in ((1900, 900, 96, 54), (700, 400, 64, 64), (3000, 1500, 64, 48)):
            oacc, _ = o.render(W, H, spp, tile=tile, threads=8)
            a, b = tile_rows(full, H, tile), tile_rows(oacc, H, tile)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)


# ---- the configurations at their own sample counts ---------------------------------------------------------------------------
# With 64 spp a wave of the path tracer starts as ONE pixel's 64 samples and k_accumulate adds them line-wise: the code path
# bench.py times.  The oracle runs the reference's loop (myapp.cpp:163-175 per sample, AddSample in sample order myapp.h:17-19).
def assert_tiles_match(desc, full, W, H, spp, tile_list):
    for tile in tile_list:
        oacc, _ = oracle_tile(desc, W, H, spp, tile)
        a, b = tile_rows(full, H, tile), tile_rows(oacc, H, tile)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), tile


def test_c3_1080p_at_its_64_spp_oracle_tiles_and_rank_share():
    """BASELINE config 3 at its own budget, 1920x1080 / 64 spp (bench.py's step): two 96x54 oracle tiles bit-exact; one rank's
    interleaved share of an 8-rank split at 64 spp equals its rows of the full film."""
    W, H, spp = 1920, 1080, 64
    desc = ag.scenes.scene_c3()
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    full, st = pt.render_to_host(g, W, H, spp)
    assert st.samples == W * H * spp and st.outliers == 0
    ctx = g.ctx
    world, r = 8, 3
    rows = tiles.local_rows(H, r, world)
    ptr = ctx.alloc(W * H * 16)
    ctx.memset(ptr, 0, W * H * 16)
    st_r = pt.render(g, W, H, spp, ptr, interleave=(tiles.BLOCK_ROWS, world, r))
    share = ctx.download(ptr, (H, W, 4))[:rows].copy()
    ctx.free(ptr)
    g.close()
    assert st_r.samples == W * rows * spp
    for (y0, h, off) in tiles.row_blocks(H, r, world):
        assert share[off:off + h].tobytes() == full[H - y0 - h:H - y0].tobytes()
    assert_tiles_match(desc, full, W, H, spp, [(912, 500, 96, 54), (200, 90, 96, 54)])


def test_c2_720p_at_its_16_spp_oracle_tiles():
    """BASELINE config 2 at its own budget, 1280x720 / 16 spp (bundles of G = 16 samples)."""
    W, H, spp = 1280, 720, 16
    desc = ag.scenes.scene_c2()
    g = gpu_scene(desc)
    full, st = ag.PathTracer(5).render_to_host(g, W, H, spp)
    g.close()
    assert st.samples == W * H * spp and st.outliers == 0
    assert_tiles_match(desc, full, W, H, spp, [(600, 330, 96, 54), (400, 500, 64, 64)])


def test_c3_192_spp_in_mixed_batches():
    """192 spp of the C3 film through batches of 64 (default), 128 + 64 and 48 x 4 samples (sample groups 64, 64 + 64, 16): the
    three films are the same bit for bit, and an oracle tile at 192 spp pins them."""
    W, H, spp = 1920, 1080, 192
    desc = ag.scenes.scene_c3()
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    a, sa = pt.render_to_host(g, W, H, spp, samples_per_batch=64)
    b, sb = pt.render_to_host(g, W, H, spp, samples_per_batch=128)
    assert a.tobytes() == b.tobytes() and sa.rays == sb.rays
    del b
    c, sc = pt.render_to_host(g, W, H, spp, samples_per_batch=48)
    g.close()
    assert a.tobytes() == c.tobytes() and sa.rays == sc.rays
    del c
    assert_tiles_match(desc, a, W, H, spp, [(1000, 420, 64, 32)])


def test_c4_4k_at_its_256_spp_oracle_tile_and_rank_share():
    """BASELINE config 4 at its own budget: the C3 scene at 3840x2160 / 256 spp (2.1 G samples, sixteen 128 Mi-path batches of 15-16
    samples: sample groups of 1 .. 16).  One 64x48 oracle tile pins the film; the last rank's share of the 8-way split at the same
    budget equals its rows of the full film."""
    W, H, spp = 3840, 2160, 256
    desc = ag.scenes.scene_c3()
    g = gpu_scene(desc)
    pt = ag.PathTracer(5)
    full, st = pt.render_to_host(g, W, H, spp)
    assert st.samples == W * H * spp and st.outliers == 0
    ctx = g.ctx
    world, r = 8, 7
    rows = tiles.local_rows(H, r, world)
    ptr = ctx.alloc(W * H * 16)
    ctx.memset(ptr, 0, W * H * 16)
    pt.render(g, W, H, spp, ptr, interleave=(tiles.BLOCK_ROWS, world, r), want_stats=False)
    share = ctx.download(ptr, (H, W, 4))[:rows].copy()
    ctx.free(ptr)
    g.close()
    for (y0, h, off) in tiles.row_blocks(H, r, world):
        assert share[off:off + h].tobytes() == full[H - y0 - h:H - y0].tobytes()
    assert_tiles_match(desc, full, W, H, spp, [(1800, 1000, 64, 48)])


def test_c5_4k_at_its_1024_spp_oracle_tile():
    """BASELINE config 5 at its own budget: the 5 M-triangle scene at 3840x2160 / 1024 spp = 8.49 G samples -- more than 2^32, so
    the 32-bit stream index wraps from sample 518 on (agpt_math.h: sample_seed; the oracle indexes the same way).  One 32x32
    oracle tile at 1024 spp pins the film, wrap included."""
    W, H, spp = 3840, 2160, 1024
    desc = ag.scenes.scene_c5()
    g = gpu_scene(desc)
    full, st = ag.PathTracer(5).render_to_host(g, W, H, spp)
    g.close()
    assert st.samples == W * H * spp
    o = oracle_scene(desc)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        tile = (1900, 900, 32, 32)
        oacc, _ = o.render(W, H, spp, tile=tile, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    a, b = tile_rows(full, H, tile), tile_rows(oacc, H, tile)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
