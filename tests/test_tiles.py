"""Multi-GPU sharding logic on the CPU: partition properties of the row-block interleave, and a world_size-2 gloo run
of the exact gather/de-interleave path bench.py uses (with the CPU oracle standing in for the per-rank renderer,
so the check needs no GPU): the gathered image must be bit-identical to a single-process render."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiles():
    import ag_pathtracer_amd  # noqa: F401
    from ag_pathtracer_amd import tiles
    return tiles


@pytest.mark.parametrize("H", [1, 7, 64, 1080, 2160])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_row_blocks_partition_the_film(H, world):
    t = _tiles()
    seen = np.zeros(H, np.int32)
    for r in range(world):
        off_expect = 0
        for (y0, h, off) in t.row_blocks(H, r, world):
            assert off == off_expect and h >= 1
            off_expect += h
            seen[y0:y0 + h] += 1
        assert off_expect == t.local_rows(H, r, world) <= t.max_local_rows(H, world)
    assert np.all(seen == 1)
    # interleave keeps the per-rank share within one block of even
    shares = [t.local_rows(H, r, world) for r in range(world)]
    assert max(shares) - min(shares) <= t.BLOCK_ROWS


def test_deinterleave_roundtrip():
    t = _tiles()
    W, H, world = 5, 37, 3
    full = np.random.RandomState(0).rand(H, W, 4).astype(np.float32)
    bufs = []
    for r in range(world):
        b = np.zeros((t.max_local_rows(H, world), W, 4), np.float32)
        for (y0, h, off) in t.row_blocks(H, r, world):
            b[off:off + h] = full[H - y0 - h:H - y0]
        bufs.append(b)
    assert np.array_equal(t.deinterleave(bufs, W, H, world), full)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import ag_pathtracer_amd as ag
    from ag_pathtracer_amd import tiles
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc = ag.scenes.scene_c1(backdrop=ob.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32))
    o = desc.instantiate(ob.OracleScene())
    local = torch.zeros((tiles.max_local_rows(H, world), W, 4), dtype=torch.float32)
    view = local.numpy()
    for (y0, h, off) in tiles.row_blocks(H, rank, world):
        # the oracle adds into a full-film buffer; copy the block's flipped rows into the rank's compact buffer
        full = np.zeros((H, W, 4), np.float32)
        o.render(W, H, spp, tile=(0, y0, W, h), accum=full)
        r0 = tiles.block_accum_row0(H, y0, h, off)
        view[(H - y0 - h) - r0:(H - y0) - r0] = full[H - y0 - h:H - y0]
    full = tiles.gather_to_rank0(local, W, H, rank, world)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gather_is_bit_identical(tmp_path, oracle):
    import torch.multiprocessing as mp
    import ag_pathtracer_amd as ag
    W, H, spp, world = 48, 36, 2, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, out), nprocs=world, join=True)
    got = np.load(out)
    desc = ag.scenes.scene_c1(backdrop=oracle.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32))
    ref, _ = desc.instantiate(oracle.OracleScene()).render(W, H, spp)
    assert got.tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_gpu_gather_tiles_c_abi():
    """agpt_gather_tiles / agpt_deinterleave_tiles (include/agpt.h): the compact per-rank buffers that agpt_render's row
    interleave fills come back as the full Accumulator::pixels image.  One GPU: the three shares of a world-3 split are
    rendered one after the other and de-interleaved by the library's kernel; the world-1 gather runs through
    agpt_gather_tiles itself (no RCCL involved).  The RCCL send/recv path needs one GPU per rank and is NOT exercised here."""
    import ag_pathtracer_amd as ag
    from ag_pathtracer_amd import tiles
    from helpers import gpu_context, gpu_scene
    ctx = gpu_context()
    scene = gpu_scene(ag.scenes.scene_c1())
    pt = ag.PathTracer(5)
    W, H, spp = 96, 77, 2          # 77 rows: the last 8-row block is short
    ref, _ = pt.render_to_host(scene, W, H, spp)
    full = ctx.alloc(W * H * 16)
    try:
        for world in (1, 3):
            ctx.memset(full, 0xFF, W * H * 16)
            for rank in range(world):
                rows = tiles.max_local_rows(H, world)
                local = ctx.alloc(rows * W * 16)
                try:
                    ctx.memset(local, 0, rows * W * 16)
                    pt.render(scene, W, H, spp, local, accum_pitch=W, interleave=(tiles.BLOCK_ROWS, world, rank))
                    if world == 1:
                        comm = ag.Comm(ctx, 1, 0)
                        comm.gather_tiles(local, W, H, tiles.BLOCK_ROWS, full)
                        comm.close()
                    else:
                        ctx.deinterleave_tiles(local, W, H, tiles.BLOCK_ROWS, world, rank, full)
                finally:
                    ctx.download(local, (1,))   # synchronises the stream before the buffer is freed
                    ctx.free(local)
            got = ctx.download(full, (H, W, 4))
            assert got[..., :3].tobytes() == ref[..., :3].tobytes(), "world %d" % world
    finally:
        ctx.free(full)
        scene.close()
