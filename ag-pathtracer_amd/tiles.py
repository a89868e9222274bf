"""Framebuffer sharding for multi-GPU rendering (one process per GPU).

Every (pixel, sample) path is independent given its seed (which depends only on the global pixel index and the
sample index), so the film is partitioned into horizontal row blocks dealt round-robin to the ranks -- interleaved
rather than contiguous strips so sky rows and atrium-floor rows are spread evenly.  Each rank accumulates its blocks
for the whole sample budget into a compact local buffer; the only exchange is one gather of those buffers to rank 0
at resolve time (RCCL over xGMI when the tensors live on GPUs), followed by a de-interleave.  The image is
bit-identical for any number of ranks.

The reference has no counterpart (single process, single thread: myapp.cpp:163-175); the accumulator semantics
(row H-1-y, sums of samples in sample order) are Accumulator::AddSample's (myapp.h:17-19).
"""
import numpy as np

BLOCK_ROWS = 8


def row_blocks(H, rank, world, block_rows=BLOCK_ROWS):
    """[(y0, h, local_row_offset)] of the row blocks owned by `rank` (block k -> rank k % world)."""
    out = []
    off = 0
    k = 0
    for y0 in range(0, H, block_rows):
        h = min(block_rows, H - y0)
        if k % world == rank:
            out.append((y0, h, off))
            off += h
        k += 1
    return out


def local_rows(H, rank, world, block_rows=BLOCK_ROWS):
    return sum(h for _, h, _ in row_blocks(H, rank, world, block_rows))


def max_local_rows(H, world, block_rows=BLOCK_ROWS):
    return max(local_rows(H, r, world, block_rows) for r in range(world))


def block_accum_row0(H, y0, h, local_off):
    """accum_row0 for agpt_render so that the block's flipped rows land at local rows [local_off, local_off+h)."""
    return H - y0 - h - local_off


def deinterleave(gathered, W, H, world, block_rows=BLOCK_ROWS):
    """gathered: list (per rank) of [max_local_rows, W, 4] arrays/tensors -> full [H, W, 4] accumulator
    (row index = flipped film row, like Accumulator::pixels)."""
    first = gathered[0]
    if hasattr(first, "new_zeros"):
        full = first.new_zeros((H, W, 4))
    else:
        full = np.zeros((H, W, 4), first.dtype)
    for r in range(world):
        for (y0, h, off) in row_blocks(H, r, world, block_rows):
            full[H - y0 - h:H - y0] = gathered[r][off:off + h]
    return full


def render_sharded(render_block, W, H, rank, world, local_accum, block_rows=BLOCK_ROWS):
    """Calls render_block(tile=(0, y0, W, h), accum_row0=..) for every block of this rank.
    `local_accum` is the rank's [max_local_rows, W, 4] buffer (render_block adds into it)."""
    out = []
    for (y0, h, off) in row_blocks(H, rank, world, block_rows):
        out.append(render_block((0, y0, W, h), block_accum_row0(H, y0, h, off)))
    return out


def alloc_gather_buffers(local_accum, rank, world):
    """Receive buffers of gather_to_rank0 on rank 0 (None elsewhere and for world == 1): allocate once, outside any
    timed region, and pass as out_bufs."""
    import torch.distributed as dist
    if world == 1 or rank != 0:
        return None
    if dist.is_initialized() and dist.get_backend() == "gloo" and local_accum.is_cuda:
        host = local_accum.cpu()
        return [host.new_zeros(host.shape) for _ in range(world)]
    return [local_accum.new_zeros(local_accum.shape) for _ in range(world)]


def gather_to_rank0(local_accum, W, H, rank, world, block_rows=BLOCK_ROWS, out_bufs=None):
    """One collective: gather every rank's compact buffer to rank 0 and de-interleave.  Uses torch.distributed
    (backend nccl = RCCL for GPU tensors, gloo for CPU tensors).  Returns the full accumulator on rank 0, else None.
    out_bufs: rank 0's receive buffers from alloc_gather_buffers (allocated here when None)."""
    import torch.distributed as dist
    if world == 1:
        return deinterleave([local_accum], W, H, 1, block_rows)
    bufs = out_bufs if out_bufs is not None else alloc_gather_buffers(local_accum, rank, world)
    # dist.gather on the nccl backend is RCCL grouped send/recv: every peer's buffer (<= 17 MB at 4K) travels its own
    # direct xGMI link to rank 0, so the exchange is not ring-bound.
    if dist.get_backend() == "gloo" and local_accum.is_cuda:
        # rehearsal path (several ranks on one GPU): gloo gathers through host memory
        dist.gather(local_accum.cpu(), bufs, dst=0)
        if rank != 0:
            return None
        return deinterleave([b.to(local_accum.device) for b in bufs], W, H, world, block_rows)
    dist.gather(local_accum, bufs, dst=0)
    if rank != 0:
        return None
    return deinterleave(bufs, W, H, world, block_rows)
