"""Developer check: one rank's share of an N-rank weak-scaling step (8-row blocks k with k % N == rank, 64*N spp) renders as
fast as the single-GPU 64-spp frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402
from ag_pathtracer_amd import tiles  # noqa: E402

W, H = 1920, 1080
ctx = ag.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = ag.scenes.scene_c3(aspect=W / H).instantiate(ag.Scene(ctx))
pt = ag.PathTracer(5)
for world in (1, 2, 4, 8):
    rows = tiles.max_local_rows(H, world)
    local = torch.zeros((rows, W, 4), device="cuda")
    il = None if world == 1 else (tiles.BLOCK_ROWS, world, world - 1)
    for rep in range(2):
        local.zero_()
        torch.cuda.synchronize()
        t0 = time.time()
        st = pt.render(scene, W, H, 64 * world, local.data_ptr(), accum_pitch=W, interleave=il)
        torch.cuda.synchronize()
        dt = time.time() - t0
    print("world %d (last rank): %.1f ms  %.0f Mrays/s  %d samples" % (world, dt * 1e3, st.rays / dt / 1e6, st.samples))
