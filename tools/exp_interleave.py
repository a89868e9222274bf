"""Developer check: what one rank's share of an N-rank run costs on ONE MI355X (no RCCL, no second GPU): rank 0's
interleaved 8-row blocks -- the largest share where the blocks do not divide evenly (1080 rows = 135 blocks: at N = 8 ranks 0..6
hold 17, rank 7 holds 16) -- rendered alone, for N = 1, 2, 4, 8.
  strong split (bench.py's default): the same film and sample budget cut N ways -> time per rank share, and the efficiency that
      share time implies, T(1) / (N * T(N)) -- an upper bound of the real curve (the gather and rank imbalance come on top);
  weak split: N times the samples, every rank keeps a full GPU's load.
python tools/exp_interleave.py [c3 1080p/64spp | c4 = the same scene at 3840x2160 / 256 spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402
from ag_pathtracer_amd import tiles  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
W, H, SPP = (1920, 1080, 64) if cfg == "c3" else (3840, 2160, 256)
ctx = ag.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = ag.scenes.scene_c3(aspect=W / H).instantiate(ag.Scene(ctx))
pt = ag.PathTracer(5)


def share(world, spp):
    rows = tiles.max_local_rows(H, world)
    local = torch.zeros((rows, W, 4), device="cuda")
    il = None if world == 1 else (tiles.BLOCK_ROWS, world, 0)
    best = None
    for rep in range(3):
        local.zero_()
        torch.cuda.synchronize()
        t0 = time.time()
        st = pt.render(scene, W, H, spp, local.data_ptr(), accum_pitch=W, interleave=il)
        torch.cuda.synchronize()
        dt = time.time() - t0
        best = dt if best is None or dt < best else best
    return best, st


print("%s: %dx%d, %d spp, one MI355X, rank 0's share (the largest) of an N-rank split (emulated, no RCCL)" % (cfg, W, H, SPP))
t1 = None
film = None
for world in (1, 2, 4, 8):
    dt, st = share(world, SPP)
    trav = st.rays - st.answered_rays
    if world == 1:
        t1, film = dt, trav
    print("strong N=%d: %8.1f ms for the largest rank share (%5.2f %% of the film's rays)  -> %6.0f Mrays/s for N ranks, efficiency %.3f"
          % (world, dt * 1e3, 100.0 * trav / film, film / dt / 1e6, t1 / (world * dt)))
if cfg == "c3":
    for world in (1, 2, 4, 8):
        dt, st = share(world, SPP * world)
        print("weak   N=%d: %8.1f ms per rank share  %6.0f Mrays/s traversed on the rank" % (world, dt * 1e3, (st.rays - st.answered_rays) / dt / 1e6))
