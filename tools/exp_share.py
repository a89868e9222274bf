"""Developer check: renders the last rank's share of an N-way split of the C3 film `reps` times (for rocprofv3 --kernel-trace --stats:
per-kernel time of a rank's share against 1/N of the whole film's).  python tools/exp_share.py [N [reps]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402
from ag_pathtracer_amd import tiles  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H, SPP = 1920, 1080, 64
ctx = ag.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = ag.scenes.scene_c3(aspect=W / H).instantiate(ag.Scene(ctx))
pt = ag.PathTracer(5)
rows = tiles.max_local_rows(H, world)
local = torch.zeros((rows, W, 4), device="cuda")
il = None if world == 1 else (tiles.BLOCK_ROWS, world, 0)
for rep in range(reps):
    local.zero_()
    torch.cuda.synchronize()
    t0 = time.time()
    st = pt.render(scene, W, H, SPP, local.data_ptr(), accum_pitch=W, interleave=il)
    torch.cuda.synchronize()
    print("N=%d rep %d: %.2f ms  trace %.2f ms  rays %d" % (world, rep, (time.time() - t0) * 1e3, st.trace_ms, st.rays))
