"""Developer check of the tolerance mode (agpt_render_params::fast_traversal): time and image difference against the exact mode
on the BASELINE scenes (1080p film).  python tools/exp_fast.py [c3|c5|c2] [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
ctx = ag.Context(0, stream=torch.cuda.current_stream().cuda_stream)
desc = {"c3": ag.scenes.scene_c3, "c5": ag.scenes.scene_c5, "c2": ag.scenes.scene_c2}[which](aspect=W / H) if which != "c2" else ag.scenes.scene_c2()
scene = desc.instantiate(ag.Scene(ctx))
pt = ag.PathTracer(5)
acc = torch.zeros((H, W, 4), device="cuda")
res = {}
for fast in (False, True, False, True):
    acc.zero_()
    torch.cuda.synchronize()
    t0 = time.time()
    st = pt.render(scene, W, H, spp, acc.data_ptr(), accum_pitch=W, fast_traversal=fast, timing=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    res[fast] = (acc[..., :3].cpu().numpy() / spp, st, dt)
    print("%s fast=%d: %.1f ms  trace %.1f (ext %.1f mis %.1f shadow %.1f)  traversed %.0f Mrays/s  queries %d" % (
        which, fast, dt * 1e3, st.trace_ms, st.ext_ms, st.mis_ms, st.shadow_ms, (st.rays - st.answered_rays) / dt / 1e6, st.rays))
a, b = res[False][0], res[True][0]
same = np.all(a.view(np.uint32) == b.view(np.uint32), axis=-1)
rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
within = np.all(np.abs(a - b) <= 1e-3 * np.abs(a) + 1e-6, axis=-1)
print("pixels bit-identical %.5f  within rel 1e-3 %.5f  max rel %.3g  mean rel diff per channel %s  rays %d vs %d" % (
    same.mean(), within.mean(), rel.max(), np.abs(a.mean((0, 1)) - b.mean((0, 1))) / a.mean((0, 1)), res[False][1].rays, res[True][1].rays))
