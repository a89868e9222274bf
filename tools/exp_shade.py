"""Developer experiment: non-trace time of one C3 1080p/64spp step per shaded vertex (for differential timing of k_shade
variants built with tools/build_variant.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ag_pathtracer_amd as ag  # noqa: E402

W, H = 1920, 1080
ctx = ag.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = ag.scenes.scene_c3(aspect=W / H).instantiate(ag.Scene(ctx))
local = torch.zeros((H, W, 4), device="cuda")
pt = ag.PathTracer(5)
for rep in range(2):
    local.zero_()
    st = pt.render(scene, W, H, 64, local.data_ptr(), accum_pitch=W, timing=True)
nt = st.total_ms - st.trace_ms
print("%-10s non-trace %.1f ms  shaded %.1f M  -> %.3f ns per shaded vertex   (rays %.0f M, trace %.1f ms)" % (
    os.environ.get("AGPT_LIB_VARIANT", "default"), nt, st.shaded_vertices / 1e6, nt * 1e6 / st.shaded_vertices, st.rays / 1e6, st.trace_ms))
