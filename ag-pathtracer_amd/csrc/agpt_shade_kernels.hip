// agpt_shade_kernels.hip -- translation unit of the shading kernels (agpt_shade_kernels.h) and their host-side launchers.
// build.py compiles it with -mllvm -disable-machine-licm (see the header for why); everything else of the library is in
// agpt_api.hip.
#include <hip/hip_runtime.h>

#include "agpt_shade_kernels.h"

namespace agpt {

bool shade_tables_fit_lds(int n_prims, int n_materials, int n_lights) {
    return n_prims <= AGPT_SHADE_LDS_PRIMS && n_materials <= AGPT_SHADE_LDS_MATERIALS && n_lights <= AGPT_SHADE_LDS_LIGHTS;
}

void launch_shade(hipStream_t stream, int grid, bool lds_tables, bool env, const DevScene& sc, const RenderConsts& rc, const PathBuffers& pb,
                  const Queues& qin, const Queues& qout, DevCounters* counters, uint32_t* tile_heads) {
    const dim3 g(grid), b(AGPT_BLOCK);
    if (lds_tables && env) hipLaunchKernelGGL((k_shade<true, true>), g, b, 0, stream, sc, rc, pb, qin, qout, counters, tile_heads);
    else if (lds_tables) hipLaunchKernelGGL((k_shade<true, false>), g, b, 0, stream, sc, rc, pb, qin, qout, counters, tile_heads);
    else if (env) hipLaunchKernelGGL((k_shade<false, true>), g, b, 0, stream, sc, rc, pb, qin, qout, counters, tile_heads);
    else hipLaunchKernelGGL((k_shade<false, false>), g, b, 0, stream, sc, rc, pb, qin, qout, counters, tile_heads);
}

void launch_resolve_pending(hipStream_t stream, int grid, const DevScene& sc, const RenderConsts& rc, const PathBuffers& pb, const Queues& qin) {
    hipLaunchKernelGGL(k_resolve_pending, dim3(grid), dim3(AGPT_BLOCK), 0, stream, sc, rc, pb, qin);
}

}  // namespace agpt
