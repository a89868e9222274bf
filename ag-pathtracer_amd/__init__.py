"""ag-pathtracer_amd: the MI355X-native path-tracing hot path of voxel-tracer/ag-pathtracer.

Hand-written gfx950 HIP kernels (BVH traversal, triangle/sphere tests, Disney/microfacet shading, NEE + MIS) behind
the C ABI of include/agpt.h, plus this thin host mirror of the reference's Scene / Integrator surface.
"""
from . import build as _build  # noqa: F401
from .binding import (MAT_DIFFUSE_ONLY, MAT_DISNEY, MAT_MIRROR, AgptError, CameraDesc, Context, PathTracer,  # noqa: F401
                      RenderParams, Scene, Stats, Comm, comm_unique_id, create_backdrop, bvh_build, toplevel_build, toplevel_pack16, load_obj, load_hdr, lib, library_path, HIT_DTYPE, NODE_DTYPE, RAY_DTYPE,
                      EXPORTS)
from . import scenes  # noqa: F401
from .scenes import SceneDesc  # noqa: F401


def build(force=False, verbose=False):
    """Compile libagpt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    return _build.build(force=force, verbose=verbose)
