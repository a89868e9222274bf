"""Builds libagpt_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libagpt_hip.so")
# (source, extra flags): agpt_shade_kernels.hip -- k_shade, k_resolve_pending -- is compiled with MachineLICM off and k_shade
# held to four waves per SIMD: the pass hoists the two v_mov of every fp64 polynomial coefficient of the trigonometry out of the
# path loop and keeps the pairs live for the whole kernel (168 registers + 7-14 spilled against 134 + 0; at the 128 of four waves
# 2 spilled).  Measured on C3: k_shade -3.5 ms per step; the same flag on the trace kernels costs them 2 ms, hence two units.
SHADE_FLAGS = ["-mllvm", "-disable-machine-licm", "-DAGPT_SHADE_WAVES=4"]
SOURCES = ["agpt_api.hip", "agpt_shade_kernels.hip", "agpt_host_scene.cpp", "agpt_obj.cpp", "agpt_image.cpp"]
SOURCE_FLAGS = {"agpt_shade_kernels.hip": SHADE_FLAGS}
HEADERS = ["agpt_math.h", "agpt_scene.h", "agpt_trace.h", "agpt_shade.h", "agpt_wavefront.h", "agpt_kernels.h", "agpt_shade_kernels.h",
           "agpt_host_scene.hpp", os.path.join("..", "..", "include", "agpt.h")]
# -ffp-contract=off + no fast-math: every fp32 op rounds on its own, exactly as written (parity with the oracle);
# explicit __builtin_fmaf calls (Markstein division in agpt_trace.h) stay fused.
# -fno-slp-vectorize: the SLP pass packs adjacent f32 adds/muls into v_pk_*_f32, which on gfx950 cost more than the two
# scalar ops they replace (measured: trace -1 %, shade -2.5 % with the pass off).
# -amdgpu-atomic-optimizer-strategy=None: the pass rewrites a one-lane atomicAdd into "atomic + readfirstlane of its result
# right behind it", which makes the trace kernel's pipelined work-queue atomic synchronous again; every atomic in this
# library is already aggregated by hand (one lane per wave / per tile).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(out, extra=(), verbose=False, objdir=None, extra_for=None):
    """Compile every source to an object with its own flags (in parallel), then link `out`.  extra: flags added for every source;
    extra_for: {source: flags} added for one (developer A/B builds, tools/build_variant.py)."""
    import concurrent.futures
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory() as tmp:
        tmp = objdir or tmp
        jobs = []
        for src in SOURCES:
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            jobs.append(([hipcc] + cflags + SOURCE_FLAGS.get(src, []) + list(extra) + list((extra_for or {}).get(src, [])) + ["-c", os.path.join(CSRC, src), "-o", obj], obj))
        if verbose:
            for cmd, _ in jobs:
                print(" ".join(cmd))
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(lambda j: subprocess.check_call(j[0]), jobs))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _, obj in jobs] + ["-o", out]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    return build_library(LIB, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
