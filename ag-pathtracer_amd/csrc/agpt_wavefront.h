// agpt_wavefront.h -- what the kernels of the wavefront path tracer share: the per-path state in HBM (PathBuffers), the path-id
// queues between the kernels, the per-batch constants, the path / pixel order and the counters.  Included by agpt_kernels.h
// (trace, generate, accumulate, ...) and by agpt_shade_kernels.h (k_shade, k_resolve_pending), which are separate translation
// units because they are compiled with different code-generation options (build.py).
#pragma once

#include "agpt_math.h"
#include "agpt_scene.h"

#define AGPT_BLOCK 256
#define AGPT_CHUNK 64

// path flags (beta4.w)
#define PF_BOUNCE_MASK 0xFFu
#define PF_SPECULAR 0x100u   // specularBounce
#define PF_DEAD 0x200u       // path ended; only the pending NEE of its last vertex is left to resolve
#define PF_PEND_C1 0x400u    // light-sampling leg contribution waits for its shadow ray
#define PF_PEND_C2 0x800u    // BSDF-sampling leg contribution waits for its MIS ray
#define PF_PENDING 0x1000u   // a vertex' UniformSampleOneLight result is still to be added to L

struct PathBuffers {
    float4 *ext_o, *ext_d;        // continuation ray (o.w = tmax)
    float4 *sh_o, *sh_d;          // shadow ray
    float4 *mis_o, *mis_d;        // MIS ray
    DevHit *hit, *mis_hit;
    uint32_t* occluded;
    uint32_t* mis_ok;             // MIS-query result of the production kernel (mis_mode 1)
    float4* beta4;                // beta.xyz, flags
    float4* L4;                   // L.xyz, rng state
    float4* fac4;                 // f*|wi.ns|/pdf of the sampled continuation, chosen light index
    float4 *c1, *c2;              // pending NEE contributions (light leg / BSDF leg)
};

#define AGPT_QSTRIDE 32  // queue counters live on separate 128-B lines: same-line atomics serialise (~11 ns each)
#ifndef AGPT_FRONTIERS
#define AGPT_FRONTIERS 8u
#endif // work-queue frontiers per trace launch (one per XCD), AGPT_QSTRIDE words apart
struct Queues {
    uint32_t *active, *ext, *mis, *shadow;   // path ids
    uint32_t* resolve;                       // ended paths whose last light sample is still to be added (k_resolve_pending)
    uint32_t* counts;                        // [q * AGPT_QSTRIDE]: q = 0 active, 1 ext, 2 mis, 3 shadow, 4 resolve
};
#define AGPT_NQUEUES 5
#define QCOUNT(q, i) ((q).counts[(i) * AGPT_QSTRIDE])

struct RenderConsts {
    int32_t W, H, x0, y0, w, h;
    int32_t s0, S;            // first sample index of the batch, samples in the batch
    uint32_t seed_base;
    int32_t max_depth;
    int32_t accum_pitch, accum_row0;
    uint32_t NP;              // pixels of the tile this call renders (w * rows)
    int32_t il_block, il_world, il_rank;  // row-block interleave (il_block == 0: off)
    int32_t mis_mode;                     // 0: MIS rays traced as closest-hit (mis_hit), 1: as MIS queries (mis_ok)
    int32_t answer_rays;                  // 1: ray queries that cannot reach the image are counted, not traced
};

// Path order inside a batch of S samples x NP pixels: the samples of a pixel are adjacent in runs of G = the largest power
// of two <= 64 that divides S (path id = ((s / G) * NP + p) * G + s % G), so that with 64 spp a wave starts on the 64 samples
// of ONE pixel: the primary rays of a wave are one bundle and the first hits share a few triangles (-3.5 ms per C3 step over
// one-sample-per-pixel waves).  Like the pixel order below, invisible outside.
__device__ __forceinline__ uint32_t sample_group(int32_t S) {
    const uint32_t low = (uint32_t)S & (0u - (uint32_t)S);
    return low < 64u ? low : 64u;
}

// local pixel index -> film pixel (x, y) and the accumulator element it adds into
__device__ __forceinline__ void pixel_of(const RenderConsts& rc, uint32_t p, int& x, int& y, size_t& accum_index) {
    // Local pixel order: 8x8-pixel blocks, row-major inside a block and over the blocks, when the region's width and its number
    // of rows are multiples of 8 (1080p is; so is every rank's share of 8-row blocks) -- a wave then starts on a compact 8x8
    // patch of the film instead of a 64x1 strip, and the queues keep that order.  Plain row-major otherwise.  (The order is
    // invisible outside: RNG streams and accumulator elements are addressed by the film pixel.)
    const uint32_t w = (uint32_t)rc.w, n_rows = rc.NP / w;
    uint32_t row, col;
    if (((w | n_rows) & 7u) == 0u) {
        const uint32_t blk = p >> 6, inner = p & 63u, per_row = w >> 3;
        const uint32_t brow = blk / per_row;
        row = brow * 8u + (inner >> 3);
        col = (blk - brow * per_row) * 8u + (inner & 7u);
    } else {
        row = p / w;
        col = p - row * w;
    }
    x = rc.x0 + (int)col;
    if (rc.il_block == 0) {
        y = rc.y0 + (int)row;
        accum_index = (size_t)((rc.H - 1 - y) - rc.accum_row0) * (size_t)rc.accum_pitch + (size_t)x;
    } else {
        const int j = (int)row / rc.il_block, within = (int)row % rc.il_block;
        const int yb = (j * rc.il_world + rc.il_rank) * rc.il_block;
        const int hb = min(rc.il_block, rc.H - yb);
        y = yb + within;
        accum_index = (size_t)(j * rc.il_block + (hb - 1 - within)) * (size_t)rc.accum_pitch + (size_t)x;
    }
}

struct DevCounters {
    unsigned long long closest_rays, anyhit_rays, interior, tris, shaded, outliers, samples, roots, answered;
#if defined(AGPT_TRACE_STATS) || defined(AGPT_SHADE_CLOCK)
    unsigned long long dbg[64];  // developer builds only (tools/build_variant.py): wave-step statistics of k_trace_fast, phase clocks
#endif
};
#ifdef AGPT_TRACE_STATS
#define TS(...) __VA_ARGS__
#else
#define TS(...)
#endif
// developer build only (-DAGPT_TRACE_STATS -DAGPT_TRACE_CLOCK): s_memtime stamps around the phases of k_trace_fast
#ifdef AGPT_TRACE_CLOCK
#define TCK(...) __VA_ARGS__
#define TCK_NOW() __builtin_amdgcn_s_memtime()
#else
#define TCK(...)
#endif

// Wave-uniform bookkeeping values that the compiler cannot prove uniform (loop-carried through regions with per-lane
// branches) are pinned to scalar registers with readfirstlane: they then cost SALU instead of VALU + exec-mask juggling.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }

// ---------------------------------------------------------------------------------------------------------
// wave-aggregated queue append: one atomic per wave, order inside the wave preserved
__device__ __forceinline__ void queue_push(uint32_t* q, uint32_t* count, bool pred, uint32_t value) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    int lane = __lane_id();
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (pred) {
        uint32_t off = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        q[base + off] = value;
    }
}


// ---------------------------------------------------------------------------------------------------------
// host-side launchers of the shading translation unit (agpt_shade_kernels.hip)
namespace agpt {
// the scene's Scene::primitives records, materials and lights fit the LDS copies k_shade<LDS_TABLES> works from
bool shade_tables_fit_lds(int n_prims, int n_materials, int n_lights);
void launch_shade(hipStream_t stream, int grid, bool lds_tables, bool env, const DevScene& sc, const RenderConsts& rc, const PathBuffers& pb,
                  const Queues& qin, const Queues& qout, DevCounters* counters, uint32_t* tile_heads);
void launch_resolve_pending(hipStream_t stream, int grid, const DevScene& sc, const RenderConsts& rc, const PathBuffers& pb, const Queues& qin);
}  // namespace agpt
