// agpt_kernels.h -- the gfx950 kernels of the wavefront path tracer.
//
// One agpt_render batch = S samples x NP tile pixels paths (path order: sample_group / pixel_of below).  Per iteration:
//     k_trace_fast<0>(ext queue)     continuation rays, closest hit                     -> hit[]
//     k_trace_fast<2>(mis queue)     BSDF-sampled MIS rays as exact early-exit queries  -> mis_ok[]
//     k_trace_fast<1>(shadow queue)  NEE shadow rays, any hit                           -> occluded[]
//     k_shade(active queue)          resolves the previous vertex' NEE, then PathTracer::Li's loop body for the
//                                    new hit; appends to the next iteration's queues with one atomic per queue per
//                                    wave-private 256-path tile (ballot counts, order-preserving offsets)
//     k_resolve_pending(resolve q.)  ended paths: adds the last vertex' light sample once its rays have been traced
// (k_shade and k_resolve_pending live in agpt_shade_kernels.h, a translation unit of their own with its own code-generation options;
// the instrumented / fallback k_trace traces MIS rays as closest hits into mis_hit[]).  Scenes with more than 64
// primitives run k_candidates (top-level tree over Scene::primitives) in front of each trace launch.  The trace kernels are
// persistent: a fixed grid of waves pulls 64-ray chunks off the queue; queue lengths stay on the device (no host round
// trip per iteration).  Path state is SoA float4 in HBM (coalesced 16 B/lane); the per-lane traversal stack lives in LDS
// (23 entries x 4 B x 256 lanes = 23 KiB/block, deeper entries in an HBM spill column).  A continuation ray that retires on a
// primitive without a material is re-cast by the closest-hit kernel itself (CUR_RECAST), so a render has max_depth + 3 iterations.
#pragma once

#include "agpt_shade.h"
#include "agpt_trace.h"

#include "agpt_wavefront.h"

// ---------------------------------------------------------------------------------------------------------
// DEPTH: per-lane stack entries (32, or 64 for BVHs deeper than AGPT_STACK_DEPTH: 64 KiB of LDS per block)
template <bool ANY, bool COUNT, int DEPTH>
__global__ void __launch_bounds__(AGPT_BLOCK)
k_trace(DevScene sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t count_imm,
        uint32_t* __restrict__ work_head, const float4* __restrict__ ray_o, const float4* __restrict__ ray_d,
        DevHit* __restrict__ hits, uint32_t* __restrict__ occluded, DevCounters* __restrict__ counters) {
    __shared__ uint32_t s_stack[DEPTH * AGPT_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t count = count_ptr ? *count_ptr : count_imm;
    TraceCounters cnt;
    cnt.interior = 0;
    cnt.tris = 0;
    cnt.roots = 0;
    cnt.last_mesh = 0;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(work_head, (uint32_t)AGPT_CHUNK);
        base = __shfl(base, 0);
        if (base >= count) break;
        uint32_t i = base + lane;
        if (i < count) {
            uint32_t pid = queue ? queue[i] : i;
            float4 o = ray_o[pid], d = ray_d[pid];
            DevHit h;
            bool hit = trace_scene<ANY, COUNT>(sc, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, h, stack, AGPT_BLOCK, cnt);
            if (ANY)
                occluded[pid] = hit ? 1u : 0u;
            else
                hits[pid] = h;
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            cnt.interior += __shfl_down(cnt.interior, off);
            cnt.tris += __shfl_down(cnt.tris, off);
            cnt.roots += __shfl_down(cnt.roots, off);
        }
        if (lane == 0) {
            atomicAdd(&counters->interior, (unsigned long long)cnt.interior);
            atomicAdd(&counters->tris, (unsigned long long)cnt.tris);
            atomicAdd(&counters->roots, (unsigned long long)cnt.roots);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (ANY)
            atomicAdd(&counters->anyhit_rays, (unsigned long long)count);
        else
            atomicAdd(&counters->closest_rays, (unsigned long long)count);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Production trace kernel (the generic k_trace above stays as the instrumented variant and the fallback for more than
// 2048 primitives; both are checked against the oracle and against each other).  Same decisions as trace_scene(),
// restructured for wave efficiency on 64-wide CDNA4 wavefronts:
//   * persistent waves with IN-FLIGHT REFILL: when at least AGPT_REFILL lanes have retired their rays the wave hands
//     them new rays from a wave-private LDS ring instead of idling until all 64 are done;
//   * PREFILTER RING: rays enter the ring 64 at a time (one atomic on the frontier counter of the wave's queue segment)
//     with a per-primitive candidate mask from a cheap conservative slab test, all 64 lanes busy; the exact
//     Bounds::Intersect is re-run per lane where the reference runs it;
//   * VOTE-SCHEDULED traversal: a lane is between primitives (A), at an interior node (B) or at a leaf (C); each step
//     the wave executes the body most lanes wait for.  A mesh's root-box test is an interior step on its root pair.
#define AGPT_REFILL 20      // idle lanes that trigger a refill, closest-hit launches (tuned on C3: 16-24 equal)
#define AGPT_REFILL_ANY 40  // same, any-hit / MIS-query launches: short traversals, refills are cheaper in bulk
#define AGPT_RING 128  // entries of the per-wave ring of pre-filtered rays (power of two, >= 64 + 63)
// per-primitive root record staged in LDS: [2k] = (bmin.xyz | sphere centre.xyz, kind), [2k+1] = (bmax.xyz, root_enc) |
// (r2, -, -, -) | (half x, half z, -, -); kind 0 = mesh, 1 = sphere, 2 = empty mesh (never hit), 3 = plane

// MODE 0: Scene::Intersect (closest hit).  MODE 1: Scene::IntersectP (any hit, shadow rays).
// MODE 2: the MIS query of EstimateDirect's BSDF-sampling leg (integrator.h:76-88).  The reference runs a full
// Scene::Intersect there but only uses "is the closest hit the sampled light's shape" (area light) or "did the ray
// escape" (infinite light).  That boolean is answered exactly by an early-exit traversal:
//   * infinite light: ok iff nothing is hit (any-hit);
//   * area light with sphere S (primitive index kS): let t_s be S's accepted root for this ray.  If S is missed the
//     answer is no; else ok iff no other primitive yields a hit the reference would have kept in front of S:
//     triangles and planes need t < t_s (their tests are strict on both list sides), another sphere P blocks iff
//     root_P < t_s when P precedes S in Scene::primitives and root_P <= t_s when it follows (Sphere::Intersect accepts
//     root <= ray.t, so a tie goes to the later primitive); degenerate triangles never block (quirk 11).
// Every MIS ray is still traced through the BVH; it stops at the first blocker instead of finishing a closest-hit search.
// LIST: scenes with more than 64 primitives.  k_candidates (below) has walked the top-level tree over the meshes' root
// boxes once per ray and left one 64-bit candidate word per chunk of 64 primitives plus a word saying which chunks have
// any (cand_mask[chunk * cand_stride + path], cand_chunks[path]); the lane walks its candidates chunk by chunk in list
// order -- exactly Scene::Intersect's sequential walk with a shared ray.t (scene.h:5-19), minus the primitives the ray
// cannot touch.  The prefilter and the LDS copy of the primitive records belong to the short-list instantiation only.
// COUNT: the same kernel also counts the work it does -- child-pair records fetched (interior), root-pair records fetched
// (roots: the mesh root-box tests that survive the prefilter) and triangle tests -- for bench.py's roofline; the counts
// are per-lane sums, deterministic for a given queue (every lane's own operation sequence is fixed).
// DEPTH = per-lane stack entries kept in LDS.  SPILL: the scene's BVHs are deeper than that -- entries DEPTH and up live
// in a per-thread column of an HBM buffer (spill[(k - DEPTH) * threads_in_grid + thread]); near-first descent rarely needs
// them.  Occupancy is LDS-bound (DEPTH KiB of stack + 8 KiB per block): 23 entries -> 5 blocks = 20 waves per CU, which
// is worth 9 % over the 4 blocks of a 32-entry stack (measured; the round-1 test of a fifth block used an exact 32-KiB fit
// that never became resident).
#ifndef AGPT_TRACE_WAVES
#define AGPT_TRACE_WAVES 1   // minimum waves per SIMD the register allocation of k_trace_fast is held to
#endif
template <int MODE, int DEPTH, bool LIST, bool COUNT, bool SPILL, bool PEEK>
__global__ void __launch_bounds__(AGPT_BLOCK, AGPT_TRACE_WAVES)
k_trace_fast(DevScene sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t count_imm,
             uint32_t* __restrict__ work_head, const float4* __restrict__ ray_o, const float4* __restrict__ ray_d,
             DevHit* __restrict__ hits, uint32_t* __restrict__ occluded, DevCounters* __restrict__ counters, int refill,
             uint32_t cand_stride, uint32_t* __restrict__ spill, const unsigned long long* __restrict__ cand_mask,
             const uint32_t* __restrict__ cand_chunks, float4* ray_o_w, float4* ray_d_w, const float4* path_beta4, float4* path_L4) {
    constexpr bool ANY = MODE != 0;
    constexpr bool MIS = MODE == 2;
    constexpr int BLOCK = AGPT_BLOCK;
    constexpr bool PRIM_LDS = !LIST;   // short lists: the primitive records live in LDS
    __shared__ uint32_t s_stack[DEPTH * BLOCK];
    __shared__ float4 s_prim[PRIM_LDS ? 2 * 64 : 2];
    uint32_t* stack = s_stack + threadIdx.x;
    uint32_t* spill_col = SPILL ? spill + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;
    const size_t spill_stride = (size_t)gridDim.x * BLOCK;
    auto stack_push = [&](int k, uint32_t v) {
        if (!SPILL || k < DEPTH)
            stack[k * BLOCK] = v;
        else
            spill_col[(size_t)(k - DEPTH) * spill_stride] = v;
    };
    auto stack_at = [&](int k) -> uint32_t {
        if (!SPILL || k < DEPTH) return stack[k * BLOCK];
        return spill_col[(size_t)(k - DEPTH) * spill_stride];
    };
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // (a value loaded through a pointer counts as divergent for the compiler: readfirstlane keeps the bookkeeping scalar)
    const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)(count_ptr ? *count_ptr : count_imm));
    // (the late iterations' queues are short or empty: a launch can use one wave per 64 rays -- the rest of the grid leaves
    // before setting up, with nothing to drain)
    if (blockIdx.x * (uint32_t)BLOCK >= count) return;
    const int n_prims = LIST ? 0 : sc.n_prims;   // (short lists: the primitive records live in LDS)
    if (PRIM_LDS && (int)threadIdx.x < n_prims) {
        const DevPrim& P = sc.prims[threadIdx.x];
        float4 a, b;
        if (P.type == AGPT_PRIM_SPHERE) {
            a.x = P.cx; a.y = P.cy; a.z = P.cz; a.w = 1.f;
            b.x = P.r2; b.y = 0.f; b.z = P.material < 0 ? 1.f : 0.f; b.w = 0.f;   // b.z: no material (an emitter's sphere)
        } else if (P.type == AGPT_PRIM_PLANE) {
            a.x = P.cx; a.y = P.cy; a.z = P.cz; a.w = 3.f;
            b.x = P.r; b.y = P.r2; b.z = P.material < 0 ? 1.f : 0.f; b.w = 0.f;
        } else {
            a.x = P.root_bmin[0]; a.y = P.root_bmin[1]; a.z = P.root_bmin[2]; a.w = P.n_tris > 0 ? 0.f : 2.f;
            b.x = P.root_bmax[0]; b.y = P.root_bmax[1]; b.z = P.root_bmax[2]; b.w = __uint_as_float(P.root_enc);
        }
        s_prim[2 * threadIdx.x] = a;
        s_prim[2 * threadIdx.x + 1] = b;
    }
    __syncthreads();
    const unsigned long long all_prims = n_prims >= 64 ? ~0ull : ((1ull << n_prims) - 1ull);

    // Work distribution: rays are taken 64 at a time from the frontier of one of AGPT_FRONTIERS queue segments (see the
    // refill code below); within a segment the frontier moves monotonically, so at any moment an XCD traces rays that are
    // neighbours in the queue (neighbouring pixels / path ids), which keeps their BVH nodes hot in its L2.  Reserving large
    // private ranges per wave instead measured 8 % slower (A/B on MI355X, same process).
    // A lane's scheduling state is carried by `cur` itself: CUR_IDLE = no ray, CUR_PICK = between meshes (state A),
    // anything below = an interior child-pair index (state B), sign bit set = a leaf encoding (state C).  One v_cmp per
    // state gives the wave's vote masks.  LIST adds CUR_FETCH (voted with state A): the current chunk's candidates are used
    // up and the next chunk's word has to be fetched.
    uint32_t pid = 0;
    unsigned long long mask = 0;
    // CUR_RECAST (closest-hit launches of the wavefront loop): the ray has retired on a primitive without a material and waits for
    // the top of the outer loop to be re-cast (see there); it takes part in no vote.
    constexpr uint32_t CUR_IDLE = 0x7FFFFFFFu, CUR_RECAST = 0x7FFFFFFEu, CUR_PICK = 0x7FFFFFFDu, CUR_FETCH = 0x7FFFFFFCu;
    constexpr uint32_t CUR_B_END = LIST ? CUR_FETCH : CUR_PICK;   // state B: cur < CUR_B_END
    const uint32_t rootpair_base = sc.rootpair_base;
    const unsigned long long mesh_mask0 = sc.mesh_masks[0];
    // LIST lane state: the mesh bits of the current chunk
    unsigned long long mmask = 0;
    uint32_t cvisit = 0, cchunk = 0;   // LIST: chunks of the ray's candidates still to visit (one bit each), the current chunk
    // the sphere of an area light as (centre, r2), wherever it sits in the primitive list
    auto light_sphere = [&](int shape, v3& c, float& r2) {
        if (PRIM_LDS) {
            const float4 sa = s_prim[2 * shape], sb = s_prim[2 * shape + 1];
            c = V3(sa.x, sa.y, sa.z);
            r2 = sb.x;
        } else {
            const DevPrim& P = sc.prims[shape];
            c = V3(P.cx, P.cy, P.cz);
            r2 = P.r2;
        }
    };
    // Scene::Intersect's walk to the next primitive of the list (scene.h:8-17).  A mesh is entered through its root
    // pair -- the root-box test becomes the lane's next interior step (state B); spheres, planes and the end of the
    // list go through state A.
    auto pick_next = [&](unsigned long long& m) -> uint32_t {
        const unsigned long long low = m & (0ull - m);
        if (LIST) {
            if (m == 0) return cvisit ? CUR_FETCH : CUR_PICK;
            if (low & mmask) {
                m ^= low;
                return rootpair_base + 2u * (64u * cchunk + (uint32_t)(__ffsll((long long)low) - 1));
            }
            return CUR_PICK;
        }
        if (low & mesh_mask0) {
            m ^= low;
            return rootpair_base + 2u * (uint32_t)(__ffsll((long long)low) - 1);
        }
        return CUR_PICK;
    };
    // LIST: start on chunk c of the lane's ray (issues the loads of its candidate word and of the chunk's mesh bits)
    auto enter_chunk = [&](uint32_t c) {
        cvisit &= ~(1u << c);
        cchunk = c;
        mask = cand_mask[(size_t)c * cand_stride + pid];
        mmask = sc.chunk_mesh_masks[c];
    };
    // Work distribution and the refill pipeline.  The queue is cut into AGPT_FRONTIERS contiguous segments, each with its own
    // frontier counter on its own 128-B line; a wave starts in the segment of its XCD (workgroups are dealt round-robin to
    // the 8 XCDs) and moves on to the next segment when its own is drained.  (One shared counter was the limit of the
    // any-hit launches: same-address atomics complete one per ~11 ns chip-wide.)  Getting 64 new rays is a chain of four
    // dependent memory operations -- frontier atomic -> queue[] -> ray_o/ray_d[] -> (prefilter) -> ring -- that took
    // 10-20 k cycles per top-up when run back to back (13 % of the closest-hit launches' wave time, 30 % of the MIS
    // launches', measured with s_memtime stamps).  It is therefore software-pipelined over successive top-ups:
    // pump_consume() prefilters the rays whose loads were issued one top-up earlier into the ring; pump_issue() -- called
    // AFTER the idle lanes have been handed their rays, because vector-memory results return in issue order and anything
    // loaded after it would wait for it -- issues the ray loads for the path ids fetched by the previous pump_issue, the
    // queue load for the offset returned by the previous atomic, and the next atomic.  The traversal steps between two
    // top-ups hide all of it; only a cold pipeline (kernel start, segment change) runs the chain synchronously.
    uint32_t seg = blockIdx.x & (AGPT_FRONTIERS - 1u), segs_left = AGPT_FRONTIERS;   // wave-uniform
    const uint32_t seg_len = ((count + AGPT_FRONTIERS - 1u) / AGPT_FRONTIERS + 63u) & ~63u;
    uint32_t cur = CUR_IDLE, hid = AGPT_HIT_MISS;
    int sp = 0;
    float rayt = 0.f, hb1 = 0.f, hb2 = 0.f;
    TraceRay r;
    r.O = V3s(0.f);
    r.D = V3(0.f, 0.f, 1.f);
    r.R = V3s(1.f);
    r.fast = true;

    // wave-private ring of pre-filtered rays (path id + primitive mask), filled 64 rays at a time
    __shared__ uint32_t s_ring_pid[BLOCK / 64][AGPT_RING];
    __shared__ unsigned long long s_ring_mask[BLOCK / 64][AGPT_RING];
    uint32_t* ring_pid = s_ring_pid[threadIdx.x >> 6];
    unsigned long long* ring_mask = s_ring_mask[threadIdx.x >> 6];
    uint32_t ring_head = 0, ring_tail = 0;  // wave-uniform, monotonically increasing
    bool mis_area = false;    // MODE 2: the sampled light is an area light
    int mis_skip = -1;        // MODE 2: primitive index of the sampled area light's sphere (never a blocker itself)
    bool mis_reach = true;    // MODE 2: infinite light, or the ray reaches the light's sphere
    bool any_slow = false;                  // wave-uniform: some active lane's ray needs the true-division slab test
    uint32_t c_int = 0, c_root = 0, c_tri = 0;   // COUNT
    TS(unsigned long long ts_steps[3] = {0, 0, 0}; unsigned long long ts_lanes[3] = {0, 0, 0}; unsigned long long ts_act = 0;
       unsigned long long ts_refills = 0; unsigned long long ts_refilled = 0; unsigned long long ts_pref = 0;
       uint32_t ts_push[5] = {0, 0, 0, 0, 0};)
    TCK(unsigned long long tk_refill = 0, tk_vote = 0, tk_bmem = 0, tk_balu = 0, tk_c = 0, tk_a = 0; const unsigned long long tk_begin = TCK_NOW();)

    uint32_t pfa_off = 0;       // lane 0: offset returned by the pending frontier atomic (stage A)
    uint32_t pfa_seg = 0;
    bool pfa_valid = false;     // wave-uniform, like every *_valid / *_n below
    uint32_t pfb_pid = 0, pfb_n = 0;   // stage B: path ids of the chunk after next (queue load pending)
    bool pfb_valid = false;
    uint32_t pfc_pid = 0, pfc_n = 0;   // stage C: the next chunk's rays (ray loads pending)
    bool pfc_valid = false;
    float4 pfc_o, pfc_d;
    pfc_o.x = 0.f; pfc_o.y = 0.f; pfc_o.z = 0.f; pfc_o.w = 0.f;
    pfc_d = pfc_o;
    auto drained = [&]() -> bool { return segs_left == 0 && !pfa_valid && !pfb_valid && !pfc_valid; };
    auto pin_state = [&]() {   // see uni(): keeps the bookkeeping in scalar registers
        seg = uni(seg); segs_left = uni(segs_left); ring_tail = uni(ring_tail); ring_head = uni(ring_head);
        pfa_seg = uni(pfa_seg); pfa_valid = uni(pfa_valid); pfb_n = uni(pfb_n); pfb_valid = uni(pfb_valid);
        pfc_n = uni(pfc_n); pfc_valid = uni(pfc_valid);
    };
    auto pump_consume = [&]() {
        // stage C -> ring: conservative per-primitive prefilter of the 64 rays that arrived, all lanes busy
        if (LIST) {
            // the candidates were collected by k_candidates: the ring carries the path id and its chunk word
            if (pfc_valid) {
                if ((uint32_t)lane < pfc_n) {
                    const uint32_t slot = (ring_tail + (uint32_t)lane) & (AGPT_RING - 1);
                    ring_pid[slot] = pfc_pid;
                    ring_mask[slot] = (unsigned long long)__float_as_uint(pfc_o.x);
                }
                ring_tail += pfc_n;
                pfc_valid = false;
                TS(ts_pref++;)
            }
        } else if (pfc_valid) {
            if ((uint32_t)lane < pfc_n) {
                const uint32_t npid = pfc_pid;
                float4 o = pfc_o;
                const float4 d = pfc_d;
                const TraceRay nr = make_trace_ray(V3(o.x, o.y, o.z), V3(d.x, d.y, d.z));
                unsigned long long skip_bit = 0;
                if (MIS) {   // MIS rays carry the sampled light's sphere (primitive index, ~0u = infinite light) in o.w; tmax = inf
                    const uint32_t shape = __float_as_uint(o.w);
                    o.w = AGPT_FLT_MAX;
                    if (shape != 0xFFFFFFFFu) {
                        v3 lc;
                        float lr2, ts;
                        light_sphere((int)shape, lc, lr2);
                        if (sphere_test_c(lc, lr2, nr, o.w, ts)) o.w = ts;
                        skip_bit = 1ull << shape;
                    }
                }
                // t ~ fma(b, R, -RN(O*R)) differs from the exact quotient (b-O)/D by at most 2^-23 |t| + 2^-24 |O*R|
                // (R = RN(1/D), one rounding of O*R, one of the fma); rejecting only when tmax*(1+2e-6) + 2E < tmin with
                // E = 2^-22 max|O*R| can therefore only over-accept -- the exact Bounds::Intersect still decides where the
                // reference runs it.
                const float nox = -(nr.O.x * nr.R.x), noy = -(nr.O.y * nr.R.y), noz = -(nr.O.z * nr.R.z);
                const float slack2 = 0x1p-21f * fmaxf(fmaxf(fabsf(nox), fabsf(noy)), fabsf(noz));
                // The records come through SCALAR loads (wave-uniform index, constant address space: scene arrays are
                // read-only for the whole launch, and only loads from that address space are selected as s_load without a
                // no-clobber proof), so the box coordinates are SGPR operands of the FMAs and the loads of several records
                // are in flight at once; reading the LDS copy made every iteration wait for its own two ds_reads.
                unsigned long long m = 0;
                typedef const float __attribute__((address_space(4))) ConstF;
                typedef const DevPrim __attribute__((address_space(4))) ConstPrim;
                ConstF* pf = (ConstF*)sc.prefilter;
                const int j0 = sc.pf_begin[0], j1 = sc.pf_begin[1];
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {   // non-empty meshes: conservative slab test of the root box
                    ConstF* q = pf + 8 * j;   // (bmin.xyz, bit index), (bmax.xyz, -)
                    float t0x = __builtin_fmaf(q[0], nr.R.x, nox), t1x = __builtin_fmaf(q[4], nr.R.x, nox);
                    float t0y = __builtin_fmaf(q[1], nr.R.y, noy), t1y = __builtin_fmaf(q[5], nr.R.y, noy);
                    float t0z = __builtin_fmaf(q[2], nr.R.z, noz), t1z = __builtin_fmaf(q[6], nr.R.z, noz);
                    float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.f));
                    float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), o.w));
                    const bool h = !(__builtin_fmaf(tmax, 1.000002f, slack2) < tmin);
                    m |= (unsigned long long)(h ? 1u : 0u) << __float_as_uint(q[3]);
                }
                ConstPrim* pr = (ConstPrim*)sc.prims;
                for (unsigned long long am = sc.analytic_masks[0]; am; am &= am - 1) {   // spheres, planes: exact
                    const int k = __ffsll((long long)am) - 1;
                    ConstPrim& P = pr[k];
                    float root;
                    const bool sphere = P.type == AGPT_PRIM_SPHERE;
                    if (analytic_test(sphere ? 1.f : 3.f, V3(P.cx, P.cy, P.cz), sphere ? P.r2 : P.r, P.r2, nr, o.w, root)) m |= 1ull << k;
                }
                const uint32_t slot = (ring_tail + (uint32_t)lane) & (AGPT_RING - 1);
                ring_pid[slot] = npid;
                ring_mask[slot] = (nr.fast ? m : all_prims) & ~skip_bit;
            }
            ring_tail += pfc_n;
            pfc_valid = false;
            TS(ts_pref++;)
        }
        pin_state();
    };
    // Loads are issued for all 64 lanes (indices clamped to valid entries): exec-mask branches around them would make
    // the compiler's s_waitcnt placement conservative (vmcnt(0) at every merge).
    auto pump_issue = [&]() {
        // (1) everything that READS a value loaded earlier comes first ...
        const uint32_t last = count ? count - 1u : 0u;
        const bool adv_b = pfb_valid, adv_a = pfa_valid;
        uint32_t b_pid = pfb_pid, offv = pfa_off;
        const uint32_t b_n = pfb_n;
        // one unconditional touch of every register a pending load may still target: the waits for them land HERE on every
        // path (free in the steady state: those loads were issued a whole top-up ago), not behind the new loads below;
        // it also keeps the readfirstlane of the atomic's result from being hoisted up to the atomic
        asm volatile("" : "+v"(b_pid), "+v"(offv) : "v"(pfc_o.x), "v"(pfc_d.x));
        bool in_range = false;
        uint32_t base = 0, n_new = 0;
        if (adv_a) {   // the frontier atomic has returned: where is that chunk (or is the segment drained)?
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)offv);
            const uint32_t seg_begin = pfa_seg * seg_len;
            const uint32_t seg_end = seg_begin + seg_len < count ? seg_begin + seg_len : count;
            in_range = seg_begin < seg_end && off < seg_end - seg_begin;
            base = in_range ? seg_begin + off : 0u;
            n_new = in_range ? (seg_end - base < 64u ? seg_end - base : 64u) : 0u;
            if (!in_range && segs_left > 0) {
                --segs_left;
                seg = (seg + 1u) & (AGPT_FRONTIERS - 1u);
                // PEEK (launches over small batches, e.g. a rank's share of a split film): segments that other waves have drained
                // meanwhile are skipped after ONE look at all the frontiers (lane k reads frontier k, past the L1) instead of one
                // failing atomic each.  At the end of a launch every wave of the grid walks through all eight segments -- 5,120 x 8
                // same-line atomics at ~11 ns, ~56 us during which the launch only drains: 2.4 % of a rank's step at an 8-way
                // split of the 1080p film, 3.4 % at a 4-way one.  A frontier only grows: a stale value costs an atomic, never a
                // ray.  Not in the instantiation for full-size batches: the code costs the hot loop two registers and 1 ms per
                // 1080p/64spp step, more than the drain it saves there.
                if (PEEK && segs_left > 0) {
                    uint32_t head = 0;
                    if (lane < (int)AGPT_FRONTIERS) head = __hip_atomic_load(work_head + lane * AGPT_QSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t fb = (uint32_t)lane * seg_len;
                    const bool live = lane < (int)AGPT_FRONTIERS && fb < count && head < (count - fb < seg_len ? count - fb : seg_len);
                    const uint32_t live_mask = (uint32_t)__ballot(live);
                    while (segs_left > 0 && !((live_mask >> seg) & 1u)) {
                        --segs_left;
                        seg = (seg + 1u) & (AGPT_FRONTIERS - 1u);
                    }
                }
            }
        }
        // (2) ... then the new loads, back to back, with no use of a loaded value in between: a merge of branches with
        // different numbers of outstanding loads makes the compiler wait with vmcnt(0), i.e. for the loads just issued
        if (adv_b) {   // B -> C: fetch the rays of the path ids that have arrived
            pfc_pid = b_pid;
            pfc_n = b_n;
            if (LIST) {
                pfc_o.x = __uint_as_float(cand_chunks[b_pid]);
            } else {
                pfc_o = ray_o[b_pid];
                pfc_d = ray_d[b_pid];
            }
            pfc_valid = true;
        }
        pfb_valid = false;
        if (adv_a) {   // A -> B: fetch that chunk's path ids
            uint32_t idx = base + (uint32_t)lane;
            idx = idx < last ? idx : last;
            pfb_pid = queue ? queue[idx] : idx;
            pfb_n = n_new;
            pfb_valid = in_range;
        }
        pfa_valid = false;
        if (segs_left > 0) {   // A: reserve the next 64 queue entries of the current segment
            if (lane == 0) pfa_off = atomicAdd(work_head + seg * AGPT_QSTRIDE, 64u);
            pfa_seg = seg;
            pfa_valid = true;
        }
        pin_state();
    };

    for (;;) {
        TCK(const unsigned long long tk_r0 = TCK_NOW();)
        pin_state();
        any_slow = uni(any_slow);
        // ---- parked lanes: emitter pass-through ----------------------------------------------------------------
        // integrator.h:152-161's re-cast of a ray that has hit a primitive without a material, made here, on the spot, in
        // k_shade's arithmetic, where the next vertex would do nothing else: the path is spared an iteration of five launches for
        // it, and the render the trail of near-empty iterations such paths used to leave (4 ms of a 1080p/64spp step).  The new
        // ray is a Scene::Intersect call of its own (counted), walks every primitive of the list, and k_shade finds it in
        // ext_o / ext_d.  A re-cast that would need the true-division slab test is left to k_shade: the hit is written after all.
        // (Out here, not in the retire branch of the vote loop, where the same code cost the loop four registers and 1 ms.)
        if (MODE == 0 && !LIST && __ballot(cur == CUR_RECAST) != 0) {
            if (cur == CUR_RECAST) {
                const v3 p = r.O + rayt * r.D;
                const v3 nO = p + AGPT_EPSILON * r.D;
                const v3 nD = normalize(r.D);
                const TraceRay nr = make_trace_ray(nO, nD);
                if (nr.fast) {
                    const float kind = ray_d_w[pid].w;
                    if (kind == 2.f) {
                        // a camera ray (d.w = 2): the vertex first adds the emitter's radiance (integrator.h:139-147, bounces == 0;
                        // nothing of an earlier vertex is pending on it), then re-casts
                        const int al = sc.prims[hid & 0x7FFFFFFFu].arealight;
                        const v3 Le = al >= 0 ? V3(sc.lights[al].L[0], sc.lights[al].L[1], sc.lights[al].L[2]) : V3s(0.f);
                        const float4 b4 = path_beta4[pid];
                        float4 l4 = path_L4[pid];
                        const v3 L = V3(l4.x, l4.y, l4.z) + V3(b4.x, b4.y, b4.z) * Le;
                        l4.x = L.x; l4.y = L.y; l4.z = L.z;
                        path_L4[pid] = l4;
                    }
                    float4 no4, nd4;
                    no4.x = nO.x; no4.y = nO.y; no4.z = nO.z; no4.w = AGPT_FLT_MAX;
                    nd4.x = nD.x; nd4.y = nD.y; nd4.z = nD.z; nd4.w = kind;
                    ray_o_w[pid] = no4;
                    ray_d_w[pid] = nd4;
                    atomicAdd(&counters->closest_rays, 1ull);
                    r = nr;
                    rayt = AGPT_FLT_MAX;
                    hid = AGPT_HIT_MISS;
                    hb1 = 0.f;
                    hb2 = 0.f;
                    sp = 0;
                    mask = all_prims;
                    cur = pick_next(mask);
                } else {
                    DevHit h;
                    h.t = rayt;
                    h.id = hid;
                    h.b1 = hb1;
                    h.b2 = hb2;
                    hits[pid] = h;
                    cur = CUR_IDLE;
                }
            }
        }
        // ---- refill ---------------------------------------------------------------------------------------
        unsigned long long act = __ballot(cur != CUR_IDLE);
        const int n_active0 = __popcll(act);
        if (n_active0 <= 64 - refill && !(drained() && ring_head == ring_tail)) {
            const uint32_t n_idle = (uint32_t)(64 - n_active0);
            // top the ring up: one pump_consume() per 64 rays in the steady state; a cold pipeline is pumped synchronously
            if (ring_tail - ring_head < n_idle && pfc_valid) pump_consume();
            while (ring_tail - ring_head < n_idle && !drained()) {
                pump_issue();
                pump_consume();
            }
            // hand ring entries to idle lanes (same wave wrote them: a wave barrier orders the LDS writes and reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t avail = ring_tail - ring_head;
            const uint32_t take = n_idle < avail ? n_idle : avail;
            if (cur == CUR_IDLE) {
                const uint32_t rank = (uint32_t)__popcll(~act & lt_mask);
                if (rank < take) {
                    const uint32_t slot = (ring_head + rank) & (AGPT_RING - 1);
                    pid = ring_pid[slot];
                    mask = ring_mask[slot];
                    if (LIST) {   // the ring word says which chunks hold candidates: start on the first of them
                        cvisit = (uint32_t)mask;
                        cchunk = 0;
                        mask = 0;
                        mmask = 0;
                        if (cvisit) enter_chunk((uint32_t)__ffs((int)cvisit) - 1u);
                    }
                    float4 o = ray_o[pid], d = ray_d[pid];
                    r = make_trace_ray(V3(o.x, o.y, o.z), V3(d.x, d.y, d.z));
                    rayt = o.w;
                    if (MIS) {
                        mis_area = false;
                        mis_skip = -1;
                        mis_reach = true;
                        const uint32_t shape = __float_as_uint(o.w);   // the sampled light's sphere, ~0u = infinite light
                        rayt = AGPT_FLT_MAX;
                        if (shape != 0xFFFFFFFFu) {
                            v3 lc;
                            float lr2, ts;
                            light_sphere((int)shape, lc, lr2);
                            mis_area = true;
                            mis_skip = (int)shape;
                            mis_reach = sphere_test_c(lc, lr2, r, rayt, ts);
                            if (mis_reach) rayt = ts;
                        }
                    }
                    hid = AGPT_HIT_MISS;
                    hb1 = 0.f;
                    hb2 = 0.f;
                    sp = 0;
                    cur = pick_next(mask);
                }
            }
            ring_head += take;
            TS(ts_refills++; ts_refilled += take;)
            if (!pfc_valid && !drained()) pump_issue();   // refill the emptied stage: nothing below waits for these loads
            act = __ballot(cur != CUR_IDLE);
            any_slow = __ballot(cur != CUR_IDLE && !r.fast) != 0;
        }
        TCK(tk_refill += TCK_NOW() - tk_r0;)
        if (act == 0) break;

        // ---- traversal round: vote scheduling -------------------------------------------------------------------
        // Each lane is in one of three states: A = at a sphere / plane of its candidate list or at the list's end, B = at
        // an interior node (or a mesh's root pair), C = at a leaf.  Every step the wave executes the body most lanes are
        // waiting for; a lane's own sequence of operations is the reference's, only the interleaving between lanes
        // changes.  (A while-while loop makes every lane wait for the longest descent in the wave: 37 % active lanes.)
        for (;;) {
            TCK(const unsigned long long tk_v0 = TCK_NOW();)
            const int nA = __popcll(__ballot(LIST ? cur - CUR_FETCH < 2u : cur == CUR_PICK)), nB = __popcll(__ballot(cur < CUR_B_END)),
                      nC = __popcll(__ballot((int32_t)cur < 0));
            const int n_active = nA + nB + nC;
            if (n_active == 0 || (n_active <= 64 - refill && !(drained() && ring_head == ring_tail))) break;
            // (nothing left to refill with: lanes parked for a re-cast would wait for the last active lane of the wave)
            if (MODE == 0 && !LIST && n_active <= 64 - refill && __ballot(cur == CUR_RECAST) != 0) break;
            TCK(const unsigned long long tk_v1 = TCK_NOW(); tk_vote += tk_v1 - tk_v0;)
            TS(ts_act += (unsigned long long)n_active; if (!(nB >= nA && nB >= nC) && !(nC >= nA)) { ts_steps[0]++; ts_lanes[0] += nA; })
            if (nB >= nA && nB >= nC) {
                TS(ts_steps[1]++; ts_lanes[1] += nB;)
                if (cur < CUR_B_END) {
                    // the wave is about to wait for these loads whatever happens: let its issue win the arbitration against
                    // waves in the middle of their arithmetic (-1.5 ms per C3 step on the closest-hit and MIS launches; the
                    // short any-hit steps lose 0.3 ms with it)
                    if (MODE != 1) __builtin_amdgcn_s_setprio(1);
                    const NodePair np = load_pair(sc, cur);
                    if (MODE != 1) {
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_s_setprio(0);
                    }
                    TCK(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tk_bmem += TCK_NOW() - tk_v1;)
                    float dl, dr;
                    bool hl, hr;
                    const uint32_t encl = __float_as_uint(np.n3.x), encr = __float_as_uint(np.n3.y);
                    pair_boxes_fast(np, r, rayt, hl, hr, dl, dr);
                    if (any_slow) {  // scalar branch: rays with a direction component below 2^-40 (true divisions)
                        if (!r.fast) pair_boxes_exact(np, r, rayt, hl, hr, dl, dr);
                    }
                    if (COUNT) {
                        if (cur < rootpair_base) c_int++; else c_root++;
                    }
                    hr = hr && cur < rootpair_base;   // a root pair has no right child
                    if (hl && hr) {
                        bool swap = ANY ? false : (dr < dl);
                        stack_push(sp, swap ? encl : encr);
                        sp++;
                        TS(ts_push[0]++; ts_push[1] += sp > 4; ts_push[2] += sp > 6; ts_push[3] += sp > 8; ts_push[4] += sp > 12;)
                        cur = swap ? encr : encl;
                    } else if (hl) {
                        cur = encl;
                    } else if (hr) {
                        cur = encr;
                    } else if (sp == 0) {
                        cur = pick_next(mask);
                    } else {
                        sp--;
                        cur = stack_at(sp);
                    }
                }
                TCK(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tk_balu += TCK_NOW() - tk_v1;)
            } else if (nC >= nA) {
                TS(ts_steps[2]++; ts_lanes[2] += nC;)
                if ((int32_t)cur < 0) {
                    uint32_t first, cnt;
                    leaf_range(sc, cur, first, cnt);
                    bool done = false;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const float4* tp = sc.tri_verts + 3 * (size_t)(first + i);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        const v3 q0 = V3(a.x, a.y, a.z), q1 = V3(b.x, b.y, b.z), q2 = V3(c.x, c.y, c.z);
                        const uint32_t q_id = __float_as_uint(a.w), q_flags = __float_as_uint(b.w);
                        float t, b1, b2;
                        if (COUNT) c_tri++;
                        if (tri_test(q0, q1, q2, r, rayt, t, b1, b2)) {
                            if (MIS) {
                                if (!(q_flags & AGPT_TRI_FLAG_REJECT)) {
                                    done = true;
                                    break;
                                }
                            } else if (ANY) {
                                done = true;
                                break;
                            } else if (!(q_flags & AGPT_TRI_FLAG_REJECT)) {
                                rayt = t;
                                hid = q_id;
                                hb1 = b1;
                                hb2 = b2;
                            }
                        }
                    }
                    if (ANY && done) {
                        occluded[pid] = MIS ? 0u : 1u;   // MODE 2 writes "ok" (1 = add the BSDF-leg contribution)
                        cur = CUR_IDLE;
                    } else if (sp == 0) {
                        cur = pick_next(mask);
                    } else {
                        sp--;
                        cur = stack_at(sp);
                    }
                }
                TCK(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tk_c += TCK_NOW() - tk_v1;)
            } else if (LIST && cur == CUR_FETCH) {
                // the next chunk that holds candidates of this ray (one dependent load; list steps are rare)
                enter_chunk((uint32_t)__ffs((int)cvisit) - 1u);
                cur = pick_next(mask);
                TCK(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tk_a += TCK_NOW() - tk_v1;)
            } else if (cur == CUR_PICK) {
                // end of the primitive list (retire), or the sphere / plane that is next in list order
                if (mask == 0) {
                    bool recast = false;
                    if (MIS) {
                        occluded[pid] = mis_reach ? 1u : 0u;
                    } else if (ANY) {
                        occluded[pid] = 0u;
                    } else {
                        // A closest hit on a primitive without a material (an emitter's sphere), on a ray whose next vertex would
                        // do nothing but re-cast it (d.w = 1, set by k_shade): the lane is parked for the re-cast at the top of the
                        // outer loop instead of retiring.
                        if (!LIST && ray_o_w && hid != AGPT_HIT_MISS && (hid & AGPT_HIT_SPHERE))
                            recast = s_prim[2 * (hid & 0x7FFFFFFFu) + 1].z != 0.f && ray_d_w[pid].w != 0.f;
                        if (!recast) {
                            DevHit h;
                            h.t = rayt;
                            h.id = hid;
                            h.b1 = hb1;
                            h.b2 = hb2;
                            hits[pid] = h;
                        }
                    }
                    cur = recast ? CUR_RECAST : CUR_IDLE;
                } else {
                    const int k = (LIST ? 64 * (int)cchunk : 0) + __ffsll((long long)mask) - 1;   // list index
                    mask &= mask - 1;
                    float4 pa, pb2;
                    if (!PRIM_LDS) {   // (the short-list instantiation keeps these records in LDS)
                        const DevPrim& P = sc.prims[k];
                        const bool sphere = P.type == AGPT_PRIM_SPHERE;
                        pa.x = P.cx; pa.y = P.cy; pa.z = P.cz; pa.w = sphere ? 1.f : (P.type == AGPT_PRIM_PLANE ? 3.f : 2.f);
                        pb2.x = sphere ? P.r2 : P.r; pb2.y = sphere ? 0.f : P.r2; pb2.z = 0.f; pb2.w = 0.f;
                    } else {
                        pa = s_prim[2 * k];
                        pb2 = s_prim[2 * k + 1];
                    }
                    if (pa.w == 1.f || pa.w == 3.f) {
                        float root;
                        if (analytic_test(pa.w, V3(pa.x, pa.y, pa.z), pb2.x, pb2.y, r, rayt, root)) {
                            if (MIS) {
                                // a sphere listed before the light's sphere only wins a strict comparison (see above)
                                const bool tie_loses = pa.w == 1.f && mis_area && mis_reach && k < mis_skip && root == rayt;
                                if (!tie_loses) {
                                    occluded[pid] = 0u;
                                    cur = CUR_IDLE;
                                }
                            } else if (ANY) {
                                occluded[pid] = 1u;
                                cur = CUR_IDLE;
                            } else {
                                rayt = root;
                                hid = AGPT_HIT_SPHERE | (uint32_t)k;
                                hb1 = 0.f;
                                hb2 = 0.f;
                            }
                        }
                    }
                    if (cur == CUR_PICK) cur = pick_next(mask);   // (meshes never get here: pick_next routes them to B)
                }
                TCK(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tk_a += TCK_NOW() - tk_v1;)
            }
        }
    }
    TS(if (lane == 0) {
        unsigned long long* d = counters->dbg + 16 * MODE;
        for (int k = 0; k < 3; ++k) { atomicAdd(d + k, ts_steps[k]); atomicAdd(d + 3 + k, ts_lanes[k]); }
        atomicAdd(d + 6, ts_act); atomicAdd(d + 7, ts_refills); atomicAdd(d + 8, ts_refilled); atomicAdd(d + 9, ts_pref);
    })
#if defined(AGPT_TRACE_STATS) && !defined(AGPT_TRACE_CLOCK)
    {   // stack-depth histogram of the pushes (per lane; slots 10..14 are the phase clocks' in the clock build)
        unsigned long long* d = counters->dbg + 16 * MODE;
        for (int k = 0; k < 5; ++k) {
            uint32_t v = ts_push[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) atomicAdd(d + 10 + k, (unsigned long long)v);
        }
    }
#endif
    TS(if (lane == 0) {
        unsigned long long* d = counters->dbg + 16 * MODE;
        TCK(atomicAdd(d + 10, tk_refill); atomicAdd(d + 11, tk_vote); atomicAdd(d + 12, tk_bmem); atomicAdd(d + 13, tk_balu);
            atomicAdd(d + 14, tk_c); atomicAdd(d + 15, tk_a); atomicAdd(counters->dbg + 48 + MODE, TCK_NOW() - tk_begin);)
    })
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            c_int += __shfl_down(c_int, off);
            c_root += __shfl_down(c_root, off);
            c_tri += __shfl_down(c_tri, off);
        }
        if (lane == 0) {
            atomicAdd(&counters->interior, (unsigned long long)c_int);
            atomicAdd(&counters->roots, (unsigned long long)c_root);
            atomicAdd(&counters->tris, (unsigned long long)c_tri);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (MODE == 1)
            atomicAdd(&counters->anyhit_rays, (unsigned long long)count);
        else
            atomicAdd(&counters->closest_rays, (unsigned long long)count);   // MIS queries are Scene::Intersect calls
    }
}

// ---------------------------------------------------------------------------------------------------------
// Top-level structure over Scene::primitives (scene.h:5-19) for lists longer than 64 entries: one thread per ray walks the
// tree over the meshes' root boxes (agpt_scene.h: depth-first order with skip links, so the walk needs no stack -- hit ->
// next node, miss -> past the subtree) with the conservative slab test of the short-list prefilter, tests the spheres and
// planes exactly, and leaves the ray's candidates as one 64-bit word per chunk of 64 primitives (cand_mask[chunk * stride +
// path]) plus the set of non-empty chunks (cand_chunks[path]).  The candidates are a superset of the primitives the
// reference's list walk can hit at the ray's initial t; k_trace_fast<LIST> visits them in list order and re-runs the exact
// root-box test where the reference runs it.  No LDS besides the per-thread words and few registers: the dependent loads of
// the walk are hidden by occupancy, which the traversal kernel (5 waves per SIMD) could not do (a per-lane tree walk inside
// its prefilter measured 15 % slower than the per-primitive loop).
static_assert(AGPT_MAX_CHUNKS <= 32, "k_trace_fast<LIST> and k_candidates keep the chunks still to visit in one 32-bit word per ray");
#define AGPT_CAND_LIST 24   // mesh candidates a thread of k_candidates collects before it gives up and marks every primitive
template <int MODE>
__global__ void __launch_bounds__(AGPT_BLOCK)
k_candidates(DevScene sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t count_imm,
             const float4* __restrict__ ray_o, const float4* __restrict__ ray_d, unsigned long long* __restrict__ cand_mask,
             uint32_t* __restrict__ cand_chunks, uint32_t cand_stride) {
    constexpr bool MIS = MODE == 2;
    // the meshes the walk finds, as list indices, one column per thread (12 KiB per block: the register count, not LDS,
    // bounds the occupancy); a ray with more than AGPT_CAND_LIST of them gets every primitive as candidate -- any superset
    // of the true candidates is valid, the traversal re-tests each root box exactly
    __shared__ uint16_t s_list[AGPT_CAND_LIST][AGPT_BLOCK];
    const uint32_t count = count_ptr ? *count_ptr : count_imm;
    const int n_chunks = (sc.n_prims + 63) / 64;
    const int tid = threadIdx.x;
    for (uint32_t i = blockIdx.x * AGPT_BLOCK + tid; i < count; i += gridDim.x * AGPT_BLOCK) {
        const uint32_t pid = queue ? queue[i] : i;
        float4 o = ray_o[pid];
        const float4 d = ray_d[pid];
        const TraceRay nr = make_trace_ray(V3(o.x, o.y, o.z), V3(d.x, d.y, d.z));
        uint32_t skip = 0xFFFFFFFFu;
        if (MIS) {   // MIS rays carry the sampled light's sphere (primitive index, ~0u = infinite light) in o.w; tmax = inf
            const uint32_t shape = __float_as_uint(o.w);
            o.w = AGPT_FLT_MAX;
            if (shape != 0xFFFFFFFFu) {
                const DevPrim& L = sc.prims[shape];
                float ts;
                if (sphere_test_c(V3(L.cx, L.cy, L.cz), L.r2, nr, o.w, ts)) o.w = ts;
                skip = shape;   // never a blocker itself
            }
        }
        int n_found = 0;
        bool all = !nr.fast;   // a direction component below 2^-40: the conservative form does not hold
        if (!all) {
            // same conservative form and slack as the short-list prefilter (k_trace_fast: pump_consume)
            const float nox = -(nr.O.x * nr.R.x), noy = -(nr.O.y * nr.R.y), noz = -(nr.O.z * nr.R.z);
            const float slack2 = 0x1p-21f * fmaxf(fmaxf(fabsf(nox), fabsf(noy)), fabsf(noz));
            uint32_t k = 0;
            const uint32_t k_end = (uint32_t)sc.n_toplevel;
            while (k < k_end) {
                const uint4 nd = sc.toplevel[k];   // six halves (box rounded outward) + skip | leaf << 16
                auto h2f = [](uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); };
                const float lox = h2f(nd.x & 0xFFFFu), loy = h2f(nd.x >> 16), loz = h2f(nd.y & 0xFFFFu);
                const float hix = h2f(nd.y >> 16), hiy = h2f(nd.z & 0xFFFFu), hiz = h2f(nd.z >> 16);
                float t0x = __builtin_fmaf(lox, nr.R.x, nox), t1x = __builtin_fmaf(hix, nr.R.x, nox);
                float t0y = __builtin_fmaf(loy, nr.R.y, noy), t1y = __builtin_fmaf(hiy, nr.R.y, noy);
                float t0z = __builtin_fmaf(loz, nr.R.z, noz), t1z = __builtin_fmaf(hiz, nr.R.z, noz);
                float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.f));
                float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), o.w));
                const bool h = !(__builtin_fmaf(tmax, 1.000002f, slack2) < tmin);
                const uint32_t prim = nd.w >> 16;
                if (h && prim != 0xFFFFu) {
                    if (n_found < AGPT_CAND_LIST) s_list[n_found][tid] = (uint16_t)prim;
                    n_found++;
                }
                k = h ? k + 1u : (nd.w & 0xFFFFu);
            }
            all = n_found > AGPT_CAND_LIST;
        }
        uint32_t chunks = 0;
        for (int c = 0; c < n_chunks; ++c) {
            unsigned long long m;
            if (all) {
                const int n_here = sc.n_prims - 64 * c < 64 ? sc.n_prims - 64 * c : 64;
                m = n_here >= 64 ? ~0ull : ((1ull << n_here) - 1ull);
            } else {
                m = 0;
                for (int j = 0; j < n_found; ++j) {
                    const uint32_t prim = s_list[j][tid];
                    if ((int)(prim >> 6) == c) m |= 1ull << (prim & 63u);
                }
                for (unsigned long long am = sc.analytic_masks[c]; am; am &= am - 1) {   // spheres, planes: exact, as in the short-list prefilter
                    const int b = __ffsll((long long)am) - 1;
                    const DevPrim& P = sc.prims[64 * c + b];
                    float root;
                    const bool sphere = P.type == AGPT_PRIM_SPHERE;
                    if (analytic_test(sphere ? 1.f : 3.f, V3(P.cx, P.cy, P.cz), sphere ? P.r2 : P.r, P.r2, nr, o.w, root)) m |= 1ull << b;
                }
            }
            if (MIS && (int)(skip >> 6) == c) m &= ~(1ull << (skip & 63u));
            cand_mask[(size_t)c * cand_stride + pid] = m;
            chunks |= (m ? 1u : 0u) << c;
        }
        cand_chunks[pid] = chunks;
    }
}

// agpt_intersect_batch: Ray ctor (camera.h:6) normalises D
__global__ void k_prepare_rays(const agpt_ray* __restrict__ in, int n, float4* __restrict__ ray_o, float4* __restrict__ ray_d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    agpt_ray r = in[i];
    v3 d = normalize(V3(r.d[0], r.d[1], r.d[2]));
    float4 o4, d4;
    o4.x = r.o[0]; o4.y = r.o[1]; o4.z = r.o[2]; o4.w = r.tmax;
    d4.x = d.x; d4.y = d.y; d4.z = d.z; d4.w = 0.f;
    ray_o[i] = o4;
    ray_d[i] = d4;
}
__global__ void k_export_hits(const DevScene sc, const DevHit* __restrict__ hits, const uint32_t* __restrict__ occluded,
                              int n, int any_hit, agpt_hit* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    agpt_hit o;
    o.hit = 0; o.prim = -1; o.tri = -1; o.t = 0; o.b1 = 0; o.b2 = 0;
    if (any_hit) {
        o.hit = occluded[i] ? 1 : 0;
    } else {
        DevHit h = hits[i];
        if (h.id != AGPT_HIT_MISS) {
            o.hit = 1;
            o.t = h.t;
            if (h.id & AGPT_HIT_SPHERE) {
                o.prim = (int)(h.id & 0x7FFFFFFFu);
            } else {
                int prim = (int)__float_as_uint(sc.tri_shade[4 * (size_t)h.id + 3].w);
                o.prim = prim;
                o.tri = 3 * ((int)h.id - sc.prims[prim].tri_base);
                o.b1 = h.b1;
                o.b2 = h.b2;
            }
        }
    }
    out[i] = o;
}

// ---------------------------------------------------------------------------------------------------------
// myapp.cpp:165-167: jittered film position -> Camera::GetRay (camera.h:58-64; lens: common.h:65-71)
__global__ void __launch_bounds__(AGPT_BLOCK)
k_generate(DevScene sc, RenderConsts rc, PathBuffers pb, Queues q) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total = rc.NP * (uint32_t)rc.S;
    if (i >= total) return;
    const uint32_t G = sample_group(rc.S);
    const uint32_t sj = i % G, tt = i / G;
    const uint32_t sg = tt / rc.NP, p = tt - sg * rc.NP;
    const uint32_t sl = sg * G + sj;
    int x, y;
    size_t unused_index;
    pixel_of(rc, p, x, y, unused_index);
    uint32_t rng = sample_seed((uint32_t)(y * rc.W + x), (uint32_t)(rc.W * rc.H), (uint32_t)(rc.s0 + (int)sl), rc.seed_base);
    float px = x + rng_float(rng);
    float py = y + rng_float(rng);
    float s = px / rc.W, t = py / rc.H;
    const DevCamera& c = sc.cam;
    v3 rd = V3s(0.f);
    if (c.lens_radius > 0.f) {
        for (;;) {
            float a = -1.f + (1.f - -1.f) * rng_float(rng);
            float b = -1.f + (1.f - -1.f) * rng_float(rng);
            v3 pd = V3(a, b, 0);
            if (sqrlen(pd) >= 1) continue;
            rd = c.lens_radius * pd;
            break;
        }
    }
    v3 offset = c.u * rd.x + c.v * rd.y;
    v3 pixel = c.lower_left_corner + s * c.horizontal + t * c.vertical;
    v3 O = c.origin + offset;
    v3 D = normalize(pixel - c.origin - offset);
    float4 o4, d4, b4, l4;
    o4.x = O.x; o4.y = O.y; o4.z = O.z; o4.w = AGPT_FLT_MAX;
    d4.x = D.x; d4.y = D.y; d4.z = D.z;
    d4.w = rc.max_depth > 0 ? 2.f : 0.f;   // 2: a camera ray that may be re-cast through emitters by the trace kernel (see k_trace_fast)
    b4.x = 1.f; b4.y = 1.f; b4.z = 1.f; b4.w = __uint_as_float(0u);
    l4.x = 0.f; l4.y = 0.f; l4.z = 0.f; l4.w = __uint_as_float(rng);
    pb.ext_o[i] = o4;
    pb.ext_d[i] = d4;
    pb.beta4[i] = b4;
    pb.L4[i] = l4;
    q.active[i] = i;
    q.ext[i] = i;
    if (i == 0) {
        QCOUNT(q, 0) = total;
        QCOUNT(q, 1) = total;
        QCOUNT(q, 2) = 0;
        QCOUNT(q, 3) = 0;
        QCOUNT(q, 4) = 0;
    }
}

// agpt_li_batch: Integrator::Li(ray, scene) (integrator.h:28-31) for n caller-supplied rays, each with its own RandomFloat() stream
// (an xorshift32 state, template/template.cpp:667-675).  The Ray ctor normalises D (camera.h:6); ray.t is the caller's tmax.
__global__ void __launch_bounds__(AGPT_BLOCK)
k_generate_li(const agpt_ray* __restrict__ rays, const uint32_t* __restrict__ rng_states, uint32_t n, PathBuffers pb, Queues q, int max_depth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const agpt_ray r = rays[i];
    const v3 d = normalize(V3(r.d[0], r.d[1], r.d[2]));
    float4 o4, d4, b4, l4;
    o4.x = r.o[0]; o4.y = r.o[1]; o4.z = r.o[2]; o4.w = r.tmax;
    d4.x = d.x; d4.y = d.y; d4.z = d.z;
    d4.w = max_depth > 0 ? 2.f : 0.f;   // (as in k_generate)
    b4.x = 1.f; b4.y = 1.f; b4.z = 1.f; b4.w = __uint_as_float(0u);
    l4.x = 0.f; l4.y = 0.f; l4.z = 0.f; l4.w = __uint_as_float(rng_states[i]);
    pb.ext_o[i] = o4;
    pb.ext_d[i] = d4;
    pb.beta4[i] = b4;
    pb.L4[i] = l4;
    q.active[i] = i;
    q.ext[i] = i;
    if (i == 0) {
        QCOUNT(q, 0) = n;
        QCOUNT(q, 1) = n;
        QCOUNT(q, 2) = 0;
        QCOUNT(q, 3) = 0;
        QCOUNT(q, 4) = 0;
    }
}
// Li's return value as it is (the NaN / inf filter belongs to MyApp::Tick, myapp.cpp:169-172, not to the integrator) and the
// stream's state after the path, so that a host can go on drawing from it
__global__ void k_export_li(PathBuffers pb, uint32_t n, float* __restrict__ radiance3, uint32_t* __restrict__ rng_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l4 = pb.L4[i];
    radiance3[3 * i] = l4.x;
    radiance3[3 * i + 1] = l4.y;
    radiance3[3 * i + 2] = l4.z;
    if (rng_out) rng_out[i] = __float_as_uint(l4.w);
}

// ---------------------------------------------------------------------------------------------------------
// myapp.cpp:169-173 NaN/inf reject + Accumulator::AddSample (myapp.h:17-19): samples are added in sample order
__global__ void __launch_bounds__(AGPT_BLOCK)
k_accumulate(RenderConsts rc, PathBuffers pb, float4* __restrict__ accum, DevCounters* __restrict__ counters) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rc.NP) return;
    int x, y;
    size_t ai;
    pixel_of(rc, p, x, y, ai);
    float4 a = accum[ai];
    uint32_t bad = 0;
    const uint32_t G = sample_group(rc.S);
    auto add = [&](float4 l4) {   // (one sample; the order of the additions is the sample order)
        v3 clr = V3(l4.x, l4.y, l4.z);
        if (isnan(clr.x) || isnan(clr.y) || isnan(clr.z) || isinf(luminance(clr))) {
            clr = V3s(0.f);
            bad++;
        }
        a.x += clr.x;
        a.y += clr.y;
        a.z += clr.z;
    };
    if ((G & 3u) == 0u) {
        // a pixel's samples lie in runs of G consecutive records: fetch a whole 64-byte line (4 samples) per step
        for (uint32_t sg = 0; sg < (uint32_t)rc.S / G; ++sg) {
            const float4* run = pb.L4 + ((size_t)sg * rc.NP + p) * G;
            for (uint32_t j = 0; j < G; j += 4) {
                const float4 l0 = run[j], l1 = run[j + 1], l2 = run[j + 2], l3 = run[j + 3];
                add(l0);
                add(l1);
                add(l2);
                add(l3);
            }
        }
    } else {
        for (int s = 0; s < rc.S; s++) add(pb.L4[((size_t)((uint32_t)s / G) * rc.NP + p) * G + (uint32_t)s % G]);
    }
    accum[ai] = a;
    if (bad) atomicAdd(&counters->outliers, (unsigned long long)bad);  // outliers are rare
}

// Accumulator::CopyToSurface (myapp.h:34-41) with lin2rgb / rgb2uint (template/common.h:41-51)
__global__ void k_resolve(const float4* __restrict__ accum, int n, int samples, uint32_t* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = accum[i];
    float e = 1 / 2.2f;
    float inv = (float)samples;
    float r = powf(a.x / inv, e), g = powf(a.y / inv, e), b = powf(a.z / inv, e);
    int ri = (int)(256 * tclampf(r, 0.0f, 0.999f));
    int gi = (int)(256 * tclampf(g, 0.0f, 0.999f));
    int bi = (int)(256 * tclampf(b, 0.0f, 0.999f));
    out[i] = (uint32_t)((ri << 16) + (gi << 8) + bi);
}

// ---------------------------------------------------------------------------------------------------------
// known-answer kernels (one lane per case)
__device__ __forceinline__ void kat_frame(Bsdf& b, const DevMaterial& m) {
    // canonical SurfaceInteraction: dpdu = +x, dpdv = +y -> n = ns = +z, ss = +x
    bsdf_setup(b, m, V3(0, 0, 1), V3(0, 0, 1), V3(1, 0, 0));
}
__global__ void k_kat_bsdf_eval(DevScene sc, int material, int n, const float* __restrict__ wo3, const float* __restrict__ wi3,
                                float* __restrict__ f3o, float* __restrict__ pdfo) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevMaterial& m = sc.materials[material];
    Bsdf b;
    kat_frame(b, m);
    v3 wo = V3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), wi = V3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);
    v3 f;
    float pdf;
    bsdf_f_pdf(b, m, bsdf_wo_cache(b, m, wo), wo, wi, true, &f, &pdf);
    f3o[3 * i] = f.x; f3o[3 * i + 1] = f.y; f3o[3 * i + 2] = f.z;
    pdfo[i] = pdf;
}
__global__ void k_kat_bsdf_sample(DevScene sc, int material, int n, const float* __restrict__ wo3, const float* __restrict__ u2,
                                  float* __restrict__ wi3o, float* __restrict__ f3o, float* __restrict__ pdfo,
                                  int32_t* __restrict__ speco) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevMaterial& m = sc.materials[material];
    Bsdf b;
    kat_frame(b, m);
    v3 wo = V3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
    v3 wi = V3s(0.f);
    float pdf = 0;
    bool spec = false;
    v3 f = bsdf_sample_f(b, m, bsdf_wo_cache(b, m, wo), wo, &wi, u2[2 * i], u2[2 * i + 1], &pdf, false, &spec);
    wi3o[3 * i] = wi.x; wi3o[3 * i + 1] = wi.y; wi3o[3 * i + 2] = wi.z;
    f3o[3 * i] = f.x; f3o[3 * i + 1] = f.y; f3o[3 * i + 2] = f.z;
    pdfo[i] = pdf;
    speco[i] = spec ? 1 : 0;
}
// known-answer kernel: Distribution1D::SampleContinuous (env_sample_continuous) for k draws, one lane each
__global__ void k_kat_distribution1d(DevEnv e, const float* __restrict__ u, int k, float* __restrict__ x_out, float* __restrict__ pdf_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    float pdf;
    x_out[i] = env_sample_continuous(e, u[i], &pdf);
    pdf_out[i] = pdf;
}
__global__ void k_kat_rng(uint32_t pixel, uint32_t wh, uint32_t sample, uint32_t seed_base, int n, float* __restrict__ out,
                          uint32_t* __restrict__ seed_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t s = sample_seed(pixel, wh, sample, seed_base);
    *seed_out = s;
    for (int i = 0; i < n; i++) out[i] = rng_float(s);
}
