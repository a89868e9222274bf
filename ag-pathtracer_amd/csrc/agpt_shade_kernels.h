// agpt_shade_kernels.h -- the shading kernels of the wavefront path tracer (see agpt_kernels.h for the iteration they are part
// of): k_shade = PathTracer::Li's loop body (integrator.h:132-188) for every live path, k_resolve_pending for the ended ones.
// Compiled as their own translation unit (agpt_shade_kernels.hip) with MachineLICM off: the pass hoists the materialisation of
// every fp64 polynomial coefficient of the trigonometry (two v_mov each) out of the path loop and keeps the pairs in registers
// for the whole kernel -- 168 registers and 7-14 spilled, against 134 and none without it, which is what lets the kernel be
// held to four waves per SIMD (128 registers, 2 spilled).
#pragma once

#include "agpt_shade.h"
#include "agpt_trace.h"   // TraceRay + sphere_test_c: MIS queries answered without a traversal
#include "agpt_wavefront.h"

// ---------------------------------------------------------------------------------------------------------
// PathTracer::Li's loop body (integrator.h:132-188) in wavefront form, for one path.  Returns which of the next
// iteration's queues the path goes to.
#define SB_ACTIVE 1u
#define SB_EXT 2u
#define SB_MIS 4u
#define SB_SHADOW 8u
#define SB_RESOLVE 16u        // (bits 0..4 = the queue the path is appended to, in Queues order)
#define SB_SHADED 32u
#define SB_MIS_ANSWERED 64u   // a MIS query answered in k_shade (counted as the Scene::Intersect call it replaces, not enqueued)
#define SB_RAY_ANSWERED 128u  // a continuation ray whose hit nothing reads (same bookkeeping)

// The light sample a vertex left pending (EstimateDirect's two terms, integrator.h:38-105), now that its shadow ray and its
// MIS query have been traced: c1 if the shadow ray got through, c2 if the MIS query's answer is yes.
__device__ __forceinline__ v3 pending_light_sample(const DevScene& sc, const RenderConsts& rc, const PathBuffers& pb, uint32_t pid,
                                                   uint32_t flags, uint32_t was_occluded, uint32_t was_mis_ok, float4 c1_4, float4 c2_4,
                                                   float4 f4) {
    v3 Ld = V3s(0.f);
    if (flags & PF_PEND_C1) {
        if (!was_occluded) {  // VisibilityTester::Unoccluded, lights.cpp:10-12
            Ld = Ld + V3(c1_4.x, c1_4.y, c1_4.z);
        }
    }
    if (flags & PF_PEND_C2) {
        uint32_t light = __float_as_uint(f4.w);
        bool add;
        if (rc.mis_mode) {
            add = was_mis_ok != 0;
        } else if (uint32_t hid = pb.mis_hit[pid].id; hid != AGPT_HIT_MISS) {  // integrator.h:82-85: hit shape's area light must be the sampled light
            int hp = (hid & AGPT_HIT_SPHERE) ? (int)(hid & 0x7FFFFFFFu)
                                             : (int)__float_as_uint(sc.tri_shade[4 * (size_t)hid + 3].w);
            add = sc.prims[hp].arealight == (int)light;
        } else {  // :86-87 light.Le(ray): non-zero only for an IsInfinite() light
            add = sc.lights[light].type != AGPT_LIGHT_AREA;
        }
        if (add) Ld = Ld + V3(c2_4.x, c2_4.y, c2_4.z);
    }
    return Ld;
}

// developer build -DAGPT_SHADE_CLOCK: s_memtime stamps at the phase boundaries of shade_path, summed over waves into
// counters->dbg[56..63] (phase k ends at SCK(k); the stamp waits for outstanding memory so a phase owns its own latency)
#ifdef AGPT_SHADE_CLOCK
#define SCK_PARAMS , unsigned long long* sck_acc, unsigned long long& sck_last
#define SCK_ARGS , sck_acc, sck_last
#define SCK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long sck_now = __builtin_amdgcn_s_memtime(); \
                    sck_acc[k] += sck_now - sck_last; sck_last = sck_now; } while (0)
#else
#define SCK_PARAMS
#define SCK_ARGS
#define SCK(k)
#endif
// ENV: the scene has an InfiniteAreaLight (lights.cpp:31-112).  Its sampling / pdf / Le helpers are out-of-line calls whose mere
// presence costs the register allocation of every path (values live across a call site go to callee-saved registers); scenes
// without an environment map -- every BASELINE configuration -- run the instantiation that does not contain them.
template <bool ENV>
__device__ __forceinline__ uint32_t shade_path(const DevScene& sc, const RenderConsts& rc, const PathBuffers& pb, uint32_t pid SCK_PARAMS) {
    uint32_t bits = 0;
    // Every per-path record is fetched up front, unconditionally: the loads then overlap in one memory round trip instead
    // of forming a chain of dependent ones behind the flag tests (a record that turns out not to be needed is a coalesced
    // 16-B read of a valid slot -- every path id owns a slot in every array).
    float4 b4 = pb.beta4[pid];
    float4 l4 = pb.L4[pid];
    const float4 f4 = pb.fac4[pid];
    const uint32_t was_occluded = pb.occluded[pid];
    const uint32_t was_mis_ok = pb.mis_ok[pid];
    const float4 c1_4 = pb.c1[pid], c2_4 = pb.c2[pid];
    const DevHit h = pb.hit[pid];
    const float4 o4 = pb.ext_o[pid], d4 = pb.ext_d[pid];
    v3 beta = V3(b4.x, b4.y, b4.z);
    v3 L = V3(l4.x, l4.y, l4.z);
    uint32_t flags = __float_as_uint(b4.w);
    uint32_t rng = __float_as_uint(l4.w);

    SCK(0);   // records arrived
    // From here on the path id is a fresh value for the compiler: the addresses of the stores below are then formed where they
    // are used (one 64-bit add each) instead of being kept from the loads above across the whole body -- seven address pairs
    // lived in registers and five more went to scratch, reloaded on every path.
    asm volatile("" : "+v"(pid));
    // ---- resolve the previous vertex' UniformSampleOneLight (integrator.h:38-105) ------------------
    if (flags & PF_PENDING) {
        const v3 Ld = pending_light_sample(sc, rc, pb, pid, flags, was_occluded, was_mis_ok, c1_4, c2_4, f4);
        float lightPdf = 1.f / sc.n_lights;           // integrator.h:100,104
        L = L + beta * (Ld / lightPdf);               // integrator.h:166
        if (!(flags & PF_DEAD)) beta = beta * V3(f4.x, f4.y, f4.z);  // integrator.h:176
        flags &= ~(PF_PENDING | PF_PEND_C1 | PF_PEND_C2);
    }

    SCK(1);   // pending light sample resolved
    if (!(flags & PF_DEAD)) {
        uint32_t bounces = flags & PF_BOUNCE_MASK;
        v3 O = V3(o4.x, o4.y, o4.z), D = V3(d4.x, d4.y, d4.z);
        const bool found = h.id != AGPT_HIT_MISS;
        Surface s;
        if (found) {
            if (h.id & AGPT_HIT_SPHERE)
                surface_from_sphere(sc, (int)(h.id & 0x7FFFFFFFu), O, D, h.t, s);
            else
                surface_from_triangle(sc, h.id, h.b1, h.b2, O, D, h.t, s);
        }
        SCK(2);   // surface record (tri_shade fetch + interpolation)
        // integrator.h:139-147
        if (bounces == 0 || (flags & PF_SPECULAR)) {
            if (found) {
                int al = sc.prims[s.prim].arealight;
                v3 Le = al >= 0 ? mat_v3(sc.lights[al].L) : V3s(0.f);
                L = L + beta * Le;
            } else {
                for (int li = 0; li < sc.n_lights; li++) {
                    if (sc.lights[li].type == AGPT_LIGHT_UNIFORM_INFINITE)
                        L = L + beta * mat_v3(sc.lights[li].L);
                    else if (ENV && sc.lights[li].type == AGPT_LIGHT_INFINITE_AREA)
                        L = L + beta * env_Le(&sc.envs[sc.lights[li].env], D);
                }
            }
        }
        if (!found || (int)bounces >= rc.max_depth) {  // :150
            flags |= PF_DEAD;
        } else if (sc.prims[s.prim].material < 0) {
            // :152-161 null material: skip the surface, re-cast without counting a bounce
            v3 nO = s.p + AGPT_EPSILON * D;
            v3 nD = normalize(D);
            float4 no4, nd4;
            no4.x = nO.x; no4.y = nO.y; no4.z = nO.z; no4.w = AGPT_FLT_MAX;
            nd4.x = nD.x; nd4.y = nD.y; nd4.z = nD.z;
            nd4.w = bounces == 0 ? 2.f : ((flags & PF_SPECULAR) ? 0.f : 1.f);   // (see below: the trace kernel may re-cast again itself)
            pb.ext_o[pid] = no4;
            pb.ext_d[pid] = nd4;
            bits |= SB_EXT;
        } else {
            SCK(3);   // emission, termination
            bits |= SB_SHADED;
            const DevMaterial& m = sc.materials[sc.prims[s.prim].material];
            Bsdf bsdf;
            bsdf_setup(bsdf, m, s.n, s.ns, s.ss_bsdf);
            const WoCache wc = bsdf_wo_cache(bsdf, m, s.wo);
            // ---- UniformSampleOneLight / EstimateDirect (integrator.h:38-105) -------------------
            const bool do_nee = !bsdf_is_perfectly_specular(bsdf) && sc.n_lights > 0;
            int numLight = 0;
            float uS0 = 0.f, uS1 = 0.f;
            float scatteringPdf = 0;
            v3 wi = V3s(0.f);
            bool light_ok = false;
            float nee_lightPdf = 0;
            v3 nee_Li = V3s(0.f);
            if (do_nee) {
                int nLights = sc.n_lights;
                numLight = (int)(rng_float(rng) * nLights);
                if (nLights - 1 < numLight) numLight = nLights - 1;
                float uL0 = rng_float(rng), uL1 = rng_float(rng);
                uS0 = rng_float(rng);
                uS1 = rng_float(rng);
                const DevLight& light = sc.lights[numLight];
                flags |= PF_PENDING;
                float lightPdf = 0;
                v3 Li;
                v3 visO = V3s(0.f), visD = V3(0, 0, 1);
                float visT = 0;
                if (light.type == AGPT_LIGHT_UNIFORM_INFINITE) {  // lights.cpp:15-24 (quirk 3)
                    wi = random_in_hemisphere(rng, s.ns);
                    lightPdf = AGPT_INV2PI;
                    visO = s.p + AGPT_EPSILON * wi;
                    visD = normalize(wi);
                    visT = AGPT_FLT_MAX;
                    Li = mat_v3(light.L);
                } else if (ENV && light.type == AGPT_LIGHT_INFINITE_AREA) {  // lights.cpp:50-90: 1 extra draw, offset along n
                    const DevEnv& env = sc.envs[light.env];
                    Li = V3s(0.f);
                    if (env_sample_li(&env, rng_float(rng), &wi, &lightPdf)) {
                        visO = s.p + AGPT_EPSILON * s.n;
                        visD = normalize(wi);
                        visT = AGPT_FLT_MAX;
                        Li = env_Le(&env, visD);
                    }
                } else {  // AreaLight::Sample_Li lights.cpp:115-126
                    v3 pS;
                    sphere_sample_ref(sc.prims[light.shape], s.p, uL0, uL1, &pS, &lightPdf);
                    if (lightPdf == 0 || sqrlen(pS - s.p) == 0) {
                        lightPdf = 0;
                        Li = V3s(0.f);
                    } else {
                        wi = pS - s.p;
                        float dist = length(wi);
                        wi = wi / dist;
                        visO = s.p + AGPT_EPSILON * wi;
                        visD = normalize(wi);
                        visT = dist - 10 * AGPT_EPSILON;
                        Li = mat_v3(light.L);
                    }
                }
                light_ok = lightPdf > 0 && !isblack(Li);   // integrator.h:44
                nee_lightPdf = lightPdf;
                nee_Li = Li;
                // the shadow ray is written where it is known (it only counts if the path joins the shadow queue below): seven
                // registers fewer across the three evaluations
                if (light_ok) {
                    float4 so4, sd4;
                    so4.x = visO.x; so4.y = visO.y; so4.z = visO.z; so4.w = visT;
                    sd4.x = visD.x; sd4.y = visD.y; sd4.z = visD.z; sd4.w = 0.f;
                    pb.sh_o[pid] = so4;
                    pb.sh_d[pid] = sd4;
                }
            }
            SCK(4);   // BSDF set-up + light sampling
            // ---- the three BSDF evaluations of a vertex run through ONE loop with one evaluator call site:
            //   it 0: BSDF::f + BSDF::Pdf for the light sample          (integrator.h:46-47, reflection.h:114-123,174-188)
            //   it 1: EstimateDirect's BSDF-sampling leg, uScattering   (integrator.h:63-90,  BSDF::Sample_f reflection.h:124-172)
            //   it 2: the continuation                                   (integrator.h:170-187, BSDF::Sample_f)
            v3 fac = V3s(1.f);
            const v3 wo_l = w2l(bsdf, s.wo);
            const float woDotNg = dot(s.wo, bsdf.ng);
#pragma unroll 1
            for (int it = do_nee ? (light_ok ? 0 : 1) : 2; it < 3; ++it) {
                const uint32_t lmask = lobes_matching(bsdf, it < 2);   // skipSpecular for both EstimateDirect evaluations
                const int matching = __popc(lmask);
                float pdf = it == 1 ? scatteringPdf : 0.f;   // EstimateDirect reuses its scatteringPdf variable (quirk 9)
                v3 f = V3s(0.f), wi_l = V3s(0.f);
                int skip_lobe = -1;
                bool eval = false, sampledSpecular = false;
                if (it == 0) {   // bsdf_f_pdf's prologue
                    if (bsdf.lobes != 0 && wo_l.z != 0) {
                        wi_l = w2l(bsdf, wi);
                        eval = true;
                    }
                } else {         // bsdf_sample_f's prologue
                    float u0, u1;
                    if (it == 1) {
                        u0 = uS0;
                        u1 = uS1;
                    } else {
                        wi = V3s(0.f);
                        u0 = rng_float(rng);
                        u1 = rng_float(rng);
                    }
                    if (matching == 0) {
                        pdf = 0;
                    } else {
                        int comp = (int)floorf(u0 * matching);
                        if (matching - 1 < comp) comp = matching - 1;
                        uint32_t mm = lmask;   // the comp-th matching lobe in list order
                        for (int i = 0; i < comp; ++i) mm &= mm - 1;
                        const int lobe = __ffs((int)mm) - 1;
                        const float ur0 = sminf(u0 * matching - comp, AGPT_ONE_MINUS_EPS);
                        if (wo_l.z != 0) {   // (wo.z == 0: returns 0 with the caller's pdf untouched, quirk 9)
                            pdf = 0;
                            sampledSpecular = lobe == LOBE_SPECULAR;
                            const v3 fs = lobe_sample(m, lobe, wo_l, &wi_l, ur0, u1, &pdf, wc);
                            if (pdf != 0) {
                                wi = l2w(bsdf, wi_l);
                                if (lobe == LOBE_SPECULAR) {
                                    if (matching > 1) pdf /= matching;
                                    f = fs;
                                } else {
                                    eval = true;   // pdf += the other lobes' Pdf (list order), f = sum of all matching lobes' f
                                    skip_lobe = matching > 1 ? lobe : -2;
                                }
                            }
                        }
                    }
                }
                if (eval) {
                    const bool reflect = dot(wi, bsdf.ng) * woDotNg > 0;
                    const LobeEval e = bsdf_eval_lobes_inl(m, lmask, wo_l, wi_l, skip_lobe, pdf, reflect, wc);
                    if (it == 0)
                        pdf = matching > 0 ? e.pdf / matching : 0.f;
                    else if (matching > 1)
                        pdf = e.pdf / matching;
                    if (reflect) f = e.f;
                }
                if (it == 0) {   // integrator.h:46-60
                    scatteringPdf = pdf;
                    f = f * absdot(wi, s.ns);
                    if (!isblack(f)) {
                        float weight = power_heuristic(1, nee_lightPdf, 1, scatteringPdf);
                        v3 c1 = f * nee_Li * weight / nee_lightPdf;
                        float4 c4;
                        c4.x = c1.x; c4.y = c1.y; c4.z = c1.z; c4.w = 0.f;
                        pb.c1[pid] = c4;
                        flags |= PF_PEND_C1;
                        bits |= SB_SHADOW;
                    }
                } else if (it == 1) {   // integrator.h:63-90
                    const DevLight& light = sc.lights[numLight];
                    f = f * absdot(wi, s.ns);
                    if (!isblack(f) && pdf > 0) {
                        float lp;
                        if (light.type == AGPT_LIGHT_UNIFORM_INFINITE)
                            lp = dot(s.n, wi) > 0 ? AGPT_INV2PI : 0.f;  // lights.cpp:26-28 (geometric n)
                        else if (ENV && light.type == AGPT_LIGHT_INFINITE_AREA)
                            lp = env_pdf_li(&sc.envs[light.env], wi);
                        else
                            lp = sphere_pdf(sc.prims[light.shape], s.p);
                        if (lp != 0) {
                            float weight = power_heuristic(1, pdf, 1, lp);
                            // Le(-wi) of the light's shape, or light.Le(ray) on a miss (ray.D = normalize(wi))
                            v3 Li2 = ENV && light.type == AGPT_LIGHT_INFINITE_AREA ? env_Le(&sc.envs[light.env], normalize(wi))
                                                                            : mat_v3(light.L);
                            // the reference traces this ray before it looks at Li (integrator.h:77-88); a black Li only
                            // means nothing is added afterwards
                            v3 c2 = isblack(Li2) ? V3s(0.f) : f * Li2 * weight / pdf;
                            v3 mO = s.p + AGPT_EPSILON * wi;
                            v3 mD = normalize(wi);
                            // The query's answer for an area light is "the closest hit is the light's sphere" (integrator.h:82-85):
                            // a ray that misses that sphere altogether cannot say yes, whatever else it meets, so it is answered
                            // here -- by the very test the MIS-query kernel starts a ray with (k_trace_fast: mis_reach) -- and
                            // never enters the queue.  BSDF-sampled directions rarely point at a small light: on C3 this
                            // settles most of the queries of the two sphere lights.  (The ray still counts as the
                            // Scene::Intersect call the reference makes.)
                            bool reach = true;
                            if (rc.answer_rays && light.type == AGPT_LIGHT_AREA) {
                                const DevPrim& LS = sc.prims[light.shape];
                                TraceRay qr;
                                qr.O = mO;
                                qr.D = mD;
                                float ts;
                                reach = sphere_test_c(V3(LS.cx, LS.cy, LS.cz), LS.r2, qr, AGPT_FLT_MAX, ts);
                            }
                            if (!reach) {
                                bits |= SB_MIS_ANSWERED;
                            } else {
                            float4 c4, mo4, md4;
                            c4.x = c2.x; c4.y = c2.y; c4.z = c2.z; c4.w = 0.f;
                            // tmax = FLT_MAX; the MIS-query kernel (mis_mode 1) is told the sampled light's sphere in its place
                            mo4.x = mO.x; mo4.y = mO.y; mo4.z = mO.z;
                            mo4.w = rc.mis_mode ? __uint_as_float(light.type == AGPT_LIGHT_AREA ? (uint32_t)light.shape : 0xFFFFFFFFu)
                                                : AGPT_FLT_MAX;
                            md4.x = mD.x; md4.y = mD.y; md4.z = mD.z; md4.w = __uint_as_float((uint32_t)numLight);
                            pb.c2[pid] = c4;
                            pb.mis_o[pid] = mo4;
                            pb.mis_d[pid] = md4;
                            flags |= PF_PEND_C2;
                            bits |= SB_MIS;
                            }
                        }
                    }
                } else if (isblack(f) || pdf == 0) {   // integrator.h:175
                    flags |= PF_DEAD;
                } else {
                    fac = f * absdot(wi, s.ns) / pdf;
                    if (!(flags & PF_PENDING)) beta = beta * fac;
                    flags = (flags & ~PF_SPECULAR) | (sampledSpecular ? PF_SPECULAR : 0u);
                    flags = (flags & ~PF_BOUNCE_MASK) | ((bounces + 1) & PF_BOUNCE_MASK);
                    if (rc.answer_rays && (int)(bounces + 1) >= rc.max_depth && !sampledSpecular) {
                        // The ray that would follow ends the path whatever it hits: at the next vertex the loop adds emission
                        // only after a specular bounce (integrator.h:139-147) and breaks on `bounces >= maxDepth` (:150) before
                        // anything else looks at the intersection.  The reference still makes that Scene::Intersect call; its
                        // answer cannot reach the image, so the production path counts the ray and does not trace it.  (The
                        // path stays in the active queue for one more pass if this vertex' light sample is still pending.)
                        flags |= PF_DEAD;
                        bits |= SB_RAY_ANSWERED;
                    } else {
                        v3 nO = s.p + AGPT_EPSILON * wi;
                        v3 nD = normalize(wi);
                        float4 no4, nd4;
                        no4.x = nO.x; no4.y = nO.y; no4.z = nO.z; no4.w = AGPT_FLT_MAX;
                        nd4.x = nD.x; nd4.y = nD.y; nd4.z = nD.z;
                        // d.w = 1: should this ray's closest hit be a primitive without a material (an emitter sphere), the next
                        // vertex does nothing but re-cast it (no emission after a non-specular bounce, :139-147; below maxDepth,
                        // :150; no bounce counted, :152-161) -- k_trace_fast may then do that re-cast itself instead of sending the
                        // path through a whole iteration for it
                        nd4.w = (!sampledSpecular && (int)(bounces + 1) < rc.max_depth) ? 1.f : 0.f;
                        pb.ext_o[pid] = no4;
                        pb.ext_d[pid] = nd4;
                        bits |= SB_EXT;
                    }
                }
            }
            SCK(5);   // the three evaluations incl. their ray / contribution stores
            if (flags & PF_PENDING) {  // the continuation factor and the chosen light wait for the resolve
                float4 f4;
                f4.x = fac.x; f4.y = fac.y; f4.z = fac.z; f4.w = __uint_as_float((uint32_t)numLight);
                pb.fac4[pid] = f4;
            }
        }
    }
    // a path that goes on has a continuation ray; one that has ended but still owes its last light sample (which waits for
    // this iteration's shadow ray / MIS query) goes to the resolve queue: k_resolve_pending adds the sample without the
    // records, the lanes and the code of a full k_shade pass
    if (bits & SB_EXT) bits |= SB_ACTIVE;
    else if (flags & PF_PENDING) bits |= SB_RESOLVE;
    b4.x = beta.x; b4.y = beta.y; b4.z = beta.z; b4.w = __uint_as_float(flags);
    pb.beta4[pid] = b4;
    l4.x = L.x; l4.y = L.y; l4.z = L.z; l4.w = __uint_as_float(rng);
    pb.L4[pid] = l4;
    SCK(6);   // final stores
    return bits;
}

// Every wave handles tiles of 64 * SHADE_U queue entries and aggregates its queue appends per tile: ballots give the per-queue
// counts, ONE atomic per queue reserves the tile's slots, then every lane writes its path id at its (order-preserving) offset
// -- same-line atomics serialise at ~11 ns each and were the kernel's bottleneck with one append per 64 paths.
#ifndef SHADE_GRAB
#define SHADE_GRAB 2   // wave tiles a wave takes per atomic
#endif
#ifndef SHADE_U
#define SHADE_U 4
#endif

// Waves per SIMD the register allocation is held to.  build.py compiles this unit with -DAGPT_SHADE_WAVES=4 and MachineLICM off:
// 128 VGPRs, 2 spilled (134 and none uncapped).  With the pass on the same source needs 168 + 7-14 spilled at three waves and
// ~90 spilled at four.  C3 step, non-trace time: 3 waves / LICM on 77.8 ms, 3 waves / off 77.2, 4 waves / off 73.9.
#ifndef AGPT_SHADE_WAVES
#define AGPT_SHADE_WAVES 3
#endif
// LDS_TABLES: the scene's small tables -- Scene::primitives records, materials, lights -- are copied into LDS at kernel
// start and every lookup of the shading code goes there.  As per-lane global loads they were ~50 separate load -> wait -> use
// points inside one long dependent chain (their addresses depend on the hit), each paying an L1/L2 round trip with only
// three to four waves per SIMD to cover it, and every store to the path state forced the compiler to re-load fields it already
// had (it cannot prove that the tables and the path buffers do not alias).  Scenes whose tables exceed the limits below
// use the global-memory instantiation.
#define AGPT_SHADE_LDS_PRIMS 256
#define AGPT_SHADE_LDS_MATERIALS 128
#define AGPT_SHADE_LDS_LIGHTS 64
template <bool LDS_TABLES, bool ENV>
__global__ void __launch_bounds__(AGPT_BLOCK, AGPT_SHADE_WAVES)
k_shade(DevScene sc_in, RenderConsts rc, PathBuffers pb, Queues qin, Queues qout, DevCounters* __restrict__ counters,
        uint32_t* __restrict__ tile_heads) {
    __shared__ DevPrim s_tab_prims[LDS_TABLES ? AGPT_SHADE_LDS_PRIMS : 1];
    __shared__ DevMaterial s_tab_materials[LDS_TABLES ? AGPT_SHADE_LDS_MATERIALS : 1];
    __shared__ DevLight s_tab_lights[LDS_TABLES ? AGPT_SHADE_LDS_LIGHTS : 1];
    // (a launch over a short queue: one wave per 256-path tile is all it can use -- the rest of the grid leaves before setting up)
    if (blockIdx.x * (AGPT_BLOCK / 64) * 64u * SHADE_U >= QCOUNT(qin, 0) && blockIdx.x > 0) return;
    DevScene sc = sc_in;
    if (LDS_TABLES) {
        auto copy_words = [&](void* dst, const void* src, int n_bytes) {
            uint32_t* d = (uint32_t*)dst;
            const uint32_t* g = (const uint32_t*)src;
            for (int i = threadIdx.x; i < n_bytes / 4; i += AGPT_BLOCK) d[i] = g[i];
        };
        copy_words(s_tab_prims, sc_in.prims, sc_in.n_prims * (int)sizeof(DevPrim));
        copy_words(s_tab_materials, sc_in.materials, sc_in.n_materials * (int)sizeof(DevMaterial));
        copy_words(s_tab_lights, sc_in.lights, sc_in.n_lights * (int)sizeof(DevLight));
        sc.prims = s_tab_prims;
        sc.materials = s_tab_materials;
        sc.lights = s_tab_lights;
        __syncthreads();
    }
    __shared__ uint32_t s_pid[SHADE_U][AGPT_BLOCK];
    __shared__ uint32_t s_bits[SHADE_U][AGPT_BLOCK];
    const uint32_t n_active = QCOUNT(qin, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t shaded = 0, mis_answered = 0;
#ifdef AGPT_SHADE_CLOCK
    unsigned long long sck_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long sck_last = __builtin_amdgcn_s_memtime();
    const unsigned long long sck_begin = sck_last;
#endif
    // Wave-private tiles of 64 * SHADE_U queue entries: the wave shades them, counts its appends per queue with ballots and
    // reserves the slots with one atomic per queue (lanes 0..3, one queue each).  No block barrier: with block-wide tiles
    // the three __syncthreads per tile made every wave wait for the slowest of its block -- 10 % of the kernel's wave
    // time (s_memtime phase clocks) -- while 4x the atomics (one per queue per 256 paths) stay far below the ~11 ns per
    // same-line atomic that had made per-wave-per-64-paths appends the bottleneck.
    const uint32_t wtile = 64u * SHADE_U;
    const uint32_t n_wtiles = (n_active + wtile - 1) / wtile;
    // Tiles are handed out dynamically, SHADE_GRAB at a time, from AGPT_FRONTIERS segment heads (one per XCD, AGPT_QSTRIDE words
    // apart: same-line atomics serialise).  Tiles differ a lot in cost (a tile of misses against a tile of surface hits); with a
    // static stride the slowest wave of a CU ran alone at the end of every launch (83.6 -> 77.7 ms of non-trace time per C3 step).
    uint32_t seg = blockIdx.x & (AGPT_FRONTIERS - 1u), segs_left = AGPT_FRONTIERS;
    const uint32_t seg_tiles = (n_wtiles + AGPT_FRONTIERS - 1u) / AGPT_FRONTIERS;
    uint32_t t_next = 0, t_end = 0;
    auto next_tile = [&](uint32_t& t) -> bool {
        if (t_next < t_end) {
            t = t_next++;
            return true;
        }
        while (segs_left) {
            uint32_t off = 0;
            if (lane == 0) off = atomicAdd(tile_heads + seg * AGPT_QSTRIDE, (uint32_t)SHADE_GRAB);
            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
            const uint32_t seg_begin = seg * seg_tiles;
            const uint32_t seg_end = min(seg_begin + seg_tiles, n_wtiles);
            if (seg_begin < seg_end && off < seg_end - seg_begin) {
                t = seg_begin + off;
                t_next = t + 1u;
                t_end = min(t + (uint32_t)SHADE_GRAB, seg_end);
                return true;
            }
            seg = (seg + 1u) & (AGPT_FRONTIERS - 1u);
            --segs_left;
            // segments that other waves have drained meanwhile are skipped after one look at all the heads (lane k reads head k,
            // past the L1) instead of one failing atomic each: without it every wave of the grid ends with a walk through all
            // eight -- 8,192 x 8 same-line atomics at ~11 ns, ~90 us in which the launch only drains.  A head only grows: a stale
            // value costs an atomic, never a tile.
            if (segs_left) {
                uint32_t head = 0;
                if (lane < (int)AGPT_FRONTIERS) head = __hip_atomic_load(tile_heads + lane * AGPT_QSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t fb = (uint32_t)lane * seg_tiles;
                const bool live = lane < (int)AGPT_FRONTIERS && fb < n_wtiles && head < min(seg_tiles, n_wtiles - fb);
                const uint32_t live_mask = (uint32_t)__ballot(live);
                while (segs_left && !((live_mask >> seg) & 1u)) {
                    seg = (seg + 1u) & (AGPT_FRONTIERS - 1u);
                    --segs_left;
                }
            }
        }
        return false;
    };
    for (uint32_t t = 0; next_tile(t);) {
        uint32_t total[AGPT_NQUEUES] = {0, 0, 0, 0, 0};
#pragma unroll 1
        for (int u = 0; u < SHADE_U; ++u) {
#ifdef AGPT_SHADE_CLOCK
            { const unsigned long long n0 = __builtin_amdgcn_s_memtime(); sck_acc[7] += n0 - sck_last; sck_last = n0; }   // queue appends
#endif
            const uint32_t qi = t * wtile + (uint32_t)u * 64u + (uint32_t)lane;
            uint32_t pid = 0, bits = 0;
            if (qi < n_active) {
                pid = qin.active[qi];
                bits = shade_path<ENV>(sc, rc, pb, pid SCK_ARGS);
            }
            s_pid[u][tid] = pid;     // (slots private to the thread: no synchronisation)
            s_bits[u][tid] = bits;
            shaded += (bits & SB_SHADED) ? 1u : 0u;
            mis_answered += ((bits & SB_MIS_ANSWERED) ? 1u : 0u) + ((bits & SB_RAY_ANSWERED) ? 1u : 0u);
            for (int q = 0; q < AGPT_NQUEUES; ++q) total[q] += (uint32_t)__popcll(__ballot((bits >> q) & 1u));
        }
        uint32_t my_total = total[0];
        for (int q = 1; q < AGPT_NQUEUES; ++q) my_total = lane == q ? total[q] : my_total;
        uint32_t base = 0;
        if (lane < AGPT_NQUEUES && my_total) base = atomicAdd(&QCOUNT(qout, lane), my_total);
        uint32_t run[AGPT_NQUEUES];
        for (int q = 0; q < AGPT_NQUEUES; ++q) run[q] = (uint32_t)__shfl((int)base, q);
        for (int u = 0; u < SHADE_U; ++u) {
            const uint32_t bits = s_bits[u][tid];
            const uint32_t pid = s_pid[u][tid];
            for (int q = 0; q < AGPT_NQUEUES; ++q) {
                const bool pred = (bits >> q) & 1u;
                const unsigned long long m = __ballot(pred);
                if (pred) {
                    uint32_t* qq = q == 0 ? qout.active : (q == 1 ? qout.ext : (q == 2 ? qout.mis : (q == 3 ? qout.shadow : qout.resolve)));
                    qq[run[q] + (uint32_t)__popcll(m & lt_mask)] = pid;
                }
                run[q] += (uint32_t)__popcll(m);
            }
        }
    }
#ifdef AGPT_SHADE_CLOCK
    if (lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&counters->dbg[56 + k], sck_acc[k]);
        atomicAdd(&counters->dbg[55], __builtin_amdgcn_s_memtime() - sck_begin);
    }
#endif
    // one atomic per wave, not per lane
    for (int off = 32; off > 0; off >>= 1) {
        shaded += __shfl_down(shaded, off);
        mis_answered += __shfl_down(mis_answered, off);
    }
    if (lane == 0 && shaded) atomicAdd(&counters->shaded, (unsigned long long)shaded);
    if (lane == 0 && mis_answered) {
        atomicAdd(&counters->closest_rays, (unsigned long long)mis_answered);   // MIS queries are Scene::Intersect calls
        atomicAdd(&counters->answered, (unsigned long long)mis_answered);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Paths that have ended (MaxDepth reached, miss, black sample) while their last vertex' light sample was still waiting for its
// shadow ray / MIS query: add the sample (integrator.h:166) and let the path go.  They used to ride through a whole
// k_shade pass for this -- after the last bounce that is EVERY path of the batch: 10 records loaded, 3 wave slots per SIMD.
__global__ void __launch_bounds__(AGPT_BLOCK)
k_resolve_pending(DevScene sc, RenderConsts rc, PathBuffers pb, Queues qin) {
    const uint32_t n = QCOUNT(qin, 4);
    for (uint32_t i = blockIdx.x * AGPT_BLOCK + threadIdx.x; i < n; i += gridDim.x * AGPT_BLOCK) {
        const uint32_t pid = qin.resolve[i];
        const float4 b4 = pb.beta4[pid];
        float4 l4 = pb.L4[pid];
        const float4 f4 = pb.fac4[pid];
        const uint32_t was_occluded = pb.occluded[pid], was_mis_ok = pb.mis_ok[pid];
        const float4 c1_4 = pb.c1[pid], c2_4 = pb.c2[pid];
        const uint32_t flags = __float_as_uint(b4.w);
        const v3 Ld = pending_light_sample(sc, rc, pb, pid, flags, was_occluded, was_mis_ok, c1_4, c2_4, f4);
        const float lightPdf = 1.f / sc.n_lights;
        const v3 L = V3(l4.x, l4.y, l4.z) + V3(b4.x, b4.y, b4.z) * (Ld / lightPdf);
        l4.x = L.x; l4.y = L.y; l4.z = L.z;
        pb.L4[pid] = l4;
    }
}

