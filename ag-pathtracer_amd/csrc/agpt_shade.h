// agpt_shade.h -- device-side shading: surface reconstruction, Disney/microfacet BSDF, light sampling.
//
// Each function names the reference code it reproduces (operation order preserved; see agpt_math.h for the
// trig definition).  Unlike the reference, nothing is virtual: a material is a POD with a lobe list
// {DisneyDiffuse, DisneyRetro, MicrofacetReflection} or {SpecularReflection}, and the SurfaceInteraction is
// rebuilt from the 16-B hit record plus the per-triangle record precomputed at upload (agpt_scene.h).
#pragma once

#include "agpt_scene.h"

enum { LOBE_DIFFUSE = 0, LOBE_RETRO = 1, LOBE_MICROFACET = 2, LOBE_SPECULAR = 3 };

// ---- sampling helpers (template/common.h) -------------------------------------------------------------------
// common.h:118-136 ConcentricSampleDisk + :139-143 CosineSampleHemisphere
__device__ __forceinline__ v3 cosine_sample_hemisphere(float u0, float u1) {
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    float dx = 0, dy = 0;
    if (!(ox == 0 && oy == 0)) {
        float theta, r;
        if (fabsf(ox) > fabsf(oy)) {
            r = ox;
            theta = (AGPT_PI / 4) * (oy / ox);
        } else {
            r = oy;
            theta = (AGPT_PI / 2) - (AGPT_PI / 4) * (ox / oy);
        }
        float s, c;
        cr_sincosf(theta, &s, &c);
        dx = r * c;
        dy = r * s;
    }
    float z = sqrtf(smaxf(0.f, 1 - dx * dx - dy * dy));
    return V3(dx, dy, z);
}
// common.h:84-89 RandomInSphere(u)
__device__ __forceinline__ v3 random_in_sphere_u(float u0, float u1) {
    float a = 1 - 2 * u0;
    float b = sqrtf(1 - a * a);
    float phi = 2 * AGPT_PI * u1;
    float s, c;
    cr_sincosf(phi, &s, &c);
    return V3(b * c, b * s, a);
}
// common.h:73-82 RandomInSphere() + :91-97 RandomInHemisphere(N): 2 RNG draws
__device__ __forceinline__ v3 random_in_hemisphere(uint32_t& rng, v3 N) {
    float a = 1 - 2 * rng_float(rng);
    float b = sqrtf(1 - a * a);
    float phi = 2 * AGPT_PI * rng_float(rng);
    float s, c;
    cr_sincosf(phi, &s, &c);
    v3 p = V3(1.f * b * c, 1.f * b * s, 1.f * a);
    if (dot(p, N) < 0) return -p;
    return p;
}
// common.h:153-156
__device__ __forceinline__ v3 spherical_direction(float sinTheta, float cosTheta, float phi, v3 x, v3 y, v3 z) {
    float s, c;
    cr_sincosf(phi, &s, &c);
    return sinTheta * c * x + sinTheta * s * y + cosTheta * z;
}

// ---- microfacet.h -------------------------------------------------------------------------------------------
__device__ __forceinline__ float CosTheta(v3 w) { return w.z; }
__device__ __forceinline__ float Cos2Theta(v3 w) { return w.z * w.z; }
__device__ __forceinline__ float AbsCosTheta(v3 w) { return fabsf(w.z); }
__device__ __forceinline__ float Sin2Theta(v3 w) { return smaxf(0.f, 1.f - Cos2Theta(w)); }
__device__ __forceinline__ float SinTheta(v3 w) { return sqrtf(Sin2Theta(w)); }
__device__ __forceinline__ float TanTheta(v3 w) { return SinTheta(w) / CosTheta(w); }
__device__ __forceinline__ float Tan2Theta(v3 w) { return Sin2Theta(w) / Cos2Theta(w); }
__device__ __forceinline__ float CosPhi(v3 w) {
    float sinTheta = SinTheta(w);
    return (sinTheta == 0) ? 1 : tclampf(w.x / sinTheta, -1.f, 1.f);
}
__device__ __forceinline__ float SinPhi(v3 w) {
    float sinTheta = SinTheta(w);
    return (sinTheta == 0) ? 0 : tclampf(w.y / sinTheta, -1.f, 1.f);
}
__device__ __forceinline__ float Cos2Phi(v3 w) { float c = CosPhi(w); return c * c; }
__device__ __forceinline__ float Sin2Phi(v3 w) { float s = SinPhi(w); return s * s; }

// microfacet.h:34-73
__device__ __forceinline__ void tr_sample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y) {
    if (cosTheta > .9999f) {
        float r = sqrtf(U1 / (1 - U1));
        float phi = 6.28318530718f * U2;
        float s, c;
        cr_sincosf(phi, &s, &c);
        *slope_x = r * c;
        *slope_y = r * s;
        return;
    }
    float sinTheta = sqrtf(smaxf(0.f, 1.f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if ((double)tmp > 1e10) tmp = 1e10f;
    float B = tanTheta;
    float D = sqrtf(smaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) {
        S = 1.f;
        U2 = 2.f * (U2 - .5f);
    } else {
        S = -1.f;
        U2 = 2.f * (.5f - U2);
    }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
              (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
}
// microfacet.h:75-94
__device__ __forceinline__ v3 tr_sample(v3 wi, float ax, float ay, float U1, float U2) {
    v3 wiS = normalize(V3(ax * wi.x, ay * wi.y, wi.z));
    float sx, sy;
    tr_sample11(CosTheta(wiS), U1, U2, &sx, &sy);
    float cp = CosPhi(wiS), sp = SinPhi(wiS);
    float tmp = cp * sx - sp * sy;
    sy = sp * sx + cp * sy;
    sx = tmp;
    sx = ax * sx;
    sy = ay * sy;
    return normalize(V3(-sx, -sy, 1.f));
}
// microfacet.h:124-132
__device__ __forceinline__ float tr_D(const DevMaterial& m, v3 wh) {
    float tan2Theta = Tan2Theta(wh);
    if (isinf(tan2Theta)) return 0.f;
    float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
    float e = (Cos2Phi(wh) / (m.alphax * m.alphax) + Sin2Phi(wh) / (m.alphay * m.alphay)) * tan2Theta;
    return 1 / (AGPT_PI * m.alphax * m.alphay * cos4Theta * (1 + e) * (1 + e));
}
// microfacet.h:142-149
__device__ __forceinline__ float tr_lambda(const DevMaterial& m, v3 w) {
    float absTanTheta = fabsf(TanTheta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = sqrtf(Cos2Phi(w) * m.alphax * m.alphax + Sin2Phi(w) * m.alphay * m.alphay);
    float a2t2 = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + a2t2)) / 2;
}
__device__ __forceinline__ float tr_G1(const DevMaterial& m, v3 w) { return 1 / (1 + tr_lambda(m, w)); }
// MicrofacetDistribution::Pdf microfacet.h:107-109
__device__ __forceinline__ float tr_pdf(const DevMaterial& m, v3 wo, v3 wh, float G1o) {  // G1o = tr_G1(m, wo)
    return tr_D(m, wh) * G1o * absdot(wo, wh) / AbsCosTheta(wo);
}
// microfacet.h:180-201
__device__ __forceinline__ float fr_dielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = tclampf(cosThetaI, -1.f, 1.f);
    bool entering = cosThetaI > 0.f;
    if (!entering) {
        float t = etaI;
        etaI = etaT;
        etaT = t;
        cosThetaI = fabsf(cosThetaI);
    }
    float sinThetaI = sqrtf(smaxf(0.f, 1.f - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = sqrtf(smaxf(0.f, 1.f - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}

// ---- disney.h -----------------------------------------------------------------------------------------------
__device__ __forceinline__ float schlick_weight(float cosTheta) {  // :12-15
    float m = tclampf(1 - cosTheta, 0.f, 1.f);
    return (m * m) * (m * m) * m;
}
__device__ __forceinline__ v3 mat_v3(const float a[3]) { return V3(a[0], a[1], a[2]); }
// DisneyFresnel::Evaluate :62-71 (FrSchlick :17-19)
__device__ __forceinline__ v3 disney_fresnel(const DevMaterial& m, float cosI) {
    v3 frs = lerp3(schlick_weight(cosI), mat_v3(m.R0), V3s(1.f));
    return lerp3(m.metallic, V3s(fr_dielectric(cosI, 1, m.eta)), frs);
}
// ---- lobes ------------------------------------------------------------------------------------------------
// A BSDF is the reference's bxdfs[] list in its fixed insertion order (material.h:51-58,79-81):
// DisneyDiffuse, DisneyRetro, MicrofacetReflection | SpecularReflection -> a 4-bit mask, bit = lobe id.
#define LOBE_BIT(l) (1u << (l))

struct LobeEval {
    v3 f;
    float pdf;
};
// Values that depend on the outgoing direction only.  One shading point evaluates its BSDF for three (wo, wi) pairs and
// samples it twice, always with the same wo: G1(wo) (two square roots, three divisions) and the Schlick weight of wo
// are computed once; the inputs are identical each time, so the shared values are the bits every call would produce.
struct WoCache {
    float G1o;  // tr_G1(m, wo)
    float Fo;   // schlick_weight(AbsCosTheta(wo))
};

// f(wo,wi) and Pdf(wo,wi) of one lobe in the local frame.  Evaluating both together lets the microfacet lobe share
// wh, D(wh) and G1(wo) between MicrofacetReflection::f (reflection.h:42-54) and ::Pdf (:67-71): wi+wo and wo+wi, and
// the two D / G1 calls of the reference, are the same fp32 values, so sharing them does not change a bit.
__device__ __forceinline__ LobeEval lobe_eval(const DevMaterial& m, int lobe, v3 wo, v3 wi, const WoCache& wc) {
    LobeEval e;
    e.f = V3s(0.f);
    e.pdf = 0.f;
    if (lobe == LOBE_DIFFUSE) {  // DisneyDiffuse::f disney.h:28-35; BxDF::Pdf reflection.h:16-18
        float Fo = wc.Fo, Fi = schlick_weight(AbsCosTheta(wi));
        e.f = mat_v3(m.diffuse_R) * AGPT_INVPI * (1 - Fo / 2) * (1 - Fi / 2);
        e.pdf = same_hemisphere(wo, wi) ? AbsCosTheta(wi) * AGPT_INVPI : 0;
    } else if (lobe == LOBE_RETRO) {  // DisneyRetro::f disney.h:43-55
        e.pdf = same_hemisphere(wo, wi) ? AbsCosTheta(wi) * AGPT_INVPI : 0;
        v3 wh = wi + wo;
        if (!(wh.x == 0 && wh.y == 0 && wh.z == 0)) {
            wh = normalize(wh);
            float cosThetaD = dot(wi, wh);
            float Fo = wc.Fo, Fi = schlick_weight(AbsCosTheta(wi));
            float Rr = 2 * m.roughness * cosThetaD * cosThetaD;
            e.f = mat_v3(m.diffuse_R) * AGPT_INVPI * Rr * (Fo + Fi + Fo * Fi * (Rr - 1));
        }
    } else if (lobe == LOBE_MICROFACET) {
        float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        v3 whs = wi + wo;
        bool f_ok = !(cosThetaI == 0 || cosThetaO == 0) && !(whs.x == 0 && whs.y == 0 && whs.z == 0);
        bool pdf_ok = same_hemisphere(wo, wi);
        if (f_ok || pdf_ok) {
            v3 wh = normalize(whs);
            float D = tr_D(m, wh);
            float G1o = wc.G1o;
            if (f_ok) {
                v3 F = disney_fresnel(m, dot(wi, faceforward(wh, V3(0, 0, 1))));
                float G = G1o * tr_G1(m, wi);  // DisneyMicrofacetDistribution::G disney.h:78-81
                e.f = V3s(1.f) * D * G * F / (4 * cosThetaI * cosThetaO);
            }
            if (pdf_ok) e.pdf = D * G1o * absdot(wo, wh) / AbsCosTheta(wo) / (4 * dot(wo, wh));
        }
    }
    // LOBE_SPECULAR: f = 0 (reflection.h:26-28), Pdf = 0 (:30)
    return e;
}

// Sum of f and of Pdf over the lobes of `mask`, in list order: what BSDF::f (reflection.h:114-123, when reflect),
// BSDF::Pdf (:174-188) and the tail of BSDF::Sample_f (:156-170) compute.  `skip_pdf_lobe` (or -1) is left out of the
// pdf sum (the sampled lobe contributes the pdf its own Sample_f returned), pdf_init is the running value.
// k_shade has exactly ONE call site of the inlined form (its three evaluations per vertex run through one loop), so
// the code stays inside the instruction cache without a function call -- a call forces everything live across it into
// the callee-saved registers and was the main source of the kernel's scratch spills.
__device__ __forceinline__ LobeEval bsdf_eval_lobes_inl(const DevMaterial& m, uint32_t mask, v3 wo, v3 wi, int skip_pdf_lobe,
                                                        float pdf_init, bool want_f, WoCache wc) {
    LobeEval acc;
    acc.f = V3s(0.f);
    acc.pdf = pdf_init;
#pragma unroll 1
    for (int lobe = 0; lobe < 4; ++lobe) {
        if (!(mask & LOBE_BIT(lobe))) continue;
        LobeEval e = lobe_eval(m, lobe, wo, wi, wc);
        if (want_f) acc.f = acc.f + e.f;
        if (lobe != skip_pdf_lobe) acc.pdf += e.pdf;
    }
    return acc;
}
// shared non-inlined copy for the helpers below (known-answer kernels)
__device__ __noinline__ LobeEval bsdf_eval_lobes(const DevMaterial* mp, uint32_t mask, v3 wo, v3 wi, int skip_pdf_lobe,
                                                 float pdf_init, bool want_f, WoCache wc) {
    return bsdf_eval_lobes_inl(*mp, mask, wo, wi, skip_pdf_lobe, pdf_init, want_f, wc);
}

// ---- BSDF (reflection.cpp:6-11, reflection.h:83-201) --------------------------------------------------------
struct Bsdf {
    v3 ng, ns, ss, ts;
    uint32_t lobes;  // LOBE_BIT mask
};
__device__ __forceinline__ void bsdf_setup(Bsdf& b, const DevMaterial& m, v3 ng, v3 ns, v3 shading_dpdu_normalized) {
    b.ng = ng;
    b.ns = ns;
    b.ss = shading_dpdu_normalized;
    b.ts = cross(ns, b.ss);
    b.lobes = (m.has_diffuse ? LOBE_BIT(LOBE_DIFFUSE) : 0u) | (m.has_retro ? LOBE_BIT(LOBE_RETRO) : 0u) |
              (m.has_microfacet ? LOBE_BIT(LOBE_MICROFACET) : 0u) | (m.has_specular ? LOBE_BIT(LOBE_SPECULAR) : 0u);
}
__device__ __forceinline__ uint32_t lobes_matching(const Bsdf& b, bool skipSpecular) {
    return skipSpecular ? (b.lobes & ~LOBE_BIT(LOBE_SPECULAR)) : b.lobes;
}
__device__ __forceinline__ v3 w2l(const Bsdf& b, v3 v) { return V3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
__device__ __forceinline__ v3 l2w(const Bsdf& b, v3 v) {
    return V3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
              b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
__device__ __forceinline__ WoCache bsdf_wo_cache(const Bsdf& b, const DevMaterial& m, v3 woW) {
    const v3 wo = w2l(b, woW);
    WoCache wc;
    wc.G1o = (b.lobes & LOBE_BIT(LOBE_MICROFACET)) ? tr_G1(m, wo) : 0.f;
    wc.Fo = schlick_weight(AbsCosTheta(wo));
    return wc;
}
__device__ __forceinline__ bool bsdf_is_perfectly_specular(const Bsdf& b) {
    return (b.lobes & ~LOBE_BIT(LOBE_SPECULAR)) == 0;  // vacuously true for an empty list, like the reference's loop
}
// BSDF::f and BSDF::Pdf for one (wo, wi) pair (reflection.h:114-123, 174-188)
__device__ __forceinline__ void bsdf_f_pdf(const Bsdf& b, const DevMaterial& m, const WoCache& wc, v3 woW, v3 wiW,
                                           bool skipSpecular, v3* f_out, float* pdf_out) {
    *f_out = V3s(0.f);
    *pdf_out = 0.f;
    if (b.lobes == 0) return;
    v3 wi = w2l(b, wiW), wo = w2l(b, woW);
    if (wo.z == 0) return;
    bool reflect = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
    uint32_t mask = lobes_matching(b, skipSpecular);
    int matching = __popc(mask);
    LobeEval e = bsdf_eval_lobes(&m, mask, wo, wi, -1, 0.f, reflect, wc);
    if (reflect) *f_out = e.f;
    *pdf_out = matching > 0 ? e.pdf / matching : 0.f;
}
// the chosen lobe's own Sample_f: returns its f (only used for the specular lobe); *pdf is written only on the paths
// the reference writes it (quirk 8)
__device__ __forceinline__ v3 lobe_sample(const DevMaterial& m, int lobe, v3 wo, v3* wi, float u0, float u1, float* pdf,
                                          const WoCache& wc) {
    if (lobe == LOBE_DIFFUSE || lobe == LOBE_RETRO) {  // BxDF::Sample_f reflection.h:8-15
        *wi = cosine_sample_hemisphere(u0, u1);
        if (wo.z < 0) wi->z *= -1;
        *pdf = same_hemisphere(wo, *wi) ? AbsCosTheta(*wi) * AGPT_INVPI : 0;
        return V3s(1.f);  // value unused: BSDF::Sample_f recomputes f over all lobes
    }
    if (lobe == LOBE_MICROFACET) {  // reflection.h:55-66, Sample_wh microfacet.h:134-140
        if (wo.z == 0) return V3s(0.f);
        bool flip = wo.z < 0;
        v3 wh = tr_sample(flip ? -wo : wo, m.alphax, m.alphay, u0, u1);
        if (flip) wh = -wh;
        if (dot(wo, wh) < 0) return V3s(0.f);
        *wi = reflect3(wo, wh);
        if (!same_hemisphere(wo, *wi)) return V3s(0.f);
        *pdf = tr_pdf(m, wo, wh, wc.G1o) / (4 * dot(wo, wh));
        return V3s(1.f);
    }
    // SpecularReflection::Sample_f reflection.cpp:13-18, FresnelNoOp
    *wi = V3(-wo.x, -wo.y, wo.z);
    *pdf = 1;
    return V3s(1.f) * mat_v3(m.mirror_R) / AbsCosTheta(*wi);
}
// BSDF::Sample_f reflection.h:124-172.  pdf_io: the caller's variable (left untouched on the wo.z==0 early out, quirk 9)
__device__ __forceinline__ v3 bsdf_sample_f(const Bsdf& b, const DevMaterial& m, const WoCache& wc, v3 woW, v3* wiW, float u0,
                                            float u1, float* pdf_io, bool skipSpecular, bool* sampledSpecular) {
    uint32_t mask = lobes_matching(b, skipSpecular);
    int matching = __popc(mask);
    if (matching == 0) {
        *pdf_io = 0;
        return V3s(0.f);
    }
    int comp = (int)floorf(u0 * matching);
    if (matching - 1 < comp) comp = matching - 1;
    // the comp-th matching lobe in list order
    uint32_t mm = mask;
    for (int i = 0; i < comp; ++i) mm &= mm - 1;
    int lobe = __ffs((int)mm) - 1;
    float ur0 = sminf(u0 * matching - comp, AGPT_ONE_MINUS_EPS);
    v3 wi = V3s(0.f), wo = w2l(b, woW);
    if (wo.z == 0) return V3s(0.f);
    *pdf_io = 0;
    if (sampledSpecular) *sampledSpecular = (lobe == LOBE_SPECULAR);
    v3 f = lobe_sample(m, lobe, wo, &wi, ur0, u1, pdf_io, wc);
    if (*pdf_io == 0) return V3s(0.f);
    *wiW = l2w(b, wi);
    if (lobe == LOBE_SPECULAR) {
        if (matching > 1) *pdf_io /= matching;
        return f;
    }
    bool reflect = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
    // *pdf += other lobes' Pdf (in list order), f = sum of all matching lobes' f if reflect
    LobeEval e = bsdf_eval_lobes(&m, mask, wo, wi, matching > 1 ? lobe : -2, *pdf_io, reflect, wc);
    if (matching > 1) *pdf_io = e.pdf / matching;
    return reflect ? e.f : V3s(0.f);
}

// ---- SurfaceInteraction reconstruction ----------------------------------------------------------------------
struct Surface {
    v3 p, wo;
    v3 n;         // geometric normal (after Faceforward for meshes with normals)
    v3 ns;        // shading.n
    v3 ss_bsdf;   // normalize(shading.dpdu)
    int prim;
};

// TriangleIntersect's accepted-hit tail (trianglemesh.cpp:82-111) + SurfaceInteraction ctor / SetShadingGeometry
// (intersectable.h:66-89, quirk 2) from the precomputed per-triangle record
__device__ __forceinline__ void surface_from_triangle(const DevScene& sc, uint32_t tri, float b1, float b2, v3 O, v3 D,
                                                      float t, Surface& s) {
    const float4* q = sc.tri_shade + 4 * (size_t)tri;
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    v3 ng = V3(q0.x, q0.y, q0.z);
    v3 ss0 = V3(q0.w, q1.x, q1.y);
    s.prim = (int)__float_as_uint(q3.w);
    s.p = O + t * D;
    s.wo = -D;
    s.n = ng;
    s.ns = ng;
    s.ss_bsdf = ss0;
    if (sc.prims[s.prim].has_normals) {
        v3 n0 = V3(q1.z, q1.w, q2.x), n1 = V3(q2.y, q2.z, q2.w), n2 = V3(q3.x, q3.y, q3.z);
        float b0 = 1.f - b1 - b2;
        v3 ns = n0 * b0 + n1 * b1 + n2 * b2;
        if (sqrlen(ns) > 0.f)
            ns = normalize(ns);
        else
            ns = ng;
        v3 ss = ss0;
        v3 ts = cross(ss, ns);
        if (sqrlen(ts) > 0.f) {
            ts = normalize(ts);
            ss = cross(ts, ns);
        } else
            coordinate_system(ns, &ss, &ts);
        s.ns = normalize(cross(ss, ts));
        s.n = faceforward(ng, s.ns);
    }
}
// Sphere::Intersect's tail (intersectable.h:183-201): dpdu/dpdv are passed swapped.  Also Plane::Intersect's
// (intersectable.h:133): SurfaceInteraction(P, uv, -D, dpdu = (0,0,1), dpdv = (1,0,0)).
__device__ __forceinline__ void surface_from_sphere(const DevScene& sc, int prim, v3 O, v3 D, float t, Surface& s) {
    const DevPrim& P = sc.prims[prim];
    v3 p = O + t * D;
    if (P.type == AGPT_PRIM_PLANE) {
        s.prim = prim;
        s.p = p;
        s.wo = -D;
        s.n = normalize(cross(V3(0, 0, 1), V3(1, 0, 0)));
        s.ns = s.n;
        s.ss_bsdf = normalize(V3(0, 0, 1));
        return;
    }
    v3 pHit = p - V3(P.cx, P.cy, P.cz);
    if (pHit.x == 0 && pHit.y == 0) pHit.x = AGPT_EPSILON * P.r;
    float theta = cr_acosf(tclampf(pHit.z / P.r, -1.f, 1.f));
    float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius;
    float sinPhi = pHit.y * invZRadius;
    v3 dpdu = V3(-AGPT_TWOPI * pHit.y, AGPT_TWOPI * pHit.x, 0);
    v3 dpdv = AGPT_PI * V3(pHit.z * cosPhi, pHit.z * sinPhi, -P.r * cr_sinf(theta));
    s.prim = prim;
    s.p = p;
    s.wo = -D;
    s.n = normalize(cross(dpdv, dpdu));
    s.ns = s.n;
    s.ss_bsdf = normalize(dpdv);
}

// ---- lights -------------------------------------------------------------------------------------------------
// Sphere::Sample(ref, u, pdf) intersectable.h:239-304 (+ Sample(u) :230-237 when ref is inside)
__device__ __forceinline__ void sphere_sample_ref(const DevPrim& P, v3 refp, float u0, float u1, v3* p_out, float* pdf) {
    v3 pCenter = V3(P.cx, P.cy, P.cz);
    if (sqrlen(refp - pCenter) <= P.r2) {
        v3 pObj = pCenter + P.r * random_in_sphere_u(u0, u1);
        v3 n = normalize(pObj);  // quirk 6
        *pdf = 1 / (4.f * AGPT_PI * P.r2);
        v3 wi = pObj - refp;
        if (sqrlen(wi) == 0)
            *pdf = 0;
        else {
            wi = normalize(wi);
            *pdf *= sqrlen(refp - pObj) / absdot(n, -wi);
        }
        if (isinf(*pdf)) *pdf = 0;
        *p_out = pObj;
        return;
    }
    float dc = length(refp - pCenter);
    float invDc = 1 / dc;
    v3 wc = (pCenter - refp) * invDc;
    v3 wcX, wcY;
    coordinate_system(wc, &wcX, &wcY);
    float sinThetaMax = P.r * invDc;
    float sinThetaMax2 = sinThetaMax * sinThetaMax;
    float invSinThetaMax = 1 / sinThetaMax;
    float cosThetaMax = sqrtf(smaxf(0.f, 1 - sinThetaMax2));
    float cosTheta = (cosThetaMax - 1) * u0 + 1;
    float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * u0;
        cosTheta = sqrtf(1 - sinTheta2);
    }
    float cosAlpha = sinTheta2 * invSinThetaMax +
                     cosTheta * sqrtf(smaxf(0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    float sinAlpha = sqrtf(smaxf(0.f, 1.f - cosAlpha * cosAlpha));
    float phi = u1 * 2 * AGPT_PI;
    v3 nWorld = spherical_direction(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    *p_out = pCenter + P.r * V3(nWorld.x, nWorld.y, nWorld.z);
    *pdf = 1 / (2 * AGPT_PI * (1 - cosThetaMax));
}
// Sphere::Pdf intersectable.h:306-317
__device__ __forceinline__ float sphere_pdf(const DevPrim& P, v3 refp) {
    v3 pCenter = V3(P.cx, P.cy, P.cz);
    if (sqrlen(refp - pCenter) <= P.r2) return 1 / (4 * AGPT_PI);
    float sinThetaMax2 = P.r2 / sqrlen(refp - pCenter);
    float cosThetaMax = sqrtf(smaxf(0.f, 1 - sinThetaMax2));
    return 1 / (2 * AGPT_PI * (1 - cosThetaMax));
}
// ---- InfiniteAreaLight (lights.cpp:50-112, #define ILS) ------------------------------------------------------
__device__ __forceinline__ int env_mod(int a, int b) {  // HDRTexture::Mod texture.h:76-79
    int result = a - (a / b) * b;
    return (result < 0) ? result + b : result;
}
__device__ __forceinline__ v3 env_value(const DevEnv& e, float u, float v) {  // HDRTexture::value texture.h:59-67
    int si = (int)floorf(u * e.width - .5f);
    int ti = (int)floorf(v * e.height - .5f);
    float4 p = e.pixels[env_mod(ti, e.height) * e.width + env_mod(si, e.width)];
    return V3(p.x, p.y, p.z);
}
__device__ __forceinline__ float spherical_theta(v3 v) { return cr_acosf(tclampf(v.z, -1.f, 1.f)); }  // common.h:158-160
__device__ __forceinline__ float spherical_phi(v3 v) {                                                // common.h:162-165
    float p = cr_atan2f(v.y, v.x);
    return (p < 0) ? (p + AGPT_TWOPI) : p;
}
__device__ __noinline__ v3 env_Le(const DevEnv* e, v3 D) {  // InfiniteAreaLight::Le lights.cpp:108-112
    v3 w = normalize(D);
    w = V3(w.x, w.z, w.y);
    return env_value(*e, spherical_phi(w) * AGPT_INV2PI, spherical_theta(w) * AGPT_INVPI);
}
// Distribution1D::SampleContinuous (sampling.h:37-52) with FindInterval (:4-17)
__device__ __forceinline__ float env_sample_continuous(const DevEnv& e, float u, float* pdf) {
    int size = e.n + 1;
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (e.cdf[middle] <= u) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    int offset = first - 1;
    if (size - 2 < offset) offset = size - 2;
    if (offset < 0) offset = 0;
    float c0 = e.cdf[offset], c1 = e.cdf[offset + 1];
    float du = u - c0;
    if ((c1 - c0) > 0) du /= c1 - c0;
    *pdf = (e.funcInt > 0) ? e.func[offset] / e.funcInt : 0;
    return (offset + du) / e.n;
}
// InfiniteAreaLight::Sample_Li lights.cpp:50-90: wi and pdf for one uniform draw u; returns false when mapPdf == 0
__device__ __noinline__ bool env_sample_li(const DevEnv* ep, float u, v3* wi, float* pdf) {
    const DevEnv& e = *ep;
    float mapPdf;
    float sample = env_sample_continuous(e, u, &mapPdf);
    if (mapPdf == 0) return false;
    int idx = (int)(sample * e.n);
    float uvx = ((idx % e.width) + .5f) / e.width;
    float uvy = ((idx / e.width) + .5f) / e.height;
    float theta = uvy * AGPT_PI, phi = uvx * AGPT_TWOPI;
    float sinTheta, cosTheta, sinPhi, cosPhi;
    cr_sincosf(theta, &sinTheta, &cosTheta);
    cr_sincosf(phi, &sinPhi, &cosPhi);
    *wi = V3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi);
    *pdf = mapPdf / (2 * AGPT_PI * AGPT_PI * sinTheta);
    if (sinTheta == 0) *pdf = 0;
    return true;
}
// InfiniteAreaLight::Pdf_Li lights.cpp:92-106
__device__ __noinline__ float env_pdf_li(const DevEnv* ep, v3 wi) {
    const DevEnv& e = *ep;
    v3 w = normalize(wi);
    w = V3(w.x, w.z, w.y);
    float theta = spherical_theta(w), phi = spherical_phi(w);
    float sinTheta = cr_sinf(theta);
    if (sinTheta == 0) return 0;
    int x = (int)(phi * AGPT_INV2PI * e.width);
    if (e.width - 1 < x) x = e.width - 1;
    if (x < 0) x = 0;
    int y = (int)(theta * AGPT_INVPI * e.height);
    if (e.height - 1 < y) y = e.height - 1;
    if (y < 0) y = 0;
    float discrete = e.func[y * e.width + x] / (e.funcInt * e.n);
    return e.n * discrete / (2 * AGPT_PI * AGPT_PI * sinTheta);
}

// integrator.h:33-36
__device__ __forceinline__ float power_heuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
