// agpt_trace.h -- device-side ray queries: Scene::Intersect / IntersectP as an iterative, per-lane-stack
// traversal of the reference's BVH.
//
// Reference semantics reproduced exactly (bit-identical t/b1/b2/primitive to a strict-IEEE build):
//   Bounds::Intersect          bvhtrimesh.h:18-36    slab test, (bmin-O)/D per axis, comparator min/max,
//                                                    reject iff tmax*1.00000024f < tmin, returns entry tmin
//   BVHTriMesh::RecursiveHit   bvhtrimesh.h:332-384  both children tested at the CURRENT ray.t, near child first
//                                                    (swap iff rightDist < leftDist), far child visited iff both
//                                                    were hit -- and NOT re-tested after ray.t shrinks
//   BVHTriMesh::RecursiveHitP  bvhtrimesh.h:386-413  any-hit, fixed left-then-right order, early out
//   TriangleIntersect / P      trianglemesh.cpp:7-43,117-155  Moeller-Trumbore, det==0 reject only, two-sided
//   Sphere::Intersect / P      intersectable.h:164-181,207-226
//   Scene::Intersect / P       scene.h:5-19          linear walk over primitives sharing ray.t
//
// The six true divisions per box of the reference are evaluated as q = a*r; q += fma(-b,q,a)*r with
// r = RN(1/D) computed once per ray (Markstein's correction): bit-identical to a/b (checked over 6.4e9
// operand pairs incl. adversarial mantissas, and by tests/test_gpu_intersect.py), 3 full-rate VALU ops
// instead of the ~11-op IEEE divide expansion.  Rays with a direction component below 2^-40 (incl. 0, where
// the reference relies on inf/NaN propagation) take the true-division path.
#pragma once

#include "agpt_scene.h"

struct TraceRay {
    v3 O, D;
    v3 R;        // RN(1/D)
    bool fast;   // all |D| >= 2^-40
};

__device__ __forceinline__ TraceRay make_trace_ray(v3 O, v3 D) {
    TraceRay r;
    r.O = O;
    r.D = D;
    r.R = V3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    const float lim = 0x1p-40f;
    r.fast = (fabsf(D.x) >= lim) && (fabsf(D.y) >= lim) && (fabsf(D.z) >= lim);
    return r;
}

__device__ __forceinline__ float mdiv(float a, float b, float r) {
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}

// one axis of Bounds::Intersect
template <bool FAST>
__device__ __forceinline__ void slab_axis(float bmin, float bmax, float O, float D, float R, float& tmin, float& tmax,
                                          bool& ok) {
    float lo = bmin - O, hi = bmax - O;
    float q0 = FAST ? mdiv(lo, D, R) : lo / D;
    float q1 = FAST ? mdiv(hi, D, R) : hi / D;
    float t0 = tminf(q0, q1);
    float t1 = tmaxf(q0, q1);
    tmin = tmaxf(t0, tmin);
    tmax = tminf(t1, tmax);
    ok = ok && !((tmax * 1.00000024f) < tmin);
}

template <bool FAST>
__device__ __forceinline__ bool box_test_t(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                           const TraceRay& r, float rayt, float& dist) {
    float tmin = 0.0f, tmax = rayt;
    bool ok = true;
    slab_axis<FAST>(bx0, bx1, r.O.x, r.D.x, r.R.x, tmin, tmax, ok);
    slab_axis<FAST>(by0, by1, r.O.y, r.D.y, r.R.y, tmin, tmax, ok);
    slab_axis<FAST>(bz0, bz1, r.O.z, r.D.z, r.R.z, tmin, tmax, ok);
    dist = tmin;
    return ok;
}

// Fast-path slab test (all |D| >= 2^-40: every quotient is a finite non-NaN number).  Equivalent to the reference's
// per-axis chain (bvhtrimesh.h:22-33) because
//   * with no NaNs the comparator min/max (a<b?a:b) and the hardware v_min/v_max agree up to the sign of zero,
//     which no later comparison can observe;
//   * tmin_k is non-decreasing and tmax_k non-increasing over the axes and x -> x*1.00000024f is monotone, so
//     "some prefix has tmax_k*c < tmin_k" holds iff it holds after the last axis: one reject test instead of three.
// 6 sub + 18 (Markstein divides) + 6 min/max + 2 max3/min3-class + 2 clamp + 2 = 36 VALU ops per box (was 57).
__device__ __forceinline__ bool box_test_fast(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                              const TraceRay& r, float rayt, float& dist) {
    float q0x = mdiv(bx0 - r.O.x, r.D.x, r.R.x), q1x = mdiv(bx1 - r.O.x, r.D.x, r.R.x);
    float q0y = mdiv(by0 - r.O.y, r.D.y, r.R.y), q1y = mdiv(by1 - r.O.y, r.D.y, r.R.y);
    float q0z = mdiv(bz0 - r.O.z, r.D.z, r.R.z), q1z = mdiv(bz1 - r.O.z, r.D.z, r.R.z);
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(q0x, q1x), __builtin_fminf(q0y, q1y)),
                                 __builtin_fmaxf(__builtin_fminf(q0z, q1z), 0.0f));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(q0x, q1x), __builtin_fmaxf(q0y, q1y)),
                                 __builtin_fminf(__builtin_fmaxf(q0z, q1z), rayt));
    dist = tmin;
    return !((tmax * 1.00000024f) < tmin);
}
// The two children of an interior node come from one interleaved 64-B pair record (agpt_host_scene.cpp:
// flatten_scene): the four 16-B loads land so that no register shuffling is needed before the slab arithmetic.
// (Issuing the two boxes as packed fp32 -- v_pk_add/mul/fma_f32 on (left, right) pairs -- was measured 5 % SLOWER on
// gfx950 than the scalar form below: a packed op costs more than two scalar ones here.  For the same reason the
// library is built with -fno-slp-vectorize, see build.py.)
struct NodePair {
    float4 n0, n1, n2, n3;
};
__device__ __forceinline__ NodePair load_pair(const DevScene& sc, uint32_t left) {
    const float4* np = sc.nodes + 2 * (size_t)left;
    NodePair p;
    p.n0 = np[0];
    p.n1 = np[1];
    p.n2 = np[2];
    p.n3 = np[3];
    return p;
}
__device__ __forceinline__ void pair_boxes_fast(const NodePair& p, const TraceRay& r, float rayt, bool& hl, bool& hr,
                                                float& dl, float& dr) {
    hl = box_test_fast(p.n0.x, p.n0.z, p.n1.x, p.n1.z, p.n2.x, p.n2.z, r, rayt, dl);
    hr = box_test_fast(p.n0.y, p.n0.w, p.n1.y, p.n1.w, p.n2.y, p.n2.w, r, rayt, dr);
}
// reference-order (true division) tests of the same record
__device__ __forceinline__ void pair_boxes_exact(const NodePair& p, const TraceRay& r, float rayt, bool& hl, bool& hr,
                                                 float& dl, float& dr) {
    hl = box_test_t<false>(p.n0.x, p.n0.z, p.n1.x, p.n1.z, p.n2.x, p.n2.z, r, rayt, dl);
    hr = box_test_t<false>(p.n0.y, p.n0.w, p.n1.y, p.n1.w, p.n2.y, p.n2.w, r, rayt, dr);
}
// (first slot, count) of a leaf from its traversal encoding
__device__ __forceinline__ void leaf_range(const DevScene& sc, uint32_t enc, uint32_t& first, uint32_t& count) {
    if ((enc & AGPT_ENC_BIGLEAF) == AGPT_ENC_BIGLEAF) {
        const uint32_t k = enc & 0x0FFFFFFFu;
        first = sc.bigleaves[2 * k];
        count = sc.bigleaves[2 * k + 1];
    } else {
        first = enc & 0x0FFFFFFFu;
        count = ((enc >> 28) & 7u) + 1u;
    }
}

__device__ __forceinline__ bool box_test(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                         const TraceRay& r, float rayt, float& dist) {
    if (__builtin_expect(r.fast, 1)) return box_test_fast(bx0, by0, bz0, bx1, by1, bz1, r, rayt, dist);
    return box_test_t<false>(bx0, by0, bz0, bx1, by1, bz1, r, rayt, dist);
}

// trianglemesh.cpp:13-43 (and :117-155): returns true and t,b1,b2 if accepted against rayt
__device__ __forceinline__ bool tri_test(v3 v0, v3 v1, v3 v2, const TraceRay& r, float rayt, float& t_out, float& b1_out,
                                         float& b2_out) {
    v3 e1 = v1 - v0;
    v3 e2 = v2 - v0;
    v3 pvec = cross(r.D, e2);
    float det = dot(e1, pvec);
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    v3 tvec = r.O - v0;
    float b1 = dot(tvec, pvec) * inv_det;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    v3 qvec = cross(tvec, e1);
    float b2 = dot(r.D, qvec) * inv_det;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    float t = dot(e2, qvec) * inv_det;
    if (t <= 0.0f || t >= rayt) return false;
    t_out = t;
    b1_out = b1;
    b2_out = b2;
    return true;
}

// intersectable.h:164-181 / :207-226: nearest root in [0, rayt]
__device__ __forceinline__ bool sphere_test_c(v3 center, float r2, const TraceRay& r, float rayt, float& root_out) {
    v3 oc = r.O - center;
    float half_b = dot(oc, r.D);
    float c = sqrlen(oc) - r2;
    float discriminant = half_b * half_b - c;
    if (discriminant < 0) return false;
    float sqrtd = sqrtf(discriminant);
    float root = -half_b - sqrtd;
    if (root < 0 || rayt < root) {
        root = -half_b + sqrtd;
        if (root < 0 || rayt < root) return false;
    }
    root_out = root;
    return true;
}
// Plane::Intersect / IntersectP (intersectable.h:123-150): XZ rectangle, accepted for 0 < t < ray.t
__device__ __forceinline__ bool plane_test_c(v3 O, float hx, float hz, const TraceRay& r, float rayt, float& t_out) {
    if (r.D.y == 0) return false;
    float t = (O.y - r.O.y) / r.D.y;
    if (t <= 0 || t >= rayt) return false;
    v3 P = r.O + t * r.D;
    float u = (P.x - O.x) / hx;
    float v = (P.z - O.z) / hz;
    if (!(fabsf(u) <= 1 && fabsf(v) <= 1)) return false;
    t_out = t;
    return true;
}
// analytic primitives: kind 1 = sphere (b.x = r2), kind 3 = plane (b.x, b.y = half size)
__device__ __forceinline__ bool analytic_test(float kind, v3 a, float bx, float by, const TraceRay& r, float rayt, float& t_out) {
    if (kind == 1.f) return sphere_test_c(a, bx, r, rayt, t_out);
    return plane_test_c(a, bx, by, r, rayt, t_out);
}
__device__ __forceinline__ bool sphere_test(const DevPrim& P, const TraceRay& r, float rayt, float& root_out) {
    if (P.type == AGPT_PRIM_PLANE) return plane_test_c(V3(P.cx, P.cy, P.cz), P.r, P.r2, r, rayt, root_out);
    return sphere_test_c(V3(P.cx, P.cy, P.cz), P.r2, r, rayt, root_out);
}

struct TraceCounters {
    uint32_t interior;
    uint32_t tris;
    uint32_t roots;
    uint32_t last_mesh;
};

// Scene::Intersect (ANY=false) / Scene::IntersectP (ANY=true) for one ray on one lane.
// stack: this lane's LDS column, element k at stack[k * stride].
//
// The reference walks Scene::primitives in order and finishes each BVHTriMesh before touching the next one
// (scene.h:5-19).  A wave doing that in lockstep idles at every mesh boundary, so the walk is split in two:
//   phase A (wave-uniform, primitive records come through scalar loads): test every primitive's root box /
//           sphere against the ray's INITIAL t and keep a per-lane bit mask of the ones that can be hit at all --
//           the slab test and the sphere test are monotone in ray.t, so a primitive missed at the initial t is
//           missed at any later (smaller) t and the reference would skip it too;
//   phase B (per lane): visit the surviving primitives in list order, RE-TESTING the root box at the current t
//           exactly where the reference tests it (bvhtrimesh.h:187,195), then traverse.  Lanes move on to their
//           next primitive independently.
// Decisions, visit order, accepted hits and the work counters are those of the reference's recursion.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool trace_scene(const DevScene& sc, v3 O, v3 D, float tmax_in, DevHit& hit,
                                            uint32_t* stack, int stride, TraceCounters& cnt) {
    TraceRay r = make_trace_ray(O, D);
    float rayt = tmax_in;
    hit.t = tmax_in;
    hit.id = AGPT_HIT_MISS;
    hit.b1 = 0.f;
    hit.b2 = 0.f;

    for (int base = 0; base < sc.n_prims; base += 64) {
        const int nchunk = min(64, sc.n_prims - base);
        // ---- phase A ------------------------------------------------------------------------------------
        unsigned long long mask = 0;
        for (int k = 0; k < nchunk; ++k) {
            const DevPrim& P = sc.prims[base + k];
            bool h;
            if (P.type != AGPT_PRIM_MESH) {
                float root;
                h = sphere_test(P, r, rayt, root);
            } else {
                float dist;
                h = P.n_tris > 0 && box_test(P.root_bmin[0], P.root_bmin[1], P.root_bmin[2], P.root_bmax[0], P.root_bmax[1],
                                             P.root_bmax[2], r, rayt, dist);
            }
            mask |= (unsigned long long)(h ? 1u : 0u) << k;
        }
        // ---- phase B ------------------------------------------------------------------------------------
        uint32_t cur = 0;
        bool in_mesh = false;
        int sp = 0;
        for (;;) {
            if (!in_mesh) {
                if (mask == 0) break;
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int pi = base + k;
                const DevPrim& P = sc.prims[pi];
                if (P.type != AGPT_PRIM_MESH) {
                    float root;
                    if (sphere_test(P, r, rayt, root)) {
                        if (ANY) {
                            if (COUNT) cnt.roots += (uint32_t)P.mesh_ordinal;
                            return true;
                        }
                        rayt = root;
                        hit.t = root;
                        hit.id = AGPT_HIT_SPHERE | (uint32_t)pi;
                        hit.b1 = 0.f;
                        hit.b2 = 0.f;
                    }
                    continue;
                }
                float dist;
                if (!box_test(P.root_bmin[0], P.root_bmin[1], P.root_bmin[2], P.root_bmax[0], P.root_bmax[1], P.root_bmax[2],
                              r, rayt, dist))
                    continue;
                cur = P.root_enc;
                sp = 0;
                in_mesh = true;
                if (ANY && COUNT) cnt.last_mesh = (uint32_t)P.mesh_ordinal + 1u;
            }
            if (!(cur & AGPT_ENC_LEAF)) {
                // interior: fetch the child pair (one 64-B line)
                const NodePair np = load_pair(sc, cur);
                if (COUNT) cnt.interior++;
                float dl, dr;
                bool hl, hr;
                if (__builtin_expect(r.fast, 1))
                    pair_boxes_fast(np, r, rayt, hl, hr, dl, dr);
                else
                    pair_boxes_exact(np, r, rayt, hl, hr, dl, dr);
                const uint32_t encl = __float_as_uint(np.n3.x), encr = __float_as_uint(np.n3.y);
                if (hl && hr) {
                    bool swap = ANY ? false : (dr < dl);
                    uint32_t nearn = swap ? encr : encl;
                    uint32_t farn = swap ? encl : encr;
                    stack[sp * stride] = farn;
                    sp++;
                    cur = nearn;
                    continue;
                } else if (hl) {
                    cur = encl;
                    continue;
                } else if (hr) {
                    cur = encr;
                    continue;
                }
            } else {
                uint32_t first, count;
                leaf_range(sc, cur, first, count);
                for (uint32_t i = 0; i < count; ++i) {
                    const float4* tp = sc.tri_verts + 3 * (size_t)(first + i);
                    float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    if (COUNT) cnt.tris++;
                    float t, b1, b2;
                    if (tri_test(V3(q0.x, q0.y, q0.z), V3(q1.x, q1.y, q1.z), V3(q2.x, q2.y, q2.z), r, rayt, t, b1, b2)) {
                        if (ANY) {
                            if (COUNT) cnt.roots += cnt.last_mesh;
                            return true;
                        }
                        // quirk 11: a degenerate triangle is rejected after the t test without touching hit / ray.t
                        if (!(__float_as_uint(q1.w) & AGPT_TRI_FLAG_REJECT)) {
                            rayt = t;
                            hit.t = t;
                            hit.id = __float_as_uint(q0.w);
                            hit.b1 = b1;
                            hit.b2 = b2;
                        }
                    }
                }
            }
            if (sp == 0) {
                in_mesh = false;
                continue;
            }
            sp--;
            cur = stack[sp * stride];
        }
    }
    // the reference root-tests every non-empty mesh of the list on a full walk (bvhtrimesh.h:187,195)
    if (COUNT) cnt.roots += (uint32_t)sc.n_meshes;
    return hit.id != AGPT_HIT_MISS;
}
