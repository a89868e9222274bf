// agpt_math.h -- fp32 vector helpers shared by the host scene builder and the gfx950 kernels.
//
// Every expression is written in the operation order the reference uses so that, compiled with
// -ffp-contract=off (no FMA fusion) and IEEE divide/sqrt, host and device produce the same bits as a
// strict-IEEE build of the reference's L1 math (template/precomp.h:364-369,671-768; template/common.h).
// Trigonometric functions are the one implementation-defined ingredient of the reference (it calls the
// platform libm); here they are defined as the correctly rounded fp32 value, computed through fp64
// (MI355X runs fp64 vector math at half the fp32 rate, and shading is <3 % of the path).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define AGPT_HD __host__ __device__ __forceinline__

#define AGPT_PI 3.14159265358979323846264f
#define AGPT_INVPI 0.31830988618379067153777f
#define AGPT_INV2PI 0.15915494309189533576888f
#define AGPT_TWOPI 6.28318530717958647692528f
#define AGPT_EPSILON 0.0001f
#define AGPT_ONE_MINUS_EPS 0x1.fffffep-1f
#define AGPT_FLT_MAX 3.402823466e+38f

struct v3 {
    float x, y, z;
};
struct v2 {
    float x, y;
};

AGPT_HD v3 V3(float x, float y, float z) {
    v3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
AGPT_HD v3 V3s(float s) { return V3(s, s, s); }
AGPT_HD v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
AGPT_HD v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
AGPT_HD v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
AGPT_HD v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
AGPT_HD v3 operator*(v3 a, float b) { return V3(a.x * b, a.y * b, a.z * b); }
AGPT_HD v3 operator*(float b, v3 a) { return V3(b * a.x, b * a.y, b * a.z); }
AGPT_HD v3 operator/(v3 a, float b) { return V3(a.x / b, a.y / b, a.z / b); }

// template/precomp.h:364-365: comparators, not libm fminf/fmaxf
AGPT_HD float tminf(float a, float b) { return a < b ? a : b; }
AGPT_HD float tmaxf(float a, float b) { return a > b ? a : b; }
// std::min / std::max
AGPT_HD float sminf(float a, float b) { return (b < a) ? b : a; }
AGPT_HD float smaxf(float a, float b) { return (a < b) ? b : a; }
AGPT_HD float tclampf(float f, float a, float b) { return tmaxf(a, tminf(f, b)); }

AGPT_HD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
AGPT_HD float absdot(v3 a, v3 b) { return fabsf(dot(a, b)); }
AGPT_HD float sqrlen(v3 a) { return dot(a, a); }
AGPT_HD float length(v3 a) { return sqrtf(dot(a, a)); }
AGPT_HD v3 cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
AGPT_HD v3 normalize(v3 v) {
    float invLen = 1.0f / sqrtf(dot(v, v));
    return v * invLen;
}
AGPT_HD v3 faceforward(v3 v, v3 v2) { return (dot(v, v2) < 0.f) ? -v : v; }
AGPT_HD bool isblack(v3 v) { return v.x == 0 && v.y == 0 && v.z == 0; }
AGPT_HD v3 lerp3(float t, v3 s1, v3 s2) { return (1 - t) * s1 + t * s2; }
AGPT_HD v3 reflect3(v3 wo, v3 n) { return -wo + 2.0f * dot(wo, n) * n; }
AGPT_HD bool same_hemisphere(v3 w, v3 wp) { return w.z * wp.z > 0; }
AGPT_HD float luminance(v3 v) { return 0.212671f * v.x + 0.715160f * v.y + 0.072169f * v.z; }
AGPT_HD float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// correctly rounded fp32 trig through fp64
// sin and cos of an fp32 angle, each correctly rounded to fp32, through a bounded-range fp64 evaluation:
// Cody-Waite reduction by pi/2 (two-term, exact for the |x| <= 8 this path ever sees: phi in [0, 2pi], concentric-map
// angles in [-pi/4, 3pi/4], acos results in [0, pi]) and the fdlibm minimax kernels on [-pi/4, pi/4].  fp64 error
// <= 1 ulp(fp64), so the fp32 rounding differs from libm's (float)sin((double)x) only if the value lies within ~2^-53
// relative of an fp32 rounding boundary (none in 2e8 sampled arguments).  ~35 fp64 ops vs ~250 for the generic routine.
AGPT_HD void cr_sincosf(float xf, float* s, float* c) {
    const double x = (double)xf;
    if (!(fabs(x) <= 8.0)) {
        // Not reachable from the hot path (every caller's argument range is listed above); NaN and +-inf arguments give NaN
        // like sin / cos.  A generic-range fallback here was inlined into every caller -- ~500 cold fp64 instructions and the
        // register peak of k_shade each.
        *s = __builtin_nanf("");
        *c = __builtin_nanf("");
        return;
    }
    const double kd = rint(x * 0.63661977236758134308);
    const int k = (int)kd;
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
    const double r2 = r * r;
    double ps = __builtin_fma(1.58969099521155010221e-10, r2, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, r2, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, r2, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, r2, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, r2, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(r * r2, ps, r);
    double pc = __builtin_fma(-1.13596475577881948265e-11, r2, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, r2, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, r2, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, r2, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, r2, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(r2 * r2, pc, __builtin_fma(-0.5, r2, 1.0));
    double ss = (k & 1) ? cs : sn, cc = (k & 1) ? sn : cs;
    if (k & 2) ss = -ss;
    if ((k + 1) & 2) cc = -cc;
    *s = (float)ss;
    *c = (float)cc;
}
AGPT_HD float cr_sinf(float x) {
    float s, c;
    cr_sincosf(x, &s, &c);
    return s;
}
AGPT_HD float cr_cosf(float x) {
    float s, c;
    cr_sincosf(x, &s, &c);
    return c;
}
AGPT_HD float cr_acosf(float x) { return (float)acos((double)x); }
AGPT_HD float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

// template/common.h:145-151
AGPT_HD void coordinate_system(v3 v1, v3* v2, v3* v3o) {
    if (fabsf(v1.x) > fabsf(v1.y))
        *v2 = V3(-v1.z, 0, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3(0, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
    *v3o = cross(v1, *v2);
}

// RNG: template/template.cpp:667-675 (xorshift32) seeded per (pixel, sample) as cl/tools.cl:1-2 prescribes
AGPT_HD uint32_t wang_hash(uint32_t s) {
    s = (s ^ 61) ^ (s >> 16);
    s *= 9;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2d;
    s = s ^ (s >> 15);
    return s;
}
// One RandomFloat() stream per (pixel, sample): WangHash((pixel + W*H*sample + 1) * 17 + seed_base) (seeding convention of
// cl/tools.cl:1-2).  The stream index is a 32-bit word like the hash's argument: it wraps once pixel + W*H*sample passes 2^32 --
// at 3840x2160 from sample 518 on -- so a budget beyond that (BASELINE configs[4]: 1024 spp at 4K) re-uses streams of other
// pixels' earlier samples.  The oracle indexes the same way (parity holds); hosts that need more than 2^32 distinct streams
// split the budget over several seed_base values.
AGPT_HD uint32_t sample_seed(uint32_t pixel, uint32_t wh, uint32_t sample, uint32_t seed_base) {
    uint32_t idx = pixel + wh * sample;
    uint32_t s = wang_hash((idx + 1u) * 17u + seed_base);
    if (s == 0) s = 0x12345678u;
    return s;
}
AGPT_HD uint32_t rng_uint(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}
AGPT_HD float rng_float(uint32_t& s) { return rng_uint(s) * 2.3283064365387e-10f; }
