/* oracle/oracle_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never
 * imported by the product path).
 *
 * Scalar/vec3 helpers restating the reference's L1 math exactly (operation order and
 * comparator semantics matter: every expression below is evaluated in fp32, left to
 * right, no FMA contraction -- build with -ffp-contract=off).
 *
 * Follows (paths relative to the reference checkout):
 *   template/precomp.h:364-365   fminf/fmaxf re-defined as  a<b?a:b  /  a>b?a:b
 *   template/precomp.h:676       Lerp(t,s1,s2) = (1-t)*s1 + t*s2
 *   template/precomp.h:678       clamp(f,a,b) = fmaxf(a, fminf(f,b))
 *   template/precomp.h:701-768   dot, absdot, Faceforward, SameHemisphere, IsBlack,
 *                                Luminance, normalize (= v * (1/sqrtf(dot))), Reflect, cross
 *   template/common.h:10-18      PI, INVPI, INV2PI, TWOPI, EPSILON, FloatOneMinusEpsilon
 */
#ifndef AGPT_ORACLE_MATH_H
#define AGPT_ORACLE_MATH_H

#include <math.h>
#include <float.h>
#include <stdint.h>

#define O_PI      3.14159265358979323846264f
#define O_INVPI   0.31830988618379067153777f
#define O_INV2PI  0.15915494309189533576888f
#define O_TWOPI   6.28318530717958647692528f
#define O_EPSILON 0.0001f
static const float O_ONE_MINUS_EPS = 0x1.fffffep-1f;

typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;

/* template/precomp.h:364-365 -- NOT libm semantics (NaN handling differs) */
static inline float t_fminf(float a, float b) { return a < b ? a : b; }
static inline float t_fmaxf(float a, float b) { return a > b ? a : b; }
/* std::min / std::max as libstdc++ defines them */
static inline float s_minf(float a, float b) { return (b < a) ? b : a; }
static inline float s_maxf(float a, float b) { return (a < b) ? b : a; }
static inline float t_clampf(float f, float a, float b) { return t_fmaxf(a, t_fminf(f, b)); }

static inline f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline f3 F3s(float s) { return F3(s, s, s); }
static inline f3 f3_add(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_mul(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 f3_muls(f3 a, float b) { return F3(a.x * b, a.y * b, a.z * b); }
static inline f3 f3_divs(f3 a, float b) { return F3(a.x / b, a.y / b, a.z / b); }
static inline f3 f3_neg(f3 a) { return F3(-a.x, -a.y, -a.z); }
static inline float f3_get(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float f3_absdot(f3 a, f3 b) { return fabsf(f3_dot(a, b)); }
static inline float f3_sqrlen(f3 a) { return f3_dot(a, a); }
static inline float f3_len(f3 a) { return sqrtf(f3_dot(a, a)); }
static inline f3 f3_cross(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 f3_normalize(f3 v) {
    float invLen = 1.0f / sqrtf(f3_dot(v, v));
    return f3_muls(v, invLen);
}
static inline f3 f3_faceforward(f3 v, f3 v2) { return (f3_dot(v, v2) < 0.f) ? f3_neg(v) : v; }
static inline int f3_isblack(f3 v) { return v.x == 0 && v.y == 0 && v.z == 0; }
static inline int f3_hasnans(f3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }
static inline float f3_luminance(f3 v) { return 0.212671f * v.x + 0.715160f * v.y + 0.072169f * v.z; }
static inline f3 f3_lerp(float t, f3 s1, f3 s2) { return f3_add(f3_muls(s1, 1 - t), f3_muls(s2, t)); }
/* Reflect(wo,n) = -wo + 2*dot(wo,n)*n */
static inline f3 f3_reflect(f3 wo, f3 n) { return f3_add(f3_neg(wo), f3_muls(n, 2.0f * f3_dot(wo, n))); }
static inline int same_hemisphere(f3 w, f3 wp) { return w.z * wp.z > 0; }

#endif
