/* oracle/oracle.c -- TEST INFRASTRUCTURE ONLY (see agpt_oracle.h).
 *
 * Plain-C restatement of the reference's CPU path tracer hot path.  Every function cites the
 * reference file:line it follows (paths relative to the reference checkout).  Deliberately literal:
 * recursive traversal, AoS records, per-hit SurfaceInteraction construction, no precomputation --
 * the product's HIP path is structured differently (wavefront, SoA, precomputed per-triangle frames)
 * and is checked against this.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  fp32 everywhere, libm
 * sinf/cosf/acosf/atan2f/tanf as the reference's std:: calls resolve to.
 */
#include "agpt_oracle.h"
#include "oracle_math.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * RNG -- template/template.cpp:667-675 (xorshift32, float = uint * 2.3283064365387e-10f);
 * per-thread seeding convention cl/tools.cl:1-2 (WangHash((idx+1)*17)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t* rng;
    oracle_stats st;
} octx;

static inline uint32_t rng_uint(uint32_t* s) {
    *s ^= *s << 13;
    *s ^= *s >> 17;
    *s ^= *s << 5;
    return *s;
}
static inline float rng_float(octx* c) { return rng_uint(c->rng) * 2.3283064365387e-10f; }

uint32_t oracle_wang_hash(uint32_t s) {
    s = (s ^ 61) ^ (s >> 16);
    s *= 9;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2d;
    s = s ^ (s >> 15);
    return s;
}
uint32_t oracle_sample_seed(uint32_t pixel, uint32_t wh, uint32_t sample, uint32_t seed_base) {
    uint32_t idx = pixel + wh * sample;
    uint32_t s = oracle_wang_hash((idx + 1u) * 17u + seed_base);
    if (s == 0) s = 0x12345678u; /* xorshift32 must not start at 0; template.cpp:667's default seed */
    return s;
}
void oracle_rng_floats(uint32_t seed, int n, float* out, uint32_t* out_u) {
    for (int i = 0; i < n; i++) {
        uint32_t u = rng_uint(&seed);
        if (out_u) out_u[i] = u;
        if (out) out[i] = u * 2.3283064365387e-10f;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Trigonometric functions are the one implementation-defined ingredient of the reference (it calls the platform
 * libm: MSVC's CRT in the shipped binary, glibc in the survey's runs).  Mode 0 (default) calls glibc's
 * sinf/cosf/acosf -- the mode the reference-output pins are checked in.  Mode 1 evaluates the correctly rounded
 * fp32 value through fp64, the definition the HIP kernels use; GPU parity tests run the oracle in mode 1, and
 * tests/test_oracle_pins.py bounds the difference between the two modes.
 * ---------------------------------------------------------------------------------------------- */
static int g_trig_mode = 0;
void oracle_set_trig_mode(int mode) { g_trig_mode = mode; }
static inline float o_sinf(float x) { return g_trig_mode ? (float)sin((double)x) : sinf(x); }
static inline float o_cosf(float x) { return g_trig_mode ? (float)cos((double)x) : cosf(x); }
static inline float o_acosf(float x) { return g_trig_mode ? (float)acos((double)x) : acosf(x); }
static inline float o_atan2f(float y, float x) { return g_trig_mode ? (float)atan2((double)y, (double)x) : atan2f(y, x); }

/* ------------------------------------------------------------------------------------------------
 * template/common.h samplers
 * ---------------------------------------------------------------------------------------------- */
/* common.h:59-63 RandomFloat(min,max) */
static inline float rng_range(octx* c, float mn, float mx) { return mn + (mx - mn) * rng_float(c); }
/* common.h:65-71 RandomInUnitDisk (args evaluated left to right, clang-oracle-defined) */
static f3 random_in_unit_disk(octx* c) {
    for (;;) {
        float px = rng_range(c, -1, 1);
        float py = rng_range(c, -1, 1);
        f3 p = F3(px, py, 0);
        if (f3_sqrlen(p) >= 1) continue;
        return p;
    }
}
/* common.h:73-82 RandomInSphere(Radius=1) */
static f3 random_in_sphere_rng(octx* c) {
    float a = 1 - 2 * rng_float(c);
    float b = sqrtf(1 - a * a);
    float phi = 2 * O_PI * rng_float(c);
    return F3(1.f * b * o_cosf(phi), 1.f * b * o_sinf(phi), 1.f * a);
}
/* common.h:84-89 RandomInSphere(u) */
static f3 random_in_sphere_u(f2 u) {
    float a = 1 - 2 * u.x;
    float b = sqrtf(1 - a * a);
    float phi = 2 * O_PI * u.y;
    return F3(b * o_cosf(phi), b * o_sinf(phi), a);
}
/* common.h:91-97 RandomInHemisphere */
static f3 random_in_hemisphere(octx* c, f3 N) {
    f3 s = random_in_sphere_rng(c);
    if (f3_dot(s, N) < 0) return f3_neg(s);
    return s;
}
/* common.h:118-136 ConcentricSampleDisk */
static f2 concentric_sample_disk(f2 u) {
    f2 uo;
    uo.x = 2.f * u.x - 1;
    uo.y = 2.f * u.y - 1;
    f2 r0 = {0, 0};
    if (uo.x == 0 && uo.y == 0) return r0;
    float theta, r;
    if (fabsf(uo.x) > fabsf(uo.y)) {
        r = uo.x;
        theta = (O_PI / 4) * (uo.y / uo.x);
    } else {
        r = uo.y;
        theta = (O_PI / 2) - (O_PI / 4) * (uo.x / uo.y);
    }
    f2 out;
    out.x = r * o_cosf(theta);
    out.y = r * o_sinf(theta);
    return out;
}
/* common.h:139-143 CosineSampleHemisphere */
static f3 cosine_sample_hemisphere(f2 u) {
    f2 d = concentric_sample_disk(u);
    float z = sqrtf(s_maxf(0.f, 1 - d.x * d.x - d.y * d.y));
    return F3(d.x, d.y, z);
}
/* common.h:145-151 CoordinateSystem */
static void coordinate_system(f3 v1, f3* v2, f3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y))
        *v2 = f3_divs(F3(-v1.z, 0, v1.x), sqrtf(v1.x * v1.x + v1.z * v1.z));
    else
        *v2 = f3_divs(F3(0, v1.z, -v1.y), sqrtf(v1.y * v1.y + v1.z * v1.z));
    *v3 = f3_cross(v1, *v2);
}
/* common.h:153-156 SphericalDirection(sinTheta, cosTheta, phi, x, y, z) */
static f3 spherical_direction(float sinTheta, float cosTheta, float phi, f3 x, f3 y, f3 z) {
    f3 a = f3_muls(x, sinTheta * o_cosf(phi));
    f3 b = f3_muls(y, sinTheta * o_sinf(phi));
    f3 cc = f3_muls(z, cosTheta);
    return f3_add(f3_add(a, b), cc);
}
/* common.h:167-169 UniformConePdf */
static float uniform_cone_pdf(float cosThetaMax) { return 1 / (2 * O_PI * (1 - cosThetaMax)); }

/* common.h:29-39 */
void oracle_rgb2lin(const float in[3], float out[3]) {
    out[0] = powf(in[0], 2.2f);
    out[1] = powf(in[1], 2.2f);
    out[2] = powf(in[2], 2.2f);
}
void oracle_hex2lin(int hex, float out[3]) {
    float rgb[3];
    rgb[0] = ((hex >> 16) & 0xFF) / 255.f;
    rgb[1] = ((hex >> 8) & 0xFF) / 255.f;
    rgb[2] = (hex & 0xFF) / 255.f;
    oracle_rgb2lin(rgb, out);
}

/* Accumulator::CopyToSurface (myapp.h:34-41): for every pixel lin2rgb(pixels[i] / (float)samples) -> rgb2uint
 * (template/common.h:41-51).  lin2rgb: pow(c, e) with float e = 1 / 2.2f (float overload = powf); rgb2uint:
 * clamp(clr.x, 0.0, 0.999) binds to the float overload (precomp.h:678) = fmaxf(a, fminf(f, b)) with the template's
 * comparators, so a NaN channel (negative or NaN sum) quantises to 255; then static_cast<int>(256 * c) and
 * (r << 16) + (g << 8) + b.  accum: n_pixels float4 (rgb + unused), the layout oracle_render writes. */
void oracle_copy_to_surface(const float* accum, int n_pixels, int samples, uint32_t* out) {
    float e = 1 / 2.2f;
    for (int i = 0; i < n_pixels; i++) {
        float cx = accum[4 * i] / (float)samples, cy = accum[4 * i + 1] / (float)samples, cz = accum[4 * i + 2] / (float)samples;
        float rx = powf(cx, e), ry = powf(cy, e), rz = powf(cz, e);
        int r = (int)(256 * t_clampf(rx, 0.0f, 0.999f));
        int g = (int)(256 * t_clampf(ry, 0.0f, 0.999f));
        int b = (int)(256 * t_clampf(rz, 0.0f, 0.999f));
        out[i] = (uint32_t)((r << 16) + (g << 8) + b);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Types: Ray (camera.h:3-15), materials (material.h), SurfaceInteraction (intersectable.h:63-115),
 * BSDF (reflection.h:83-201)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    f3 O, D;
    float t; /* `mutable float t` */
} Ray;

static Ray make_ray(f3 o, f3 d, float t) { /* camera.h:6 -- D(normalize(d)) */
    Ray r;
    r.O = o;
    r.D = f3_normalize(d);
    r.t = t;
    return r;
}
static inline f3 ray_at(const Ray* r, float t) { return f3_add(r->O, f3_muls(r->D, t)); }

enum { BX_DIFFUSE = 0, BX_RETRO = 1, BX_MICROFACET = 2, BX_SPECULAR = 3 };

typedef struct {
    int type;
    /* DisneyMaterial (material.h:14-58) */
    int has_diffuse, has_retro;
    f3 diffuse_R;   /* diffuseWeight * c */
    float roughness;
    float alphax, alphay;
    f3 R0;          /* Cspec0 */
    float metallic, eta;
    /* MirrorMaterial */
    f3 mirror_R;
} Material;

typedef struct {
    float eta;
    f3 ng, ns, ss, ts;
    int nBxDFs;
    int bxdfs[3];
    const Material* m;
} BSDF;

typedef struct {
    f3 p, wo, n;
    f2 uv;
    f3 dpdu, dpdv;
    int shape; /* scene primitive index, -1 = none */
    int tri;
    float b1, b2;
    struct { f3 n, dpdu, dpdv; } shading;
    BSDF bsdf;
} SurfaceInteraction;

/* intersectable.h:66-78 */
static void si_init(SurfaceInteraction* si, f3 p, f2 uv, f3 wo, f3 dpdu, f3 dpdv, int shape) {
    si->p = p;
    si->wo = wo;
    si->n = f3_normalize(f3_cross(dpdu, dpdv));
    si->uv = uv;
    si->dpdu = dpdu;
    si->dpdv = dpdv;
    si->shape = shape;
    si->shading.n = si->n;
    si->shading.dpdu = dpdu;
    si->shading.dpdv = dpdv;
    si->tri = -1;
    si->b1 = si->b2 = 0;
}
/* intersectable.h:80-89 (quirk: overwrites the geometric dpdu/dpdv, leaves shading.dpdu/dpdv) */
static void si_set_shading_geometry(SurfaceInteraction* si, f3 dpdus, f3 dpdvs, int orientationIsAuthorative) {
    si->shading.n = f3_normalize(f3_cross(dpdus, dpdvs));
    if (orientationIsAuthorative)
        si->n = f3_faceforward(si->n, si->shading.n);
    else
        si->shading.n = f3_faceforward(si->shading.n, si->n);
    si->dpdu = dpdus;
    si->dpdv = dpdvs;
}

/* ------------------------------------------------------------------------------------------------
 * Scene containers
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int v, n, t; } index_type; /* trianglemesh.h:5-12 */
typedef struct {
    int index;
    float bmin[3], bmax[3];
    f3 centroid;
} Primitive; /* bvhtrimesh.h:132-145 */

typedef struct BuildNode {
    float bmin[3], bmax[3];
    struct BuildNode *left, *right;
    int firstPrimOffset, nPrimitives;
} BuildNode;

typedef struct {
    f3* vertices; int n_vertices;
    f3* normals; int n_normals;
    f2* texcoords; int n_texcoords;
    index_type* indices; int n_indices;
    Primitive* primitives; int n_primitives;
    oracle_bvh_node* nodes; int totalNodes;
} Mesh;

typedef struct { f3 Center; float r, r2; } Sphere;
typedef struct { f3 O; f2 HalfSize; } Plane; /* intersectable.h:119-157: XZ plane, normal +Y */

enum { PRIM_MESH = 0, PRIM_SPHERE = 1, PRIM_PLANE = 2 };
typedef struct {
    int type;
    int index;     /* into meshes / spheres */
    int material;  /* -1 = nullptr */
    int arealight; /* index into lights, -1 = none */
} ScenePrim;

enum { LIGHT_AREA = 0, LIGHT_UNIFORM_INFINITE = 1, LIGHT_INFINITE_AREA = 2 };
/* HDRTexture (texture.h:41-84) + Distribution1D (sampling.h:19-69) of an InfiniteAreaLight (lights.cpp:31-48) */
typedef struct {
    int width, height;
    f3* pixels;
    float* func; /* [n] */
    float* cdf;  /* [n+1] */
    float funcInt;
    int n;
} EnvMap;
typedef struct {
    int type;
    int shape; /* scene primitive index (area lights) */
    f3 Lemit;
    EnvMap* env; /* LIGHT_INFINITE_AREA */
} Light;

typedef struct {
    f3 origin, u, v, w, lower_left_corner, horizontal, vertical;
    float lens_radius, focus_dist, viewport_width, viewport_height;
} Camera;

struct oracle_scene {
    Material* materials; int n_materials;
    Mesh* meshes; int n_meshes;
    Sphere* spheres; int n_spheres;
    Plane* planes; int n_planes;
    ScenePrim* prims; int n_prims;
    Light* lights; int n_lights;
    Camera cam;
    int MaxDepth;
};

oracle_scene* oracle_scene_new(void) {
    oracle_scene* s = (oracle_scene*)calloc(1, sizeof(oracle_scene));
    s->MaxDepth = 5; /* integrator.h:122 */
    return s;
}
void oracle_scene_free(oracle_scene* s) {
    if (!s) return;
    for (int i = 0; i < s->n_meshes; i++) {
        Mesh* m = &s->meshes[i];
        free(m->vertices); free(m->normals); free(m->texcoords); free(m->indices);
        free(m->primitives); free(m->nodes);
    }
    for (int i = 0; i < s->n_lights; i++)
        if (s->lights[i].env) { free(s->lights[i].env->pixels); free(s->lights[i].env->func); free(s->lights[i].env->cdf); free(s->lights[i].env); }
    free(s->materials); free(s->meshes); free(s->spheres); free(s->planes); free(s->prims); free(s->lights);
    free(s);
}
void oracle_set_max_depth(oracle_scene* s, int d) { s->MaxDepth = d; }

#define PUSH(arr, n, type) (arr = (type*)realloc(arr, sizeof(type) * (size_t)((n) + 1)), &arr[(n)++])

/* disney.h:23 */
static float SchlickR0FromEta(float eta) { return ((eta - 1) * (eta - 1)) / ((eta + 1) * (eta + 1)); }

/* material.h:14-49 DisneyMaterial ctor; material.h:72-77 MirrorMaterial */
int oracle_add_material(oracle_scene* s, int type, const float color[3], float roughness, float metallic) {
    Material* m = PUSH(s->materials, s->n_materials, Material);
    memset(m, 0, sizeof(*m));
    m->type = type;
    f3 c = F3(color[0], color[1], color[2]);
    if (type == ORACLE_MAT_DISNEY) {
        float metallicWeight = metallic;
        float e = 1.5f; /* material.h:65 `float eta = 1.5f` */
        float strans = 0.f;
        float diffuseWeight = (1 - metallicWeight) * (1 - strans);
        float rough = roughness;
        f3 Ctint = F3s(1.f);
        if (diffuseWeight > 0) {
            m->has_diffuse = 1;
            m->has_retro = 1;
            m->diffuse_R = f3_muls(c, diffuseWeight);
        }
        m->roughness = rough;
        float aspect = 1.f;
        float ax = s_maxf(.001f, (rough * rough) / aspect);
        float ay = s_maxf(.001f, (rough * rough) * aspect);
        /* TrowbridgeReitzDistribution ctor (microfacet.h:120-122) clamps again */
        m->alphax = s_maxf(0.001f, ax);
        m->alphay = s_maxf(0.001f, ay);
        float specTint = 0.f;
        f3 inner = f3_muls(f3_lerp(specTint, F3s(1.f), Ctint), SchlickR0FromEta(e));
        m->R0 = f3_lerp(metallicWeight, inner, c);
        m->metallic = metallicWeight;
        m->eta = e;
    } else if (type == ORACLE_MAT_MIRROR) {
        m->mirror_R = c;
    } else { /* single DisneyDiffuse lobe ("Lambertian" of BASELINE config 2; SURVEY naming corrections) */
        m->has_diffuse = 1;
        m->diffuse_R = c;
    }
    return s->n_materials - 1;
}

/* ------------------------------------------------------------------------------------------------
 * BVH build -- bvhtrimesh.h:154-178 (ctor), :213-310 (BuildRecursive), :312-330 (FlattenBVHTree)
 * ---------------------------------------------------------------------------------------------- */
static void bounds_init(float bmin[3], float bmax[3]) { /* bvhtrimesh.h:8-16 */
    for (int a = 0; a < 3; a++) { bmin[a] = 1e34f; bmax[a] = -1e34f; }
}
static void bounds_grow_b(float bmin[3], float bmax[3], const float omin[3], const float omax[3]) { /* :38-46 */
    for (int a = 0; a < 3; a++) { bmin[a] = t_fminf(bmin[a], omin[a]); bmax[a] = t_fmaxf(bmax[a], omax[a]); }
}
static void bounds_grow_p(float bmin[3], float bmax[3], f3 p) { /* :48-56 */
    bmin[0] = t_fminf(bmin[0], p.x); bmin[1] = t_fminf(bmin[1], p.y); bmin[2] = t_fminf(bmin[2], p.z);
    bmax[0] = t_fmaxf(bmax[0], p.x); bmax[1] = t_fmaxf(bmax[1], p.y); bmax[2] = t_fmaxf(bmax[2], p.z);
}
static float bounds_surface_area(const float bmin[3], const float bmax[3]) { /* :84-87 */
    float dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
static int bounds_longest_axis(const float bmin[3], const float bmax[3]) { /* :60-66 */
    int a = 0;
    if ((bmax[1] - bmin[1]) > (bmax[0] - bmin[0])) a = 1;
    if ((bmax[2] - bmin[2]) > (bmax[a] - bmin[a])) a = 2;
    return a;
}
static float bounds_offset_axis(const float bmin[3], const float bmax[3], f3 p, int axis) { /* :76-82 */
    float o = f3_get(p, axis) - bmin[axis];
    if (bmax[axis] > bmin[axis]) o /= bmax[axis] - bmin[axis];
    return o;
}

#define NBUCKETS 12
static int sah_bucket(const float cbmin[3], const float cbmax[3], f3 centroid, int axis) {
    int b = (int)(NBUCKETS * bounds_offset_axis(cbmin, cbmax, centroid, axis));
    if (b == NBUCKETS) b = NBUCKETS - 1;
    return b;
}

static BuildNode* build_recursive(Mesh* m, int start, int end, int maxPrimsInNode, int* totalNodes) {
    BuildNode* node = (BuildNode*)calloc(1, sizeof(BuildNode));
    (*totalNodes)++;
    Primitive* P = m->primitives;

    float bmin[3], bmax[3];
    bounds_init(bmin, bmax);
    for (int i = start; i < end; i++) bounds_grow_b(bmin, bmax, P[i].bmin, P[i].bmax);

    int nPrimitives = end - start;
    if (nPrimitives == 1) {
        node->firstPrimOffset = start; node->nPrimitives = nPrimitives;
        memcpy(node->bmin, bmin, 12); memcpy(node->bmax, bmax, 12);
        return node;
    }
    float cbmin[3], cbmax[3];
    bounds_init(cbmin, cbmax);
    for (int i = start; i < end; i++) bounds_grow_p(cbmin, cbmax, P[i].centroid);
    int axis = bounds_longest_axis(cbmin, cbmax);

    if (cbmin[axis] == cbmax[axis]) { /* :235-238 */
        node->firstPrimOffset = start; node->nPrimitives = nPrimitives;
        memcpy(node->bmin, bmin, 12); memcpy(node->bmax, bmax, 12);
        return node;
    }

    int mid = (start + end) / 2;
    if (nPrimitives <= 2) {
        /* std::nth_element on 2 elements (libstdc++: insertion sort) -- ascending centroid[axis] */
        if (f3_get(P[start + 1].centroid, axis) < f3_get(P[start].centroid, axis)) {
            Primitive tmp = P[start]; P[start] = P[start + 1]; P[start + 1] = tmp;
        }
    } else {
        int count[NBUCKETS];
        float bbmin[NBUCKETS][3], bbmax[NBUCKETS][3];
        for (int i = 0; i < NBUCKETS; i++) { count[i] = 0; bounds_init(bbmin[i], bbmax[i]); }
        for (int i = start; i < end; i++) {
            int b = sah_bucket(cbmin, cbmax, P[i].centroid, axis);
            count[b]++;
            bounds_grow_b(bbmin[b], bbmax[b], P[i].bmin, P[i].bmax);
        }
        float cost[NBUCKETS - 1];
        for (int i = 0; i < NBUCKETS - 1; i++) {
            float b0min[3], b0max[3], b1min[3], b1max[3];
            bounds_init(b0min, b0max); bounds_init(b1min, b1max);
            int count0 = 0, count1 = 0;
            for (int j = 0; j <= i; j++) { bounds_grow_b(b0min, b0max, bbmin[j], bbmax[j]); count0 += count[j]; }
            for (int j = i + 1; j < NBUCKETS; j++) { bounds_grow_b(b1min, b1max, bbmin[j], bbmax[j]); count1 += count[j]; }
            cost[i] = 1 + (count0 * bounds_surface_area(b0min, b0max) + count1 * bounds_surface_area(b1min, b1max)) /
                              bounds_surface_area(bmin, bmax);
        }
        float minCost = cost[0];
        int minCostSplitBucket = 0;
        for (int i = 1; i < NBUCKETS - 1; i++)
            if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }

        float leafCost = (float)nPrimitives;
        if (nPrimitives > maxPrimsInNode || minCost < leafCost) {
            /* std::partition, libstdc++ bidirectional-iterator algorithm */
            int first = start, last = end;
            for (;;) {
                for (;;) {
                    if (first == last) goto part_done;
                    else if (sah_bucket(cbmin, cbmax, P[first].centroid, axis) <= minCostSplitBucket) ++first;
                    else break;
                }
                --last;
                for (;;) {
                    if (first == last) goto part_done;
                    else if (!(sah_bucket(cbmin, cbmax, P[last].centroid, axis) <= minCostSplitBucket)) --last;
                    else break;
                }
                { Primitive tmp = P[first]; P[first] = P[last]; P[last] = tmp; }
                ++first;
            }
        part_done:
            mid = first;
        } else {
            node->firstPrimOffset = start; node->nPrimitives = nPrimitives;
            memcpy(node->bmin, bmin, 12); memcpy(node->bmax, bmax, 12);
            return node;
        }
    }
    node->left = build_recursive(m, start, mid, maxPrimsInNode, totalNodes);
    node->right = build_recursive(m, mid, end, maxPrimsInNode, totalNodes);
    /* InitInterior: Bounds::Union (bvhtrimesh.h:89-99,113-118) */
    for (int a = 0; a < 3; a++) {
        node->bmin[a] = t_fminf(node->left->bmin[a], node->right->bmin[a]);
        node->bmax[a] = t_fmaxf(node->left->bmax[a], node->right->bmax[a]);
    }
    node->nPrimitives = 0;
    return node;
}

static void flatten(Mesh* m, const BuildNode* node, int offset, int* firstChildOffset) { /* :312-330 */
    oracle_bvh_node* ln = &m->nodes[offset];
    memcpy(ln->bmin, node->bmin, 12);
    memcpy(ln->bmax, node->bmax, 12);
    if (node->nPrimitives > 0) {
        ln->first = node->firstPrimOffset;
        ln->count = node->nPrimitives;
    } else {
        ln->count = 0;
        ln->first = *firstChildOffset;
        *firstChildOffset += 2;
        flatten(m, node->left, ln->first, firstChildOffset);
        flatten(m, node->right, ln->first + 1, firstChildOffset);
    }
}
static void free_build(BuildNode* n) {
    if (!n) return;
    free_build(n->left); free_build(n->right); free(n);
}

int oracle_add_mesh(oracle_scene* s, const float* verts, int n_verts, const float* normals, int n_normals,
                    const float* uvs, int n_uvs, const int32_t* indices, int n_indices, int material,
                    int max_prims_in_node) {
    Mesh* m = PUSH(s->meshes, s->n_meshes, Mesh);
    memset(m, 0, sizeof(*m));
    m->n_vertices = n_verts; m->vertices = (f3*)malloc(sizeof(f3) * (size_t)(n_verts > 0 ? n_verts : 1));
    for (int i = 0; i < n_verts; i++) m->vertices[i] = F3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    m->n_normals = n_normals; m->normals = (f3*)malloc(sizeof(f3) * (size_t)(n_normals > 0 ? n_normals : 1));
    for (int i = 0; i < n_normals; i++) m->normals[i] = F3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    m->n_texcoords = n_uvs; m->texcoords = (f2*)malloc(sizeof(f2) * (size_t)(n_uvs > 0 ? n_uvs : 1));
    for (int i = 0; i < n_uvs; i++) { m->texcoords[i].x = uvs[2 * i]; m->texcoords[i].y = uvs[2 * i + 1]; }
    m->n_indices = n_indices; m->indices = (index_type*)malloc(sizeof(index_type) * (size_t)(n_indices > 0 ? n_indices : 1));
    for (int i = 0; i < n_indices; i++) {
        m->indices[i].v = indices[3 * i]; m->indices[i].n = indices[3 * i + 1]; m->indices[i].t = indices[3 * i + 2];
    }
    /* bvhtrimesh.h:156-161 */
    m->n_primitives = n_indices / 3;
    m->primitives = (Primitive*)malloc(sizeof(Primitive) * (size_t)(m->n_primitives > 0 ? m->n_primitives : 1));
    for (int i = 0, k = 0; i + 2 < n_indices; i += 3, k++) {
        Primitive* p = &m->primitives[k];
        p->index = i;
        bounds_init(p->bmin, p->bmax);
        bounds_grow_p(p->bmin, p->bmax, m->vertices[m->indices[i].v]);
        bounds_grow_p(p->bmin, p->bmax, m->vertices[m->indices[i + 1].v]);
        bounds_grow_p(p->bmin, p->bmax, m->vertices[m->indices[i + 2].v]);
        p->centroid = F3((p->bmin[0] + p->bmax[0]) * 0.5f, (p->bmin[1] + p->bmax[1]) * 0.5f,
                         (p->bmin[2] + p->bmax[2]) * 0.5f);
    }
    int totalNodes = 0;
    BuildNode* root = build_recursive(m, 0, m->n_primitives, max_prims_in_node, &totalNodes);
    m->totalNodes = totalNodes;
    m->nodes = (oracle_bvh_node*)calloc((size_t)totalNodes + 1, sizeof(oracle_bvh_node));
    int offset = 2;
    flatten(m, root, 0, &offset);
    free_build(root);

    ScenePrim* sp = PUSH(s->prims, s->n_prims, ScenePrim);
    sp->type = PRIM_MESH; sp->index = s->n_meshes - 1; sp->material = material; sp->arealight = -1;
    return s->n_prims - 1;
}

int oracle_add_sphere(oracle_scene* s, const float center[3], float radius, int material) {
    Sphere* sp = PUSH(s->spheres, s->n_spheres, Sphere);
    sp->Center = F3(center[0], center[1], center[2]);
    sp->r = radius; sp->r2 = radius * radius; /* intersectable.h:161-162 */
    ScenePrim* p = PUSH(s->prims, s->n_prims, ScenePrim);
    p->type = PRIM_SPHERE; p->index = s->n_spheres - 1; p->material = material; p->arealight = -1;
    return s->n_prims - 1;
}
/* Plane(o, size, material) pushed to Scene::primitives (intersectable.h:121) */
int oracle_add_plane(oracle_scene* s, const float o[3], const float size[2], int material) {
    Plane* pl = PUSH(s->planes, s->n_planes, Plane);
    pl->O = F3(o[0], o[1], o[2]);
    pl->HalfSize.x = size[0] / 2; pl->HalfSize.y = size[1] / 2;
    ScenePrim* p = PUSH(s->prims, s->n_prims, ScenePrim);
    p->type = PRIM_PLANE; p->index = s->n_planes - 1; p->material = material; p->arealight = -1;
    return s->n_prims - 1;
}
int oracle_add_area_light(oracle_scene* s, const float center[3], float radius, const float L[3]) { /* scene.h:21-25 */
    int prim = oracle_add_sphere(s, center, radius, -1);
    Light* l = PUSH(s->lights, s->n_lights, Light);
    l->type = LIGHT_AREA; l->shape = prim; l->Lemit = F3(L[0], L[1], L[2]); l->env = NULL;
    s->prims[prim].arealight = s->n_lights - 1;
    return prim;
}
int oracle_add_uniform_infinite_light(oracle_scene* s, const float L[3]) {
    Light* l = PUSH(s->lights, s->n_lights, Light);
    l->type = LIGHT_UNIFORM_INFINITE; l->shape = -1; l->Lemit = F3(L[0], L[1], L[2]); l->env = NULL;
    return s->n_lights - 1;
}

/* Distribution1D ctor (sampling.h:21-33): cdf[n + 1] and funcInt of the step function func[n] */
static void dist1d_build(const float* func, int n, float* cdf, float* funcInt) {
    cdf[0] = 0;
    for (int i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) {
        for (int i = 1; i < n + 1; i++) cdf[i] = (float)i / (float)n;
    } else {
        for (int i = 1; i < n + 1; i++) cdf[i] /= *funcInt;
    }
}
/* known-answer entry points for the Distribution1D restatement (pinned against the reference's own sampling.h compiled in place:
 * oracle/ref_sampling.cpp, tests/golden/dist1d.npz) */
void oracle_dist1d_build(const float* func, int n, float* cdf_out, float* funcInt_out) { dist1d_build(func, n, cdf_out, funcInt_out); }

/* InfiniteAreaLight ctor (lights.cpp:31-48, #define ILS on) from an in-memory RGB float image (the reference loads it
 * with stbi_loadf; no .hdr asset ships) + Distribution1D ctor (sampling.h:21-33) */
int oracle_add_infinite_area_light(oracle_scene* s, const float* rgb, int width, int height) {
    EnvMap* e = (EnvMap*)calloc(1, sizeof(EnvMap));
    e->width = width; e->height = height; e->n = width * height;
    e->pixels = (f3*)malloc(sizeof(f3) * (size_t)e->n);
    for (int i = 0; i < e->n; i++) e->pixels[i] = F3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    e->func = (float*)malloc(sizeof(float) * (size_t)e->n);
    for (int idx = 0; idx < e->n; idx++) {
        int y = idx / width;
        float th = (y + .5f) * O_PI / height;
        f3 value = e->pixels[idx];
        float maxComponent = s_maxf(value.x, s_maxf(value.y, value.z));
        e->func[idx] = maxComponent * sinf(th); /* host-side: libm in both trig modes */
    }
    int n = e->n;
    e->cdf = (float*)malloc(sizeof(float) * ((size_t)n + 1));
    dist1d_build(e->func, n, e->cdf, &e->funcInt);
    Light* l = PUSH(s->lights, s->n_lights, Light);
    l->type = LIGHT_INFINITE_AREA; l->shape = -1; l->Lemit = F3s(0.f); l->env = e;
    return s->n_lights - 1;
}

int oracle_mesh_num_nodes(const oracle_scene* s, int prim) { return s->meshes[s->prims[prim].index].totalNodes; }
int oracle_mesh_num_prims(const oracle_scene* s, int prim) { return s->meshes[s->prims[prim].index].n_primitives; }
void oracle_mesh_get_bvh(const oracle_scene* s, int prim, oracle_bvh_node* nodes_out, int32_t* prim_index_out) {
    const Mesh* m = &s->meshes[s->prims[prim].index];
    memcpy(nodes_out, m->nodes, sizeof(oracle_bvh_node) * ((size_t)m->totalNodes + 1));
    for (int i = 0; i < m->n_primitives; i++) prim_index_out[i] = m->primitives[i].index;
}

/* ------------------------------------------------------------------------------------------------
 * Camera -- camera.h:38-56 (ctor), :77-90 (updateCoords), :58-64 (GetRay)
 * ---------------------------------------------------------------------------------------------- */
void oracle_set_camera(oracle_scene* s, const float lookfrom_[3], const float lookat_[3], const float vup_[3],
                       float aspect_ratio, float vfov, float aperture) {
    Camera* c = &s->cam;
    f3 lookfrom = F3(lookfrom_[0], lookfrom_[1], lookfrom_[2]);
    f3 lookat = F3(lookat_[0], lookat_[1], lookat_[2]);
    f3 vup = F3(vup_[0], vup_[1], vup_[2]);
    float theta = vfov * O_PI / 180.0f; /* common.h:21-23 radians() */
    float h = tanf(theta / 2);
    c->viewport_height = 2 * h;
    c->viewport_width = aspect_ratio * c->viewport_height;
    c->lens_radius = aperture / 2;
    c->focus_dist = f3_len(f3_sub(lookat, lookfrom));
    c->w = f3_normalize(f3_sub(lookfrom, lookat));
    c->u = f3_normalize(f3_cross(vup, c->w));
    c->v = f3_cross(c->w, c->u);
    c->origin = lookfrom;
    c->horizontal = f3_muls(c->u, c->focus_dist * c->viewport_width);
    c->vertical = f3_muls(c->v, c->focus_dist * c->viewport_height);
    c->lower_left_corner = f3_sub(f3_sub(f3_sub(c->origin, f3_divs(c->horizontal, 2)), f3_divs(c->vertical, 2)),
                                  f3_muls(c->w, c->focus_dist));
}
static Ray camera_get_ray(const Camera* c, octx* ctx, float s, float t) {
    f3 rd = F3s(0.f);
    if (c->lens_radius > 0.f) rd = f3_muls(random_in_unit_disk(ctx), c->lens_radius);
    f3 offset = f3_add(f3_muls(c->u, rd.x), f3_muls(c->v, rd.y));
    f3 pixel = f3_add(f3_add(c->lower_left_corner, f3_muls(c->horizontal, s)), f3_muls(c->vertical, t));
    return make_ray(f3_add(c->origin, offset), f3_sub(f3_sub(pixel, c->origin), offset), FLT_MAX);
}

/* ------------------------------------------------------------------------------------------------
 * Bounds::Intersect -- bvhtrimesh.h:18-36
 * ---------------------------------------------------------------------------------------------- */
static int bounds_intersect(const float bmin3[3], const float bmax3[3], const Ray* ray, float* t, octx* c) {
    if (c) c->st.box_tests++;
    float tmin = 0.0f;
    float tmax = ray->t;
    for (int a = 0; a < 3; a++) {
        float Oa = f3_get(ray->O, a), Da = f3_get(ray->D, a);
        float t0 = t_fminf((bmin3[a] - Oa) / Da, (bmax3[a] - Oa) / Da);
        float t1 = t_fmaxf((bmin3[a] - Oa) / Da, (bmax3[a] - Oa) / Da);
        tmin = t_fmaxf(t0, tmin);
        tmax = t_fminf(t1, tmax);
        if ((tmax * 1.00000024f) < tmin) return 0;
    }
    *t = tmin;
    return 1;
}
int oracle_bounds_intersect(const float bmin[3], const float bmax[3], const oracle_ray* r, float* tmin_out) {
    Ray ray = make_ray(F3(r->o[0], r->o[1], r->o[2]), F3(r->d[0], r->d[1], r->d[2]), r->tmax);
    float t = 0;
    int h = bounds_intersect(bmin, bmax, &ray, &t, NULL);
    if (tmin_out) *tmin_out = t;
    return h;
}

/* ------------------------------------------------------------------------------------------------
 * TriangleMesh::TriangleIntersect -- trianglemesh.cpp:7-114; TriangleIntersectP :117-155
 * ---------------------------------------------------------------------------------------------- */
static int triangle_intersect(const Mesh* m, int shape, Ray* ray, int tridx, SurfaceInteraction* hit, octx* c) {
    c->st.tri_tests++;
    f3 v0 = m->vertices[m->indices[tridx + 0].v];
    f3 v1 = m->vertices[m->indices[tridx + 1].v];
    f3 v2 = m->vertices[m->indices[tridx + 2].v];
    f3 e1 = f3_sub(v1, v0);
    f3 e2 = f3_sub(v2, v0);
    f3 pvec = f3_cross(ray->D, e2);
    float det = f3_dot(e1, pvec);
    if (det == 0.0f) return 0;
    float inv_det = 1.0f / det;
    f3 tvec = f3_sub(ray->O, v0);
    float b1 = f3_dot(tvec, pvec) * inv_det;
    if (b1 < 0.0f || b1 > 1.0f) return 0;
    f3 qvec = f3_cross(tvec, e1);
    float b2 = f3_dot(ray->D, qvec) * inv_det;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return 0;
    float b0 = 1.f - b1 - b2;
    float t = f3_dot(e2, qvec) * inv_det;
    if (t <= 0.0f || t >= ray->t) return 0;

    f2 uv[3];
    if (m->n_texcoords > 0) {
        uv[0] = m->texcoords[m->indices[tridx + 0].t];
        uv[1] = m->texcoords[m->indices[tridx + 1].t];
        uv[2] = m->texcoords[m->indices[tridx + 2].t];
    } else {
        uv[0].x = 0; uv[0].y = 0; uv[1].x = 1; uv[1].y = 0; uv[2].x = 1; uv[2].y = 1;
    }
    f2 tc;
    tc.x = uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2;
    tc.y = uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2;

    f3 dpdu = F3s(0), dpdv = F3s(0);
    f2 duv02, duv12;
    duv02.x = uv[0].x - uv[2].x; duv02.y = uv[0].y - uv[2].y;
    duv12.x = uv[1].x - uv[2].x; duv12.y = uv[1].y - uv[2].y;
    f3 dp02 = f3_sub(v0, v2), dp12 = f3_sub(v1, v2);
    float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
    int degenerateUV = (double)fabsf(determinant) < 1e-8; /* float vs double literal, :65 */
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = f3_muls(f3_sub(f3_muls(dp02, duv12.y), f3_muls(dp12, duv02.y)), invdet);
        dpdv = f3_muls(f3_add(f3_muls(dp02, -duv12.x), f3_muls(dp12, duv02.x)), invdet);
    }
    if (degenerateUV || f3_sqrlen(f3_cross(dpdu, dpdv)) == 0) {
        f3 ng = f3_cross(f3_sub(v2, v0), f3_sub(v1, v0));
        if (f3_sqrlen(ng) == 0) return 0; /* quirk 11: reject after the t test, hit/ray.t untouched */
        coordinate_system(f3_normalize(ng), &dpdu, &dpdv);
    }

    si_init(hit, ray_at(ray, t), tc, f3_neg(ray->D), dpdu, dpdv, shape);
    hit->tri = tridx; hit->b1 = b1; hit->b2 = b2;
    ray->t = t;

    if (m->n_normals > 0) {
        f3 n0 = m->normals[m->indices[tridx + 0].n];
        f3 n1 = m->normals[m->indices[tridx + 1].n];
        f3 n2 = m->normals[m->indices[tridx + 2].n];
        f3 ns = f3_add(f3_add(f3_muls(n0, b0), f3_muls(n1, b1)), f3_muls(n2, b2));
        if (f3_sqrlen(ns) > 0.f) ns = f3_normalize(ns);
        else ns = hit->n;
        f3 ss = f3_normalize(hit->dpdu);
        f3 ts = f3_cross(ss, ns);
        if (f3_sqrlen(ts) > 0.f) {
            ts = f3_normalize(ts);
            ss = f3_cross(ts, ns);
        } else
            coordinate_system(ns, &ss, &ts);
        si_set_shading_geometry(hit, ss, ts, 1);
    }
    return 1;
}

static int triangle_intersect_p(const Mesh* m, const Ray* ray, int tridx, octx* c) {
    c->st.tri_tests++;
    f3 v0 = m->vertices[m->indices[tridx + 0].v];
    f3 v1 = m->vertices[m->indices[tridx + 1].v];
    f3 v2 = m->vertices[m->indices[tridx + 2].v];
    f3 e1 = f3_sub(v1, v0);
    f3 e2 = f3_sub(v2, v0);
    f3 pvec = f3_cross(ray->D, e2);
    float det = f3_dot(e1, pvec);
    if (det == 0.0f) return 0;
    float inv_det = 1.0f / det;
    f3 tvec = f3_sub(ray->O, v0);
    float b1 = f3_dot(tvec, pvec) * inv_det;
    if (b1 < 0.0f || b1 > 1.0f) return 0;
    f3 qvec = f3_cross(tvec, e1);
    float b2 = f3_dot(ray->D, qvec) * inv_det;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return 0;
    float t = f3_dot(e2, qvec) * inv_det;
    if (t <= 0.0f || t >= ray->t) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * BVHTriMesh::RecursiveHit (bvhtrimesh.h:332-384), RecursiveHitP (:386-413), Intersect/IntersectP (:185-198)
 * ---------------------------------------------------------------------------------------------- */
static int recursive_hit(const Mesh* m, int shape, oracle_bvh_node node, Ray* ray, SurfaceInteraction* hit, octx* c) {
    int hit_anything = 0;
    if (node.count > 0) {
        for (int i = 0; i < node.count; i++) {
            int idx = m->primitives[node.first + i].index;
            if (triangle_intersect(m, shape, ray, idx, hit, c)) hit_anything = 1;
        }
        return hit_anything;
    }
    c->st.interior_visits++;
    oracle_bvh_node left = m->nodes[node.first];
    oracle_bvh_node right = m->nodes[node.first + 1];
    float leftDist, rightDist;
    int traverseLeft = bounds_intersect(left.bmin, left.bmax, ray, &leftDist, c);
    int traverseRight = bounds_intersect(right.bmin, right.bmax, ray, &rightDist, c);
    int swap;
    if (traverseLeft && traverseRight) swap = rightDist < leftDist;
    else if (traverseLeft || traverseRight) swap = !traverseLeft;
    else return 0;
    if (swap) { oracle_bvh_node tmp = left; left = right; right = tmp; }
    if (recursive_hit(m, shape, left, ray, hit, c)) hit_anything = 1;
    if (traverseLeft && traverseRight && recursive_hit(m, shape, right, ray, hit, c)) hit_anything = 1;
    return hit_anything;
}
static int recursive_hit_p(const Mesh* m, oracle_bvh_node node, const Ray* ray, octx* c) {
    if (node.count > 0) {
        for (int i = 0; i < node.count; i++) {
            int idx = m->primitives[node.first + i].index;
            if (triangle_intersect_p(m, ray, idx, c)) return 1;
        }
        return 0;
    }
    c->st.interior_visits++;
    {
        oracle_bvh_node child = m->nodes[node.first];
        float tmp;
        if (bounds_intersect(child.bmin, child.bmax, ray, &tmp, c) && recursive_hit_p(m, child, ray, c)) return 1;
    }
    {
        oracle_bvh_node child = m->nodes[node.first + 1];
        float tmp;
        if (bounds_intersect(child.bmin, child.bmax, ray, &tmp, c) && recursive_hit_p(m, child, ray, c)) return 1;
    }
    return 0;
}
static int mesh_intersect(const Mesh* m, int shape, Ray* ray, SurfaceInteraction* hit, octx* c) {
    float dist;
    if (m->n_primitives == 0) return 0;
    if (!bounds_intersect(m->nodes[0].bmin, m->nodes[0].bmax, ray, &dist, c)) return 0;
    return recursive_hit(m, shape, m->nodes[0], ray, hit, c);
}
static int mesh_intersect_p(const Mesh* m, const Ray* ray, octx* c) {
    float dist;
    if (m->n_primitives == 0) return 0;
    if (!bounds_intersect(m->nodes[0].bmin, m->nodes[0].bmax, ray, &dist, c)) return 0;
    return recursive_hit_p(m, m->nodes[0], ray, c);
}

/* ------------------------------------------------------------------------------------------------
 * Sphere -- intersectable.h:164-226 (Intersect/IntersectP), :228-317 (Area/Sample/Pdf)
 * ---------------------------------------------------------------------------------------------- */
static int sphere_intersect(const Sphere* sp, int shape, Ray* ray, SurfaceInteraction* hit) {
    f3 oc = f3_sub(ray->O, sp->Center);
    float half_b = f3_dot(oc, ray->D);
    float cc = f3_sqrlen(oc) - sp->r2;
    float discriminant = half_b * half_b - cc;
    if (discriminant < 0) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = -half_b - sqrtd;
    if (root < 0 || ray->t < root) {
        root = -half_b + sqrtd;
        if (root < 0 || ray->t < root) return 0;
    }
    f3 p = ray_at(ray, root);
    f3 pHit = f3_sub(p, sp->Center);
    if (pHit.x == 0 && pHit.y == 0) pHit.x = O_EPSILON * sp->r;
    float phi = o_atan2f(pHit.y, pHit.x);   /* (only DbgIntegrator reads u) */
    if (phi < 0) phi += O_TWOPI;
    float u = phi * O_INV2PI;
    float theta = o_acosf(t_clampf(pHit.z / sp->r, -1.f, 1.f));
    float v = theta * O_INVPI;
    float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius;
    float sinPhi = pHit.y * invZRadius;
    f3 dpdu = F3(-O_TWOPI * pHit.y, O_TWOPI * pHit.x, 0);
    f3 dpdv = f3_muls(F3(pHit.z * cosPhi, pHit.z * sinPhi, -sp->r * o_sinf(theta)), O_PI);
    f2 uv; uv.x = u; uv.y = v;
    si_init(hit, p, uv, f3_neg(ray->D), dpdv, dpdu, shape); /* swapped on purpose, :200-201 */
    ray->t = root;
    return 1;
}
static int sphere_intersect_p(const Sphere* sp, const Ray* ray) {
    f3 oc = f3_sub(ray->O, sp->Center);
    float half_b = f3_dot(oc, ray->D);
    float cc = f3_sqrlen(oc) - sp->r2;
    float discriminant = half_b * half_b - cc;
    if (discriminant < 0) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = -half_b - sqrtd;
    if (root < 0 || ray->t < root) {
        root = -half_b + sqrtd;
        if (root < 0 || ray->t < root) return 0;
    }
    return 1;
}
static float sphere_area(const Sphere* sp) { return 4.f * O_PI * sp->r2; }

typedef struct { f3 p, n; } Interaction;

static Interaction sphere_sample_u(const Sphere* sp, f2 u, float* pdf) { /* :230-237 */
    f3 pObj = f3_add(sp->Center, f3_muls(random_in_sphere_u(u), sp->r));
    Interaction it;
    it.n = f3_normalize(pObj); /* quirk 6 */
    it.p = pObj;
    *pdf = 1 / sphere_area(sp);
    return it;
}
static Interaction sphere_sample_ref(const Sphere* sp, const SurfaceInteraction* ref, f2 u, float* pdf) { /* :239-304 */
    f3 pCenter = sp->Center;
    f3 pOrigin = ref->p;
    if (f3_sqrlen(f3_sub(pOrigin, pCenter)) <= sp->r2) {
        Interaction intr = sphere_sample_u(sp, u, pdf);
        f3 wi = f3_sub(intr.p, ref->p);
        if (f3_sqrlen(wi) == 0)
            *pdf = 0;
        else {
            wi = f3_normalize(wi);
            *pdf *= f3_sqrlen(f3_sub(ref->p, intr.p)) / f3_absdot(intr.n, f3_neg(wi));
        }
        if (isinf(*pdf)) *pdf = 0;
        return intr;
    }
    float dc = f3_len(f3_sub(ref->p, pCenter));
    float invDc = 1 / dc;
    f3 wc = f3_muls(f3_sub(pCenter, ref->p), invDc);
    f3 wcX, wcY;
    coordinate_system(wc, &wcX, &wcY);
    float sinThetaMax = sp->r * invDc;
    float sinThetaMax2 = sinThetaMax * sinThetaMax;
    float invSinThetaMax = 1 / sinThetaMax;
    float cosThetaMax = sqrtf(s_maxf(0.f, 1 - sinThetaMax2));
    float cosTheta = (cosThetaMax - 1) * u.x + 1;
    float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * u.x;
        cosTheta = sqrtf(1 - sinTheta2);
    }
    float cosAlpha = sinTheta2 * invSinThetaMax +
                     cosTheta * sqrtf(s_maxf(0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    float sinAlpha = sqrtf(s_maxf(0.f, 1.f - cosAlpha * cosAlpha));
    float phi = u.y * 2 * O_PI;
    f3 nWorld = spherical_direction(sinAlpha, cosAlpha, phi, f3_neg(wcX), f3_neg(wcY), f3_neg(wc));
    f3 pWorld = f3_add(pCenter, f3_muls(F3(nWorld.x, nWorld.y, nWorld.z), sp->r));
    Interaction it;
    it.p = pWorld;
    it.n = nWorld;
    *pdf = 1 / (2 * O_PI * (1 - cosThetaMax));
    return it;
}
static float sphere_pdf(const Sphere* sp, const SurfaceInteraction* ref, f3 wi) { /* :306-317 */
    (void)wi;
    f3 pCenter = sp->Center;
    f3 pOrigin = ref->p;
    if (f3_sqrlen(f3_sub(pOrigin, pCenter)) <= sp->r2) return 1 / (4 * O_PI);
    float sinThetaMax2 = sp->r2 / f3_sqrlen(f3_sub(ref->p, pCenter));
    float cosThetaMax = sqrtf(s_maxf(0.f, 1 - sinThetaMax2));
    return uniform_cone_pdf(cosThetaMax);
}

/* Plane::Intersect / IntersectP -- intersectable.h:123-150 */
static int plane_intersect(const Plane* pl, int shape, Ray* ray, SurfaceInteraction* hit) {
    if (ray->D.y == 0) return 0;
    float t = (pl->O.y - ray->O.y) / ray->D.y;
    if (t <= 0 || t >= ray->t) return 0;
    f3 P = ray_at(ray, t);
    float u = (P.x - pl->O.x) / pl->HalfSize.x;
    float v = (P.z - pl->O.z) / pl->HalfSize.y;
    if (fabsf(u) <= 1 && fabsf(v) <= 1) { /* fabs(float v) promotes to double: same comparison result */
        f2 uv; uv.x = (u + 1) * .5f; uv.y = (v + 1) * .5f;
        si_init(hit, P, uv, f3_neg(ray->D), F3(0, 0, 1), F3(1, 0, 0), shape);
        ray->t = t;
        return 1;
    }
    return 0;
}
static int plane_intersect_p(const Plane* pl, const Ray* ray) {
    if (ray->D.y == 0) return 0;
    float t = (pl->O.y - ray->O.y) / ray->D.y;
    if (t <= 0 || t >= ray->t) return 0;
    f3 P = ray_at(ray, t);
    float u = (P.x - pl->O.x) / pl->HalfSize.x;
    float v = (P.z - pl->O.z) / pl->HalfSize.y;
    return (fabsf(u) <= 1 && fabsf(v) <= 1);
}

/* ------------------------------------------------------------------------------------------------
 * Scene::Intersect / IntersectP -- scene.h:5-19
 * ---------------------------------------------------------------------------------------------- */
static int scene_intersect(const oracle_scene* s, Ray* ray, SurfaceInteraction* hit, octx* c) {
    c->st.closest_rays++;
    int found = 0;
    for (int i = 0; i < s->n_prims; i++) {
        const ScenePrim* p = &s->prims[i];
        int h = (p->type == PRIM_MESH)    ? mesh_intersect(&s->meshes[p->index], i, ray, hit, c)
                : (p->type == PRIM_SPHERE) ? sphere_intersect(&s->spheres[p->index], i, ray, hit)
                                           : plane_intersect(&s->planes[p->index], i, ray, hit);
        if (h) found = 1;
    }
    return found;
}
static int scene_intersect_p(const oracle_scene* s, const Ray* ray, octx* c) {
    c->st.anyhit_rays++;
    for (int i = 0; i < s->n_prims; i++) {
        const ScenePrim* p = &s->prims[i];
        int h = (p->type == PRIM_MESH)    ? mesh_intersect_p(&s->meshes[p->index], ray, c)
                : (p->type == PRIM_SPHERE) ? sphere_intersect_p(&s->spheres[p->index], ray)
                                           : plane_intersect_p(&s->planes[p->index], ray);
        if (h) return 1;
    }
    return 0;
}

void oracle_intersect_batch(const oracle_scene* s, const oracle_ray* rays, int n, oracle_hit* out, int any_hit,
                            oracle_stats* stats) {
    octx c; memset(&c, 0, sizeof(c));
    for (int i = 0; i < n; i++) {
        Ray ray = make_ray(F3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), F3(rays[i].d[0], rays[i].d[1], rays[i].d[2]),
                           rays[i].tmax);
        oracle_hit* h = &out[i];
        memset(h, 0, sizeof(*h));
        h->prim = -1; h->tri = -1;
        if (any_hit) {
            h->hit = scene_intersect_p(s, &ray, &c);
        } else {
            SurfaceInteraction si;
            si.shape = -1; si.tri = -1;
            h->hit = scene_intersect(s, &ray, &si, &c);
            if (h->hit) { h->prim = si.shape; h->tri = si.tri; h->t = ray.t; h->b1 = si.b1; h->b2 = si.b2; }
        }
    }
    if (stats) *stats = c.st;
}

/* ------------------------------------------------------------------------------------------------
 * microfacet.h
 * ---------------------------------------------------------------------------------------------- */
static inline float CosTheta(f3 w) { return w.z; }
static inline float Cos2Theta(f3 w) { return w.z * w.z; }
static inline float AbsCosTheta(f3 w) { return fabsf(w.z); }
static inline float Sin2Theta(f3 w) { return s_maxf(0.f, 1.f - Cos2Theta(w)); }
static inline float SinTheta(f3 w) { return sqrtf(Sin2Theta(w)); }
static inline float TanTheta(f3 w) { return SinTheta(w) / CosTheta(w); }
static inline float Tan2Theta(f3 w) { return Sin2Theta(w) / Cos2Theta(w); }
static inline float CosPhi(f3 w) {
    float sinTheta = SinTheta(w);
    return (sinTheta == 0) ? 1 : t_clampf(w.x / sinTheta, -1.f, 1.f);
}
static inline float SinPhi(f3 w) {
    float sinTheta = SinTheta(w);
    return (sinTheta == 0) ? 0 : t_clampf(w.y / sinTheta, -1.f, 1.f);
}
static inline float Cos2Phi(f3 w) { return CosPhi(w) * CosPhi(w); }
static inline float Sin2Phi(f3 w) { return SinPhi(w) * SinPhi(w); }

/* microfacet.h:34-73 */
static void TrowbridgeReitzSample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y) {
    if (cosTheta > .9999f) {
        float r = sqrtf(U1 / (1 - U1));
        float phi = 6.28318530718f * U2;
        *slope_x = r * o_cosf(phi);
        *slope_y = r * o_sinf(phi);
        return;
    }
    float sinTheta = sqrtf(s_maxf(0.f, 1.f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if ((double)tmp > 1e10) tmp = (float)1e10;
    float B = tanTheta;
    float D = sqrtf(s_maxf((float)(B * B * tmp * tmp - (A * A - B * B) * tmp), 0.f));
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
/* microfacet.h:75-94 */
static f3 TrowbridgeReitzSample(f3 wi, float alpha_x, float alpha_y, float U1, float U2) {
    f3 wiStretched = f3_normalize(F3(alpha_x * wi.x, alpha_y * wi.y, wi.z));
    float slope_x, slope_y;
    TrowbridgeReitzSample11(CosTheta(wiStretched), U1, U2, &slope_x, &slope_y);
    float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
    slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
    slope_x = tmp;
    slope_x = alpha_x * slope_x;
    slope_y = alpha_y * slope_y;
    return f3_normalize(F3(-slope_x, -slope_y, 1.f));
}
/* microfacet.h:124-132 */
static float TR_D(const Material* m, f3 wh) {
    float tan2Theta = Tan2Theta(wh);
    if (isinf(tan2Theta)) return 0.f;
    const float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
    float e = (Cos2Phi(wh) / (m->alphax * m->alphax) + Sin2Phi(wh) / (m->alphay * m->alphay)) * tan2Theta;
    return 1 / (O_PI * m->alphax * m->alphay * cos4Theta * (1 + e) * (1 + e));
}
/* microfacet.h:142-149 */
static float TR_Lambda(const Material* m, f3 w) {
    float absTanTheta = fabsf(TanTheta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = sqrtf(Cos2Phi(w) * m->alphax * m->alphax + Sin2Phi(w) * m->alphay * m->alphay);
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
static float TR_G1(const Material* m, f3 w) { return 1 / (1 + TR_Lambda(m, w)); }          /* :100-102 */
static float Disney_G(const Material* m, f3 wo, f3 wi) { return TR_G1(m, wo) * TR_G1(m, wi); } /* disney.h:78-81 */
static float TR_Pdf(const Material* m, f3 wo, f3 wh) {                                      /* :107-109 */
    return TR_D(m, wh) * TR_G1(m, wo) * f3_absdot(wo, wh) / AbsCosTheta(wo);
}
static f3 TR_Sample_wh(const Material* m, f3 wo, f2 u) { /* :134-140 */
    int flip = wo.z < 0;
    f3 wh = TrowbridgeReitzSample(flip ? f3_neg(wo) : wo, m->alphax, m->alphay, u.x, u.y);
    if (flip) wh = f3_neg(wh);
    return wh;
}
/* microfacet.h:180-201 */
static float FrDielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = t_clampf(cosThetaI, -1.f, 1.f);
    int entering = cosThetaI > 0.f;
    if (!entering) {
        float tmp = etaI; etaI = etaT; etaT = tmp;
        cosThetaI = fabsf(cosThetaI);
    }
    float sinThetaI = sqrtf(s_maxf(0.f, 1.f - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = sqrtf(s_maxf(0.f, 1.f - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}

/* ------------------------------------------------------------------------------------------------
 * disney.h
 * ---------------------------------------------------------------------------------------------- */
static float SchlickWeight(float cosTheta) { /* :12-15 */
    float m = t_clampf(1 - cosTheta, 0.f, 1.f);
    return (m * m) * (m * m) * m;
}
static f3 FrSchlick(f3 R0, float cosTheta) { return f3_lerp(SchlickWeight(cosTheta), R0, F3s(1.f)); } /* :17-19 */
static f3 DisneyFresnel_Evaluate(const Material* m, float cosI) { /* :62-71 */
    return f3_lerp(m->metallic, F3s(FrDielectric(cosI, 1, m->eta)), FrSchlick(m->R0, cosI));
}
static f3 DisneyDiffuse_f(const Material* m, f3 wo, f3 wi) { /* :28-35 */
    float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
    return f3_muls(f3_muls(f3_muls(m->diffuse_R, O_INVPI), (1 - Fo / 2)), (1 - Fi / 2));
}
static f3 DisneyRetro_f(const Material* m, f3 wo, f3 wi) { /* :43-55 */
    f3 wh = f3_add(wi, wo);
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return F3s(0.f);
    wh = f3_normalize(wh);
    float cosThetaD = f3_dot(wi, wh);
    float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
    float Rr = 2 * m->roughness * cosThetaD * cosThetaD;
    return f3_muls(f3_muls(f3_muls(m->diffuse_R, O_INVPI), Rr), (Fo + Fi + Fo * Fi * (Rr - 1)));
}

/* ------------------------------------------------------------------------------------------------
 * reflection.h -- BxDFs
 * ---------------------------------------------------------------------------------------------- */
static f3 Microfacet_f(const Material* m, f3 wo, f3 wi) { /* :42-54 */
    float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
    f3 wh = f3_add(wi, wo);
    if (cosThetaI == 0 || cosThetaO == 0) return F3s(0.f);
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return F3s(0.f);
    wh = f3_normalize(wh);
    f3 F = DisneyFresnel_Evaluate(m, f3_dot(wi, f3_faceforward(wh, F3(0, 0, 1))));
    f3 R = F3s(1.f); /* material.h:48 */
    f3 num = f3_mul(f3_muls(f3_muls(R, TR_D(m, wh)), Disney_G(m, wo, wi)), F);
    return f3_divs(num, (4 * cosThetaI * cosThetaO));
}
static float Microfacet_Pdf(const Material* m, f3 wo, f3 wi) { /* :67-71 */
    if (!same_hemisphere(wo, wi)) return 0;
    f3 wh = f3_normalize(f3_add(wo, wi));
    return TR_Pdf(m, wo, wh) / (4 * f3_dot(wo, wh));
}
static f3 Microfacet_Sample_f(const Material* m, f3 wo, f3* wi, f2 u, float* pdf) { /* :55-66 */
    if (wo.z == 0) return F3s(0.f);
    f3 wh = TR_Sample_wh(m, wo, u);
    if (f3_dot(wo, wh) < 0) return F3s(0.f);
    *wi = f3_reflect(wo, wh);
    if (!same_hemisphere(wo, *wi)) return F3s(0.f);
    *pdf = TR_Pdf(m, wo, wh) / (4 * f3_dot(wo, wh));
    return Microfacet_f(m, wo, *wi);
}

static int bxdf_is_specular(int kind) { return kind == BX_SPECULAR; }
static f3 bxdf_f(const Material* m, int kind, f3 wo, f3 wi) {
    switch (kind) {
        case BX_DIFFUSE: return DisneyDiffuse_f(m, wo, wi);
        case BX_RETRO: return DisneyRetro_f(m, wo, wi);
        case BX_MICROFACET: return Microfacet_f(m, wo, wi);
        default: return F3s(0.f); /* SpecularReflection::f, reflection.h:26-28 */
    }
}
static float bxdf_pdf(const Material* m, int kind, f3 wo, f3 wi) {
    switch (kind) {
        case BX_DIFFUSE:
        case BX_RETRO: return same_hemisphere(wo, wi) ? AbsCosTheta(wi) * O_INVPI : 0; /* :16-18 */
        case BX_MICROFACET: return Microfacet_Pdf(m, wo, wi);
        default: return 0; /* :30 */
    }
}
static f3 bxdf_sample_f(const Material* m, int kind, f3 wo, f3* wi, f2 u, float* pdf) {
    switch (kind) {
        case BX_DIFFUSE:
        case BX_RETRO: { /* BxDF::Sample_f, :8-15 */
            *wi = cosine_sample_hemisphere(u);
            if (wo.z < 0) wi->z *= -1;
            *pdf = bxdf_pdf(m, kind, wo, *wi);
            return bxdf_f(m, kind, wo, *wi);
        }
        case BX_MICROFACET: return Microfacet_Sample_f(m, wo, wi, u, pdf);
        default: { /* SpecularReflection::Sample_f, reflection.cpp:13-18; FresnelNoOp = 1 */
            *wi = F3(-wo.x, -wo.y, wo.z);
            *pdf = 1;
            return f3_divs(f3_mul(F3s(1.f), m->mirror_R), AbsCosTheta(*wi));
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * BSDF -- reflection.cpp:6-11 (ctor), reflection.h:88-201; Material::SetupBSDF material.h:51-58,79-81
 * ---------------------------------------------------------------------------------------------- */
static void bsdf_init(BSDF* b, const SurfaceInteraction* si, const Material* m) {
    b->eta = 1.f;
    b->ng = si->n;
    b->ns = si->shading.n;
    b->ss = f3_normalize(si->shading.dpdu);
    b->ts = f3_cross(b->ns, b->ss);
    b->nBxDFs = 0;
    b->m = m;
    if (m->type == ORACLE_MAT_DISNEY) {
        if (m->has_diffuse) b->bxdfs[b->nBxDFs++] = BX_DIFFUSE;
        if (m->has_retro) b->bxdfs[b->nBxDFs++] = BX_RETRO;
        b->bxdfs[b->nBxDFs++] = BX_MICROFACET;
    } else if (m->type == ORACLE_MAT_MIRROR) {
        b->bxdfs[b->nBxDFs++] = BX_SPECULAR;
    } else {
        b->bxdfs[b->nBxDFs++] = BX_DIFFUSE;
    }
}
static int matches_flag(int kind, int skipSpecular) { return !skipSpecular || !bxdf_is_specular(kind); }
static int bsdf_num_components(const BSDF* b, int skipSpecular) {
    int num = 0;
    for (int i = 0; i < b->nBxDFs; i++) if (matches_flag(b->bxdfs[i], skipSpecular)) num++;
    return num;
}
static int bsdf_is_perfectly_specular(const BSDF* b) {
    for (int i = 0; i < b->nBxDFs; i++) if (!bxdf_is_specular(b->bxdfs[i])) return 0;
    return 1;
}
static f3 bsdf_world_to_local(const BSDF* b, f3 v) { return F3(f3_dot(v, b->ss), f3_dot(v, b->ts), f3_dot(v, b->ns)); }
static f3 bsdf_local_to_world(const BSDF* b, f3 v) {
    return F3(b->ss.x * v.x + b->ts.x * v.y + b->ns.x * v.z,
              b->ss.y * v.x + b->ts.y * v.y + b->ns.y * v.z,
              b->ss.z * v.x + b->ts.z * v.y + b->ns.z * v.z);
}
static f3 bsdf_f(const BSDF* b, f3 woW, f3 wiW, int skipSpecular) { /* :114-123 */
    f3 wi = bsdf_world_to_local(b, wiW), wo = bsdf_world_to_local(b, woW);
    if (wo.z == 0) return F3s(0.f);
    int reflect = f3_dot(wiW, b->ng) * f3_dot(woW, b->ng) > 0;
    f3 f = F3s(0.f);
    for (int i = 0; i < b->nBxDFs; i++)
        if (matches_flag(b->bxdfs[i], skipSpecular) && reflect) f = f3_add(f, bxdf_f(b->m, b->bxdfs[i], wo, wi));
    return f;
}
static f3 bsdf_sample_f(const BSDF* b, f3 woWorld, f3* wiWorld, f2 u, float* pdf, int skipSpecular,
                        int* sampledSpecular) { /* :124-172 */
    int matchingComps = bsdf_num_components(b, skipSpecular);
    if (matchingComps == 0) {
        *pdf = 0;
        return F3s(0.f);
    }
    int comp = (int)floorf(u.x * matchingComps);
    if (matchingComps - 1 < comp) comp = matchingComps - 1;
    int bxdf = -1, bxdf_slot = -1;
    int count = comp;
    for (int i = 0; i < b->nBxDFs; i++)
        if (matches_flag(b->bxdfs[i], skipSpecular) && count-- == 0) {
            bxdf = b->bxdfs[i]; bxdf_slot = i;
            break;
        }
    f2 uRemapped;
    uRemapped.x = s_minf(u.x * matchingComps - comp, O_ONE_MINUS_EPS);
    uRemapped.y = u.y;
    f3 wi = F3s(0.f), wo = bsdf_world_to_local(b, woWorld);
    if (wo.z == 0) return F3s(0.f); /* quirk 9: *pdf not written */
    *pdf = 0;
    if (sampledSpecular) *sampledSpecular = bxdf_is_specular(bxdf);
    f3 f = bxdf_sample_f(b->m, bxdf, wo, &wi, uRemapped, pdf);
    if (*pdf == 0) return F3s(0.f);
    *wiWorld = bsdf_local_to_world(b, wi);
    if (!bxdf_is_specular(bxdf) && matchingComps > 1)
        for (int i = 0; i < b->nBxDFs; i++)
            if (i != bxdf_slot && matches_flag(b->bxdfs[i], skipSpecular)) *pdf += bxdf_pdf(b->m, b->bxdfs[i], wo, wi);
    if (matchingComps > 1) *pdf /= matchingComps;
    if (!bxdf_is_specular(bxdf)) {
        int reflect = f3_dot(*wiWorld, b->ng) * f3_dot(woWorld, b->ng) > 0;
        f = F3s(0.f);
        for (int i = 0; i < b->nBxDFs; i++)
            if (matches_flag(b->bxdfs[i], skipSpecular) && reflect) f = f3_add(f, bxdf_f(b->m, b->bxdfs[i], wo, wi));
    }
    return f;
}
static float bsdf_pdf(const BSDF* b, f3 woWorld, f3 wiWorld, int skipSpecular) { /* :174-188 */
    if (b->nBxDFs == 0) return 0.f;
    f3 wo = bsdf_world_to_local(b, woWorld), wi = bsdf_world_to_local(b, wiWorld);
    if (wo.z == 0) return 0.f;
    float pdf = 0.f;
    int matchingComps = 0;
    for (int i = 0; i < b->nBxDFs; i++)
        if (matches_flag(b->bxdfs[i], skipSpecular)) {
            ++matchingComps;
            pdf += bxdf_pdf(b->m, b->bxdfs[i], wo, wi);
        }
    return matchingComps > 0 ? pdf / matchingComps : 0.f;
}

/* intersectable.h:91-96 */
static int si_eval_material(const oracle_scene* s, SurfaceInteraction* si) {
    int mat = s->prims[si->shape].material;
    if (mat < 0) return 0;
    bsdf_init(&si->bsdf, si, &s->materials[mat]);
    return 1;
}
/* Intersectable.cpp:5-8 + lights.h:82 */
static f3 si_Le(const oracle_scene* s, const SurfaceInteraction* si) {
    int al = s->prims[si->shape].arealight;
    return al >= 0 ? s->lights[al].Lemit : F3s(0.f);
}

/* ------------------------------------------------------------------------------------------------
 * Lights -- lights.cpp:10-28 (VisibilityTester, UniformInfiniteLight), :115-130 (AreaLight)
 * ---------------------------------------------------------------------------------------------- */
/* texture.h:76-79,59-67 HDRTexture::value / Mod */
static int env_mod(int a, int b) {
    int result = a - (a / b) * b;
    return (result < 0) ? result + b : result;
}
static f3 env_value(const EnvMap* e, float u, float v) {
    int si = (int)floorf(u * e->width - .5f);
    int ti = (int)floorf(v * e->height - .5f);
    return e->pixels[env_mod(ti, e->height) * e->width + env_mod(si, e->width)];
}
/* common.h:158-165 */
static float spherical_theta(f3 v) { return o_acosf(t_clampf(v.z, -1.f, 1.f)); }
static float spherical_phi(f3 v) {
    float p = o_atan2f(v.y, v.x);
    return (p < 0) ? (p + O_TWOPI) : p;
}
/* InfiniteAreaLight::Le lights.cpp:108-112 */
static f3 env_Le(const EnvMap* e, f3 D) {
    f3 w = f3_normalize(D);
    w = F3(w.x, w.z, w.y);
    return env_value(e, spherical_phi(w) * O_INV2PI, spherical_theta(w) * O_INVPI);
}
/* sampling.h:4-17 FindInterval with pred(index) = cdf[index] <= u */
static int env_find_interval(const EnvMap* e, float u) {
    int size = e->n + 1;
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (e->cdf[middle] <= u) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    int r = first - 1;
    if (size - 2 < r) r = size - 2; /* clamp(int): max(a, min(f, b)) */
    if (r < 0) r = 0;
    return r;
}
/* Distribution1D::SampleContinuous sampling.h:37-52 */
static float env_sample_continuous(const EnvMap* e, float u, float* pdf) {
    int offset = env_find_interval(e, u);
    float du = u - e->cdf[offset];
    if ((e->cdf[offset + 1] - e->cdf[offset]) > 0) du /= e->cdf[offset + 1] - e->cdf[offset];
    *pdf = (e->funcInt > 0) ? e->func[offset] / e->funcInt : 0;
    return (offset + du) / e->n;
}

/* Distribution1D::SampleContinuous for k values of u on caller-supplied arrays (known-answer entry point) */
void oracle_dist1d_sample(const float* func, const float* cdf, float funcInt, int n, const float* u, int k, float* x_out,
                          float* pdf_out, int* offset_out) {
    EnvMap e;
    memset(&e, 0, sizeof(e));
    e.func = (float*)func; e.cdf = (float*)cdf; e.funcInt = funcInt; e.n = n;
    for (int i = 0; i < k; i++) {
        offset_out[i] = env_find_interval(&e, u[i]);
        x_out[i] = env_sample_continuous(&e, u[i], &pdf_out[i]);
    }
}

static f3 light_sample_li(const oracle_scene* s, const Light* l, const SurfaceInteraction* ref, f2 u, f3* wi,
                          float* pdf, Ray* vis, octx* c) {
    if (l->type == LIGHT_INFINITE_AREA) { /* lights.cpp:50-90 (ILS): ignores u, 1 extra draw */
        const EnvMap* e = l->env;
        float mapPdf;
        float sample = env_sample_continuous(e, rng_float(c), &mapPdf);
        if (mapPdf == 0) return F3s(0.f);
        int idx = (int)(sample * e->n);
        float uvx = ((idx % e->width) + .5f) / e->width;
        float uvy = ((idx / e->width) + .5f) / e->height;
        float theta = uvy * O_PI, phi = uvx * O_TWOPI;
        float cosTheta = o_cosf(theta), sinTheta = o_sinf(theta);
        float sinPhi = o_sinf(phi), cosPhi = o_cosf(phi);
        *wi = F3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi);
        *pdf = mapPdf / (2 * O_PI * O_PI * sinTheta);
        if (sinTheta == 0) *pdf = 0;
        *vis = make_ray(f3_add(ref->p, f3_muls(ref->n, O_EPSILON)), *wi, FLT_MAX); /* quirk 4: offset along n */
        return env_Le(e, vis->D);
    }
    if (l->type == LIGHT_UNIFORM_INFINITE) {
        *wi = random_in_hemisphere(c, ref->shading.n); /* quirk 3: ignores u, 2 extra draws */
        *pdf = O_INV2PI;
        *vis = make_ray(f3_add(ref->p, f3_muls(*wi, O_EPSILON)), *wi, FLT_MAX);
        return l->Lemit;
    }
    const Sphere* sp = &s->spheres[s->prims[l->shape].index];
    Interaction pShape = sphere_sample_ref(sp, ref, u, pdf);
    if (*pdf == 0 || f3_sqrlen(f3_sub(pShape.p, ref->p)) == 0) {
        *pdf = 0;
        return F3s(0.f);
    }
    *wi = f3_sub(pShape.p, ref->p);
    float dist = f3_len(*wi);
    *wi = f3_divs(*wi, dist);
    *vis = make_ray(f3_add(ref->p, f3_muls(*wi, O_EPSILON)), *wi, dist - 10 * O_EPSILON);
    return l->Lemit;
}
static float light_pdf_li(const oracle_scene* s, const Light* l, const SurfaceInteraction* ref, f3 wi) {
    if (l->type == LIGHT_UNIFORM_INFINITE) return f3_dot(ref->n, wi) > 0 ? O_INV2PI : 0.f;
    if (l->type == LIGHT_INFINITE_AREA) { /* lights.cpp:92-106 */
        const EnvMap* e = l->env;
        f3 w = f3_normalize(wi);
        w = F3(w.x, w.z, w.y);
        float theta = spherical_theta(w), phi = spherical_phi(w);
        float sinTheta = o_sinf(theta);
        if (sinTheta == 0) return 0;
        int x = (int)(phi * O_INV2PI * e->width);
        if (e->width - 1 < x) x = e->width - 1;
        if (x < 0) x = 0;
        int y = (int)(theta * O_INVPI * e->height);
        if (e->height - 1 < y) y = e->height - 1;
        if (y < 0) y = 0;
        float discrete = e->func[y * e->width + x] / (e->funcInt * e->n); /* DiscretePDF sampling.h:63-65 */
        return e->n * discrete / (2 * O_PI * O_PI * sinTheta);
    }
    return sphere_pdf(&s->spheres[s->prims[l->shape].index], ref, wi);
}
static f3 light_Le(const Light* l, f3 D) {
    if (l->type == LIGHT_UNIFORM_INFINITE) return l->Lemit;
    if (l->type == LIGHT_INFINITE_AREA) return env_Le(l->env, D);
    return F3s(0.f);
}

/* ------------------------------------------------------------------------------------------------
 * integrator.h -- PowerHeuristic :33-36, EstimateDirect :38-93, UniformSampleOneLight :95-105,
 * PathTracer::Li :124-191
 * ---------------------------------------------------------------------------------------------- */
static float PowerHeuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
static f3 EstimateDirect(const oracle_scene* s, const SurfaceInteraction* si, f2 uScattering, int lightIdx, f2 uLight,
                         octx* c) {
    const Light* light = &s->lights[lightIdx];
    f3 Ld = F3s(0.f);
    f3 wi = F3s(0.f);
    float lightPdf = 0, scatteringPdf = 0;
    Ray visibility;
    memset(&visibility, 0, sizeof(visibility));
    f3 Li = light_sample_li(s, light, si, uLight, &wi, &lightPdf, &visibility, c);
    if (lightPdf > 0 && !f3_isblack(Li)) {
        f3 f = f3_muls(bsdf_f(&si->bsdf, si->wo, wi, 1), f3_absdot(wi, si->shading.n));
        scatteringPdf = bsdf_pdf(&si->bsdf, si->wo, wi, 1);
        if (!f3_isblack(f)) {
            if (scene_intersect_p(s, &visibility, c)) Li = F3s(0.f);
            if (!f3_isblack(Li)) {
                float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                Ld = f3_add(Ld, f3_divs(f3_muls(f3_mul(f, Li), weight), lightPdf));
            }
        }
    }
    {
        f3 f = bsdf_sample_f(&si->bsdf, si->wo, &wi, uScattering, &scatteringPdf, 1, NULL);
        f = f3_muls(f, f3_absdot(wi, si->shading.n));
        if (!f3_isblack(f) && scatteringPdf > 0) {
            float weight = 1;
            lightPdf = light_pdf_li(s, light, si, wi);
            if (lightPdf == 0) return Ld;
            weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            SurfaceInteraction lightIsect;
            lightIsect.shape = -1;
            Ray ray = make_ray(f3_add(si->p, f3_muls(wi, O_EPSILON)), wi, FLT_MAX);
            int found = scene_intersect(s, &ray, &lightIsect, c);
            f3 Li2 = F3s(0.f);
            if (found) {
                if (s->prims[lightIsect.shape].arealight == lightIdx) Li2 = si_Le(s, &lightIsect);
            } else
                Li2 = light_Le(light, ray.D);
            if (!f3_isblack(Li2)) Ld = f3_add(Ld, f3_divs(f3_muls(f3_mul(f, Li2), weight), scatteringPdf));
        }
    }
    return Ld;
}
static f3 UniformSampleOneLight(const oracle_scene* s, const SurfaceInteraction* si, octx* c) {
    int nLights = s->n_lights;
    if (nLights == 0) return F3s(0.f);
    int numLight = (int)(rng_float(c) * nLights);
    if (nLights - 1 < numLight) numLight = nLights - 1;
    float lightPdf = 1.f / nLights;
    f2 uLight, uScattering;
    uLight.x = rng_float(c); uLight.y = rng_float(c);
    uScattering.x = rng_float(c); uScattering.y = rng_float(c);
    return f3_divs(EstimateDirect(s, si, uScattering, numLight, uLight, c), lightPdf);
}

static f3 PathTracer_Li(const oracle_scene* s, const Ray* r, octx* c) {
    f3 beta = F3s(1.f);
    f3 L = F3s(0.f);
    Ray ray = *r;
    int specularBounce = 0;
    int bounces;
    for (bounces = 0;; bounces++) {
        SurfaceInteraction isect;
        isect.shape = -1;
        int foundIntersection = scene_intersect(s, &ray, &isect, c);
        if (bounces == 0 || specularBounce) {
            if (foundIntersection) {
                L = f3_add(L, f3_mul(beta, si_Le(s, &isect)));
            } else {
                for (int i = 0; i < s->n_lights; i++)
                    if (s->lights[i].type != LIGHT_AREA) L = f3_add(L, f3_mul(beta, light_Le(&s->lights[i], ray.D)));
            }
        }
        if (!foundIntersection || bounces >= s->MaxDepth) break;
        if (!si_eval_material(s, &isect)) {
            ray = make_ray(f3_add(isect.p, f3_muls(ray.D, O_EPSILON)), ray.D, FLT_MAX);
            bounces--;
            continue;
        }
        c->st.shaded_vertices++;
        if (!bsdf_is_perfectly_specular(&isect.bsdf)) L = f3_add(L, f3_mul(beta, UniformSampleOneLight(s, &isect, c)));
        f3 wo = f3_neg(ray.D), wi = F3s(0.f);
        f2 u;
        u.x = rng_float(c); u.y = rng_float(c);
        float pdf = 0;
        int sampledSpecular = 0;
        f3 f = bsdf_sample_f(&isect.bsdf, wo, &wi, u, &pdf, 0, &sampledSpecular);
        if (f3_isblack(f) || pdf == 0) break;
        beta = f3_mul(beta, f3_divs(f3_muls(f, f3_absdot(wi, isect.shading.n)), pdf));
        specularBounce = sampledSpecular;
        /* Russian roulette :180-185 tests the `depth` parameter (always 0): never fires, 0 draws */
        ray = make_ray(f3_add(isect.p, f3_muls(wi, O_EPSILON)), wi, FLT_MAX);
    }
    return L;
}

/* ------------------------------------------------------------------------------------------------
 * MyApp::Tick per-pixel loop -- myapp.cpp:163-175; Accumulator myapp.h:17-19,57-59
 * ---------------------------------------------------------------------------------------------- */
static void stats_add(oracle_stats* a, const oracle_stats* b) {
    a->closest_rays += b->closest_rays; a->anyhit_rays += b->anyhit_rays; a->box_tests += b->box_tests;
    a->interior_visits += b->interior_visits; a->tri_tests += b->tri_tests;
    a->shaded_vertices += b->shaded_vertices; a->samples += b->samples; a->outliers += b->outliers;
}

static void render_sample(const oracle_scene* s, octx* c, int W, int H, int x, int y, float* accum) {
    float px = x + rng_float(c);
    float py = y + rng_float(c);
    float uvx = px / W, uvy = py / H;
    Ray ray = camera_get_ray(&s->cam, c, uvx, uvy);
    f3 clr = PathTracer_Li(s, &ray, c);
    if (f3_hasnans(clr) || isinf(f3_luminance(clr))) {
        clr = F3s(0.f);
        c->st.outliers++;
    }
    c->st.samples++;
    float* px_out = &accum[((size_t)(H - 1 - y) * W + x) * 4];
    px_out[0] += clr.x; px_out[1] += clr.y; px_out[2] += clr.z;
}

void oracle_render(const oracle_scene* s, int W, int H, int x0, int y0, int w, int h, int spp_begin, int spp_count,
                   uint32_t seed_base, int rng_mode, int threads, float* accum, oracle_stats* stats) {
    oracle_stats total; memset(&total, 0, sizeof(total));
    if (rng_mode == ORACLE_RNG_GLOBAL) {
        uint32_t seed = seed_base;
        octx c; memset(&c, 0, sizeof(c)); c.rng = &seed;
        for (int sp = spp_begin; sp < spp_begin + spp_count; sp++)
            for (int y = y0; y < y0 + h; y++)
                for (int x = x0; x < x0 + w; x++) render_sample(s, &c, W, H, x, y, accum);
        total = c.st;
    } else {
        if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
        {
            octx c; memset(&c, 0, sizeof(c));
            uint32_t seed;
            c.rng = &seed;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (int y = y0; y < y0 + h; y++)
                for (int x = x0; x < x0 + w; x++)
                    for (int sp = spp_begin; sp < spp_begin + spp_count; sp++) {
                        seed = oracle_sample_seed((uint32_t)(y * W + x), (uint32_t)(W * H), (uint32_t)sp, seed_base);
                        render_sample(s, &c, W, H, x, y, accum);
                    }
#ifdef _OPENMP
#pragma omp critical
#endif
            stats_add(&total, &c.st);
        }
    }
    if (stats) *stats = total;
}

/* ------------------------------------------------------------------------------------------------
 * Known-answer helpers
 * ---------------------------------------------------------------------------------------------- */
static void canonical_si(SurfaceInteraction* si) {
    f2 uv = {0, 0};
    si_init(si, F3s(0), uv, F3(0, 0, 1), F3(1, 0, 0), F3(0, 1, 0), 0);
}
void oracle_bsdf_eval(const oracle_scene* s, int material, const float wo_[3], const float wi_[3], float f_out[3],
                      float* pdf_out) {
    SurfaceInteraction si; canonical_si(&si);
    BSDF b; bsdf_init(&b, &si, &s->materials[material]);
    f3 wo = F3(wo_[0], wo_[1], wo_[2]), wi = F3(wi_[0], wi_[1], wi_[2]);
    f3 f = bsdf_f(&b, wo, wi, 1);
    f_out[0] = f.x; f_out[1] = f.y; f_out[2] = f.z;
    *pdf_out = bsdf_pdf(&b, wo, wi, 1);
}
void oracle_bsdf_sample(const oracle_scene* s, int material, const float wo_[3], const float u_[2], float wi_out[3],
                        float f_out[3], float* pdf_out, int* specular_out) {
    SurfaceInteraction si; canonical_si(&si);
    BSDF b; bsdf_init(&b, &si, &s->materials[material]);
    f3 wo = F3(wo_[0], wo_[1], wo_[2]), wi = F3s(0);
    f2 u; u.x = u_[0]; u.y = u_[1];
    float pdf = 0; int spec = 0;
    f3 f = bsdf_sample_f(&b, wo, &wi, u, &pdf, 0, &spec);
    wi_out[0] = wi.x; wi_out[1] = wi.y; wi_out[2] = wi.z;
    f_out[0] = f.x; f_out[1] = f.y; f_out[2] = f.z;
    *pdf_out = pdf; *specular_out = spec;
}
void oracle_camera_ray(const oracle_scene* s, float sx, float ty, uint32_t* rng, oracle_ray* out) {
    octx c; memset(&c, 0, sizeof(c)); c.rng = rng;
    Ray r = camera_get_ray(&s->cam, &c, sx, ty);
    out->o[0] = r.O.x; out->o[1] = r.O.y; out->o[2] = r.O.z;
    out->d[0] = r.D.x; out->d[1] = r.D.y; out->d[2] = r.D.z;
    out->tmax = r.t;
}
void oracle_li(const oracle_scene* s, const oracle_ray* ray, uint32_t* rng, float L_out[3], oracle_stats* stats) {
    octx c; memset(&c, 0, sizeof(c)); c.rng = rng;
    Ray r = make_ray(F3(ray->o[0], ray->o[1], ray->o[2]), F3(ray->d[0], ray->d[1], ray->d[2]), ray->tmax);
    f3 L = PathTracer_Li(s, &r, &c);
    L_out[0] = L.x; L_out[1] = L.y; L_out[2] = L.z;
    if (stats) *stats = c.st;
}

/* DbgIntegrator::Li -- integrator.h:107-118: the hit's uv as a colour ((u, v, 0) / 5; red where u or v is exactly 0), black on a miss */
void oracle_dbg_li(const oracle_scene* s, const oracle_ray* ray, float L_out[3]) {
    octx c; memset(&c, 0, sizeof(c));
    Ray r = make_ray(F3(ray->o[0], ray->o[1], ray->o[2]), F3(ray->d[0], ray->d[1], ray->d[2]), ray->tmax);
    SurfaceInteraction si;
    f3 L = F3s(0.f);
    if (scene_intersect(s, &r, &si, &c)) {
        if (si.uv.x == 0 || si.uv.y == 0) L = F3(1, 0, 0);
        else { L.x = si.uv.x / 5; L.y = si.uv.y / 5; L.z = 0.f / 5; }
    }
    L_out[0] = L.x; L_out[1] = L.y; L_out[2] = L.z;
}

/* ------------------------------------------------------------------------------------------------
 * TriangleMesh::CreateBackdrop -- trianglemesh.cpp:232-318 (host-side scene prep)
 * ---------------------------------------------------------------------------------------------- */
void oracle_create_backdrop(const float origin_[3], const float size[3], float radius, int steps, float* verts,
                            float* normals, float* uvs, int32_t* indices, int* n_verts, int* n_indices) {
    f3 origin = F3(origin_[0], origin_[1], origin_[2]);
    float width = size[0], height = size[1], depth = size[2];
    int nv = 0;
#define PUSHV(P, N, U, V) do { f3 p_ = (P); f3 n_ = (N); verts[3*nv]=p_.x; verts[3*nv+1]=p_.y; verts[3*nv+2]=p_.z; \
        normals[3*nv]=n_.x; normals[3*nv+1]=n_.y; normals[3*nv+2]=n_.z; uvs[2*nv]=(U); uvs[2*nv+1]=(V); nv++; } while (0)
    PUSHV(f3_add(origin, F3(width / 2, height, 0)), F3(0, 0, -1), 0, 0);
    PUSHV(f3_add(origin, F3(-width / 2, height, 0)), F3(0, 0, -1), 0, 1);
    PUSHV(f3_add(origin, F3(width / 2, radius * 1.1f, 0)), F3(0, 0, -1), 1, 0);
    PUSHV(f3_add(origin, F3(-width / 2, radius * 1.1f, 0)), F3(0, 0, -1), 1, 1);
    float stepAngle = O_PI / (2 * steps);
    for (int i = 0; i <= steps; i++) {
        float zRot = cosf(stepAngle * i);
        float yRot = -sinf(stepAngle * i);
        f3 n = f3_normalize(F3(0, -yRot, -zRot));
        PUSHV(f3_add(origin, F3(width / 2, yRot * radius + radius, zRot * radius - radius)), n, 2.f + i, 0);
        PUSHV(f3_add(origin, F3(-width / 2, yRot * radius + radius, zRot * radius - radius)), n, 2.f + i, 1);
    }
    PUSHV(f3_add(origin, F3(width / 2, 0, -radius * 1.1f)), F3(0, 1, 0), 3.f + steps, 0);
    PUSHV(f3_add(origin, F3(-width / 2, 0, -radius * 1.1f)), F3(0, 1, 0), 3.f + steps, 1);
    PUSHV(f3_add(origin, F3(width / 2, 0, -depth)), F3(0, 1, 0), 4.f + steps, 0);
    PUSHV(f3_add(origin, F3(-width / 2, 0, -depth)), F3(0, 1, 0), 4.f + steps, 1);
#undef PUSHV
    int ni = 0;
    int numparts = 4 + steps;
#define PUSHI(I) do { indices[3*ni]=(I); indices[3*ni+1]=(I); indices[3*ni+2]=(I); ni++; } while (0)
    for (int i = 0; i < numparts; i++) {
        PUSHI(i * 2); PUSHI((i + 1) * 2); PUSHI(i * 2 + 1);
        PUSHI((i + 1) * 2); PUSHI((i + 1) * 2 + 1); PUSHI(i * 2 + 1);
    }
#undef PUSHI
    *n_verts = nv;
    *n_indices = ni;
}

/* ------------------------------------------------------------------------------------------------
 * Pin helper: the synthetic heightfield the survey stage rendered with the unmodified reference
 * (n x n quads over [-S,S]^2, h = .35 sin3.1x cos2.7z + .15 sin(9.3x+1) sin7.9z,
 * central-difference smooth normals).  Uses libm sinf/cosf so the geometry is bit-identical to that run.
 * Capacities: verts/normals 3*(n+1)^2, uvs 2*(n+1)^2, indices 3*6*n*n.
 * ---------------------------------------------------------------------------------------------- */
static float pin_h(float x, float z) {
    return 0.35f * sinf(3.1f * x) * cosf(2.7f * z) + 0.15f * sinf(9.3f * x + 1.f) * sinf(7.9f * z);
}
void oracle_pin_heightfield(int n, float S, float* verts, float* normals, float* uvs, int32_t* indices) {
    int k = 0;
    for (int j = 0; j <= n; j++)
        for (int i = 0; i <= n; i++, k++) {
            float x = -S + 2 * S * i / n, z = -S + 2 * S * j / n;
            verts[3 * k] = x; verts[3 * k + 1] = pin_h(x, z); verts[3 * k + 2] = z;
            float e = 1e-3f;
            f3 dx = F3(2 * e, pin_h(x + e, z) - pin_h(x - e, z), 0), dz = F3(0, pin_h(x, z + e) - pin_h(x, z - e), 2 * e);
            f3 nn = f3_normalize(f3_cross(dz, dx));
            normals[3 * k] = nn.x; normals[3 * k + 1] = nn.y; normals[3 * k + 2] = nn.z;
            uvs[2 * k] = (float)i / n; uvs[2 * k + 1] = (float)j / n;
        }
    int q = 0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            int a = j * (n + 1) + i, b = a + 1, c = a + n + 1, d = c + 1;
            int tri[6] = {a, c, b, b, c, d};
            for (int t = 0; t < 6; t++, q++) { indices[3 * q] = tri[t]; indices[3 * q + 1] = tri[t]; indices[3 * q + 2] = tri[t]; }
        }
}
