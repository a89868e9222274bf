/* oracle/agpt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C API of the CPU oracle: a plain-C restatement of the reference's path-tracing hot path
 * (voxel-tracer/ag-pathtracer: integrator.h, bvhtrimesh.h, trianglemesh.cpp, intersectable.h,
 * reflection.h, microfacet.h, disney.h, material.h, lights.cpp, camera.h, template/common.h).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped product (ag-pathtracer_amd/, libagpt_hip.so) never links or imports it.
 *
 * PARITY PIN: the reference itself cannot be compiled in this image without stand-in headers
 * (template/precomp.h:20,41 include <io.h>/<windows.h> unconditionally), so this restatement is
 * pinned against outputs of the *unmodified* reference recorded in BASELINE.md section 2 / SURVEY.md
 * section 6 (smoke-scene mean RGB to 6 digits, exact ray counts of two 1-spp renders, BVH node counts,
 * first RNG float) -- see tests/test_oracle_pins.py.  What those do not cover is "parity unpinned".
 */
#ifndef AGPT_ORACLE_H
#define AGPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

enum { ORACLE_MAT_DISNEY = 0, ORACLE_MAT_MIRROR = 1, ORACLE_MAT_DIFFUSE_ONLY = 2 };
enum { ORACLE_RNG_PER_SAMPLE = 0, ORACLE_RNG_GLOBAL = 1 };

typedef struct {
    float bmin[3];
    float bmax[3];
    int32_t first;
    int32_t count;
} oracle_bvh_node; /* 32 B, bvhtrimesh.h:126-130 */

typedef struct { float o[3]; float d[3]; float tmax; } oracle_ray; /* d is normalised by the callee like Ray's ctor */
typedef struct {
    int32_t hit;    /* 0/1 */
    int32_t prim;   /* scene primitive index (order of insertion) */
    int32_t tri;    /* index of the first index_type of the triangle (= 3*triangle), -1 for spheres */
    float t, b1, b2;
} oracle_hit;

typedef struct {
    uint64_t closest_rays;  /* Scene::Intersect calls */
    uint64_t anyhit_rays;   /* Scene::IntersectP calls */
    uint64_t box_tests;     /* Bounds::Intersect calls */
    uint64_t interior_visits; /* interior nodes whose two children were fetched */
    uint64_t tri_tests;     /* TriangleIntersect + TriangleIntersectP calls */
    uint64_t shaded_vertices;
    uint64_t samples;
    uint64_t outliers;      /* NaN/inf samples zeroed, myapp.cpp:169-172 */
} oracle_stats;

oracle_scene* oracle_scene_new(void);
void oracle_scene_free(oracle_scene*);

/* materials: returns material id */
int oracle_add_material(oracle_scene*, int type, const float color[3], float roughness, float metallic);

/* BVHTriMesh(TriangleMesh(indices, vertices, normals, texcoords, mat), mat, maxPrimsInNode):
 * indices are (vertex,normal,texcoord) triplets, n_indices = 3 * triangles. returns primitive index */
int oracle_add_mesh(oracle_scene*, const float* verts, int n_verts, const float* normals, int n_normals,
                    const float* uvs, int n_uvs, const int32_t* indices, int n_indices, int material,
                    int max_prims_in_node);
/* Sphere(center, r, material) pushed to Scene::primitives; material -1 = nullptr */
int oracle_add_sphere(oracle_scene*, const float center[3], float radius, int material);
/* Plane(o, size, material) pushed to Scene::primitives (intersectable.h:119-157) */
int oracle_add_plane(oracle_scene*, const float o[3], const float size[2], int material);
/* Scene::addAreaLight(make_shared<Sphere>(center, r, nullptr), L): returns primitive index */
int oracle_add_area_light(oracle_scene*, const float center[3], float radius, const float L[3]);
/* scene.lights.push_back(make_shared<UniformInfiniteLight>(L)) */
int oracle_add_uniform_infinite_light(oracle_scene*, const float L[3]);
/* scene.lights.push_back(make_shared<InfiniteAreaLight>(texmap)) with the HDR image given as W*H RGB floats */
int oracle_add_infinite_area_light(oracle_scene*, const float* rgb, int width, int height);
/* CameraDesc + Camera ctor (camera.h:17-56,77-90) */
void oracle_set_camera(oracle_scene*, const float lookfrom[3], const float lookat[3], const float vup[3],
                       float aspect_ratio, float vfov, float aperture);
void oracle_set_max_depth(oracle_scene*, int max_depth);

/* BVH inspection (for pinning the product's builder) */
int oracle_mesh_num_nodes(const oracle_scene*, int prim);         /* totalNodes (slot 1 unused -> array has +1) */
int oracle_mesh_num_prims(const oracle_scene*, int prim);
void oracle_mesh_get_bvh(const oracle_scene*, int prim, oracle_bvh_node* nodes_out /*[num_nodes+1]*/,
                         int32_t* prim_index_out /*[num_prims] Primitive::index in reordered order*/);

/* Scene::Intersect / IntersectP over a batch of rays */
void oracle_intersect_batch(const oracle_scene*, const oracle_ray* rays, int n, oracle_hit* out, int any_hit,
                            oracle_stats* stats);

/* The per-pixel loop of MyApp::Tick (myapp.cpp:163-175) for a tile [x0,x0+w) x [y0,y0+h) of a W x H film,
 * samples [spp_begin, spp_begin+spp_count).  accum is float[W*H*4] (rgb + unused), row (H-1-y) like
 * Accumulator::AddSample; samples are ADDED.  rng_mode GLOBAL = one serial xorshift32 stream seeded with
 * seed_base (the reference as shipped); PER_SAMPLE = seed WangHash((pixel + W*H*sample + 1)*17 + seed_base).
 * threads <= 1 runs serially; GLOBAL mode is always serial. */
void oracle_render(const oracle_scene*, int W, int H, int x0, int y0, int w, int h, int spp_begin, int spp_count,
                   uint32_t seed_base, int rng_mode, int threads, float* accum, oracle_stats* stats);

/* Known-answer helpers */
uint32_t oracle_wang_hash(uint32_t s);
uint32_t oracle_sample_seed(uint32_t pixel, uint32_t wh, uint32_t sample, uint32_t seed_base);
void oracle_rng_floats(uint32_t seed, int n, float* out, uint32_t* out_u);
int oracle_bounds_intersect(const float bmin[3], const float bmax[3], const oracle_ray* ray, float* tmin_out);
/* BSDF known answers for a material on a canonical frame (ng = ns = +z, ss = +x):
 * f(wo,wi), Pdf(wo,wi), Sample_f(wo,u) -> wi, f, pdf */
void oracle_bsdf_eval(const oracle_scene*, int material, const float wo[3], const float wi[3], float f_out[3],
                      float* pdf_out);
void oracle_bsdf_sample(const oracle_scene*, int material, const float wo[3], const float u[2], float wi_out[3],
                        float f_out[3], float* pdf_out, int* specular_out);
/* camera ray for film coords (s,t) with the scene camera; consumes RNG from *rng if aperture>0 */
/* Distribution1D (sampling.h:19-69): ctor and SampleContinuous on caller-supplied arrays */
void oracle_dist1d_build(const float* func, int n, float* cdf_out /*[n+1]*/, float* funcInt_out);
void oracle_dist1d_sample(const float* func, const float* cdf, float funcInt, int n, const float* u, int k, float* x_out,
                          float* pdf_out, int* offset_out);
/* DbgIntegrator::Li (integrator.h:107-118) */
void oracle_dbg_li(const oracle_scene*, const oracle_ray* ray, float L_out[3]);
void oracle_camera_ray(const oracle_scene*, float s, float t, uint32_t* rng, oracle_ray* out);
/* one full path (PathTracer::Li) from an explicit RNG state; returns radiance */
void oracle_li(const oracle_scene*, const oracle_ray* ray, uint32_t* rng, float L_out[3], oracle_stats* stats);

/* Host-side scene prep restated for tests: TriangleMesh::CreateBackdrop (trianglemesh.cpp:232-318).
 * Output capacities: verts/normals 3*(2*(steps+5)) floats, uvs 2*(2*(steps+5)), indices 3*6*(steps+4) ints */
void oracle_create_backdrop(const float origin[3], const float size[3], float radius, int steps, float* verts,
                            float* normals, float* uvs, int32_t* indices, int* n_verts, int* n_indices);
/* rgb2lin / hex2lin (template/common.h:29-39) */
void oracle_hex2lin(int hex, float out[3]);
void oracle_rgb2lin(const float in[3], float out[3]);
/* Accumulator::CopyToSurface (myapp.h:34-41) with lin2rgb / rgb2uint (template/common.h:41-51):
 * accum = n_pixels float4 sums (rgb + unused) -> 0x00RRGGBB */
void oracle_copy_to_surface(const float* accum, int n_pixels, int samples, uint32_t* out);

/* 0 = libm sinf/cosf/acosf (pinned against the reference outputs), 1 = correctly rounded through fp64 (what the HIP
 * kernels compute); see oracle.c */
void oracle_set_trig_mode(int mode);

/* Pin helper: heightfield mesh of the survey's reference runs. */
void oracle_pin_heightfield(int n, float S, float* verts, float* normals, float* uvs, int32_t* indices);

#ifdef __cplusplus
}
#endif
#endif
