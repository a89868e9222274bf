/* include/agpt.h -- C ABI of libagpt_hip.so, the MI355X (gfx950) path-tracing hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The reference has no
 * FFI -- its seam is three in-process C++ abstract classes:
 *     Intersectable  intersectable.h:17-61      Scene  scene.h:3-30      Integrator  integrator.h:28-31
 * and the caller is the per-pixel loop of MyApp::Tick (myapp.cpp:163-175).  Each entry point below
 * names the reference interface it replaces (paths relative to the reference checkout); the C++
 * adapter classes that keep reference-style host code unchanged are in include/agpt_host.hpp and the
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: every function returns AGPT_OK (0) or a negative agpt_status and records a message
 * retrievable with agpt_last_error().  The caller owns host buffers; the library owns device memory
 * except where a parameter is documented as a DEVICE pointer.  One context per GPU/process; calls on
 * one context are serialised by the caller.  All arithmetic is fp32, indices int32.
 */
#ifndef AGPT_H
#define AGPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    AGPT_OK = 0,
    AGPT_ERR_INVALID = -1,   /* bad argument / call order */
    AGPT_ERR_DEVICE = -2,    /* HIP runtime error, no device */
    AGPT_ERR_NOMEM = -3,
    AGPT_ERR_LIMIT = -4,     /* scene exceeds a kernel limit (e.g. BVH deeper than the traversal stack) */
    AGPT_ERR_IO = -5         /* file could not be opened / written (image writers) */
} agpt_status;

typedef struct agpt_ctx agpt_ctx;
typedef struct agpt_scene agpt_scene;

/* material kinds: material.h:11-58 DisneyMaterial, material.h:60-81 MirrorMaterial; DIFFUSE_ONLY is a
 * Material whose SetupBSDF adds a single DisneyDiffuse lobe (disney.h:25-38) -- BASELINE config 2's
 * "Lambertian" (the reference has no Lambertian material). */
enum { AGPT_MAT_DISNEY = 0, AGPT_MAT_MIRROR = 1, AGPT_MAT_DIFFUSE_ONLY = 2 };

/* 32-byte BVH node, identical to the reference's BVHNode (bvhtrimesh.h:126-130) */
typedef struct {
    float bmin[3];
    float bmax[3];
    int32_t first; /* interior: index of the first child (pair first, first+1); leaf: first primitive */
    int32_t count; /* 0 = interior */
} agpt_bvh_node;

/* Ray (camera.h:3-15); d need not be normalised, the library normalises it like Ray's ctor */
typedef struct { float o[3]; float d[3]; float tmax; } agpt_ray;

/* result of Scene::Intersect (scene.h:5-13) reduced to what identifies the hit */
typedef struct {
    int32_t hit;   /* 0/1 */
    int32_t prim;  /* index into Scene::primitives (insertion order); -1 on miss / any-hit queries */
    int32_t tri;   /* triangle meshes: index of the triangle's first index_type (= 3*triangle); else -1 */
    float t, b1, b2;
} agpt_hit;

/* CameraDesc (camera.h:17-25) */
typedef struct {
    float lookfrom[3];
    float lookat[3];
    float vup[3];
    float aspect_ratio;
    float vfov;       /* degrees; reference default 45 */
    float aperture;   /* reference default 0 */
} agpt_camera_desc;

typedef struct {
    uint64_t closest_rays;    /* Scene::Intersect-equivalents: primary + continuation + MIS + emitter pass-through */
    uint64_t anyhit_rays;     /* Scene::IntersectP-equivalents: shadow rays */
    uint64_t interior_visits; /* interior nodes whose child pair was fetched (only if counters enabled) */
    uint64_t tri_tests;       /* triangle tests (only if counters enabled) */
    uint64_t shaded_vertices;
    uint64_t samples;
    uint64_t outliers;        /* NaN/inf samples zeroed (myapp.cpp:169-172) */
    uint64_t iterations;      /* wavefront iterations executed */
    double   trace_ms;        /* HIP-event time of the trace kernels over the call (stream the kernels ran on) */
    double   total_ms;        /* HIP-event time of the whole call */
    uint64_t trace_launches;
    uint64_t root_tests;      /* mesh root-box tests, bvhtrimesh.h:187,195 (only if counters enabled) */
    double   ext_ms, mis_ms, shadow_ms; /* trace_ms split: continuation / MIS closest-hit launches, any-hit launches
                                           (only if agpt_render_params::enable_timing) */
    uint64_t answered_rays;   /* of closest_rays: Scene::Intersect calls of the reference that the production path settles without
                                 a traversal, with the same result for the image -- MIS queries (integrator.h:76-88) towards a
                                 sphere light whose ray misses the light's sphere, and the ray after the last bounce of a path
                                 whose hit nothing reads (integrator.h:139-150).  0 with enable_counters = 1 (reference order). */
} agpt_stats;

/* parameters of one agpt_render call = MyApp::Tick's per-pixel loop (myapp.cpp:163-175) for a tile */
typedef struct {
    int32_t width, height;        /* film size W x H (Accumulator width/height, myapp.h:10) */
    int32_t x0, y0, w, h;         /* tile to render, in film pixels */
    int32_t spp_begin, spp_count; /* samples [spp_begin, spp_begin+spp_count) of every pixel */
    uint32_t seed_base;           /* per-(pixel,sample) stream: WangHash((pixel + W*H*sample + 1)*17 + seed_base) */
    int32_t max_depth;            /* PathTracer::MaxDepth (integrator.h:122), reference default 5 */
    int32_t accum_pitch;          /* row pitch of accum in float4 pixels */
    int32_t accum_row0;           /* film row (after the y flip) that accum's row 0 holds; tile rows map to
                                     accum row (H-1-y) - accum_row0 */
    int32_t samples_per_batch;    /* 0 = library default */
    int32_t enable_counters;      /* 1 = also count interior visits / triangle tests in the reference's order (the instrumented
                                   * reference-order kernel, slower); 2 = the production trace kernel counts the work it
                                   * does itself (child-pair / root-pair records fetched, triangle tests) */
    int32_t enable_timing;        /* 1 = bracket every trace launch with HIP events on the launch stream */
    /* multi-GPU row interleave (0 = off): of the film rows, cut into blocks of interleave_block rows, render only the
     * blocks k with k % interleave_world == interleave_rank (the tile must then be the whole film: x0=y0=0, w=W, h=H).
     * accum is the rank's COMPACT buffer: block j of this rank occupies rows [j*block, j*block+h_j), flipped inside the
     * block like Accumulator::AddSample; accum_row0 is ignored. */
    int32_t interleave_block, interleave_world, interleave_rank;
    int32_t trace_all_rays;       /* 1 = send every Scene::Intersect call of the reference through the BVH.  By default (0) the
                                   * production path only COUNTS the two kinds of call whose answer cannot reach the image (MIS
                                   * queries whose ray misses the sampled sphere light, the ray after a path's last bounce; see
                                   * agpt_stats::answered_rays): same image bit for bit, same ray totals, less work */
} agpt_render_params;

const char* agpt_last_error(void);
int agpt_version(void);

/* context: picks the GPU.  stream = hipStream_t to launch on, or NULL for the null stream */
int agpt_init(int device, agpt_ctx** out);
int agpt_set_stream(agpt_ctx*, void* hip_stream);
void agpt_destroy(agpt_ctx*);

/* ---- scene building: mirrors the reference's scene-construction calls (myapp.cpp:13-114) ---------- */
int agpt_scene_create(agpt_ctx*, agpt_scene** out);
void agpt_scene_destroy(agpt_scene*);
/* DisneyMaterial::Make / MirrorMaterial::Make (material.h:60-62,83-85): returns material id >= 0 */
int agpt_scene_add_material(agpt_scene*, int type, const float color[3], float roughness, float metallic);
/* scene->primitives.push_back(make_shared<BVHTriMesh>(TriangleMesh(indices, vertices, normals, texcoords), mat,
 * maxPrimsInNode)) (bvhtrimesh.h:154-178): builds the binned-SAH BVH on the host.  indices are
 * (vertex, normal, texcoord) triplets = index_type (trianglemesh.h:5-12); n_normals / n_texcoords may be 0.
 * returns the primitive index (position in Scene::primitives) */
int agpt_scene_add_mesh(agpt_scene*, const float* vertices, int n_vertices, const float* normals, int n_normals,
                        const float* texcoords, int n_texcoords, const int32_t* indices, int n_indices,
                        int material, int max_prims_in_node);
/* scene->primitives.push_back(make_shared<Sphere>(center, r, material)); material -1 = nullptr */
int agpt_scene_add_sphere(agpt_scene*, const float center[3], float radius, int material);
/* scene->primitives.push_back(make_shared<Plane>(o, size, material)) (intersectable.h:119-157): XZ rectangle, +Y normal */
int agpt_scene_add_plane(agpt_scene*, const float o[3], const float size[2], int material);
/* Scene::addAreaLight(make_shared<Sphere>(center, r, nullptr), L) (scene.h:21-25): returns primitive index */
int agpt_scene_add_area_light(agpt_scene*, const float center[3], float radius, const float L[3]);
/* scene->lights.push_back(make_shared<UniformInfiniteLight>(L)) (lights.h:37-51): returns light index */
int agpt_scene_add_uniform_infinite_light(agpt_scene*, const float L[3]);
/* scene->lights.push_back(make_shared<InfiniteAreaLight>(texmap)) (lights.cpp:31-112 with ILS): the HDR environment map is
 * passed as width*height RGB floats (row-major, row 0 = top), i.e. what stbi_loadf returns in the reference (texture.h:43-53);
 * builds the max(rgb)*sin(theta) Distribution1D (sampling.h:19-69) on the host.  returns light index */
int agpt_scene_add_infinite_area_light(agpt_scene*, const float* rgb, int width, int height);
/* scene->camera = desc; Camera(desc) (camera.h:29-56,77-90) */
int agpt_scene_set_camera(agpt_scene*, const agpt_camera_desc*);
/* flatten + upload to HBM; must be called after the last add_* and before render/intersect */
int agpt_scene_commit(agpt_scene*);

/* BVH inspection (host copy, reference layout): nodes_out has agpt_mesh_num_nodes()+1 entries (slot 1 unused),
 * prim_index_out the reordered Primitive::index list (bvhtrimesh.h:132-145,208) */
int agpt_mesh_num_nodes(const agpt_scene*, int prim);
int agpt_mesh_num_prims(const agpt_scene*, int prim);
int agpt_mesh_get_bvh(const agpt_scene*, int prim, agpt_bvh_node* nodes_out, int32_t* prim_index_out);
/* BVHTriMesh's constructor alone (bvhtrimesh.h:154-178,213-330), host-only -- needs no GPU and no context.
 * indices as in agpt_scene_add_mesh.  nodes_out needs 2*(n_indices/3)+2 entries of capacity (total_nodes+1 are
 * written, slot 1 unused); prim_index_out n_indices/3 entries.  Either output may be NULL. */
int agpt_bvh_build(const float* vertices, int n_vertices, const int32_t* indices, int n_indices, int max_prims_in_node,
                   agpt_bvh_node* nodes_out, int32_t* prim_index_out, int* total_nodes_out, int* max_depth_out);
/* The top-level structure the library builds over Scene::primitives (scene.h:5-19) when the list is longer than 64 entries,
 * host-only (no GPU, no context): a binary tree over n boxes (6 floats each: bmin.xyz, bmax.xyz) in depth-first order with
 * skip links.  Writes 2n-1 nodes of 8 floats -- (bmin.xyz, uint32 index of the node after this node's subtree), (bmax.xyz,
 * uint32 leaf: index of the box / interior: 0xFFFFFFFF) -- and returns the node count.  A ray collects its candidate
 * primitives with "hit -> next node, miss -> skip link"; the walk over the candidates stays in list order. */
int agpt_toplevel_build(const float* boxes6, int n, float* nodes8_out);
/* The 16-byte node form the GPU reads: per node 4 words -- bmin.x | bmin.y << 16, bmin.z | bmax.x << 16, bmax.y | bmax.z << 16
 * as IEEE halves rounded OUTWARD (bmin down, bmax up; no half denormals; beyond +-65504 -> +-inf), skip | leaf << 16 as
 * 16-bit indices (leaf 0xFFFF = interior).  nodes8 = agpt_toplevel_build's output.  Host-only. */
int agpt_toplevel_pack16(const float* nodes8, int n_nodes, uint32_t* packed4_out);
/* TriangleMesh::CreateBackdrop (trianglemesh.cpp:232-318), host-side scene prep.
 * capacities: vertices/normals 3*2*(steps+5) floats, texcoords 2*2*(steps+5), indices 3*6*(steps+4) ints */
int agpt_create_backdrop(const float origin[3], const float size[3], float radius, int steps, float* vertices,
                         float* normals, float* texcoords, int32_t* indices, int* n_vertices, int* n_indices);

/* TriangleMesh::LoadObj (trianglemesh.cpp:157-230): OBJ text -> the four arrays agpt_scene_add_mesh takes.  Host-only.
 * transform16 = row-major mat4 (template/precomp.h:845-1030) applied to points (TransformPoint) and, as inverse
 * transpose, to normals; NULL = identity.  Polygons are triangulated like the tinyobjloader 2.0.0 the reference vendors.
 * Errors (unreadable file, malformed / zero face index, face referencing a missing normal or texcoord) return
 * AGPT_ERR_INVALID with agpt_obj_last_error() -- the reference calls exit(1) / reads out of bounds instead. */
typedef struct agpt_obj_mesh agpt_obj_mesh;
int agpt_obj_load(const char* path, const float* transform16, int ignore_normals, agpt_obj_mesh** out);
int agpt_obj_parse(const char* text, size_t length, const float* transform16, int ignore_normals, agpt_obj_mesh** out);
int agpt_obj_counts(const agpt_obj_mesh*, int* n_vertices, int* n_normals, int* n_texcoords, int* n_indices);
int agpt_obj_get(const agpt_obj_mesh*, float* vertices, float* normals, float* texcoords, int32_t* indices);
void agpt_obj_free(agpt_obj_mesh*);
const char* agpt_obj_last_error(void);

/* ---- hot path -------------------------------------------------------------------------------------- */
/* Scene::Intersect (any_hit=0, scene.h:5-13) / Scene::IntersectP (any_hit=1, scene.h:15-19) for n rays.
 * rays/out are HOST arrays (copied in/out); BVHTriMesh::RecursiveHit/RecursiveHitP (bvhtrimesh.h:332-413),
 * TriangleIntersect/P (trianglemesh.cpp:7-155), Sphere::Intersect/P (intersectable.h:164-226) run on the GPU. */
int agpt_intersect_batch(agpt_scene*, const agpt_ray* rays, int n, agpt_hit* out, int any_hit, agpt_stats* stats);
/* Same, with rays_dev / out_dev DEVICE arrays (e.g. from agpt_device_alloc or the host application's own HIP
 * allocations): nothing crosses PCIe; enqueued on the context's stream and synchronised before returning. */
int agpt_intersect_device(agpt_scene*, const agpt_ray* rays_dev, int n, agpt_hit* out_dev, int any_hit, agpt_stats* stats);

/* MyApp::Tick's loop body for every pixel of the tile and every sample in the range:
 *   jitter -> Camera::GetRay -> PathTracer::Li -> NaN/inf reject -> Accumulator::AddSample  (myapp.cpp:165-173)
 * accum_dev is a DEVICE pointer to float4 pixels (rgb + unused w); samples are ADDED in sample order, so
 * successive calls continue a progressive render (Accumulator, myapp.h:17-32).  Returns after the work
 * has been enqueued and, if stats != NULL, synchronised. */
int agpt_render(agpt_scene*, const agpt_render_params*, float* accum_dev, agpt_stats* stats);

/* Integrator::Li(const Ray&, const Scene&) (integrator.h:28-31) of PathTracer (integrator.h:120-191) for n rays of the caller --
 * the single-ray entry the reference's split-screen compare (myapp.cpp:168) and mouse picking (myapp.cpp:197-201) use, batched.
 * rays / rng_states / radiance3_out are HOST arrays.  rng_states[i] is the xorshift32 state the path's RandomFloat() calls start
 * from (template/template.cpp:667-675; the reference has one global state -- a host that wants one stream per ray derives them
 * as agpt_render does: WangHash((pixel + W*H*sample + 1)*17 + seed_base)); rng_states_out (may be NULL) receives the state after
 * the path.  radiance3_out[3i..3i+2] is Li's return value unfiltered (the NaN / inf reject belongs to the caller's loop,
 * myapp.cpp:169-172).  Ray::t = rays[i].tmax; the direction is normalised like Ray's ctor. */
int agpt_li_batch(agpt_scene*, const agpt_ray* rays, const uint32_t* rng_states, int n, int max_depth, float* radiance3_out,
                  uint32_t* rng_states_out, agpt_stats* stats);

/* DbgIntegrator::Li (integrator.h:107-118), the reference's debug view: (u, v, 0) / 5 of the closest hit's texture coordinates, red
 * where u or v is exactly 0, black on a miss.  Scene::Intersect runs on the GPU; the uv comes from the host copy of the scene
 * (the device keeps no texture coordinates).  rays / radiance3_out: HOST arrays of n rays and 3n floats. */
int agpt_dbg_li_batch(agpt_scene*, const agpt_ray* rays, int n, float* radiance3_out);

/* ---- multi-GPU (one process / context per GPU) -----------------------------------------------------------
 * The reference is single-process (myapp.cpp:163-175 is its whole frame loop); a multi-GPU host splits the film into
 * row blocks dealt round-robin to the ranks (agpt_render_params::interleave_*), every rank accumulates its blocks for
 * the whole sample budget into its compact buffer, and ONE exchange at resolve time brings the buffers to rank 0.
 * agpt_comm_unique_id: rank 0 fills a 128-byte id (ncclUniqueId) that the host passes to the other ranks by its own
 * means (MPI_Bcast, a file, an environment variable); agpt_comm_init: every rank, collectively (world == 1 needs no id
 * and never loads RCCL).  agpt_gather_tiles: every rank, collectively, enqueued on the context's stream -- grouped
 * RCCL send/recv of the compact buffers to rank 0 (each peer over its own direct xGMI link) followed, on rank 0, by the
 * de-interleave into the full accumulator full_accum_dev[height][width] float4 in Accumulator::pixels order (row
 * H-1-y, myapp.h:17-19); full_accum_dev is ignored on the other ranks. */
typedef struct agpt_comm agpt_comm;
int agpt_comm_unique_id(void* id128);
int agpt_comm_init(agpt_ctx*, const void* id128, int world, int rank, agpt_comm** out);
void agpt_comm_destroy(agpt_comm*);
int agpt_gather_tiles(agpt_comm*, const float* local_accum_dev, int width, int height, int block_rows, float* full_accum_dev);
/* The de-interleave step alone (no communication; enqueued on the context's stream): writes the rows of `rank`'s compact
 * buffer to their places in the full accumulator.  For hosts that move the buffers themselves (MPI, hipMemcpyPeer). */
int agpt_deinterleave_tiles(agpt_ctx*, const float* compact_dev, int width, int height, int block_rows, int world, int rank,
                            float* full_accum_dev);

/* Accumulator::CopyToSurface (myapp.h:34-41): lin2rgb(sum/samples) -> 0x00RRGGBB.  accum_dev DEVICE float4,
 * out_rgb HOST uint32[w*h] */
int agpt_resolve(agpt_ctx*, const float* accum_dev, int n_pixels, int samples, uint32_t* out_rgb);

/* Host-only image writers (no GPU work) for headless use; the reference only blits to an OpenGL window
 * (myapp.cpp:177).  agpt_write_png: rgb = width*height 0x00RRGGBB words as agpt_resolve returns them, top row first
 * (Accumulator::CopyToSurface order).  agpt_write_pfm: accum_host = a HOST copy of the float4 accumulator (row 0 =
 * top image row, myapp.h:24); writes sum/samples as little-endian RGB float32, bottom row first. */
int agpt_write_png(const char* path, const uint32_t* rgb, int width, int height);
int agpt_write_pfm(const char* path, const float* accum_host, int width, int height, int samples);

/* Radiance RGBE (.hdr) reader, host only: what HDRTexture's constructor gets from stbi_loadf(filename, &w, &h, &n, 0)
 * (texture.h:41-52; the vendored stb_image v2.27, lib/stb_image.h:7005-7215) -- *rgb_out = width*height RGB float32, top row
 * first, malloc'ed, released with agpt_hdr_free; pass it to agpt_scene_add_infinite_area_light.  Anything that is not a
 * well-formed Radiance file fails with AGPT_ERR_INVALID (the reference goes on to stb's LDR decoders, or reads past the end). */
int agpt_hdr_load(const char* path, int* width_out, int* height_out, float** rgb_out);
int agpt_hdr_parse(const void* bytes, size_t length, int* width_out, int* height_out, float** rgb_out);
void agpt_hdr_free(float* rgb);

/* device memory helpers for hosts that do not bring their own allocator (tests, the C++ adapter) */
int agpt_device_alloc(agpt_ctx*, size_t bytes, void** out_dev);
int agpt_device_free(agpt_ctx*, void* dev);
int agpt_device_memset(agpt_ctx*, void* dev, int value, size_t bytes);
int agpt_device_download(agpt_ctx*, void* host_dst, const void* dev_src, size_t bytes);
int agpt_device_upload(agpt_ctx*, void* dev_dst, const void* host_src, size_t bytes);

/* known-answer entry points (each runs the device implementation of one hot-path function on one lane) */
/* BSDF::f + BSDF::Pdf (reflection.h:114-123,174-188) on the canonical frame ng = ns = +z, ss = +x */
int agpt_kat_bsdf_eval(agpt_scene*, int material, int n, const float* wo3, const float* wi3, float* f3_out, float* pdf_out);
/* BSDF::Sample_f (reflection.h:124-172) */
int agpt_kat_bsdf_sample(agpt_scene*, int material, int n, const float* wo3, const float* u2, float* wi3_out,
                         float* f3_out, float* pdf_out, int32_t* specular_out);
/* RNG stream of a (pixel, sample): first n floats (template.cpp:667-675 + cl/tools.cl:1-2) */
int agpt_kat_rng(agpt_ctx*, uint32_t pixel, uint32_t wh, uint32_t sample, uint32_t seed_base, int n, float* out,
                 uint32_t* seed_out);
/* Distribution1D (sampling.h:19-69), the importance table of InfiniteAreaLight (lights.cpp:31-48): the ctor on the host
 * (cdf_out[n + 1], *func_int_out; either may be NULL) and SampleContinuous (:37-52, with FindInterval :4-17) on the device
 * for k draws u[k] -> x_out[k], pdf_out[k] */
int agpt_kat_distribution1d(agpt_ctx*, const float* func, int n, const float* u, int k, float* cdf_out, float* func_int_out,
                            float* x_out, float* pdf_out);

#ifdef __cplusplus
}
#endif
#endif
