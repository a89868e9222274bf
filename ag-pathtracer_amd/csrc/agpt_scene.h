// agpt_scene.h -- flat, HBM-resident scene representation shared by the host builder and the kernels.
//
// Layout (all arrays 64-B aligned, replicated per GPU):
//   nodes      float4[2*N]  the reference's 32-B BVHNode content (bvhtrimesh.h:126-130: bounds, first, count), stored per
//                           SIBLING PAIR: nodes 2k and 2k+1 (always fetched together, bvhtrimesh.h:350-351) share one
//                           64-B line with their boxes interleaved component-wise --
//                             [4k+0] lx0 rx0 ly0 ry0  [4k+1] lz0 rz0 lx1 rx1  [4k+2] ly1 ry1 lz1 rz1  [4k+3] zL zR wL wR
//                           z = the node's precomputed traversal encoding (interior: child-pair index; leaf:
//                           AGPT_ENC_LEAF|count-1|first slot; leaf with > 7 primitives: AGPT_ENC_BIGLEAF|k, with
//                           (first, count) in bigleaves[2k..]), w = count; all meshes concatenated, indices rebased
//                           to global node / triangle-slot indices
//   tri_verts  float4[3*T]  per REORDERED primitive slot (BVH leaf order): v0,v1,v2; v0.w = global triangle id,
//                           v1.w = flags (bit0: degenerate-reject, quirk 11), v2.w unused.  Replaces the reference's
//                           primitives[] -> indices[] -> vertices[] double indirection (88 B/test -> 48 B/test)
//   tri_shade  float4[4*T]  per global triangle id, ray-independent parts of the SurfaceInteraction the reference
//                           rebuilds on every accepted hit (trianglemesh.cpp:46-111): geometric normal ng,
//                           ss = normalize(dpdu), the three vertex normals, owning primitive
//   prims      DevPrim[P]   Scene::primitives in insertion order (scene.h:5-19 walks them linearly)
//   materials  DevMaterial[M], lights DevLight[L]
#pragma once

#include "agpt_math.h"

#define AGPT_PRIM_MESH 0
#define AGPT_PRIM_SPHERE 1
#define AGPT_PRIM_PLANE 2   // intersectable.h:119-157: cx,cy,cz = O, r = HalfSize.x, r2 = HalfSize.y
#define AGPT_LIGHT_AREA 0
#define AGPT_LIGHT_UNIFORM_INFINITE 1
#define AGPT_LIGHT_INFINITE_AREA 2

#define AGPT_HIT_MISS 0xFFFFFFFFu
#define AGPT_HIT_SPHERE 0x80000000u   // analytic primitive (sphere or plane): low bits = primitive index

#define AGPT_TRI_FLAG_REJECT 1u     // degenerate triangle: TriangleIntersect returns false after the t test
#ifndef AGPT_STACK_DEPTH
#define AGPT_STACK_DEPTH 32         // per-lane traversal stack entries staged in LDS (production kernel)
#endif
#define AGPT_MAX_CHUNKS 32          // the production trace kernels walk lists of up to 32 x 64 primitives (beyond: the reference-order kernel)
#define AGPT_STACK_DEPTH_MAX 64     // deepest BVH the generic kernel's 64-entry variant can walk

// traversal-stack / node encoding: bit31 = leaf.  leaf with count<=7: bits 28..30 = count-1, bits 0..27 = first slot.
// leaf with more prims (identical centroids, bvhtrimesh.h:235-238): 0xF0000000 | node index (node is re-fetched).
#define AGPT_ENC_LEAF 0x80000000u
#define AGPT_ENC_BIGLEAF 0xF0000000u

struct DevPrim {
    int32_t type;
    int32_t material;   // -1 = nullptr (emitter spheres)
    int32_t arealight;  // index into lights, -1 = none
    int32_t root;       // mesh: global index of the root node
    int32_t tri_base;   // mesh: global id of its first triangle
    int32_t has_normals;
    int32_t n_tris;
    int32_t mesh_ordinal;  // number of non-empty meshes before this primitive in Scene::primitives
    float cx, cy, cz, r;  // sphere (intersectable.h:161-162)
    float r2;
    float root_bmin[3];   // mesh root bounds (bvhtrimesh.h:185-198 tests them before descending)
    float root_bmax[3];
    uint32_t root_enc;    // encoded root node
};

struct DevMaterial {
    int32_t type;
    int32_t has_diffuse, has_retro, has_microfacet, has_specular;
    float diffuse_R[3];  // diffuseWeight * color
    float roughness;
    float alphax, alphay;
    float R0[3];         // Cspec0
    float metallic, eta;
    float mirror_R[3];
};

struct DevLight {
    int32_t type;
    int32_t shape;  // primitive index of the emitting sphere
    float L[3];
    int32_t env;    // AGPT_LIGHT_INFINITE_AREA: index into DevScene::envs
};

// InfiniteAreaLight's HDRTexture + Distribution1D (lights.cpp:31-48, texture.h:41-84, sampling.h:19-69) in HBM
struct DevEnv {
    const float4* pixels;  // [height*width] rgb
    const float* func;     // [n] max(rgb) * sin(theta_row)
    const float* cdf;      // [n+1]
    int32_t width, height, n;
    float funcInt;
};

struct DevCamera {
    v3 origin, u, v, w, lower_left_corner, horizontal, vertical;
    float lens_radius;
};

struct DevScene {
    const float4* nodes;
    const uint32_t* bigleaves;  // (first, count) pairs, see above
    const float4* tri_verts;
    const float4* tri_shade;
    const DevPrim* prims;
    const DevMaterial* materials;
    const DevLight* lights;
    const DevEnv* envs;
    int32_t n_prims, n_lights, n_materials;
    int32_t n_infinite;        // number of IsInfinite() lights
    int32_t n_meshes;          // non-empty mesh primitives
    int32_t max_depth;         // deepest BVH of the scene (selects the traversal-stack size)
    // Root pairs: for each of the first 64 * AGPT_MAX_CHUNKS primitives that is a non-empty mesh, nodes[] holds one extra pair record at
    // node index rootpair_base + 2k whose LEFT box is the mesh's root box and whose encoding is the mesh's root -- the
    // reference's root-box test at BVHTriMesh::Intersect (bvhtrimesh.h:187,195) then runs as an ordinary interior step
    // of the traversal kernel (the right half of the record is ignored).  mesh_masks: bit k of word c set for primitive
    // 64c + k if it is such a mesh.
    uint32_t rootpair_base;
    unsigned long long mesh_masks[AGPT_MAX_CHUNKS];   // per chunk of 64 primitives
    unsigned long long analytic_masks[AGPT_MAX_CHUNKS];   // spheres and planes, same layout
    // Prefilter table: per non-empty mesh in list order two float4 -- (root bmin.xyz, bit index in its chunk), (root
    // bmax.xyz, -) -- read through scalar loads by the trace kernel's conservative per-primitive filter; chunk c owns the
    // records [pf_begin[c], pf_begin[c + 1]).
    const float4* prefilter;
    int32_t pf_begin[AGPT_MAX_CHUNKS + 1];
    // Top-level tree (primitive lists longer than 64 entries; agpt_host_scene.cpp: build_skip_tree, pack_skip_tree16): a binary
    // tree over the root boxes of all non-empty meshes in depth-first order with skip links, one 16-byte node each: the box
    // as six IEEE halves rounded OUTWARD (x: bmin.x | bmin.y << 16, y: bmin.z | bmax.x << 16, z: bmax.y | bmax.z << 16) and
    // w: index of the node after this node's subtree | (leaf: list index of the primitive / interior: 0xFFFF) << 16.
    // k_candidates walks it once per ray (one 16-byte load per visit: the walk is bound by the vector L1's rate for divergent
    // loads) and leaves one 64-bit candidate word per chunk of 64 primitives; k_trace_fast<LIST> walks the candidates in
    // list order and tests the exact fp32 root boxes.
    const uint4* toplevel;   // 16-byte nodes, see below
    int32_t n_toplevel;
    const unsigned long long* chunk_mesh_masks;   // mesh_masks[] again, in global memory (per-lane chunk index)
    DevCamera cam;
};

// 16-B hit record written by the trace kernel
struct DevHit {
    float t;
    uint32_t id;  // AGPT_HIT_MISS | AGPT_HIT_SPHERE|prim | global triangle id
    float b1, b2;
};
