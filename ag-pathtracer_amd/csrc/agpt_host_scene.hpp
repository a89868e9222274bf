// agpt_host_scene.hpp -- host-side scene container behind the C ABI: keeps what the reference's Scene /
// BVHTriMesh / Sphere / Material / Light objects hold (scene.h, bvhtrimesh.h, intersectable.h, material.h,
// lights.h) as flat arrays, builds the binned-SAH BVH on the host and flattens everything into the
// HBM layout of agpt_scene.h.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/agpt.h"
#include "agpt_scene.h"

namespace agpt {

struct HostMesh {
    std::vector<v3> vertices, normals;
    std::vector<v2> texcoords;
    std::vector<int32_t> indices;  // (v, n, t) triplets
    int material = -1;
    // BVH in the reference layout (bvhtrimesh.h:126-130, 312-330); nodes[1] is the unused slot
    std::vector<agpt_bvh_node> nodes;
    int total_nodes = 0;
    std::vector<int32_t> prim_index;  // reordered Primitive::index (= 3 * triangle)
    int max_depth = 0;
};

struct HostSphere {
    v3 center;
    float r, r2;
};

struct HostPrim {
    int type;
    int index;  // into meshes / spheres
    int material;
    int arealight;
};

struct HostLight {
    int type;
    int shape;
    v3 L;
    int env = -1;
};

// InfiniteAreaLight ctor data (lights.cpp:31-48) + Distribution1D ctor (sampling.h:21-33)
struct HostEnv {
    int width = 0, height = 0;
    std::vector<float4> pixels;
    std::vector<float> func, cdf;
    float funcInt = 0;
};
float build_distribution1d(const float* func, int n, float* cdf /*[n + 1]*/);
HostEnv make_env(const float* rgb, int width, int height);

// BVHTriMesh ctor: Primitive list + BuildRecursive + FlattenBVHTree (bvhtrimesh.h:154-178,213-330)
void build_bvh(HostMesh& mesh, int max_prims_in_node);

// DisneyMaterial / MirrorMaterial ctor parameterisation (material.h:14-49,72-77)
DevMaterial make_material(int type, const float color[3], float roughness, float metallic);

// Camera ctor + updateCoords (camera.h:38-56,77-90)
DevCamera make_camera(const agpt_camera_desc& d);

struct FlatScene {
    std::vector<float4> nodes, tri_verts, tri_shade;
    std::vector<DevPrim> prims;
    std::vector<uint32_t> bigleaves;  // (first slot, count) pairs of leaves too big for the inline encoding
    int max_depth = 0;
    uint32_t rootpair_base = 0;        // node index of primitive 0's root pair (agpt_scene.h)
    unsigned long long mesh_masks[AGPT_MAX_CHUNKS] = {};  // non-empty meshes, one word per chunk of 64 primitives
    unsigned long long analytic_masks[AGPT_MAX_CHUNKS] = {};  // spheres and planes, same layout
    // prefilter table (agpt_scene.h): two float4 per non-empty mesh in list order, chunk c = records [pf_begin[c], pf_begin[c+1])
    std::vector<float4> prefilter;
    int32_t pf_begin[AGPT_MAX_CHUNKS + 1] = {};
    // top-level tree (agpt_scene.h), lists longer than 64 primitives only
    std::vector<float4> toplevel;       // builder output (fp32 boxes)
    std::vector<uint32_t> toplevel16;   // the same tree as the device reads it: 4 words per node, boxes rounded outward to fp16
    int32_t n_toplevel = 0;
};
// 16-byte nodes from build_skip_tree's output: (lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16, skip | leaf << 16),
// box coordinates as IEEE half rounded outward (lo down, hi up), skip / leaf as 16-bit indices (leaf 0xFFFF = interior)
void pack_skip_tree16(const std::vector<float4>& nodes, std::vector<uint32_t>& packed);
// one top-level tree (depth-first order, skip links) over n boxes (bmin.xyz, bmax.xyz each), appended to nodes_out;
// payload[k] = the value stored in leaf k (the primitive's list index)
void build_skip_tree(const float* boxes6, const uint32_t* payload, int n, std::vector<float4>& nodes_out);
// per-mesh concatenation + per-triangle precomputation (trianglemesh.cpp:46-111's ray-independent part)
void flatten_scene(const std::vector<HostMesh>& meshes, const std::vector<HostSphere>& spheres,
                   const std::vector<HostPrim>& prims, FlatScene& out);

// TriangleMesh::CreateBackdrop (trianglemesh.cpp:232-318)
void create_backdrop(v3 origin, v3 size, float radius, int steps, std::vector<v3>& vertices, std::vector<v3>& normals,
                     std::vector<v2>& texcoords, std::vector<int32_t>& indices);

}  // namespace agpt
