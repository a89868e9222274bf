// agpt_host_scene.cpp -- host-side scene preparation (runs once per scene; the reference does the same work in
// BVHTriMesh's constructor, DisneyMaterial's constructor and Camera's constructor).
#include "agpt_host_scene.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <future>
#include <thread>

namespace agpt {

using uv_t = v2;

// ------------------------------------------------------------------------------------------------------------
// Binned-SAH BVH, built straight into the flattened reference layout.
//   Primitive            bvhtrimesh.h:132-145   bounds of the 3 vertices, centroid = bounds centre
//   BuildRecursive       bvhtrimesh.h:213-310   12 buckets along the longest centroid axis, leaf when one prim is
//                                               left or all centroids coincide on that axis; nth_element for n<=2,
//                                               std::partition on the winning bucket otherwise
//   FlattenBVHTree       bvhtrimesh.h:312-330   root at 0, child pairs allocated in pre-order from slot 2
// std::partition / std::nth_element are the same libstdc++ algorithms the reference calls, so the primitive order
// inside every subtree -- and with it the traversal order of tied hits -- is reproduced.
// ------------------------------------------------------------------------------------------------------------
namespace {

struct Box {
    float lo[3], hi[3];
    Box() {
        for (int a = 0; a < 3; a++) {
            lo[a] = 1e34f;
            hi[a] = -1e34f;
        }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = tminf(lo[a], b.lo[a]);
            hi[a] = tmaxf(hi[a], b.hi[a]);
        }
    }
    void grow(v3 p) {
        lo[0] = tminf(lo[0], p.x);
        lo[1] = tminf(lo[1], p.y);
        lo[2] = tminf(lo[2], p.z);
        hi[0] = tmaxf(hi[0], p.x);
        hi[1] = tmaxf(hi[1], p.y);
        hi[2] = tmaxf(hi[2], p.z);
    }
    float extent(int a) const { return hi[a] - lo[a]; }
    int longest_axis() const {
        int a = 0;
        if (extent(1) > extent(0)) a = 1;
        if (extent(2) > extent(a)) a = 2;
        return a;
    }
    float area() const {
        float dx = extent(0), dy = extent(1), dz = extent(2);
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    float offset(float p, int a) const {
        float o = p - lo[a];
        if (hi[a] > lo[a]) o /= hi[a] - lo[a];
        return o;
    }
};

struct BuildPrim {
    int32_t index;
    Box box;
    float c[3];
};

constexpr int kBuckets = 12;

struct Builder {
    std::vector<BuildPrim>& prims;  // shared; every task works on its own [start, end) range
    std::vector<agpt_bvh_node>& nodes;
    int max_prims;
    int next_pair = 2;
    int total = 0;
    int max_depth = 0;

    Builder(std::vector<agpt_bvh_node>& n, std::vector<BuildPrim>& p, int mp) : prims(p), nodes(n), max_prims(mp) {}

    static int bucket_of(const Box& cb, const BuildPrim& p, int axis) {
        int b = (int)(kBuckets * cb.offset(p.c[axis], axis));
        if (b == kBuckets) b = kBuckets - 1;
        return b;
    }

    void leaf(int slot, int start, int n, const Box& b, int depth) {
        agpt_bvh_node& nd = nodes[slot];
        std::memcpy(nd.bmin, b.lo, 12);
        std::memcpy(nd.bmax, b.hi, 12);
        nd.first = start;
        nd.count = n;
        max_depth = std::max(max_depth, depth);
    }

    // the split decision of BuildRecursive (bvhtrimesh.h:217-305): returns the partition point, or -1 for a leaf;
    // `bounds` receives the node's primitive bounds.  Reorders prims[start, end).
    int choose_split(int start, int end, Box& bounds) {
        bounds = Box();
        for (int i = start; i < end; i++) bounds.grow(prims[i].box);
        const int n = end - start;
        if (n == 1) return -1;
        Box cb;
        for (int i = start; i < end; i++) cb.grow(V3(prims[i].c[0], prims[i].c[1], prims[i].c[2]));
        const int axis = cb.longest_axis();
        if (cb.lo[axis] == cb.hi[axis]) return -1;
        int mid = (start + end) / 2;
        if (n <= 2) {
            std::nth_element(prims.begin() + start, prims.begin() + mid, prims.begin() + end,
                             [axis](const BuildPrim& a, const BuildPrim& b) { return a.c[axis] < b.c[axis]; });
            return mid;
        }
        int count[kBuckets] = {0};
        Box bb[kBuckets];
        for (int i = start; i < end; i++) {
            int b = bucket_of(cb, prims[i], axis);
            count[b]++;
            bb[b].grow(prims[i].box);
        }
        float cost[kBuckets - 1];
        for (int i = 0; i < kBuckets - 1; i++) {
            Box b0, b1;
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; j++) {
                b0.grow(bb[j]);
                c0 += count[j];
            }
            for (int j = i + 1; j < kBuckets; j++) {
                b1.grow(bb[j]);
                c1 += count[j];
            }
            cost[i] = 1 + (c0 * b0.area() + c1 * b1.area()) / bounds.area();
        }
        float min_cost = cost[0];
        int split = 0;
        for (int i = 1; i < kBuckets - 1; i++)
            if (cost[i] < min_cost) {
                min_cost = cost[i];
                split = i;
            }
        const float leaf_cost = (float)n;
        if (n > max_prims || min_cost < leaf_cost) {
            BuildPrim* pmid = std::partition(&prims[start], &prims[end - 1] + 1,
                                             [&](const BuildPrim& p) { return bucket_of(cb, p, axis) <= split; });
            return (int)(pmid - &prims[0]);
        }
        return -1;
    }

    void build(int start, int end, int slot, int depth) {
        total++;
        Box bounds;
        const int mid = choose_split(start, end, bounds);
        if (mid < 0) return leaf(slot, start, end - start, bounds, depth);
        const int first = next_pair;
        next_pair += 2;
        build(start, mid, first, depth + 1);
        build(mid, end, first + 1, depth + 1);
        agpt_bvh_node& nd = nodes[slot];  // Bounds::Union of the children (bvhtrimesh.h:113-118)
        for (int a = 0; a < 3; a++) {
            nd.bmin[a] = tminf(nodes[first].bmin[a], nodes[first + 1].bmin[a]);
            nd.bmax[a] = tmaxf(nodes[first].bmax[a], nodes[first + 1].bmax[a]);
        }
        nd.first = first;
        nd.count = 0;
    }
};

// A subtree built on its own: `nodes` in the layout the sequential builder produces for a tree of its own (root at 0,
// slot 1 unused, child pairs from 2).  Embedding it below a parent only rebases the interior `first` fields, because the
// reference's flatten hands out pair slots in pre-order: parent pair, then every pair of the left subtree, then the right.
struct SubTree {
    std::vector<agpt_bvh_node> nodes;  // total + 1 entries
    int total = 0;
    int max_depth = 0;
};

SubTree build_subtree(std::vector<BuildPrim>& shared, int start, int end, int max_prims, int depth, int par_levels) {
    SubTree out;
    const int n = end - start;
    if (par_levels <= 0 || n < (1 << 16)) {
        // sequential: the Builder works on the shared primitive array in place (disjoint ranges per task)
        out.nodes.assign((size_t)2 * n + 2, agpt_bvh_node{});
        Builder b(out.nodes, shared, max_prims);
        b.build(start, end, 0, depth);
        out.total = b.total;
        out.max_depth = b.max_depth;
        out.nodes.resize((size_t)b.total + 1);
        return out;
    }
    std::vector<agpt_bvh_node> scratch(2);
    Builder top(scratch, shared, max_prims);
    Box bounds;
    const int mid = top.choose_split(start, end, bounds);
    if (mid < 0) return build_subtree(shared, start, end, max_prims, depth, 0);
    auto fut = std::async(std::launch::async, [&]() { return build_subtree(shared, start, mid, max_prims, depth + 1, par_levels - 1); });
    SubTree R = build_subtree(shared, mid, end, max_prims, depth + 1, par_levels - 1);
    SubTree L = fut.get();
    out.total = 1 + L.total + R.total;
    out.max_depth = std::max(L.max_depth, R.max_depth);
    out.nodes.resize((size_t)out.total + 1);
    // children at slots 2,3; left's pairs follow from 4, right's after them
    const int baseL = 4, baseR = 4 + (L.total - 1);
    auto embed = [&](const SubTree& T, int root_slot, int base) {
        for (int i = 0; i <= T.total; i++) {
            if (i == 1) continue;
            agpt_bvh_node nd = T.nodes[i];
            if (nd.count == 0) nd.first = base + (nd.first - 2);
            out.nodes[i == 0 ? root_slot : base + (i - 2)] = nd;
        }
    };
    embed(L, 2, baseL);
    embed(R, 3, baseR);
    agpt_bvh_node& nd = out.nodes[0];
    for (int a = 0; a < 3; a++) {
        nd.bmin[a] = tminf(out.nodes[2].bmin[a], out.nodes[3].bmin[a]);
        nd.bmax[a] = tmaxf(out.nodes[2].bmax[a], out.nodes[3].bmax[a]);
    }
    nd.first = 2;
    nd.count = 0;
    return out;
}

}  // namespace

void build_bvh(HostMesh& mesh, int max_prims_in_node) {
    const int n_tris = (int)mesh.indices.size() / 9;
    std::vector<BuildPrim> prims(n_tris);
    for (int t = 0; t < n_tris; t++) {
        BuildPrim& p = prims[t];
        p.index = 3 * t;
        for (int k = 0; k < 3; k++) p.box.grow(mesh.vertices[mesh.indices[9 * t + 3 * k]]);
        for (int a = 0; a < 3; a++) p.c[a] = (p.box.lo[a] + p.box.hi[a]) * 0.5f;
    }
    // the top levels of the recursion fan out over host threads (subtrees touch disjoint primitive ranges); the result
    // is byte-identical to the sequential build
    int par_levels = 0;
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (const char* e = std::getenv("AGPT_BVH_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    while ((1u << par_levels) < hw && par_levels < 5) par_levels++;
    SubTree tree = build_subtree(prims, 0, n_tris, max_prims_in_node, 0, par_levels);
    mesh.nodes.swap(tree.nodes);
    mesh.total_nodes = tree.total;
    mesh.max_depth = tree.max_depth;
    mesh.prim_index.resize(n_tris);
    for (int t = 0; t < n_tris; t++) mesh.prim_index[t] = prims[t].index;
}

// ------------------------------------------------------------------------------------------------------------
DevMaterial make_material(int type, const float color[3], float roughness, float metallic) {
    DevMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.type = type;
    const v3 c = V3(color[0], color[1], color[2]);
    if (type == AGPT_MAT_DISNEY) {
        // material.h:14-49
        const float eta = 1.5f;
        const float strans = 0.f;
        const float diffuse_weight = (1 - metallic) * (1 - strans);
        if (diffuse_weight > 0) {
            m.has_diffuse = 1;
            m.has_retro = 1;
            v3 d = diffuse_weight * c;
            m.diffuse_R[0] = d.x;
            m.diffuse_R[1] = d.y;
            m.diffuse_R[2] = d.z;
        }
        m.roughness = roughness;
        const float aspect = 1.f;
        float ax = smaxf(.001f, (roughness * roughness) / aspect);
        float ay = smaxf(.001f, (roughness * roughness) * aspect);
        m.alphax = smaxf(0.001f, ax);  // TrowbridgeReitzDistribution ctor, microfacet.h:120-122
        m.alphay = smaxf(0.001f, ay);
        const float spec_tint = 0.f;
        const v3 ctint = V3s(1.f);
        const float r0 = ((eta - 1) * (eta - 1)) / ((eta + 1) * (eta + 1));  // SchlickR0FromEta, disney.h:23
        v3 cspec0 = lerp3(metallic, r0 * lerp3(spec_tint, V3s(1.f), ctint), c);
        m.R0[0] = cspec0.x;
        m.R0[1] = cspec0.y;
        m.R0[2] = cspec0.z;
        m.metallic = metallic;
        m.eta = eta;
        m.has_microfacet = 1;
    } else if (type == AGPT_MAT_MIRROR) {
        m.has_specular = 1;
        m.mirror_R[0] = c.x;
        m.mirror_R[1] = c.y;
        m.mirror_R[2] = c.z;
    } else {
        m.has_diffuse = 1;
        m.diffuse_R[0] = c.x;
        m.diffuse_R[1] = c.y;
        m.diffuse_R[2] = c.z;
    }
    return m;
}

DevCamera make_camera(const agpt_camera_desc& d) {
    DevCamera c;
    const v3 lookfrom = V3(d.lookfrom[0], d.lookfrom[1], d.lookfrom[2]);
    const v3 lookat = V3(d.lookat[0], d.lookat[1], d.lookat[2]);
    const v3 vup = V3(d.vup[0], d.vup[1], d.vup[2]);
    const float theta = d.vfov * AGPT_PI / 180.0f;
    const float h = tanf(theta / 2);
    const float viewport_height = 2 * h;
    const float viewport_width = d.aspect_ratio * viewport_height;
    c.lens_radius = d.aperture / 2;
    const float focus_dist = length(lookat - lookfrom);
    c.w = normalize(lookfrom - lookat);
    c.u = normalize(cross(vup, c.w));
    c.v = cross(c.w, c.u);
    c.origin = lookfrom;
    c.horizontal = focus_dist * viewport_width * c.u;
    c.vertical = focus_dist * viewport_height * c.v;
    c.lower_left_corner = c.origin - c.horizontal / 2 - c.vertical / 2 - focus_dist * c.w;
    return c;
}

// Distribution1D ctor (sampling.h:21-33): running sum of func / n, normalised by its last entry (or the uniform ramp when
// func integrates to 0).  Sequential fp32 sums in this order; returns funcInt.
float build_distribution1d(const float* func, int n, float* cdf) {
    cdf[0] = 0;
    for (int i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    const float func_int = cdf[n];
    if (func_int == 0) {
        for (int i = 1; i < n + 1; i++) cdf[i] = float(i) / float(n);
    } else {
        for (int i = 1; i < n + 1; i++) cdf[i] /= func_int;
    }
    return func_int;
}

HostEnv make_env(const float* rgb, int width, int height) {
    HostEnv e;
    e.width = width;
    e.height = height;
    const int n = width * height;
    e.pixels.resize(n);
    e.func.resize(n);
    for (int idx = 0; idx < n; idx++) {
        float4 p;
        p.x = rgb[3 * idx]; p.y = rgb[3 * idx + 1]; p.z = rgb[3 * idx + 2]; p.w = 0.f;
        e.pixels[idx] = p;
        const int y = idx / width;
        const float th = (y + .5f) * AGPT_PI / height;
        const float max_component = smaxf(p.x, smaxf(p.y, p.z));
        e.func[idx] = max_component * sinf(th);
    }
    e.cdf.resize((size_t)n + 1);
    e.funcInt = build_distribution1d(e.func.data(), n, e.cdf.data());
    return e;
}

// ------------------------------------------------------------------------------------------------------------
static inline float4 F4(float x, float y, float z, float w) {
    float4 r;
    r.x = x;
    r.y = y;
    r.z = z;
    r.w = w;
    return r;
}
static inline float bits(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// Top-level structure over Scene::primitives (scene.h:5-19) for lists longer than 64 entries: ONE binary tree over the root
// boxes of the non-empty meshes, laid out in depth-first order with skip links -- node i is two float4, (bmin.xyz, index of
// the node that follows i's subtree) and (bmax.xyz, leaf: the primitive's list index / interior: ~0u) -- so a ray collects
// its candidate meshes with a stackless walk: hit -> i + 1, miss -> skip.  The tree only selects CANDIDATES (a superset of
// the meshes the reference's list walk can enter); the walk over them stays in list order.  Median split of the box
// centres along their widest axis.
static void toplevel_subtree(std::vector<float4>& nodes, std::vector<std::array<float, 7>>& e, size_t lo, size_t hi) {
    float bmin[3] = {e[lo][0], e[lo][1], e[lo][2]}, bmax[3] = {e[lo][3], e[lo][4], e[lo][5]};
    float cmin[3], cmax[3];
    for (int a = 0; a < 3; a++) cmin[a] = cmax[a] = e[lo][a] + e[lo][3 + a];
    for (size_t k = lo + 1; k < hi; k++)
        for (int a = 0; a < 3; a++) {
            bmin[a] = std::min(bmin[a], e[k][a]);
            bmax[a] = std::max(bmax[a], e[k][3 + a]);
            cmin[a] = std::min(cmin[a], e[k][a] + e[k][3 + a]);
            cmax[a] = std::max(cmax[a], e[k][a] + e[k][3 + a]);
        }
    const size_t self = nodes.size() / 2;
    nodes.push_back(F4(bmin[0], bmin[1], bmin[2], 0.f));
    nodes.push_back(F4(bmax[0], bmax[1], bmax[2], hi - lo == 1 ? e[lo][6] : bits(0xFFFFFFFFu)));
    if (hi - lo > 1) {
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(e.begin() + lo, e.begin() + mid, e.begin() + hi, [axis](const std::array<float, 7>& x, const std::array<float, 7>& y) {
            return x[axis] + x[3 + axis] < y[axis] + y[3 + axis];
        });
        toplevel_subtree(nodes, e, lo, mid);
        toplevel_subtree(nodes, e, mid, hi);
    }
    nodes[2 * self].w = bits((uint32_t)(nodes.size() / 2));
}
// fp16 with directed rounding (k_candidates reads the tree as 16-byte nodes: boxes rounded OUTWARD to half precision stay
// conservative).  Values below the smallest normal half (2^-14) go to 0 or +-2^-14, whichever is outward, so the result
// does not depend on how the device treats half denormals; beyond +-65504 the outward value is +-inf.
static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {   // denormal: man * 2^-24
            float v = (float)man * 0x1p-24f;
            std::memcpy(&f, &v, 4);
            f |= sign;
        }
    } else if (exp == 31) f = sign | 0x7F800000u | (man << 13);
    else f = sign | ((exp + 112u) << 23) | (man << 13);
    float out;
    std::memcpy(&out, &f, 4);
    return out;
}
static uint16_t half_step(uint16_t h, bool up) {   // the neighbouring half in the given direction (h is not NaN)
    const bool neg = (h & 0x8000u) != 0;
    if ((h & 0x7FFFu) == 0) return up ? 0x0001u : 0x8001u;   // +-0 -> smallest denormal of the wanted sign
    if (neg == up) return (uint16_t)(h - 1);                   // towards zero
    return (h & 0x7FFFu) == 0x7C00u ? h : (uint16_t)(h + 1);  // away from zero, saturating at inf
}
static uint16_t half_outward(float v, bool up) {
    if (v != v) return up ? 0x7C00u : 0xFC00u;
    // start from a truncated conversion, then walk to the nearest half on the wanted side
    uint32_t f;
    std::memcpy(&f, &v, 4);
    const uint32_t sign = (f >> 16) & 0x8000u;
    const int32_t e = (int32_t)((f >> 23) & 0xFFu) - 127;
    uint16_t h;
    if (e > 15) h = (uint16_t)(sign | 0x7C00u);
    else if (e < -24) h = (uint16_t)sign;
    else if (e < -14) h = (uint16_t)(sign | (((f & 0x7FFFFFu) | 0x800000u) >> (-e - 1)));
    else h = (uint16_t)(sign | ((uint32_t)(e + 15) << 10) | ((f & 0x7FFFFFu) >> 13));
    for (int guard = 0; guard < 4 && (up ? half_bits_to_float(h) < v : half_bits_to_float(h) > v); ++guard) h = half_step(h, up);
    for (int guard = 0; guard < 4 && (up ? half_bits_to_float(half_step(h, !up)) >= v : half_bits_to_float(half_step(h, !up)) <= v) &&
                        half_step(h, !up) != h; ++guard)
        h = half_step(h, !up);   // (tighten: the truncated start may lie a step too far out)
    if ((h & 0x7C00u) == 0 && (h & 0x3FFu) != 0) {   // a denormal: snap outward to 0 or +-2^-14
        const bool neg = (h & 0x8000u) != 0;
        h = (neg == up) ? (uint16_t)(neg ? 0x8000u : 0x0000u) : (uint16_t)((h & 0x8000u) | 0x0400u);
    }
    return h;
}
void pack_skip_tree16(const std::vector<float4>& nodes, std::vector<uint32_t>& packed) {
    packed.clear();
    for (size_t i = 0; 2 * i + 1 < nodes.size(); i++) {
        const float4 &lo = nodes[2 * i], &hi = nodes[2 * i + 1];
        uint32_t skip, leaf;
        std::memcpy(&skip, &lo.w, 4);
        std::memcpy(&leaf, &hi.w, 4);
        const uint16_t lx = half_outward(lo.x, false), ly = half_outward(lo.y, false), lz = half_outward(lo.z, false);
        const uint16_t hx = half_outward(hi.x, true), hy = half_outward(hi.y, true), hz = half_outward(hi.z, true);
        packed.push_back((uint32_t)lx | ((uint32_t)ly << 16));
        packed.push_back((uint32_t)lz | ((uint32_t)hx << 16));
        packed.push_back((uint32_t)hy | ((uint32_t)hz << 16));
        packed.push_back((skip & 0xFFFFu) | ((leaf == 0xFFFFFFFFu ? 0xFFFFu : (leaf & 0xFFFFu)) << 16));
    }
}
void build_skip_tree(const float* boxes6, const uint32_t* payload, int n, std::vector<float4>& nodes_out) {
    std::vector<std::array<float, 7>> e((size_t)n);
    for (int k = 0; k < n; k++) {
        for (int a = 0; a < 6; a++) e[k][a] = boxes6[6 * k + a];
        e[k][6] = bits(payload[k]);
    }
    if (n > 0) toplevel_subtree(nodes_out, e, 0, e.size());
}

void flatten_scene(const std::vector<HostMesh>& meshes, const std::vector<HostSphere>& spheres,
                   const std::vector<HostPrim>& prims, FlatScene& out) {
    out.nodes.clear();
    out.tri_verts.clear();
    out.tri_shade.clear();
    out.prims.clear();
    out.bigleaves.clear();
    out.max_depth = 0;
    size_t node_base = 0, tri_base = 0;
    std::vector<size_t> mesh_node_base(meshes.size()), mesh_tri_base(meshes.size());
    for (size_t m = 0; m < meshes.size(); m++) {
        mesh_node_base[m] = node_base;
        mesh_tri_base[m] = tri_base;
        size_t nn = meshes[m].nodes.size();
        nn = (nn + 1) & ~size_t(1);  // keep every mesh's child pairs 64-B aligned
        node_base += nn;
        tri_base += meshes[m].prim_index.size();
    }
    out.rootpair_base = (uint32_t)node_base;
    for (int c = 0; c < AGPT_MAX_CHUNKS; c++) out.mesh_masks[c] = 0;
    const size_t n_rootpairs = std::min<size_t>(prims.size(), 64 * AGPT_MAX_CHUNKS);
    out.nodes.assign(2 * (node_base + 2 * n_rootpairs), F4(0, 0, 0, 0));   // + the root-pair records (agpt_scene.h)
    out.tri_verts.resize(3 * tri_base);
    out.tri_shade.resize(4 * tri_base);

    int mesh_ordinal = 0;
    for (size_t pi = 0; pi < prims.size(); pi++) {
        const HostPrim& hp = prims[pi];
        DevPrim dp;
        std::memset(&dp, 0, sizeof(dp));
        dp.mesh_ordinal = mesh_ordinal;
        dp.type = hp.type;
        dp.material = hp.material;
        dp.arealight = hp.arealight;
        if (hp.type == AGPT_PRIM_SPHERE || hp.type == AGPT_PRIM_PLANE) {
            const HostSphere& s = spheres[hp.index];
            dp.cx = s.center.x;
            dp.cy = s.center.y;
            dp.cz = s.center.z;
            dp.r = s.r;
            dp.r2 = s.r2;
            out.prims.push_back(dp);
            continue;
        }
        const HostMesh& mesh = meshes[hp.index];
        const size_t nb = mesh_node_base[hp.index], tb = mesh_tri_base[hp.index];
        const int n_tris = (int)mesh.prim_index.size();
        dp.root = (int32_t)nb;
        dp.tri_base = (int32_t)tb;
        dp.n_tris = n_tris;
        if (n_tris > 0) mesh_ordinal++;
        dp.has_normals = mesh.normals.empty() ? 0 : 1;
        out.max_depth = std::max(out.max_depth, mesh.max_depth);
        // nodes: rebase `first` to global node / slot indices
        for (size_t i = 0; i < mesh.nodes.size(); i++) {
            if (i == 1) continue;  // unused slot
            const agpt_bvh_node& n = mesh.nodes[i];
            uint32_t first = (uint32_t)n.first + (uint32_t)(n.count > 0 ? tb : nb);
            // .z = the traversal encoding of this node (what a parent pushes / descends to), precomputed here so the
            // kernels do not rebuild it per visit; big leaves keep first/count and are flagged in .w
            uint32_t z = first, w = (uint32_t)n.count;
            if (n.count > 0) {
                if (n.count <= 7 && first < 0x10000000u)
                    z = AGPT_ENC_LEAF | ((uint32_t)(n.count - 1) << 28) | first;
                else {
                    z = AGPT_ENC_BIGLEAF | (uint32_t)out.bigleaves.size();
                    out.bigleaves.push_back(first);
                    out.bigleaves.push_back((uint32_t)n.count);
                }
            }
            // sibling pairs (node indices 2k, 2k+1) share one 64-B record with the two boxes interleaved component by
            // component, so a lane's four 16-B loads land as (left, right) register pairs for packed-fp32 slab maths:
            //   [0] lx0 rx0 ly0 ry0   [1] lz0 rz0 lx1 rx1   [2] ly1 ry1 lz1 rz1   [3] zL zR wL wR
            float* rec = reinterpret_cast<float*>(&out.nodes[4 * ((nb + i) >> 1)]);
            const size_t side = (nb + i) & 1;
            for (int a = 0; a < 3; a++) {
                rec[2 * a + side] = n.bmin[a];
                rec[6 + 2 * a + side] = n.bmax[a];
            }
            rec[12 + side] = bits(z);
            rec[14 + side] = bits(w);
        }
        {
            const agpt_bvh_node& r = mesh.nodes[0];
            std::memcpy(dp.root_bmin, r.bmin, 12);
            std::memcpy(dp.root_bmax, r.bmax, 12);
            uint32_t first = (uint32_t)r.first + (uint32_t)(r.count > 0 ? tb : nb);
            if (r.count == 0)
                dp.root_enc = first;
            else if (r.count <= 7 && first < 0x10000000u)
                dp.root_enc = AGPT_ENC_LEAF | ((uint32_t)(r.count - 1) << 28) | first;
            else {
                dp.root_enc = AGPT_ENC_BIGLEAF | (uint32_t)(out.bigleaves.size() / 2);
                out.bigleaves.push_back(first);
                out.bigleaves.push_back((uint32_t)r.count);
            }
        }
        if (pi < 64 * AGPT_MAX_CHUNKS && n_tris > 0) {
            float* rec = reinterpret_cast<float*>(&out.nodes[4 * ((out.rootpair_base + 2 * pi) >> 1)]);
            for (int a = 0; a < 3; a++) {
                rec[2 * a] = rec[2 * a + 1] = dp.root_bmin[a];
                rec[6 + 2 * a] = rec[6 + 2 * a + 1] = dp.root_bmax[a];
            }
            rec[12] = rec[13] = bits(dp.root_enc);
            out.mesh_masks[pi / 64] |= 1ull << (pi % 64);
        }
        // triangles
        const bool has_uv = !mesh.texcoords.empty();
        std::vector<uint32_t> reject(n_tris, 0);
        for (int t = 0; t < n_tris; t++) {
            const int32_t* ix = &mesh.indices[9 * t];
            const v3 v0 = mesh.vertices[ix[0]], v1 = mesh.vertices[ix[3]], v2 = mesh.vertices[ix[6]];
            // trianglemesh.cpp:46-80 -- ray-independent: uv deltas, dpdu/dpdv, degenerate handling
            uv_t uv0, uv1, uv2;
            if (has_uv) {
                uv0 = mesh.texcoords[ix[2]];
                uv1 = mesh.texcoords[ix[5]];
                uv2 = mesh.texcoords[ix[8]];
            } else {
                uv0.x = 0; uv0.y = 0;
                uv1.x = 1; uv1.y = 0;
                uv2.x = 1; uv2.y = 1;
            }
            const float du02x = uv0.x - uv2.x, du02y = uv0.y - uv2.y;
            const float du12x = uv1.x - uv2.x, du12y = uv1.y - uv2.y;
            const v3 dp02 = v0 - v2, dp12 = v1 - v2;
            const float determinant = du02x * du12y - du02y * du12x;
            const bool degenerate_uv = (double)fabsf(determinant) < 1e-8;
            v3 dpdu = V3s(0), dpdv = V3s(0);
            if (!degenerate_uv) {
                const float invdet = 1 / determinant;
                dpdu = (du12y * dp02 - du02y * dp12) * invdet;
                dpdv = (-du12x * dp02 + du02x * dp12) * invdet;
            }
            if (degenerate_uv || sqrlen(cross(dpdu, dpdv)) == 0) {
                v3 ng = cross(v2 - v0, v1 - v0);
                if (sqrlen(ng) == 0) {
                    reject[t] = AGPT_TRI_FLAG_REJECT;
                    dpdu = V3(1, 0, 0);
                    dpdv = V3(0, 1, 0);
                } else {
                    coordinate_system(normalize(ng), &dpdu, &dpdv);
                }
            }
            const v3 ng = normalize(cross(dpdu, dpdv));  // SurfaceInteraction ctor, intersectable.h:69
            const v3 ss = normalize(dpdu);                // BSDF::ss (reflection.cpp:10) and trianglemesh.cpp:100
            v3 n0 = V3s(0), n1 = V3s(0), n2 = V3s(0);
            if (dp.has_normals) {
                n0 = mesh.normals[ix[1]];
                n1 = mesh.normals[ix[4]];
                n2 = mesh.normals[ix[7]];
            }
            float4* q = &out.tri_shade[4 * (tb + t)];
            q[0] = F4(ng.x, ng.y, ng.z, ss.x);
            q[1] = F4(ss.y, ss.z, n0.x, n0.y);
            q[2] = F4(n0.z, n1.x, n1.y, n1.z);
            q[3] = F4(n2.x, n2.y, n2.z, bits((uint32_t)pi));
        }
        for (int s = 0; s < n_tris; s++) {
            const int t = mesh.prim_index[s] / 3;
            const int32_t* ix = &mesh.indices[9 * t];
            const v3 v0 = mesh.vertices[ix[0]], v1 = mesh.vertices[ix[3]], v2 = mesh.vertices[ix[6]];
            float4* q = &out.tri_verts[3 * (tb + s)];
            q[0] = F4(v0.x, v0.y, v0.z, bits((uint32_t)(tb + t)));
            q[1] = F4(v1.x, v1.y, v1.z, bits(reject[t]));
            q[2] = F4(v2.x, v2.y, v2.z, 0.f);
        }
        out.prims.push_back(dp);
    }
    // prefilter table + analytic-primitive masks of the production trace kernel (agpt_scene.h)
    out.prefilter.clear();
    for (int c = 0; c <= AGPT_MAX_CHUNKS; c++) out.pf_begin[c] = 0;
    for (int c = 0; c < AGPT_MAX_CHUNKS; c++) {
        out.analytic_masks[c] = 0;
        out.pf_begin[c] = (int32_t)(out.prefilter.size() / 2);
        for (size_t pi = 64 * (size_t)c; pi < std::min<size_t>(out.prims.size(), 64 * (size_t)(c + 1)); pi++) {
            const DevPrim& dp = out.prims[pi];
            if (dp.type != AGPT_PRIM_MESH)
                out.analytic_masks[c] |= 1ull << (pi % 64);
            else if (dp.n_tris > 0) {
                out.prefilter.push_back(F4(dp.root_bmin[0], dp.root_bmin[1], dp.root_bmin[2], bits((uint32_t)(pi % 64))));
                out.prefilter.push_back(F4(dp.root_bmax[0], dp.root_bmax[1], dp.root_bmax[2], 0.f));
            }
        }
    }
    out.pf_begin[AGPT_MAX_CHUNKS] = (int32_t)(out.prefilter.size() / 2);
    if (out.prefilter.empty()) out.prefilter.assign(2, F4(0, 0, 0, 0));
    // top-level tree over the root boxes of all non-empty meshes (lists longer than 64 primitives, agpt_scene.h)
    out.toplevel.clear();
    if (out.prims.size() > 64) {
        std::vector<float> boxes;
        std::vector<uint32_t> index;
        for (size_t pi = 0; pi < std::min<size_t>(out.prims.size(), 64 * (size_t)AGPT_MAX_CHUNKS); pi++) {
            const DevPrim& dp = out.prims[pi];
            if (dp.type == AGPT_PRIM_MESH && dp.n_tris > 0) {
                boxes.insert(boxes.end(), {dp.root_bmin[0], dp.root_bmin[1], dp.root_bmin[2], dp.root_bmax[0], dp.root_bmax[1], dp.root_bmax[2]});
                index.push_back((uint32_t)pi);
            }
        }
        build_skip_tree(boxes.data(), index.data(), (int)index.size(), out.toplevel);
    }
    out.n_toplevel = (int32_t)(out.toplevel.size() / 2);
    pack_skip_tree16(out.toplevel, out.toplevel16);
    if (out.toplevel16.empty()) out.toplevel16.assign(4, 0u);
}

// ------------------------------------------------------------------------------------------------------------
void create_backdrop(v3 origin, v3 size, float radius, int steps, std::vector<v3>& vertices, std::vector<v3>& normals,
                     std::vector<v2>& texcoords, std::vector<int32_t>& indices) {
    const float width = size.x, height = size.y, depth = size.z;
    auto push = [&](v3 p, v3 n, float u, float v) {
        vertices.push_back(p);
        normals.push_back(n);
        v2 t;
        t.x = u;
        t.y = v;
        texcoords.push_back(t);
    };
    push(origin + V3(width / 2, height, 0), V3(0, 0, -1), 0, 0);
    push(origin + V3(-width / 2, height, 0), V3(0, 0, -1), 0, 1);
    push(origin + V3(width / 2, radius * 1.1f, 0), V3(0, 0, -1), 1, 0);
    push(origin + V3(-width / 2, radius * 1.1f, 0), V3(0, 0, -1), 1, 1);
    const float step_angle = AGPT_PI / (2 * steps);
    for (int i = 0; i <= steps; i++) {
        const float z_rot = cosf(step_angle * i);
        const float y_rot = -sinf(step_angle * i);
        const v3 n = normalize(V3(0, -y_rot, -z_rot));
        push(origin + V3(width / 2, y_rot * radius + radius, z_rot * radius - radius), n, 2.f + i, 0);
        push(origin + V3(-width / 2, y_rot * radius + radius, z_rot * radius - radius), n, 2.f + i, 1);
    }
    push(origin + V3(width / 2, 0, -radius * 1.1f), V3(0, 1, 0), 3.f + steps, 0);
    push(origin + V3(-width / 2, 0, -radius * 1.1f), V3(0, 1, 0), 3.f + steps, 1);
    push(origin + V3(width / 2, 0, -depth), V3(0, 1, 0), 4.f + steps, 0);
    push(origin + V3(-width / 2, 0, -depth), V3(0, 1, 0), 4.f + steps, 1);
    auto idx = [&](int i) {
        indices.push_back(i);
        indices.push_back(i);
        indices.push_back(i);
    };
    const int parts = 4 + steps;
    for (int i = 0; i < parts; i++) {
        idx(i * 2);
        idx((i + 1) * 2);
        idx(i * 2 + 1);
        idx((i + 1) * 2);
        idx((i + 1) * 2 + 1);
        idx(i * 2 + 1);
    }
}

}  // namespace agpt
