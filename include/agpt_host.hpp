// include/agpt_host.hpp -- C++ host adapter over the C ABI of include/agpt.h.
//
// Header-only mirror of the reference's host-side surface for the path-tracing hot path, so that code written
// like MyApp::Init / MyApp::Tick (myapp.cpp:13-135,141-190) keeps its shape while the per-pixel loop, BVH traversal
// and shading run on the MI355X:
//
//   reference (paths relative to the reference checkout)            this header
//   ---------------------------------------------------------------------------------------------------------
//   DisneyMaterial::Make(color, rough, metallic)  material.h:60     agpt::DisneyMaterial::Make(scene, ...)
//   MirrorMaterial::Make(r)                        material.h:83     agpt::MirrorMaterial::Make(scene, r)
//   TriangleMesh::CreateBackdrop(...)              trianglemesh.cpp:232   agpt::TriangleMesh::CreateBackdrop(...)
//   TriangleMesh::LoadObj(file, mat, transform)    trianglemesh.cpp:157   agpt::TriangleMesh::LoadObj(file, transform16)
//   make_shared<BVHTriMesh>(mesh, mat, 1)          bvhtrimesh.h:154  scene.primitives_push_back(mesh, mat, 1)
//   make_shared<Sphere>(c, r, mat)                 intersectable.h:161    scene.primitives_push_back(Sphere{c, r}, mat)
//   scene->addAreaLight(sphere, L)                 scene.h:21        scene.addAreaLight(Sphere{c, r}, L)
//   scene->lights.push_back(UniformInfiniteLight)  lights.h:37       scene.lights_push_back(UniformInfiniteLight{L})
//   scene->lights.push_back(InfiniteAreaLight(hdr)) lights.h:53      scene.lights_push_back(InfiniteAreaLight("x.hdr"))
//   scene->camera = CameraDesc{...}                camera.h:17       scene.camera = ...; (applied at commit)
//   Scene::Intersect / IntersectP                  scene.h:5-19      scene.Intersect(rays, n, hits) / IntersectP
//   PathTracer(maxDepth).Li per pixel + Accumulator::AddSample       agpt::PathTracer(maxDepth).Render(scene, accum, spp)
//   Accumulator (sum buffer, y flip, sample count) myapp.h:8-68      agpt::Accumulator
//   RotatingCamera(desc).update(angle)             camera.h:109-162  agpt::RotatingCamera(desc).update(dx, dy) -> CameraDesc
//   DbgIntegrator().Li(ray, scene)                 integrator.h:107  agpt::DbgIntegrator().Li(scene, ray)
//
// Errors: the reference has no error returns (bool hit/miss, exit() on load failure); here every failing C call
// throws agpt::Error carrying agpt_last_error().
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "agpt.h"

namespace agpt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what + ": " + agpt_last_error()), code(c) {}
};
inline int check(int rc, const char* what) {
    if (rc < 0) throw Error(rc, what);
    return rc;
}

struct float3 {
    float x, y, z;
};

class Context {
public:
    explicit Context(int device = 0, void* hip_stream = nullptr) {
        check(agpt_init(device, &h_), "agpt_init");
        if (hip_stream) check(agpt_set_stream(h_, hip_stream), "agpt_set_stream");
    }
    ~Context() { agpt_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    agpt_ctx* handle() const { return h_; }

private:
    agpt_ctx* h_ = nullptr;
};

// TriangleMesh (trianglemesh.h:14-57): AoS arrays + (v, n, t) index triplets
struct TriangleMesh {
    std::vector<float> vertices, normals, texcoords;
    std::vector<int32_t> indices;

    // TriangleMesh::LoadObj(inputfile, mat, transform, ignore_normals) (trianglemesh.cpp:157-230); transform16 is a
    // row-major mat4 or nullptr for identity.  Throws agpt::Error where the reference calls exit(1).
    static TriangleMesh LoadObj(const std::string& inputfile, const float* transform16 = nullptr, bool ignore_normals = false) {
        agpt_obj_mesh* h = nullptr;
        if (agpt_obj_load(inputfile.c_str(), transform16, ignore_normals ? 1 : 0, &h) < 0)
            throw std::runtime_error(std::string("agpt_obj_load: ") + agpt_obj_last_error());
        TriangleMesh m;
        int nv = 0, nn = 0, nt = 0, ni = 0;
        agpt_obj_counts(h, &nv, &nn, &nt, &ni);
        m.vertices.resize(3 * (size_t)nv);
        m.normals.resize(3 * (size_t)nn);
        m.texcoords.resize(2 * (size_t)nt);
        m.indices.resize(3 * (size_t)ni);
        agpt_obj_get(h, m.vertices.data(), m.normals.data(), m.texcoords.data(), m.indices.data());
        agpt_obj_free(h);
        return m;
    }

    static TriangleMesh CreateBackdrop(float3 origin, float3 size, float radius, int steps) {
        TriangleMesh m;
        const int nv = 2 * (steps + 5), ni = 6 * (steps + 4);
        m.vertices.resize(3 * nv);
        m.normals.resize(3 * nv);
        m.texcoords.resize(2 * nv);
        m.indices.resize(3 * ni);
        const float o[3] = {origin.x, origin.y, origin.z}, s[3] = {size.x, size.y, size.z};
        int n_v = 0, n_i = 0;
        check(agpt_create_backdrop(o, s, radius, steps, m.vertices.data(), m.normals.data(), m.texcoords.data(),
                                   m.indices.data(), &n_v, &n_i),
              "agpt_create_backdrop");
        return m;
    }
};

struct Sphere {
    float3 center;
    float radius;
};
// Plane (intersectable.h:119-157): XZ rectangle with +Y normal, size = full extents
struct Plane {
    float3 o;
    float size_x, size_z;
};
struct UniformInfiniteLight {
    float3 L;
};
// InfiniteAreaLight (lights.h:53-70): the HDR environment map as width*height RGB floats (what stbi_loadf returns)
// Built from a Radiance .hdr file as in the reference (myapp.cpp:113: InfiniteAreaLight("small_workshop_1k.hdr"); HDRTexture's
// stbi_loadf, texture.h:41-52 = agpt_hdr_load) or from pixels already in memory.
struct InfiniteAreaLight {
    const float* rgb = nullptr;
    int width = 0, height = 0;
    InfiniteAreaLight(const float* pixels, int w, int h) : rgb(pixels), width(w), height(h) {}
    explicit InfiniteAreaLight(const std::string& filename) {
        float* p = nullptr;
        check(agpt_hdr_load(filename.c_str(), &width, &height, &p), "agpt_hdr_load");
        owned_.reset(p, agpt_hdr_free);
        rgb = p;
    }

private:
    std::shared_ptr<float> owned_;
};
using CameraDesc = agpt_camera_desc;

class Scene {
public:
    explicit Scene(Context& ctx) : ctx_(ctx) {
        check(agpt_scene_create(ctx.handle(), &h_), "agpt_scene_create");
        camera = CameraDesc{{0, 0, 1}, {0, 0, 0}, {0, 1, 0}, 1.f, 45.f, 0.f};
    }
    ~Scene() { agpt_scene_destroy(h_); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    int add_material(int type, float3 color, float roughness, float metallic) {
        const float c[3] = {color.x, color.y, color.z};
        return check(agpt_scene_add_material(h_, type, c, roughness, metallic), "agpt_scene_add_material");
    }
    // scene->primitives.push_back(make_shared<BVHTriMesh>(mesh, material, maxPrimsInNode))
    int primitives_push_back(const TriangleMesh& m, int material, int maxPrimsInNode = 1) {
        return check(agpt_scene_add_mesh(h_, m.vertices.data(), (int)m.vertices.size() / 3, m.normals.data(),
                                         (int)m.normals.size() / 3, m.texcoords.data(), (int)m.texcoords.size() / 2,
                                         m.indices.data(), (int)m.indices.size() / 3, material, maxPrimsInNode),
                     "agpt_scene_add_mesh");
    }
    // scene->primitives.push_back(make_shared<Sphere>(center, radius, material)); material -1 = nullptr
    int primitives_push_back(const Sphere& s, int material) {
        const float c[3] = {s.center.x, s.center.y, s.center.z};
        return check(agpt_scene_add_sphere(h_, c, s.radius, material), "agpt_scene_add_sphere");
    }
    int primitives_push_back(const Plane& p, int material) {
        const float o[3] = {p.o.x, p.o.y, p.o.z}, sz[2] = {p.size_x, p.size_z};
        return check(agpt_scene_add_plane(h_, o, sz, material), "agpt_scene_add_plane");
    }
    int addAreaLight(const Sphere& s, float3 L) {
        const float c[3] = {s.center.x, s.center.y, s.center.z}, l[3] = {L.x, L.y, L.z};
        return check(agpt_scene_add_area_light(h_, c, s.radius, l), "agpt_scene_add_area_light");
    }
    int lights_push_back(const UniformInfiniteLight& l) {
        const float c[3] = {l.L.x, l.L.y, l.L.z};
        return check(agpt_scene_add_uniform_infinite_light(h_, c), "agpt_scene_add_uniform_infinite_light");
    }
    int lights_push_back(const InfiniteAreaLight& l) {
        return check(agpt_scene_add_infinite_area_light(h_, l.rgb, l.width, l.height), "agpt_scene_add_infinite_area_light");
    }
    // upload to HBM; call once after the scene is built (and again after changing `camera`)
    void commit() {
        check(agpt_scene_set_camera(h_, &camera), "agpt_scene_set_camera");
        check(agpt_scene_commit(h_), "agpt_scene_commit");
    }
    // camera moved (RotatingCamera::update): re-derive the camera only, geometry stays in HBM
    void set_camera() { check(agpt_scene_set_camera(h_, &camera), "agpt_scene_set_camera"); }
    void Intersect(const agpt_ray* rays, int n, agpt_hit* hits) const {
        check(agpt_intersect_batch(h_, rays, n, hits, 0, nullptr), "agpt_intersect_batch");
    }
    void IntersectP(const agpt_ray* rays, int n, agpt_hit* hits) const {
        check(agpt_intersect_batch(h_, rays, n, hits, 1, nullptr), "agpt_intersect_batch");
    }
    // single-ray convenience for picking / debugging (myapp.cpp:200-201)
    bool Intersect(const agpt_ray& ray, agpt_hit& hit) const {
        Intersect(&ray, 1, &hit);
        return hit.hit != 0;
    }

    CameraDesc camera;
    agpt_scene* handle() const { return h_; }
    Context& context() const { return ctx_; }

private:
    Context& ctx_;
    agpt_scene* h_ = nullptr;
};

// RotatingCamera (camera.h:109-162): orbit around lookat at a fixed distance.  The elevation angle is clamped to
// [-(pi/2 - 1e-4), 0] and lookfrom = RotateY(yAngle) * RotateX(xAngle) * (0,0,1) * dist + lookat.  The host applies the
// result with `scene.camera = cam.update(dx, dy); scene.set_camera(); accumulator.Clear();` (myapp.cpp:143-149).
class RotatingCamera {
public:
    explicit RotatingCamera(const CameraDesc& d) : desc_(d) {
        const float fx = d.lookfrom[0] - d.lookat[0], fy = d.lookfrom[1] - d.lookat[1], fz = d.lookfrom[2] - d.lookat[2];
        dist_ = std::sqrt(fx * fx + fy * fy + fz * fz);
        const float inv = 1.0f / dist_;
        const float nx = fx * inv, ny = fy * inv, nz = fz * inv;
        const float horiz = std::sqrt(nx * nx + nz * nz);
        xAngle_ = std::acos(horiz);
        if (ny > 0) xAngle_ = -xAngle_;
        yAngle_ = std::acos(nz / horiz);
        if (nx < 0) yAngle_ = -yAngle_;
    }
    const CameraDesc& update(float dxAngle, float dyAngle) {
        const float lim = 3.14159265358979323846f / 2 - 0.0001f;
        xAngle_ += dxAngle;
        xAngle_ = xAngle_ < -lim ? -lim : (xAngle_ > 0.f ? 0.f : xAngle_);
        yAngle_ += dyAngle;
        const float cx = std::cos(xAngle_), sx = std::sin(xAngle_), cy = std::cos(yAngle_), sy = std::sin(yAngle_);
        // third column of RotateY * RotateX (row-major mat4, template/precomp.h:875-876)
        desc_.lookfrom[0] = sy * cx * dist_ + desc_.lookat[0];
        desc_.lookfrom[1] = -sx * dist_ + desc_.lookat[1];
        desc_.lookfrom[2] = cy * cx * dist_ + desc_.lookat[2];
        return desc_;
    }
    const CameraDesc& desc() const { return desc_; }
    float xAngle() const { return xAngle_; }
    float yAngle() const { return yAngle_; }

private:
    CameraDesc desc_;
    float dist_, xAngle_, yAngle_;
};

struct DisneyMaterial {
    static int Make(Scene& s, float3 color, float roughness, float metallic) {
        return s.add_material(AGPT_MAT_DISNEY, color, roughness, metallic);
    }
};
struct MirrorMaterial {
    static int Make(Scene& s, float3 r) { return s.add_material(AGPT_MAT_MIRROR, r, 0.f, 0.f); }
};

// Accumulator (myapp.h:8-68): float sum buffer in HBM (float4 per pixel, row H-1-y), sample count, clear, resolve
class Accumulator {
public:
    Accumulator(Context& ctx, int w, int h) : width(w), height(h), ctx_(ctx) {
        check(agpt_device_alloc(ctx.handle(), (size_t)w * h * 16, &pixels_), "agpt_device_alloc");
        Clear();
    }
    ~Accumulator() { agpt_device_free(ctx_.handle(), pixels_); }
    Accumulator(const Accumulator&) = delete;
    Accumulator& operator=(const Accumulator&) = delete;
    void Clear() {
        check(agpt_device_memset(ctx_.handle(), pixels_, 0, (size_t)width * height * 16), "agpt_device_memset");
        samples_ = 0;
    }
    int NumSamples() const { return samples_; }
    void AddSamples(int n) { samples_ += n; }
    void SetSamples(int n) { samples_ = n; }
    float* device_pixels() const { return static_cast<float*>(pixels_); }
    // CopyToSurface: gamma 2.2 + 0x00RRGGBB (myapp.h:34-41)
    std::vector<uint32_t> CopyToSurface() const {
        std::vector<uint32_t> out((size_t)width * height);
        check(agpt_resolve(ctx_.handle(), device_pixels(), width * height, samples_, out.data()), "agpt_resolve");
        return out;
    }
    std::vector<float> Download() const {
        std::vector<float> out((size_t)width * height * 4);
        check(agpt_device_download(ctx_.handle(), out.data(), pixels_, out.size() * 4), "agpt_device_download");
        return out;
    }
    const int width, height;

private:
    Context& ctx_;
    void* pixels_ = nullptr;
    int samples_ = 0;
};

// Integrator / PathTracer (integrator.h:28-31,120-196).  Li is evaluated for whole frames: one Render() call is
// `spp` iterations of MyApp::Tick's per-pixel loop (myapp.cpp:161-175).
// One rank's share of a film split over `world` GPUs: the 8-row blocks k with k % world == rank.
// The rank's Accumulator then holds only its own rows, compacted (Accumulator(ctx, width, RowsOfRank(height, ...))).
struct RankShare {
    int block_rows = 8, world = 1, rank = 0;
    // rows of a `film_height`-row film that belong to this rank
    int Rows(int film_height) const {
        int rows = 0;
        for (int y0 = 0, k = 0; y0 < film_height; y0 += block_rows, k++)
            if (k % world == rank) rows += (film_height - y0 < block_rows) ? film_height - y0 : block_rows;
        return rows;
    }
};

// The RCCL communicator of a multi-GPU render (one process and one Context per GPU).  Rank 0 obtains a 128-byte id with
// Comm::UniqueId() and passes it to the other ranks by the host application's own means (MPI_Bcast, a file, ...); every rank
// then constructs Comm(ctx, id, world, rank) collectively.  GatherTiles brings every rank's compact Accumulator to rank 0's
// full-film Accumulator (agpt_gather_tiles: grouped send/recv over xGMI + de-interleave), once per displayed frame.
class Comm {
public:
    static std::vector<unsigned char> UniqueId() {
        std::vector<unsigned char> id(128);
        check(agpt_comm_unique_id(id.data()), "agpt_comm_unique_id");
        return id;
    }
    Comm(Context& ctx, const std::vector<unsigned char>& id, int world, int rank) {
        check(agpt_comm_init(ctx.handle(), world > 1 ? id.data() : nullptr, world, rank, &h_), "agpt_comm_init");
    }
    ~Comm() { agpt_comm_destroy(h_); }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    // local: this rank's compact accumulator; full: the whole film on rank 0 (ignored, may be null, elsewhere)
    void GatherTiles(const Accumulator& local, int film_height, const RankShare& share, Accumulator* full) {
        check(agpt_gather_tiles(h_, local.device_pixels(), local.width, film_height, share.block_rows, full ? full->device_pixels() : nullptr),
              "agpt_gather_tiles");
        if (full) full->SetSamples(local.NumSamples());
    }

private:
    agpt_comm* h_ = nullptr;
};

class PathTracer {
public:
    explicit PathTracer(int maxDepth = 5) : MaxDepth(maxDepth) {}
    // multi-GPU: `acc` is this rank's compact accumulator (width x share.Rows(film_height)); seeds depend on the global
    // pixel index, so the gathered film is bit-identical to a single-GPU render
    agpt_stats Render(Scene& scene, Accumulator& acc, int film_height, const RankShare& share, int spp, uint32_t seed_base = 0) const {
        agpt_render_params p{};
        p.width = acc.width;
        p.height = film_height;
        p.w = acc.width;
        p.h = film_height;
        p.spp_begin = acc.NumSamples();
        p.spp_count = spp;
        p.seed_base = seed_base;
        p.max_depth = MaxDepth;
        p.accum_pitch = acc.width;
        p.interleave_block = share.block_rows;
        p.interleave_world = share.world;
        p.interleave_rank = share.rank;
        agpt_stats st{};
        check(agpt_render(scene.handle(), &p, acc.device_pixels(), &st), "agpt_render");
        acc.AddSamples(spp);
        return st;
    }
    agpt_stats Render(Scene& scene, Accumulator& acc, int spp, uint32_t seed_base = 0) const {
        agpt_render_params p{};
        p.width = acc.width;
        p.height = acc.height;
        p.x0 = 0;
        p.y0 = 0;
        p.w = acc.width;
        p.h = acc.height;
        p.spp_begin = acc.NumSamples();
        p.spp_count = spp;
        p.seed_base = seed_base;
        p.max_depth = MaxDepth;
        p.accum_pitch = acc.width;
        agpt_stats st{};
        check(agpt_render(scene.handle(), &p, acc.device_pixels(), &st), "agpt_render");
        acc.AddSamples(spp);
        return st;
    }
    // Integrator::Li(const Ray&, const Scene&) (integrator.h:28-31) for n rays -- what MyApp::Tick calls per pixel (myapp.cpp:168)
    // and the mouse-pick overlay per click (myapp.cpp:197-201).  rng[i] is the RandomFloat() state path i starts from and receives
    // the state it ends with (the reference keeps ONE such state, template/template.cpp:667: pass it for a single ray and store it
    // back to continue its stream).  radiance[i] is Li's return value, unfiltered.
    void Li(Scene& scene, const agpt_ray* rays, uint32_t* rng, int n, float3* radiance) const {
        std::vector<float> out(3 * (size_t)n);
        check(agpt_li_batch(scene.handle(), rays, rng, n, MaxDepth, out.data(), rng, nullptr), "agpt_li_batch");
        for (int i = 0; i < n; i++) radiance[i] = float3{out[3 * i], out[3 * i + 1], out[3 * i + 2]};
    }
    float3 Li(Scene& scene, const agpt_ray& ray, uint32_t& rng) const {
        float3 L{0.f, 0.f, 0.f};
        Li(scene, &ray, &rng, 1, &L);
        return L;
    }
    int MaxDepth;
};

// DbgIntegrator (integrator.h:107-118), the uv view MyApp can put on the left half of the split screen (myapp.cpp:129-130)
class DbgIntegrator {
public:
    void Li(Scene& scene, const agpt_ray* rays, int n, float3* radiance) const {
        std::vector<float> out(3 * (size_t)n);
        check(agpt_dbg_li_batch(scene.handle(), rays, n, out.data()), "agpt_dbg_li_batch");
        for (int i = 0; i < n; i++) radiance[i] = float3{out[3 * i], out[3 * i + 1], out[3 * i + 2]};
    }
    float3 Li(Scene& scene, const agpt_ray& ray) const {
        float3 L{0.f, 0.f, 0.f};
        Li(scene, &ray, 1, &L);
        return L;
    }
};

}  // namespace agpt
