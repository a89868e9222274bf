// agpt_api.hip -- the C ABI of include/agpt.h on top of the kernels in agpt_kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/agpt.h"
#include "agpt_host_scene.hpp"
#include "agpt_kernels.h"

// LDS stack entries of the production trace kernel (23 KiB of stack + 8 KiB = 31 KiB per block -> five blocks per CU) and the
// blocks per CU that go with it; deeper BVHs spill the entries beyond into agpt_ctx::spill (HBM)
#ifndef AGPT_FAST_STACK
#define AGPT_FAST_STACK 23
#endif
#ifndef AGPT_FAST_BLOCKS_PER_CU
#define AGPT_FAST_BLOCKS_PER_CU 5
#endif


namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

}  // namespace
namespace agpt {
// error reporting for the library's other translation units (agpt_image.cpp)
int report_error(int code, const std::string& msg) { return fail(code, msg); }
}  // namespace agpt
namespace {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(AGPT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n) return AGPT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) return fail(AGPT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        n = count;
        return AGPT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

struct agpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t aux_stream[2] = {nullptr, nullptr};   // small batches: the MIS and shadow launches of an iteration run beside the closest-hit one
    hipEvent_t aux_ev[3] = {nullptr, nullptr, nullptr};
    int multi_stream = 0;
    // path-state pool (sized for the largest batch seen)
    size_t pool_paths = 0;
    DevBuf<float4> ext_o, ext_d, sh_o, sh_d, mis_o, mis_d, beta4, L4, fac4, c1, c2;
    DevBuf<DevHit> hit, mis_hit;
    DevBuf<uint32_t> occluded, mis_ok;
    DevBuf<uint32_t> q[2][AGPT_NQUEUES];
    DevBuf<uint32_t> qcounts;   // 2 x AGPT_NQUEUES queue lengths
    DevBuf<uint32_t> work;      // work-queue heads of the persistent trace launches
    DevBuf<DevCounters> counters;
    uint32_t* host_pinned = nullptr;
    int blocks_per_cu = 4;       // generic trace kernel (32-entry LDS stacks)
    bool trace_recast = false;        // the closest-hit launch in flight is the wavefront loop's own (its rays carry d.w, see k_trace_fast)
    bool trace_small_batch = false;   // the rays of the current batch number fewer than AGPT_SMALL_BATCH: k_trace_fast<PEEK>
    int shade_blocks_per_cu = 8;   // AGPT_SHADE_BLOCKS_PER_CU: k_shade's grid (its waves take wave tiles from work heads)
    int fast_blocks_per_cu = AGPT_FAST_BLOCKS_PER_CU;  // AGPT_BLOCKS_PER_CU: production trace kernel (AGPT_FAST_STACK-entry LDS stacks)
    DevBuf<uint32_t> spill;      // traversal-stack entries beyond the LDS part (BVHs deeper than AGPT_FAST_STACK)
    // primitive lists longer than 64 entries: per-path candidate words written by k_candidates, read by k_trace_fast<LIST>
    DevBuf<unsigned long long> cand_mask;   // [chunk][pool_paths]
    DevBuf<uint32_t> cand_chunks;           // [pool_paths]
    int refill = AGPT_REFILL;    // AGPT_REFILL: idle lanes that trigger an in-flight refill (closest-hit launches)
    int refill_any = AGPT_REFILL_ANY;  // AGPT_REFILL_ANY: same for the any-hit / MIS-query launches
    bool mis_closest = false;    // AGPT_MIS_CLOSEST=1: trace MIS rays as full closest-hit queries (A/B, tests)
    bool force_generic = false;  // AGPT_FORCE_GENERIC=1: always use the generic k_trace (tests)
    // per-launch HIP-event timing of the trace kernels (agpt_render_params::enable_timing)
    std::vector<hipEvent_t> tev;
    size_t tev_used = 0;
    std::vector<int> tev_kind;  // 0 = closest (continuation), 1 = closest (MIS), 2 = any-hit
    // first error of an asynchronous helper (memset / event record / kernel launch inside the launch helpers); checked and
    // cleared by the entry point that enqueued the work
    hipError_t deferred = hipSuccess;
    void note(hipError_t e) {
        if (e != hipSuccess && deferred == hipSuccess) deferred = e;
    }
    hipError_t take_deferred() {
        const hipError_t e = deferred;
        deferred = hipSuccess;
        return e;
    }
};

struct agpt_scene {
    agpt_ctx* ctx = nullptr;
    std::vector<agpt::HostMesh> meshes;
    std::vector<agpt::HostSphere> spheres;
    std::vector<agpt::HostPrim> prims;
    std::vector<agpt::HostLight> lights;
    std::vector<agpt::HostEnv> envs;
    std::vector<DevMaterial> materials;
    DevCamera cam{};
    bool has_camera = false;
    bool committed = false;
    int max_depth = 0;
    DevBuf<float4> d_nodes, d_tri_verts, d_tri_shade, d_prefilter;
    DevBuf<uint32_t> d_toplevel;
    DevBuf<unsigned long long> d_chunk_mesh_masks;
    DevBuf<uint32_t> d_bigleaves;
    DevBuf<DevPrim> d_prims;
    DevBuf<DevMaterial> d_materials;
    DevBuf<DevLight> d_lights;
    DevBuf<DevEnv> d_envs;
    std::vector<DevBuf<float4>> d_env_pixels;
    std::vector<DevBuf<float>> d_env_func, d_env_cdf;
    DevScene dev{};
};

// scoped device allocation for the entry points that need a temporary: freed on every return path
template <class T>
struct TempBuf {
    T* p = nullptr;
    ~TempBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t count) { return hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)); }
};

template <class T>
static int upload(DevBuf<T>& buf, const std::vector<T>& host, hipStream_t st) {
    int rc = buf.ensure(host.empty() ? 1 : host.size());
    if (rc) return rc;
    if (!host.empty()) HIP_TRY(hipMemcpyAsync(buf.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return AGPT_OK;
}

static int trace_grid(const agpt_ctx* c) { return c->num_cus * c->blocks_per_cu; }           // generic kernel
static int fast_grid(const agpt_ctx* c) { return c->num_cus * c->fast_blocks_per_cu; }       // production kernel

// count: 0 = off, 1 = reference-order counters (the generic kernel: what the reference's recursion does, equal to the
// oracle's counters), 2 = the production kernel counting its own work (bench.py's roofline)
static bool use_fast_trace(const agpt_ctx* c, const DevScene& sc, int count) {
    return count != 1 && sc.n_prims <= 64 * AGPT_MAX_CHUNKS && !c->force_generic;
}

// LDS stack entries of the production kernel (23 KiB of stack + 8 KiB = 31 KiB per block -> five blocks per CU) and the
// blocks per CU that go with it; deeper BVHs spill the entries beyond into c->spill (HBM)
#define AGPT_SMALL_BATCH (48ull << 20)   // paths: below this the trace launches run the PEEK instantiation (see k_trace_fast)
template <int MODE, bool COUNT, bool SPILL, bool PEEK>
static void launch_trace_fast(agpt_ctx* c, const DevScene& sc, const uint32_t* queue, const uint32_t* count_ptr, uint32_t count_imm,
                              uint32_t* work_head, const float4* ro, const float4* rd, DevHit* hits, uint32_t* occ, int refill) {
    const dim3 block(AGPT_BLOCK), g(fast_grid(c));
    if (sc.n_prims <= 64) {
        // (closest-hit launches of the wavefront loop may re-cast a ray in place: see the retire branch of k_trace_fast)
        float4* ro_w = MODE == 0 && c->trace_recast ? const_cast<float4*>(ro) : nullptr;
        float4* rd_w = MODE == 0 && c->trace_recast ? const_cast<float4*>(rd) : nullptr;
        hipLaunchKernelGGL((k_trace_fast<MODE, AGPT_FAST_STACK, false, COUNT, SPILL, PEEK>), g, block, 0, c->stream, sc, queue, count_ptr,
                           count_imm, work_head, ro, rd, hits, occ, c->counters.p, refill, 0u, c->spill.p,
                           (const unsigned long long*)nullptr, (const uint32_t*)nullptr, ro_w, rd_w, (const float4*)c->beta4.p, c->L4.p);
        return;
    }
    // more than 64 primitives: the top-level tree gives every ray its candidates (one word per chunk of 64 primitives), then
    // ONE traversal launch walks them in list order
    const uint32_t stride = (uint32_t)c->pool_paths;
    if (c->cand_chunks.n < c->pool_paths || c->cand_mask.n < (size_t)((sc.n_prims + 63) / 64) * c->pool_paths) {
        c->note(hipErrorOutOfMemory);   // (ensure_pool sizes both for the scene: not reached)
        return;
    }
    hipLaunchKernelGGL((k_candidates<MODE>), dim3(c->num_cus * 8), block, 0, c->stream, sc, queue, count_ptr, count_imm, ro, rd,
                       c->cand_mask.p, c->cand_chunks.p, stride);
    hipLaunchKernelGGL((k_trace_fast<MODE, AGPT_FAST_STACK, true, COUNT, SPILL, PEEK>), g, block, 0, c->stream, sc, queue, count_ptr,
                       count_imm, work_head, ro, rd, hits, occ, c->counters.p, refill, stride, c->spill.p,
                       (const unsigned long long*)c->cand_mask.p, (const uint32_t*)c->cand_chunks.p, (float4*)nullptr, (float4*)nullptr,
                       (const float4*)nullptr, (float4*)nullptr);
}

template <int MODE>
static void launch_trace_fast_c(agpt_ctx* c, int count, const DevScene& sc, const uint32_t* queue, const uint32_t* count_ptr,
                                uint32_t count_imm, uint32_t* work_head, const float4* ro, const float4* rd, DevHit* hits,
                                uint32_t* occ, int refill) {
    const bool spill = sc.max_depth > AGPT_FAST_STACK;
    if (spill) {
        // one column of (max_depth - AGPT_FAST_STACK) entries per thread of the grid
        const size_t need = (size_t)(sc.max_depth - AGPT_FAST_STACK) * (size_t)fast_grid(c) * AGPT_BLOCK;
        if (c->spill.ensure(need) != AGPT_OK) {
            c->note(hipErrorOutOfMemory);
            return;
        }
    }
    if (count) {
        if (spill) launch_trace_fast<MODE, true, true, false>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
        else launch_trace_fast<MODE, true, false, false>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
    } else if (c->trace_small_batch) {   // (see PEEK in k_trace_fast)
        if (spill) launch_trace_fast<MODE, false, true, true>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
        else launch_trace_fast<MODE, false, false, true>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
    } else {
        if (spill) launch_trace_fast<MODE, false, true, false>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
        else launch_trace_fast<MODE, false, false, false>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
    }
}

// MODE 0 closest, 1 any-hit, 2 MIS query (production kernel only; the generic kernel traces MIS rays as closest hits)
template <int MODE>
static void launch_trace(agpt_ctx* c, const DevScene& sc, const uint32_t* queue, const uint32_t* count_ptr, uint32_t count_imm,
                         uint32_t* work_head, const float4* ro, const float4* rd, DevHit* hits, uint32_t* occ, int count) {
    constexpr bool ANY = MODE == 1;
    dim3 grid(trace_grid(c)), block(AGPT_BLOCK);
    if (use_fast_trace(c, sc, count)) {
        const int refill = MODE == 0 ? c->refill : c->refill_any;
        launch_trace_fast_c<MODE>(c, count, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, refill);
    } else if (sc.max_depth > AGPT_STACK_DEPTH) {
        dim3 grid_deep(c->num_cus * 2);
        if (count)
            hipLaunchKernelGGL((k_trace<ANY, true, AGPT_STACK_DEPTH_MAX>), grid_deep, block, 0, c->stream, sc, queue, count_ptr,
                               count_imm, work_head, ro, rd, hits, occ, c->counters.p);
        else
            hipLaunchKernelGGL((k_trace<ANY, false, AGPT_STACK_DEPTH_MAX>), grid_deep, block, 0, c->stream, sc, queue, count_ptr,
                               count_imm, work_head, ro, rd, hits, occ, c->counters.p);
    } else if (count)
        hipLaunchKernelGGL((k_trace<ANY, true, AGPT_STACK_DEPTH>), grid, block, 0, c->stream, sc, queue, count_ptr, count_imm,
                           work_head, ro, rd, hits, occ, c->counters.p);
    else
        hipLaunchKernelGGL((k_trace<ANY, false, AGPT_STACK_DEPTH>), grid, block, 0, c->stream, sc, queue, count_ptr, count_imm,
                           work_head, ro, rd, hits, occ, c->counters.p);
}

static void release_pool(agpt_ctx* c) {
    for (DevBuf<float4>* b : {&c->ext_o, &c->ext_d, &c->sh_o, &c->sh_d, &c->mis_o, &c->mis_d, &c->beta4, &c->L4, &c->fac4, &c->c1,
                              &c->c2})
        b->release();
    c->hit.release();
    c->mis_hit.release();
    c->occluded.release();
    c->mis_ok.release();
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < AGPT_NQUEUES; j++) c->q[i][j].release();
    c->cand_mask.release();
    c->cand_chunks.release();
    c->pool_paths = 0;
}
static hipEvent_t next_event(agpt_ctx* c) {
    if (c->tev_used == c->tev.size()) {
        hipEvent_t e = nullptr;
        c->note(hipEventCreate(&e));
        c->tev.push_back(e);
    }
    return c->tev[c->tev_used++];
}

template <int MODE>
static void launch_trace_timed(agpt_ctx* c, bool timing, int kind, const DevScene& sc, const uint32_t* queue,
                               const uint32_t* count_ptr, uint32_t count_imm, uint32_t* work_head, const float4* ro,
                               const float4* rd, DevHit* hits, uint32_t* occ, int count) {
    if (timing) {
        c->note(hipEventRecord(next_event(c), c->stream));
        c->tev_kind.push_back(kind);
    }
    launch_trace<MODE>(c, sc, queue, count_ptr, count_imm, work_head, ro, rd, hits, occ, count);
    c->note(hipGetLastError());
    if (timing) c->note(hipEventRecord(next_event(c), c->stream));
}

extern "C" {

const char* agpt_last_error(void) { return g_error.c_str(); }
int agpt_version(void) { return 1; }

int agpt_init(int device, agpt_ctx** out) {
    if (!out) return fail(AGPT_ERR_INVALID, "agpt_init: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(AGPT_ERR_DEVICE, "agpt_init: no HIP device (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= n) return fail(AGPT_ERR_INVALID, "agpt_init: bad device index");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<agpt_ctx> c(new agpt_ctx());
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (auto& ev : c->ev) HIP_TRY(hipEventCreate(&ev));
    for (auto& st : c->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto& ev : c->aux_ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (const char* ms = getenv("AGPT_MULTI_STREAM")) c->multi_stream = atoi(ms);
    HIP_TRY(hipHostMalloc((void**)&c->host_pinned, 2 * AGPT_NQUEUES * AGPT_QSTRIDE * sizeof(uint32_t)));
    const char* fg = getenv("AGPT_FORCE_GENERIC");
    c->force_generic = fg && fg[0] == '1';
    if (const char* mc = getenv("AGPT_MIS_CLOSEST")) c->mis_closest = mc[0] == '1';
    if (const char* bp = getenv("AGPT_BLOCKS_PER_CU")) c->fast_blocks_per_cu = std::max(1, atoi(bp));
    if (const char* sb = getenv("AGPT_SHADE_BLOCKS_PER_CU")) c->shade_blocks_per_cu = std::min(64, std::max(1, atoi(sb)));
    if (const char* rf = getenv("AGPT_REFILL")) c->refill = std::min(64, std::max(1, atoi(rf)));
    if (const char* rf = getenv("AGPT_REFILL_ANY")) c->refill_any = std::min(64, std::max(1, atoi(rf)));
    *out = c.release();
    return AGPT_OK;
}

int agpt_set_stream(agpt_ctx* c, void* s) {
    if (!c) return fail(AGPT_ERR_INVALID, "agpt_set_stream: ctx is NULL");
    c->stream = (hipStream_t)s;
    return AGPT_OK;
}

void agpt_destroy(agpt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    release_pool(c);   // every per-path buffer (one list, shared with the batch-halving path of agpt_render)
    c->qcounts.release();
    c->spill.release();
    c->work.release();
    c->counters.release();
    if (c->host_pinned) (void)hipHostFree(c->host_pinned);
    for (auto& ev : c->tev) (void)hipEventDestroy(ev);
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->aux_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& st : c->aux_stream)
        if (st) (void)hipStreamDestroy(st);
    delete c;
}

// ---- scene building -----------------------------------------------------------------------------------------
int agpt_scene_create(agpt_ctx* c, agpt_scene** out) {
    if (!c || !out) return fail(AGPT_ERR_INVALID, "agpt_scene_create: NULL argument");
    agpt_scene* s = new agpt_scene();
    s->ctx = c;
    *out = s;
    return AGPT_OK;
}

void agpt_scene_destroy(agpt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipDeviceSynchronize();
    s->d_nodes.release();
    s->d_bigleaves.release();
    s->d_tri_verts.release();
    s->d_tri_shade.release();
    s->d_prefilter.release();
    s->d_toplevel.release();
    s->d_chunk_mesh_masks.release();
    s->d_prims.release();
    s->d_materials.release();
    s->d_lights.release();
    s->d_envs.release();
    for (auto& b : s->d_env_pixels) b.release();
    for (auto& b : s->d_env_func) b.release();
    for (auto& b : s->d_env_cdf) b.release();
    delete s;
}

int agpt_scene_add_material(agpt_scene* s, int type, const float color[3], float roughness, float metallic) {
    if (!s || !color) return fail(AGPT_ERR_INVALID, "agpt_scene_add_material: NULL argument");
    if (type < AGPT_MAT_DISNEY || type > AGPT_MAT_DIFFUSE_ONLY) return fail(AGPT_ERR_INVALID, "unknown material type");
    s->materials.push_back(agpt::make_material(type, color, roughness, metallic));
    s->committed = false;
    return (int)s->materials.size() - 1;
}

int agpt_scene_add_mesh(agpt_scene* s, const float* vertices, int n_vertices, const float* normals, int n_normals,
                        const float* texcoords, int n_texcoords, const int32_t* indices, int n_indices, int material,
                        int max_prims_in_node) {
    if (!s || !vertices || !indices) return fail(AGPT_ERR_INVALID, "agpt_scene_add_mesh: NULL argument");
    if (n_indices < 3 || n_indices % 3 != 0 || n_vertices <= 0)
        return fail(AGPT_ERR_INVALID, "agpt_scene_add_mesh: need at least one triangle (the reference's BVH build does not terminate on an empty mesh)");
    if (material < -1 || material >= (int)s->materials.size()) return fail(AGPT_ERR_INVALID, "agpt_scene_add_mesh: bad material id");
    for (int i = 0; i < n_indices; i++) {
        const int32_t* ix = indices + 3 * i;
        if (ix[0] < 0 || ix[0] >= n_vertices) return fail(AGPT_ERR_INVALID, "agpt_scene_add_mesh: vertex index out of range");
        if (n_normals > 0 && (ix[1] < 0 || ix[1] >= n_normals)) return fail(AGPT_ERR_INVALID, "agpt_scene_add_mesh: normal index out of range");
        if (n_texcoords > 0 && (ix[2] < 0 || ix[2] >= n_texcoords)) return fail(AGPT_ERR_INVALID, "agpt_scene_add_mesh: texcoord index out of range");
    }
    agpt::HostMesh m;
    m.vertices.resize(n_vertices);
    for (int i = 0; i < n_vertices; i++) m.vertices[i] = V3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
    if (normals && n_normals > 0) {
        m.normals.resize(n_normals);
        for (int i = 0; i < n_normals; i++) m.normals[i] = V3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    }
    if (texcoords && n_texcoords > 0) {
        m.texcoords.resize(n_texcoords);
        for (int i = 0; i < n_texcoords; i++) {
            m.texcoords[i].x = texcoords[2 * i];
            m.texcoords[i].y = texcoords[2 * i + 1];
        }
    }
    m.indices.assign(indices, indices + (size_t)3 * n_indices);
    m.material = material;
    agpt::build_bvh(m, max_prims_in_node);
    s->meshes.push_back(std::move(m));
    agpt::HostPrim p;
    p.type = AGPT_PRIM_MESH;
    p.index = (int)s->meshes.size() - 1;
    p.material = material;
    p.arealight = -1;
    s->prims.push_back(p);
    s->committed = false;
    return (int)s->prims.size() - 1;
}

int agpt_scene_add_sphere(agpt_scene* s, const float center[3], float radius, int material) {
    if (!s || !center) return fail(AGPT_ERR_INVALID, "agpt_scene_add_sphere: NULL argument");
    if (material < -1 || material >= (int)s->materials.size()) return fail(AGPT_ERR_INVALID, "agpt_scene_add_sphere: bad material id");
    agpt::HostSphere sp;
    sp.center = V3(center[0], center[1], center[2]);
    sp.r = radius;
    sp.r2 = radius * radius;
    s->spheres.push_back(sp);
    agpt::HostPrim p;
    p.type = AGPT_PRIM_SPHERE;
    p.index = (int)s->spheres.size() - 1;
    p.material = material;
    p.arealight = -1;
    s->prims.push_back(p);
    s->committed = false;
    return (int)s->prims.size() - 1;
}

int agpt_scene_add_plane(agpt_scene* s, const float o[3], const float size[2], int material) {
    if (!s || !o || !size) return fail(AGPT_ERR_INVALID, "agpt_scene_add_plane: NULL argument");
    if (material < -1 || material >= (int)s->materials.size()) return fail(AGPT_ERR_INVALID, "agpt_scene_add_plane: bad material id");
    agpt::HostSphere sp;  // reused record: center = O, r = HalfSize.x, r2 = HalfSize.y
    sp.center = V3(o[0], o[1], o[2]);
    sp.r = size[0] / 2;
    sp.r2 = size[1] / 2;
    s->spheres.push_back(sp);
    agpt::HostPrim p;
    p.type = AGPT_PRIM_PLANE;
    p.index = (int)s->spheres.size() - 1;
    p.material = material;
    p.arealight = -1;
    s->prims.push_back(p);
    s->committed = false;
    return (int)s->prims.size() - 1;
}

int agpt_scene_add_area_light(agpt_scene* s, const float center[3], float radius, const float L[3]) {
    if (!s || !center || !L) return fail(AGPT_ERR_INVALID, "agpt_scene_add_area_light: NULL argument");
    int prim = agpt_scene_add_sphere(s, center, radius, -1);
    if (prim < 0) return prim;
    agpt::HostLight l;
    l.type = AGPT_LIGHT_AREA;
    l.shape = prim;
    l.L = V3(L[0], L[1], L[2]);
    s->lights.push_back(l);
    s->prims[prim].arealight = (int)s->lights.size() - 1;
    return prim;
}

int agpt_scene_add_uniform_infinite_light(agpt_scene* s, const float L[3]) {
    if (!s || !L) return fail(AGPT_ERR_INVALID, "agpt_scene_add_uniform_infinite_light: NULL argument");
    agpt::HostLight l;
    l.type = AGPT_LIGHT_UNIFORM_INFINITE;
    l.shape = -1;
    l.L = V3(L[0], L[1], L[2]);
    s->lights.push_back(l);
    s->committed = false;
    return (int)s->lights.size() - 1;
}

int agpt_scene_add_infinite_area_light(agpt_scene* s, const float* rgb, int width, int height) {
    if (!s || !rgb || width <= 0 || height <= 0 || (long long)width * height > (1ll << 28))
        return fail(AGPT_ERR_INVALID, "agpt_scene_add_infinite_area_light: bad argument");
    s->envs.push_back(agpt::make_env(rgb, width, height));
    agpt::HostLight l;
    l.type = AGPT_LIGHT_INFINITE_AREA;
    l.shape = -1;
    l.L = V3s(0.f);
    l.env = (int)s->envs.size() - 1;
    s->lights.push_back(l);
    s->committed = false;
    return (int)s->lights.size() - 1;
}

int agpt_scene_set_camera(agpt_scene* s, const agpt_camera_desc* d) {
    if (!s || !d) return fail(AGPT_ERR_INVALID, "agpt_scene_set_camera: NULL argument");
    s->cam = agpt::make_camera(*d);
    s->has_camera = true;
    s->dev.cam = s->cam;
    return AGPT_OK;
}

int agpt_scene_commit(agpt_scene* s) {
    if (!s) return fail(AGPT_ERR_INVALID, "agpt_scene_commit: scene is NULL");
    HIP_TRY(hipSetDevice(s->ctx->device));
    agpt::FlatScene flat;
    agpt::flatten_scene(s->meshes, s->spheres, s->prims, flat);
    s->max_depth = flat.max_depth;
    if (flat.max_depth > AGPT_STACK_DEPTH_MAX)
        return fail(AGPT_ERR_LIMIT, "agpt_scene_commit: BVH depth " + std::to_string(flat.max_depth) +
                                        " exceeds the deepest traversal stack (" + std::to_string(AGPT_STACK_DEPTH_MAX) + ")");
    std::vector<DevLight> lights(s->lights.size());
    int n_inf = 0;
    for (size_t i = 0; i < lights.size(); i++) {
        lights[i].type = s->lights[i].type;
        lights[i].shape = s->lights[i].shape;
        lights[i].L[0] = s->lights[i].L.x;
        lights[i].L[1] = s->lights[i].L.y;
        lights[i].L[2] = s->lights[i].L.z;
        lights[i].env = s->lights[i].env;
        if (lights[i].type != AGPT_LIGHT_AREA) n_inf++;
    }
    hipStream_t st = s->ctx->stream;
    int rc;
    if ((rc = upload(s->d_nodes, flat.nodes, st))) return rc;
    if (flat.bigleaves.empty()) flat.bigleaves.assign(2, 0u);
    if ((rc = upload(s->d_bigleaves, flat.bigleaves, st))) return rc;
    if ((rc = upload(s->d_tri_verts, flat.tri_verts, st))) return rc;
    if ((rc = upload(s->d_tri_shade, flat.tri_shade, st))) return rc;
    if ((rc = upload(s->d_prefilter, flat.prefilter, st))) return rc;
    if ((rc = upload(s->d_toplevel, flat.toplevel16, st))) return rc;
    {
        std::vector<unsigned long long> mm(flat.mesh_masks, flat.mesh_masks + AGPT_MAX_CHUNKS);
        if ((rc = upload(s->d_chunk_mesh_masks, mm, st))) return rc;
    }
    if ((rc = upload(s->d_prims, flat.prims, st))) return rc;
    if ((rc = upload(s->d_materials, s->materials, st))) return rc;
    if ((rc = upload(s->d_lights, lights, st))) return rc;
    std::vector<DevEnv> envs(s->envs.size());
    s->d_env_pixels.resize(envs.size());
    s->d_env_func.resize(envs.size());
    s->d_env_cdf.resize(envs.size());
    for (size_t i = 0; i < envs.size(); i++) {
        const agpt::HostEnv& he = s->envs[i];
        if ((rc = upload(s->d_env_pixels[i], he.pixels, st))) return rc;
        if ((rc = upload(s->d_env_func[i], he.func, st))) return rc;
        if ((rc = upload(s->d_env_cdf[i], he.cdf, st))) return rc;
        envs[i].pixels = s->d_env_pixels[i].p;
        envs[i].func = s->d_env_func[i].p;
        envs[i].cdf = s->d_env_cdf[i].p;
        envs[i].width = he.width;
        envs[i].height = he.height;
        envs[i].n = he.width * he.height;
        envs[i].funcInt = he.funcInt;
    }
    if ((rc = upload(s->d_envs, envs, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    s->dev.envs = s->d_envs.p;
    s->dev.nodes = s->d_nodes.p;
    s->dev.bigleaves = s->d_bigleaves.p;
    s->dev.tri_verts = s->d_tri_verts.p;
    s->dev.tri_shade = s->d_tri_shade.p;
    s->dev.prims = s->d_prims.p;
    s->dev.materials = s->d_materials.p;
    s->dev.lights = s->d_lights.p;
    s->dev.n_prims = (int)flat.prims.size();
    s->dev.n_lights = (int)lights.size();
    s->dev.n_materials = (int)s->materials.size();
    s->dev.n_infinite = n_inf;
    s->dev.max_depth = flat.max_depth;
    s->dev.rootpair_base = flat.rootpair_base;
    for (int ch = 0; ch < AGPT_MAX_CHUNKS; ch++) s->dev.mesh_masks[ch] = flat.mesh_masks[ch];
    for (int ch = 0; ch < AGPT_MAX_CHUNKS; ch++) s->dev.analytic_masks[ch] = flat.analytic_masks[ch];
    for (int ch = 0; ch <= AGPT_MAX_CHUNKS; ch++) s->dev.pf_begin[ch] = flat.pf_begin[ch];
    s->dev.prefilter = s->d_prefilter.p;
    s->dev.toplevel = reinterpret_cast<const uint4*>(s->d_toplevel.p);
    s->dev.n_toplevel = flat.n_toplevel;
    s->dev.chunk_mesh_masks = s->d_chunk_mesh_masks.p;
    s->dev.n_meshes = 0;
    for (const DevPrim& dp : flat.prims)
        if (dp.type == AGPT_PRIM_MESH && dp.n_tris > 0) s->dev.n_meshes++;
    s->dev.cam = s->cam;
    s->committed = true;
    return AGPT_OK;
}

static const agpt::HostMesh* mesh_of(const agpt_scene* s, int prim) {
    if (!s || prim < 0 || prim >= (int)s->prims.size() || s->prims[prim].type != AGPT_PRIM_MESH) return nullptr;
    return &s->meshes[s->prims[prim].index];
}
int agpt_mesh_num_nodes(const agpt_scene* s, int prim) {
    const agpt::HostMesh* m = mesh_of(s, prim);
    return m ? m->total_nodes : fail(AGPT_ERR_INVALID, "agpt_mesh_num_nodes: not a mesh primitive");
}
int agpt_mesh_num_prims(const agpt_scene* s, int prim) {
    const agpt::HostMesh* m = mesh_of(s, prim);
    return m ? (int)m->prim_index.size() : fail(AGPT_ERR_INVALID, "agpt_mesh_num_prims: not a mesh primitive");
}
int agpt_mesh_get_bvh(const agpt_scene* s, int prim, agpt_bvh_node* nodes_out, int32_t* prim_index_out) {
    const agpt::HostMesh* m = mesh_of(s, prim);
    if (!m) return fail(AGPT_ERR_INVALID, "agpt_mesh_get_bvh: not a mesh primitive");
    if (nodes_out) std::memcpy(nodes_out, m->nodes.data(), m->nodes.size() * sizeof(agpt_bvh_node));
    if (prim_index_out) std::memcpy(prim_index_out, m->prim_index.data(), m->prim_index.size() * sizeof(int32_t));
    return AGPT_OK;
}

int agpt_bvh_build(const float* vertices, int n_vertices, const int32_t* indices, int n_indices, int max_prims_in_node,
                   agpt_bvh_node* nodes_out, int32_t* prim_index_out, int* total_nodes_out, int* max_depth_out) {
    if (!vertices || !indices || n_vertices <= 0 || n_indices < 3 || n_indices % 3 != 0)
        return fail(AGPT_ERR_INVALID, "agpt_bvh_build: need at least one triangle");
    for (int i = 0; i < n_indices; i++)
        if (indices[3 * i] < 0 || indices[3 * i] >= n_vertices) return fail(AGPT_ERR_INVALID, "agpt_bvh_build: vertex index out of range");
    agpt::HostMesh m;
    m.vertices.resize(n_vertices);
    for (int i = 0; i < n_vertices; i++) m.vertices[i] = V3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
    m.indices.assign(indices, indices + (size_t)3 * n_indices);
    agpt::build_bvh(m, max_prims_in_node);
    if (nodes_out) std::memcpy(nodes_out, m.nodes.data(), m.nodes.size() * sizeof(agpt_bvh_node));
    if (prim_index_out) std::memcpy(prim_index_out, m.prim_index.data(), m.prim_index.size() * sizeof(int32_t));
    if (total_nodes_out) *total_nodes_out = m.total_nodes;
    if (max_depth_out) *max_depth_out = m.max_depth;
    return AGPT_OK;
}

int agpt_toplevel_build(const float* boxes6, int n, float* nodes8_out) {
    if (!boxes6 || !nodes8_out || n < 1) return fail(AGPT_ERR_INVALID, "agpt_toplevel_build: need at least one box");
    std::vector<uint32_t> payload((size_t)n);
    for (int k = 0; k < n; k++) payload[k] = (uint32_t)k;
    std::vector<float4> nodes;
    agpt::build_skip_tree(boxes6, payload.data(), n, nodes);
    std::memcpy(nodes8_out, nodes.data(), nodes.size() * sizeof(float4));
    return (int)(nodes.size() / 2);
}

int agpt_toplevel_pack16(const float* nodes8, int n_nodes, uint32_t* packed4_out) {
    if (!nodes8 || !packed4_out || n_nodes < 1) return fail(AGPT_ERR_INVALID, "agpt_toplevel_pack16: need at least one node");
    std::vector<float4> nodes((size_t)2 * n_nodes);
    std::memcpy(nodes.data(), nodes8, nodes.size() * sizeof(float4));
    std::vector<uint32_t> packed;
    agpt::pack_skip_tree16(nodes, packed);
    std::memcpy(packed4_out, packed.data(), packed.size() * sizeof(uint32_t));
    return n_nodes;
}

int agpt_create_backdrop(const float origin[3], const float size[3], float radius, int steps, float* vertices, float* normals,
                         float* texcoords, int32_t* indices, int* n_vertices, int* n_indices) {
    if (!origin || !size || !vertices || !normals || !texcoords || !indices || steps < 1)
        return fail(AGPT_ERR_INVALID, "agpt_create_backdrop: bad argument");
    std::vector<v3> v, n;
    std::vector<v2> t;
    std::vector<int32_t> ix;
    agpt::create_backdrop(V3(origin[0], origin[1], origin[2]), V3(size[0], size[1], size[2]), radius, steps, v, n, t, ix);
    for (size_t i = 0; i < v.size(); i++) {
        vertices[3 * i] = v[i].x; vertices[3 * i + 1] = v[i].y; vertices[3 * i + 2] = v[i].z;
        normals[3 * i] = n[i].x; normals[3 * i + 1] = n[i].y; normals[3 * i + 2] = n[i].z;
        texcoords[2 * i] = t[i].x; texcoords[2 * i + 1] = t[i].y;
    }
    std::memcpy(indices, ix.data(), ix.size() * sizeof(int32_t));
    if (n_vertices) *n_vertices = (int)v.size();
    if (n_indices) *n_indices = (int)ix.size() / 3;
    return AGPT_OK;
}

// ---- device helpers -----------------------------------------------------------------------------------------
int agpt_device_alloc(agpt_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return fail(AGPT_ERR_INVALID, "agpt_device_alloc: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(AGPT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return AGPT_OK;
}
int agpt_device_free(agpt_ctx* c, void* p) {
    if (!c) return fail(AGPT_ERR_INVALID, "agpt_device_free: ctx is NULL");
    if (p) HIP_TRY(hipFree(p));
    return AGPT_OK;
}
int agpt_device_memset(agpt_ctx* c, void* p, int value, size_t bytes) {
    if (!c || !p) return fail(AGPT_ERR_INVALID, "agpt_device_memset: NULL argument");
    HIP_TRY(hipMemsetAsync(p, value, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return AGPT_OK;
}
int agpt_device_download(agpt_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return fail(AGPT_ERR_INVALID, "agpt_device_download: NULL argument");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return AGPT_OK;
}
int agpt_device_upload(agpt_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return fail(AGPT_ERR_INVALID, "agpt_device_upload: NULL argument");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return AGPT_OK;
}

// ---- hot path -----------------------------------------------------------------------------------------------
// bytes of pool state per path: 11 float4 arrays + 2 hit arrays (16 B) + 2 flag words + 2 x AGPT_NQUEUES queue words
#define AGPT_BYTES_PER_PATH (11 * 16 + 2 * 16 + 2 * 4 + 2 * AGPT_NQUEUES * 4)
// extra pool bytes per path for a scene: the candidate words of lists longer than 64 primitives (k_candidates)
static size_t candidate_bytes_per_path(int n_prims) { return n_prims > 64 ? 4 + 8 * (size_t)((n_prims + 63) / 64) : 0; }
static int ensure_pool(agpt_ctx* c, size_t paths, int n_prims) {
    int rc;
    // the pool never shrinks: a small call after a large one keeps the capacity (agpt_render sizes its batches by it, and the
    // candidate words are laid out [chunk][pool_paths])
    paths = std::max(paths, c->pool_paths);
    if (n_prims > 64) {
        if ((rc = c->cand_chunks.ensure(paths))) return rc;
        if ((rc = c->cand_mask.ensure(paths * (size_t)((n_prims + 63) / 64)))) return rc;
    }
    for (DevBuf<float4>* b : {&c->ext_o, &c->ext_d, &c->sh_o, &c->sh_d, &c->mis_o, &c->mis_d, &c->beta4, &c->L4, &c->fac4,
                              &c->c1, &c->c2})
        if ((rc = b->ensure(paths))) return rc;
    if ((rc = c->hit.ensure(paths))) return rc;
    if ((rc = c->mis_hit.ensure(paths))) return rc;
    if ((rc = c->occluded.ensure(paths))) return rc;
    if ((rc = c->mis_ok.ensure(paths))) return rc;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < AGPT_NQUEUES; j++)
            if ((rc = c->q[i][j].ensure(paths))) return rc;
    if ((rc = c->qcounts.ensure(2 * AGPT_NQUEUES * AGPT_QSTRIDE))) return rc;
    if ((rc = c->work.ensure(4096))) return rc;
    if ((rc = c->counters.ensure(1))) return rc;
    c->pool_paths = paths;
    return AGPT_OK;
}

// The wavefront loop of PathTracer::Li (integrator.h:124-191) over the paths k_generate / k_generate_li have set up in q[0]:
// per iteration the three trace launches, k_shade and k_resolve_pending.  A path needs at most max_depth + 2 iterations unless it
// crosses emitter spheres (pass-through re-casts do not count as bounces, integrator.h:152-161): that many run without a host
// round trip, then the loop keeps going while the active or the resolve queue is non-empty.
struct WavefrontRun {
    bool mis_mode, timing, shade_lds_tables, shade_env;
    int count;              // enable_counters, normalised
    uint64_t iterations = 0, launches = 0;
};
static int run_wavefront(agpt_ctx* c, agpt_scene* s, const RenderConsts& rcn, const PathBuffers& pb, Queues q[2], WavefrontRun& run) {
    const int count = run.count;
    const bool mis_mode = run.mis_mode, timing = run.timing;
    const int shade_grid = c->num_cus * c->shade_blocks_per_cu;
    c->trace_small_batch = (unsigned long long)rcn.NP * (unsigned long long)rcn.S < AGPT_SMALL_BATCH;
    const bool recast = !getenv("AGPT_NO_RECAST");   // (developer knob: emitter pass-throughs through k_shade, an iteration each)
    // side streams: not where the launches share scratch buffers (the HBM stack spill of deep BVHs, the candidate words of long
    // lists), nor with the reference-order kernels
    const bool side = (c->multi_stream == 2 || (c->multi_stream == 1 && c->trace_small_batch)) && use_fast_trace(c, s->dev, count) &&
                      s->dev.max_depth <= AGPT_FAST_STACK && s->dev.n_prims <= 64;
    int cur = 0;
    const int planned = rcn.max_depth + 2;
    for (int it = 0;; it++) {
        if (it >= planned) {
            // Termination check, one iteration behind: this iteration's queue counters (active and, 4 counters further on,
            // resolve) are copied out asynchronously, and what is looked at is the copy made an iteration ago, which has
            // arrived by now -- the stream never waits for the host (a synchronous check left it idle for ~50 us per iteration,
            // and the iterations out here are a few hundred microseconds long).  The price is one iteration of launches over
            // empty queues at the very end (the kernels return at once on those).
            uint32_t* slot = c->host_pinned + (it & 1) * (AGPT_NQUEUES * AGPT_QSTRIDE);
            HIP_TRY(hipMemcpyAsync(slot, q[cur].counts, (4 * AGPT_QSTRIDE + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(c->ev[2 + (it & 1)], c->stream));
            if (it > planned) {
                const uint32_t* seen = c->host_pinned + ((it - 1) & 1) * (AGPT_NQUEUES * AGPT_QSTRIDE);
                HIP_TRY(hipEventSynchronize(c->ev[2 + ((it - 1) & 1)]));
                if (seen[0] == 0 && seen[4 * AGPT_QSTRIDE] == 0) break;
            }
            if (it > 4096) return fail(AGPT_ERR_LIMIT, "agpt: path did not terminate");
        }
        const int nxt = cur ^ 1;
        HIP_TRY(hipMemsetAsync(q[nxt].counts, 0, AGPT_NQUEUES * AGPT_QSTRIDE * sizeof(uint32_t), c->stream));
        HIP_TRY(hipMemsetAsync(c->work.p, 0, 4 * AGPT_FRONTIERS * AGPT_QSTRIDE * sizeof(uint32_t), c->stream));
        auto trace_ext = [&]() {
            c->trace_recast = recast;   // (the closest-hit launch over the continuation rays may re-cast in place, see k_trace_fast)
            launch_trace_timed<0>(c, timing, 0, s->dev, q[cur].ext, &QCOUNT(q[cur], 1), 0, c->work.p + 0, pb.ext_o, pb.ext_d, pb.hit, nullptr, count);
            c->trace_recast = false;
        };
        auto trace_mis = [&]() {
            if (mis_mode)
                launch_trace_timed<2>(c, timing, 1, s->dev, q[cur].mis, &QCOUNT(q[cur], 2), 0, c->work.p + AGPT_FRONTIERS * AGPT_QSTRIDE, pb.mis_o,
                                      pb.mis_d, nullptr, pb.mis_ok, count);
            else
                launch_trace_timed<0>(c, timing, 1, s->dev, q[cur].mis, &QCOUNT(q[cur], 2), 0, c->work.p + AGPT_FRONTIERS * AGPT_QSTRIDE, pb.mis_o,
                                      pb.mis_d, pb.mis_hit, nullptr, count);
        };
        auto trace_shadow = [&]() {
            launch_trace_timed<1>(c, timing, 2, s->dev, q[cur].shadow, &QCOUNT(q[cur], 3), 0, c->work.p + 2 * AGPT_FRONTIERS * AGPT_QSTRIDE, pb.sh_o,
                                  pb.sh_d, nullptr, pb.occluded, count);
        };
        if (side) {
            // AGPT_MULTI_STREAM: the three trace launches of an iteration are independent -- MIS and shadow rays on streams of their
            // own, beside the closest-hit launch (c->stream is what the launch helpers enqueue on; restored whatever happens)
            struct Restore {
                agpt_ctx* c;
                hipStream_t s;
                ~Restore() { c->stream = s; }
            } restore{c, c->stream};
            hipStream_t main_stream = c->stream;
            HIP_TRY(hipEventRecord(c->aux_ev[0], main_stream));
            HIP_TRY(hipStreamWaitEvent(c->aux_stream[0], c->aux_ev[0], 0));
            HIP_TRY(hipStreamWaitEvent(c->aux_stream[1], c->aux_ev[0], 0));
            c->stream = c->aux_stream[0];
            trace_mis();
            HIP_TRY(hipEventRecord(c->aux_ev[1], c->stream));
            c->stream = c->aux_stream[1];
            trace_shadow();
            HIP_TRY(hipEventRecord(c->aux_ev[2], c->stream));
            c->stream = main_stream;
            trace_ext();
            HIP_TRY(hipStreamWaitEvent(main_stream, c->aux_ev[1], 0));
            HIP_TRY(hipStreamWaitEvent(main_stream, c->aux_ev[2], 0));
        } else {
            trace_ext();
            trace_mis();
            trace_shadow();
        }
        agpt::launch_shade(c->stream, shade_grid, run.shade_lds_tables, run.shade_env, s->dev, rcn, pb, q[cur], q[nxt], c->counters.p,
                           c->work.p + 3 * AGPT_FRONTIERS * AGPT_QSTRIDE /* k_shade's tile heads, zeroed above */);
        // (independent of k_shade: different paths)
        agpt::launch_resolve_pending(c->stream, c->num_cus * 8, s->dev, rcn, pb, q[cur]);
        cur = nxt;
        run.iterations++;
        run.launches += 3;
    }
    return AGPT_OK;
}

static void bind_pool(agpt_ctx* c, PathBuffers& pb, Queues q[2]) {
    pb.ext_o = c->ext_o.p; pb.ext_d = c->ext_d.p; pb.sh_o = c->sh_o.p; pb.sh_d = c->sh_d.p;
    pb.mis_o = c->mis_o.p; pb.mis_d = c->mis_d.p; pb.hit = c->hit.p; pb.mis_hit = c->mis_hit.p;
    pb.occluded = c->occluded.p; pb.mis_ok = c->mis_ok.p; pb.beta4 = c->beta4.p; pb.L4 = c->L4.p; pb.fac4 = c->fac4.p;
    pb.c1 = c->c1.p; pb.c2 = c->c2.p;
    for (int i = 0; i < 2; i++) {
        q[i].active = c->q[i][0].p; q[i].ext = c->q[i][1].p; q[i].mis = c->q[i][2].p; q[i].shadow = c->q[i][3].p;
        q[i].resolve = c->q[i][4].p;
        q[i].counts = c->qcounts.p + AGPT_NQUEUES * AGPT_QSTRIDE * i;
    }
}
static bool shade_tables_fit_lds(const DevScene& d) {
    return agpt::shade_tables_fit_lds(d.n_prims, d.n_materials, d.n_lights) && !getenv("AGPT_SHADE_GLOBAL_TABLES");
}

static void read_counters(const DevCounters& d, agpt_stats* st) {
    st->closest_rays = d.closest_rays;
    st->anyhit_rays = d.anyhit_rays;
    st->interior_visits = d.interior;
    st->root_tests = d.roots;
    st->tri_tests = d.tris;
    st->shaded_vertices = d.shaded;
    st->outliers = d.outliers;
    st->samples = d.samples;
    st->answered_rays = d.answered;
}

int agpt_intersect_device(agpt_scene* s, const agpt_ray* d_rays, int n, agpt_hit* d_out, int any_hit, agpt_stats* stats) {
    if (!s || !d_rays || !d_out || n < 0) return fail(AGPT_ERR_INVALID, "agpt_intersect_device: bad argument");
    if (!s->committed) return fail(AGPT_ERR_INVALID, "agpt_intersect_device: scene not committed");
    if (n == 0) return AGPT_OK;
    agpt_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_pool(c, (size_t)n, s->dev.n_prims);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(DevCounters), c->stream));
    HIP_TRY(hipMemsetAsync(c->work.p, 0, AGPT_FRONTIERS * AGPT_QSTRIDE * sizeof(uint32_t), c->stream));
    int blocks = (n + AGPT_BLOCK - 1) / AGPT_BLOCK;
    hipLaunchKernelGGL(k_prepare_rays, dim3(blocks), dim3(AGPT_BLOCK), 0, c->stream, d_rays, n, c->ext_o.p, c->ext_d.p);
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    // stats requested -> the instrumented reference-order kernel (work counters); AGPT_INTERSECT_TIMING=1 (developer knob)
    // keeps the production kernel so that stats->trace_ms times it
    const int instrumented = (stats != nullptr && !getenv("AGPT_INTERSECT_TIMING")) ? 1 : 0;
    c->trace_small_batch = (unsigned long long)n < AGPT_SMALL_BATCH;
    if (any_hit)
        launch_trace<1>(c, s->dev, nullptr, nullptr, (uint32_t)n, c->work.p, c->ext_o.p, c->ext_d.p, c->hit.p, c->occluded.p,
                        instrumented);
    else
        launch_trace<0>(c, s->dev, nullptr, nullptr, (uint32_t)n, c->work.p, c->ext_o.p, c->ext_d.p, c->hit.p, c->occluded.p,
                        instrumented);
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    hipLaunchKernelGGL(k_export_hits, dim3(blocks), dim3(AGPT_BLOCK), 0, c->stream, s->dev, c->hit.p, c->occluded.p, n, any_hit,
                       d_out);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(c->take_deferred());
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        DevCounters dc;
        HIP_TRY(hipMemcpy(&dc, c->counters.p, sizeof(dc), hipMemcpyDeviceToHost));
        read_counters(dc, stats);
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        stats->trace_ms = ms;
        stats->total_ms = ms;
        stats->trace_launches = 1;
    }
    return AGPT_OK;
}

int agpt_intersect_batch(agpt_scene* s, const agpt_ray* rays, int n, agpt_hit* out, int any_hit, agpt_stats* stats) {
    if (!s || !rays || !out || n < 0) return fail(AGPT_ERR_INVALID, "agpt_intersect_batch: bad argument");
    if (!s->committed) return fail(AGPT_ERR_INVALID, "agpt_intersect_batch: scene not committed");
    if (n == 0) return AGPT_OK;
    agpt_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    TempBuf<agpt_ray> d_rays;
    TempBuf<agpt_hit> d_out;
    if (d_rays.alloc((size_t)n) != hipSuccess || d_out.alloc((size_t)n) != hipSuccess)
        return fail(AGPT_ERR_NOMEM, "agpt_intersect_batch: out of device memory");
    HIP_TRY(hipMemcpyAsync(d_rays.p, rays, (size_t)n * sizeof(agpt_ray), hipMemcpyHostToDevice, c->stream));
    int rc = agpt_intersect_device(s, d_rays.p, n, d_out.p, any_hit, stats);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * sizeof(agpt_hit), hipMemcpyDeviceToHost));
    return AGPT_OK;
}

// DbgIntegrator::Li (integrator.h:107-118): Scene::Intersect on the GPU, then the hit's uv -- which only this debug view reads, so
// the device keeps no texture coordinates -- from the host copy of the scene in the arithmetic of trianglemesh.cpp:46-57,
// intersectable.h:133 and :187-201.
int agpt_dbg_li_batch(agpt_scene* s, const agpt_ray* rays, int n, float* radiance3_out) {
    if (!s || !rays || !radiance3_out || n < 0) return fail(AGPT_ERR_INVALID, "agpt_dbg_li_batch: bad argument");
    if (n == 0) return AGPT_OK;
    std::vector<agpt_hit> hits((size_t)n);
    const int rc = agpt_intersect_batch(s, rays, n, hits.data(), 0, nullptr);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        float* L = radiance3_out + 3 * (size_t)i;
        L[0] = L[1] = L[2] = 0.f;
        const agpt_hit& h = hits[i];
        if (!h.hit) continue;
        const agpt::HostPrim& hp = s->prims[h.prim];
        float u, v;
        if (hp.type == AGPT_PRIM_MESH) {
            const agpt::HostMesh& m = s->meshes[hp.index];
            float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};   // a mesh without texture coordinates (trianglemesh.cpp:52-56)
            if (!m.texcoords.empty())
                for (int k = 0; k < 3; ++k) {
                    const v2 t = m.texcoords[m.indices[3 * (h.tri + k) + 2]];
                    uv[k][0] = t.x;
                    uv[k][1] = t.y;
                }
            const float b0 = 1.f - h.b1 - h.b2;
            u = uv[0][0] * b0 + uv[1][0] * h.b1 + uv[2][0] * h.b2;
            v = uv[0][1] * b0 + uv[1][1] * h.b1 + uv[2][1] * h.b2;
        } else {
            const agpt::HostSphere& sp = s->spheres[hp.index];
            const v3 D = normalize(V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]));   // the Ray ctor's (camera.h:6), as k_prepare_rays
            const v3 P = V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]) + h.t * D;
            if (hp.type == AGPT_PRIM_PLANE) {   // centre = O, r / r2 = HalfSize.x / .y
                u = ((P.x - sp.center.x) / sp.r + 1) * .5f;
                v = ((P.z - sp.center.z) / sp.r2 + 1) * .5f;
            } else {
                v3 pHit = P - sp.center;
                if (pHit.x == 0 && pHit.y == 0) pHit.x = AGPT_EPSILON * sp.r;
                float phi = cr_atan2f(pHit.y, pHit.x);
                if (phi < 0) phi += AGPT_TWOPI;
                u = phi * AGPT_INV2PI;
                v = cr_acosf(tclampf(pHit.z / sp.r, -1.f, 1.f)) * AGPT_INVPI;
            }
        }
        if (u == 0 || v == 0) {
            L[0] = 1.f;
        } else {
            L[0] = u / 5;
            L[1] = v / 5;
        }
    }
    return AGPT_OK;
}

int agpt_render(agpt_scene* s, const agpt_render_params* rp, float* accum_dev, agpt_stats* stats) {
    if (!s || !rp || !accum_dev) return fail(AGPT_ERR_INVALID, "agpt_render: NULL argument");
    if (!s->committed || !s->has_camera) return fail(AGPT_ERR_INVALID, "agpt_render: scene not committed or camera not set");
    if (rp->width <= 0 || rp->height <= 0 || rp->w <= 0 || rp->h <= 0 || rp->x0 < 0 || rp->y0 < 0 ||
        rp->x0 + rp->w > rp->width || rp->y0 + rp->h > rp->height || rp->spp_count < 0 || rp->max_depth < 0 ||
        rp->max_depth > 200 || rp->accum_pitch < rp->x0 + rp->w)
        return fail(AGPT_ERR_INVALID, "agpt_render: bad tile / film / sample range");
    agpt_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t rows = (uint32_t)rp->h;
    if (rp->interleave_block > 0) {
        if (rp->interleave_world < 1 || rp->interleave_rank < 0 || rp->interleave_rank >= rp->interleave_world || rp->x0 != 0 ||
            rp->y0 != 0 || rp->w != rp->width || rp->h != rp->height)
            return fail(AGPT_ERR_INVALID, "agpt_render: row interleave needs the whole film as tile and 0 <= rank < world");
        rows = 0;
        for (int k = rp->interleave_rank, y = k * rp->interleave_block; y < rp->height;
             k += rp->interleave_world, y = k * rp->interleave_block)
            rows += (uint32_t)std::min(rp->interleave_block, rp->height - y);
        if (rows == 0) return AGPT_OK;
    }
    const uint32_t NP = (uint32_t)rp->w * rows;
    // default batch: up to 128 Mi paths (27 GB of path state -- MI355X has 288 GB); fewer, larger wavefront launches
    int S = rp->samples_per_batch > 0 ? rp->samples_per_batch : (int)std::max<uint64_t>(1, (128ull << 20) / NP);
    S = std::min(S, std::max(1, rp->spp_count));
    if ((uint64_t)NP * (uint64_t)S > 0x7FFFFFFFull) return fail(AGPT_ERR_LIMIT, "agpt_render: batch too large");
    // The batch is a performance choice, not a semantic one (the image is batch-split invariant): when the pool has to
    // grow, keep it within the memory that is actually free (other ranks or applications may share the GPU), and halve
    // the batch if an allocation still fails.
    if ((size_t)NP * (size_t)S > c->pool_paths) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_path = (size_t)AGPT_BYTES_PER_PATH + candidate_bytes_per_path(s->dev.n_prims);
            const size_t avail = free_b + c->pool_paths * per_path;
            const size_t fit = (size_t)((double)avail * 0.9) / per_path;
            while (S > 1 && (size_t)NP * (size_t)S > fit) S = (S + 1) / 2;
        }
    }
    int rc = ensure_pool(c, (size_t)NP * S, s->dev.n_prims);
    while (rc == AGPT_ERR_NOMEM && S > 1) {
        release_pool(c);
        S = (S + 1) / 2;
        rc = ensure_pool(c, (size_t)NP * S, s->dev.n_prims);
    }
    if (rc) return rc;
    s->dev.cam = s->cam;

    PathBuffers pb;
    Queues q[2];
    bind_pool(c, pb, q);
    WavefrontRun run;
    run.count = rp->enable_counters == 2 ? 2 : (rp->enable_counters != 0 ? 1 : 0);
    run.mis_mode = use_fast_trace(c, s->dev, run.count) && !c->mis_closest;
    run.timing = rp->enable_timing != 0 && stats != nullptr;
    run.shade_lds_tables = shade_tables_fit_lds(s->dev);
    run.shade_env = !s->envs.empty();   // an InfiniteAreaLight is present
    const bool timing = run.timing;
    c->tev_used = 0;
    c->tev_kind.clear();
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(DevCounters), c->stream));
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));

    for (int s0 = rp->spp_begin; s0 < rp->spp_begin + rp->spp_count; s0 += S) {
        RenderConsts rcn;
        rcn.W = rp->width; rcn.H = rp->height; rcn.x0 = rp->x0; rcn.y0 = rp->y0; rcn.w = rp->w; rcn.h = rp->h;
        rcn.s0 = s0; rcn.S = std::min(S, rp->spp_begin + rp->spp_count - s0);
        rcn.seed_base = rp->seed_base; rcn.max_depth = rp->max_depth;
        rcn.accum_pitch = rp->accum_pitch; rcn.accum_row0 = rp->accum_row0; rcn.NP = NP;
        rcn.mis_mode = run.mis_mode ? 1 : 0;
        rcn.answer_rays = (run.mis_mode && !rp->trace_all_rays) ? 1 : 0;
        rcn.il_block = rp->interleave_block; rcn.il_world = rp->interleave_world; rcn.il_rank = rp->interleave_rank;
        const uint32_t total = NP * (uint32_t)rcn.S;
        hipLaunchKernelGGL(k_generate, dim3((total + AGPT_BLOCK - 1) / AGPT_BLOCK), dim3(AGPT_BLOCK), 0, c->stream, s->dev, rcn, pb,
                           q[0]);
        {
            const int rc_run = run_wavefront(c, s, rcn, pb, q, run);
            if (rc_run) return rc_run;
        }
        hipLaunchKernelGGL(k_accumulate, dim3((NP + AGPT_BLOCK - 1) / AGPT_BLOCK), dim3(AGPT_BLOCK), 0, c->stream, rcn, pb,
                           (float4*)accum_dev, c->counters.p);
    }
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(c->take_deferred());
    if (stats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memset(stats, 0, sizeof(*stats));
        DevCounters dc;
        HIP_TRY(hipMemcpy(&dc, c->counters.p, sizeof(dc), hipMemcpyDeviceToHost));
        read_counters(dc, stats);
#ifdef AGPT_SHADE_CLOCK
        {
            const double tot = (double)dc.dbg[55];
            static const char* names[8] = {"record loads", "resolve pending", "surface", "emission/termination", "BSDF set-up + light sampling",
                                           "evaluations + ray stores", "final stores", "queue appends + barriers"};
            std::fprintf(stderr, "[shade clock] wave-cycles %.4g:", tot);
            for (int k = 0; k < 8; ++k) std::fprintf(stderr, "  %s %.1f%%", names[k], 100. * dc.dbg[56 + k] / tot);
            std::fprintf(stderr, "\n");
        }
#endif
#ifdef AGPT_TRACE_STATS
        for (int mode = 0; mode < 3; ++mode) {
            const unsigned long long* d = dc.dbg + 16 * mode;
            const double steps = (double)(d[0] + d[1] + d[2]);
            if (steps == 0) continue;
            std::fprintf(stderr,
                         "[trace stats mode %d] steps A/B/C %.3g/%.3g/%.3g (%.1f%%/%.1f%%/%.1f%%)  lanes per step A %.1f B %.1f C %.1f  "
                         "active lanes per step %.1f  refills %.3g (%.1f lanes each)  prefilter batches %.3g\n", mode,
                         (double)d[0], (double)d[1], (double)d[2], 100 * d[0] / steps, 100 * d[1] / steps, 100 * d[2] / steps,
                         d[0] ? (double)d[3] / d[0] : 0., d[1] ? (double)d[4] / d[1] : 0., d[2] ? (double)d[5] / d[2] : 0.,
                         (double)d[6] / steps, (double)d[7], d[7] ? (double)d[8] / d[7] : 0., (double)d[9]);
#ifndef AGPT_TRACE_CLOCK
            std::fprintf(stderr, "[trace stats mode %d] stack pushes %.3g: to depth > 4 %.2f%%  > 6 %.3f%%  > 8 %.4f%%  > 12 %.5f%%\n", mode, (double)d[10],
                         d[10] ? 100. * d[11] / d[10] : 0., d[10] ? 100. * d[12] / d[10] : 0., d[10] ? 100. * d[13] / d[10] : 0.,
                         d[10] ? 100. * d[14] / d[10] : 0.);
#endif
#ifdef AGPT_TRACE_CLOCK
            const double tot = (double)dc.dbg[48 + mode];
            std::fprintf(stderr,
                         "[trace clock mode %d] wave-cycles %.4g: refill %.1f%%  vote %.1f%%  B %.1f%% (of which load wait %.1f%%)  C %.1f%%  A %.1f%%;"
                         "  cycles per step: B %.0f (wait %.0f)  C %.0f  A %.0f  vote %.0f; per refill %.0f\n", mode, tot,
                         100 * d[10] / tot, 100 * d[11] / tot, 100 * d[13] / tot, 100 * d[12] / tot, 100 * d[14] / tot, 100 * d[15] / tot,
                         d[1] ? (double)d[13] / d[1] : 0., d[1] ? (double)d[12] / d[1] : 0., d[2] ? (double)d[14] / d[2] : 0.,
                         d[0] ? (double)d[15] / d[0] : 0., (double)d[11] / steps, d[7] ? (double)d[10] / d[7] : 0.);
#endif
        }
#endif
        stats->samples = (uint64_t)NP * (uint64_t)rp->spp_count;
        stats->iterations = run.iterations;
        stats->trace_launches = run.launches;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        stats->total_ms = ms;
        if (timing) {
            for (size_t i = 0; i + 1 < c->tev_used; i += 2) {
                float t = 0;
                HIP_TRY(hipEventElapsedTime(&t, c->tev[i], c->tev[i + 1]));
                int kind = c->tev_kind[i / 2];
                stats->trace_ms += t;
                if (kind == 0) stats->ext_ms += t;
                else if (kind == 1) stats->mis_ms += t;
                else stats->shadow_ms += t;
            }
        }
    }
    return AGPT_OK;
}

int agpt_li_batch(agpt_scene* s, const agpt_ray* rays, const uint32_t* rng_states, int n, int max_depth, float* radiance3_out,
                  uint32_t* rng_states_out, agpt_stats* stats) {
    if (!s || !rays || !rng_states || !radiance3_out || n < 0 || max_depth < 0 || max_depth > 200)
        return fail(AGPT_ERR_INVALID, "agpt_li_batch: bad argument");
    if (!s->committed) return fail(AGPT_ERR_INVALID, "agpt_li_batch: scene not committed");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return AGPT_OK;
    agpt_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    TempBuf<agpt_ray> d_rays;
    TempBuf<uint32_t> d_rng;
    TempBuf<float> d_out;
    if (d_rays.alloc((size_t)n) != hipSuccess || d_rng.alloc((size_t)n) != hipSuccess || d_out.alloc(3 * (size_t)n) != hipSuccess)
        return fail(AGPT_ERR_NOMEM, "agpt_li_batch: out of device memory");
    HIP_TRY(hipMemcpyAsync(d_rays.p, rays, (size_t)n * sizeof(agpt_ray), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_rng.p, rng_states, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    int rc = ensure_pool(c, (size_t)n, s->dev.n_prims);
    if (rc) return rc;
    PathBuffers pb;
    Queues q[2];
    bind_pool(c, pb, q);
    WavefrontRun run;
    run.count = 0;
    run.mis_mode = use_fast_trace(c, s->dev, 0) && !c->mis_closest;
    run.timing = false;
    run.shade_lds_tables = shade_tables_fit_lds(s->dev);
    run.shade_env = !s->envs.empty();
    RenderConsts rcn{};
    rcn.max_depth = max_depth;
    rcn.mis_mode = run.mis_mode ? 1 : 0;
    rcn.answer_rays = run.mis_mode ? 1 : 0;
    c->tev_used = 0;
    c->tev_kind.clear();
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(DevCounters), c->stream));
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    const dim3 grid((unsigned)((n + AGPT_BLOCK - 1) / AGPT_BLOCK)), block(AGPT_BLOCK);
    hipLaunchKernelGGL(k_generate_li, grid, block, 0, c->stream, (const agpt_ray*)d_rays.p, (const uint32_t*)d_rng.p, (uint32_t)n, pb, q[0], max_depth);
    rc = run_wavefront(c, s, rcn, pb, q, run);
    if (rc) return rc;
    hipLaunchKernelGGL(k_export_li, grid, block, 0, c->stream, pb, (uint32_t)n, d_out.p, rng_states_out ? d_rng.p : (uint32_t*)nullptr);
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(c->take_deferred());
    HIP_TRY(hipMemcpyAsync(radiance3_out, d_out.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (rng_states_out) HIP_TRY(hipMemcpyAsync(rng_states_out, d_rng.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats) {
        DevCounters dc;
        HIP_TRY(hipMemcpy(&dc, c->counters.p, sizeof(dc), hipMemcpyDeviceToHost));
        read_counters(dc, stats);
        stats->samples = (uint64_t)n;
        stats->iterations = run.iterations;
        stats->trace_launches = run.launches;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        stats->total_ms = ms;
    }
    return AGPT_OK;
}

int agpt_resolve(agpt_ctx* c, const float* accum_dev, int n_pixels, int samples, uint32_t* out_rgb) {
    if (!c || !accum_dev || !out_rgb || n_pixels <= 0 || samples <= 0) return fail(AGPT_ERR_INVALID, "agpt_resolve: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    TempBuf<uint32_t> d;
    HIP_TRY(d.alloc((size_t)n_pixels));
    hipLaunchKernelGGL(k_resolve, dim3((n_pixels + 255) / 256), dim3(256), 0, c->stream, (const float4*)accum_dev, n_pixels, samples, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rgb, d.p, (size_t)n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return AGPT_OK;
}

// ---- known-answer entry points ------------------------------------------------------------------------------
int agpt_kat_bsdf_eval(agpt_scene* s, int material, int n, const float* wo3, const float* wi3, float* f3_out, float* pdf_out) {
    if (!s || !s->committed || material < 0 || material >= (int)s->materials.size() || n <= 0 || !wo3 || !wi3 || !f3_out || !pdf_out)
        return fail(AGPT_ERR_INVALID, "agpt_kat_bsdf_eval: bad argument");
    agpt_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    TempBuf<float> d_wo, d_wi, d_f, d_p;
    HIP_TRY(d_wo.alloc(3 * (size_t)n));
    HIP_TRY(d_wi.alloc(3 * (size_t)n));
    HIP_TRY(d_f.alloc(3 * (size_t)n));
    HIP_TRY(d_p.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_wo.p, wo3, 12 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_wi.p, wi3, 12 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_kat_bsdf_eval, dim3((n + 63) / 64), dim3(64), 0, c->stream, s->dev, material, n, d_wo.p, d_wi.p, d_f.p, d_p.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(f3_out, d_f.p, 12 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf_out, d_p.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return AGPT_OK;
}

int agpt_kat_bsdf_sample(agpt_scene* s, int material, int n, const float* wo3, const float* u2, float* wi3_out, float* f3_out,
                         float* pdf_out, int32_t* specular_out) {
    if (!s || !s->committed || material < 0 || material >= (int)s->materials.size() || n <= 0 || !wo3 || !u2 || !wi3_out ||
        !f3_out || !pdf_out || !specular_out)
        return fail(AGPT_ERR_INVALID, "agpt_kat_bsdf_sample: bad argument");
    agpt_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    TempBuf<float> d_wo, d_u, d_wi, d_f, d_p;
    TempBuf<int32_t> d_s;
    HIP_TRY(d_wo.alloc(3 * (size_t)n));
    HIP_TRY(d_u.alloc(2 * (size_t)n));
    HIP_TRY(d_wi.alloc(3 * (size_t)n));
    HIP_TRY(d_f.alloc(3 * (size_t)n));
    HIP_TRY(d_p.alloc((size_t)n));
    HIP_TRY(d_s.alloc((size_t)n));
    HIP_TRY(hipMemcpy(d_wo.p, wo3, 12 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_u.p, u2, 8 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_kat_bsdf_sample, dim3((n + 63) / 64), dim3(64), 0, c->stream, s->dev, material, n, d_wo.p, d_u.p, d_wi.p, d_f.p,
                       d_p.p, d_s.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(wi3_out, d_wi.p, 12 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(f3_out, d_f.p, 12 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf_out, d_p.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(specular_out, d_s.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return AGPT_OK;
}

int agpt_kat_rng(agpt_ctx* c, uint32_t pixel, uint32_t wh, uint32_t sample, uint32_t seed_base, int n, float* out,
                 uint32_t* seed_out) {
    if (!c || n <= 0 || !out || !seed_out) return fail(AGPT_ERR_INVALID, "agpt_kat_rng: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    TempBuf<float> d_o;
    TempBuf<uint32_t> d_s;
    HIP_TRY(d_o.alloc((size_t)n));
    HIP_TRY(d_s.alloc(1));
    hipLaunchKernelGGL(k_kat_rng, dim3(1), dim3(64), 0, c->stream, pixel, wh, sample, seed_base, n, d_o.p, d_s.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, d_o.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(seed_out, d_s.p, 4, hipMemcpyDeviceToHost));
    return AGPT_OK;
}

int agpt_kat_distribution1d(agpt_ctx* c, const float* func, int n, const float* u, int k, float* cdf_out, float* func_int_out,
                            float* x_out, float* pdf_out) {
    if (!c || !func || n <= 0 || k < 0 || (k && (!u || !x_out || !pdf_out)))
        return fail(AGPT_ERR_INVALID, "agpt_kat_distribution1d: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float> cdf((size_t)n + 1);
    const float func_int = agpt::build_distribution1d(func, n, cdf.data());
    if (cdf_out) memcpy(cdf_out, cdf.data(), 4 * cdf.size());
    if (func_int_out) *func_int_out = func_int;
    if (k == 0) return AGPT_OK;
    TempBuf<float> d_f, d_c, d_u, d_x, d_p;
    HIP_TRY(d_f.alloc((size_t)n));
    HIP_TRY(d_c.alloc((size_t)n + 1));
    HIP_TRY(d_u.alloc((size_t)k));
    HIP_TRY(d_x.alloc((size_t)k));
    HIP_TRY(d_p.alloc((size_t)k));
    HIP_TRY(hipMemcpy(d_f.p, func, 4 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_c.p, cdf.data(), 4 * cdf.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_u.p, u, 4 * (size_t)k, hipMemcpyHostToDevice));
    DevEnv e{};
    e.func = d_f.p;
    e.cdf = d_c.p;
    e.n = n;
    e.funcInt = func_int;
    hipLaunchKernelGGL(k_kat_distribution1d, dim3((k + 63) / 64), dim3(64), 0, c->stream, e, d_u.p, k, d_x.p, d_p.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(x_out, d_x.p, 4 * (size_t)k, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf_out, d_p.p, 4 * (size_t)k, hipMemcpyDeviceToHost));
    return AGPT_OK;
}


// ---- multi-GPU: gather of the per-rank tile buffers ------------------------------------------------
// RCCL is bound at run time (dlopen of the librccl already in the process, else the ROCm one): a single-GPU host never
// loads it, and a host that also uses PyTorch shares PyTorch's copy instead of getting a second set of nccl* symbols.
}  // extern "C"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) return nullptr;
    bool ok = true;
    auto sym = [&](const char* n) {
        void* p = dlsym(api.lib, n);
        ok = ok && p != nullptr;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(api.lib);
        api.lib = nullptr;
        return nullptr;
    }
    return &api;
}

// compact rank buffer -> full accumulator.  Rank r owns the film's row blocks k with k % world == r; its j-th block sits at
// compact rows [j*block, j*block + h) with the rows flipped inside the block (agpt_render's interleave layout), the full
// accumulator is Accumulator::pixels: row (H-1-y) (myapp.h:17-19).  One thread per float4.
__global__ void k_deinterleave(const float4* __restrict__ compact, float4* __restrict__ full, int W, int H, int block, int world,
                               int rank, int rows_local) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows_local * (size_t)W) return;
    const int row = (int)(i / (size_t)W), x = (int)(i - (size_t)row * (size_t)W);
    const int j = row / block, r_in = row % block;
    const int yb = (j * world + rank) * block;      // first film row of the block
    const int hb = min(block, H - yb);
    if (r_in >= hb) return;
    const int y = yb + (hb - 1 - r_in);             // compact row j*block + (hb-1-within) holds film row yb + within
    full[(size_t)(H - 1 - y) * (size_t)W + (size_t)x] = compact[i];
}

int rows_of_rank(int H, int block, int world, int rank) {
    int rows = 0;
    for (int k = rank, y = k * block; y < H; k += world, y = k * block) rows += std::min(block, H - y);
    return rows;
}

}  // namespace

struct agpt_comm {
    agpt_ctx* ctx = nullptr;
    int world = 1, rank = 0;
    ncclComm_t comm = nullptr;
    DevBuf<float4> staging;   // rank 0: one compact buffer per peer
};

extern "C" {

int agpt_comm_unique_id(void* id128) {
    if (!id128) return fail(AGPT_ERR_INVALID, "agpt_comm_unique_id: NULL argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    RcclApi* R = rccl();
    if (!R) return fail(AGPT_ERR_DEVICE, "agpt_comm_unique_id: librccl.so could not be loaded");
    ncclResult_t e = R->GetUniqueId((ncclUniqueId*)id128);
    if (e != ncclSuccess) return fail(AGPT_ERR_DEVICE, std::string("ncclGetUniqueId: ") + R->GetErrorString(e));
    return AGPT_OK;
}

int agpt_comm_init(agpt_ctx* c, const void* id128, int world, int rank, agpt_comm** out) {
    if (!c || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128))
        return fail(AGPT_ERR_INVALID, "agpt_comm_init: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    std::unique_ptr<agpt_comm> m(new agpt_comm());
    m->ctx = c;
    m->world = world;
    m->rank = rank;
    if (world > 1) {   // a single rank needs no communicator (and no RCCL)
        RcclApi* R = rccl();
        if (!R) return fail(AGPT_ERR_DEVICE, "agpt_comm_init: librccl.so could not be loaded");
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        ncclResult_t e = R->CommInitRank(&m->comm, world, id, rank);
        if (e != ncclSuccess) return fail(AGPT_ERR_DEVICE, std::string("ncclCommInitRank: ") + R->GetErrorString(e));
    }
    *out = m.release();
    return AGPT_OK;
}

void agpt_comm_destroy(agpt_comm* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->comm) (void)rccl()->CommDestroy(m->comm);
    m->staging.release();
    delete m;
}

int agpt_deinterleave_tiles(agpt_ctx* c, const float* compact_dev, int width, int height, int block_rows, int world, int rank,
                            float* full_accum_dev) {
    if (!c || !compact_dev || !full_accum_dev || width <= 0 || height <= 0 || block_rows <= 0 || world < 1 || rank < 0 || rank >= world)
        return fail(AGPT_ERR_INVALID, "agpt_deinterleave_tiles: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const int rows = rows_of_rank(height, block_rows, world, rank);
    if (!rows) return AGPT_OK;
    const size_t n = (size_t)rows * (size_t)width;
    hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float4*)compact_dev,
                       (float4*)full_accum_dev, width, height, block_rows, world, rank, rows);
    HIP_TRY(hipGetLastError());
    return AGPT_OK;
}

int agpt_gather_tiles(agpt_comm* m, const float* local_accum_dev, int width, int height, int block_rows, float* full_accum_dev) {
    if (!m || !local_accum_dev || width <= 0 || height <= 0 || block_rows <= 0 || (m->rank == 0 && !full_accum_dev))
        return fail(AGPT_ERR_INVALID, "agpt_gather_tiles: bad argument");
    agpt_ctx* c = m->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int world = m->world;
    int max_rows = 0;
    for (int r = 0; r < world; r++) max_rows = std::max(max_rows, rows_of_rank(height, block_rows, world, r));
    const size_t slot = (size_t)max_rows * (size_t)width;   // float4 per rank buffer
    if (world > 1) {
        RcclApi* R = rccl();
        ncclResult_t e = ncclSuccess;
        if (m->rank == 0) {
            int rc = m->staging.ensure(slot * (size_t)(world - 1));
            if (rc) return rc;
            // grouped point-to-point: every peer's buffer travels its own direct xGMI link to rank 0 (not a ring)
            // (a group that was started is always ended, also when a call inside it fails: the first error is reported)
            e = R->GroupStart();
            if (e == ncclSuccess) {
                for (int r = 1; r < world && e == ncclSuccess; r++) {
                    const size_t n = (size_t)rows_of_rank(height, block_rows, world, r) * (size_t)width * 4;
                    if (n) e = R->Recv(m->staging.p + slot * (size_t)(r - 1), n, ncclFloat, r, m->comm, c->stream);
                }
                const ncclResult_t e_end = R->GroupEnd();
                if (e == ncclSuccess) e = e_end;
            }
        } else {
            const size_t n = (size_t)rows_of_rank(height, block_rows, world, m->rank) * (size_t)width * 4;
            e = R->GroupStart();
            if (e == ncclSuccess) {
                if (n) e = R->Send(local_accum_dev, n, ncclFloat, 0, m->comm, c->stream);
                const ncclResult_t e_end = R->GroupEnd();
                if (e == ncclSuccess) e = e_end;
            }
        }
        if (e != ncclSuccess) return fail(AGPT_ERR_DEVICE, std::string("agpt_gather_tiles: ") + R->GetErrorString(e));
    }
    if (m->rank == 0) {
        for (int r = 0; r < world; r++) {
            const float* src = r == 0 ? local_accum_dev : (const float*)(m->staging.p + slot * (size_t)(r - 1));
            int rc = agpt_deinterleave_tiles(c, src, width, height, block_rows, world, r, full_accum_dev);
            if (rc) return rc;
        }
    }
    return AGPT_OK;
}

}  // extern "C"
