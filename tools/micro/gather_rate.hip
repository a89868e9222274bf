// Developer microbenchmark: what a divergent gather of 64-byte records costs on gfx950, by access shape.
// Every lane (or lane pair / quad) chases its own chain of random 64-B records of a 12.8 MB table (the size of the C3 scene's
// BVH nodes: L2-missing, Infinity-Cache resident); the next index depends on the loaded data, like a BVH descent.
//   KIND 0  one record per lane,  4 x global_load_dwordx4 of the lane's own 64-B line   (k_trace_fast's interior step, round 2)
//   KIND 1  one record per lane,  1 x dwordx4 (first quarter only)
//   KIND 7  one record per lane,  2 x dwordx4
//   KIND 6  one record per lane,  4 x dword
//   KIND 2  one record per lane PAIR: 2 x dwordx4, the pair reads 32 contiguous bytes per instruction
//   KIND 3  one record per lane QUAD: 1 x dwordx4, the quad reads the 64 contiguous bytes
//   KIND 4  one record per lane, fetched quad-cooperatively: instruction j reads the record of the quad's lane j (results land transposed)
//   KIND 5  as 4 through global_load_lds_dwordx4, then 4 x ds_read_b128 of the lane's own record
//   KIND 8  one record per lane, 4 x global_load_lds_dwordx4 of the lane's OWN line (row j = quarter j), then 4 x ds_read_b128
//   KIND 9  as 8 with 3 x LDS-DMA (48 B) + 1 x global_load_dwordx2 into registers
// `active` (2nd argument): only lanes < active take part (the trace kernel's interior step runs with ~36 of 64 lanes).
// Reported per variant and occupancy: records per second and CU cycles per 64 records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ uint32_t next_index(uint32_t idx, uint32_t data, uint32_t n) {
    uint32_t h = (idx * 1664525u + 1013904223u) ^ data;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return h % n;
}

template <int KIND>
__global__ void __launch_bounds__(256) gk(const float4* __restrict__ table, uint32_t n, int iters, uint32_t* __restrict__ out, int active) {
    __shared__ float4 s_stage[KIND == 5 || KIND == 8 || KIND == 9 ? 4 : 1][4][65];   // per wave: 4 instruction rows of 1 KiB + 16 B pad
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t group = KIND == 2 ? gid >> 1 : (KIND == 3 ? gid >> 2 : gid);
    uint32_t idx = (group * 2654435761u) % n;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t data = 0;
        if (lane < active) {
        if (KIND == 0 || KIND == 1 || KIND == 7) {
            const float4* p = table + 4 * (size_t)idx;
            float4 a = p[0];
            uint32_t x = __float_as_uint(a.x) ^ __float_as_uint(a.w);
            if (KIND != 1) { float4 b = p[1]; x ^= __float_as_uint(b.y); }
            if (KIND == 0) { float4 c = p[2], d = p[3]; x ^= __float_as_uint(c.z) ^ __float_as_uint(d.w); }
            data = x;
        } else if (KIND == 6) {
            const float* p = (const float*)(table + 4 * (size_t)idx);
            data = __float_as_uint(p[0]) ^ __float_as_uint(p[4]) ^ __float_as_uint(p[8]) ^ __float_as_uint(p[12]);
        } else if (KIND == 2) {
            const float4* p = table + 4 * (size_t)idx + (lane & 1);
            float4 a = p[0], b = p[2];
            uint32_t x = __float_as_uint(a.x) ^ __float_as_uint(b.y);
            x ^= (uint32_t)__shfl_xor((int)x, 1);
            data = x;
        } else if (KIND == 3) {
            const float4* p = table + 4 * (size_t)idx + (lane & 3);
            float4 a = p[0];
            uint32_t x = __float_as_uint(a.x);
            x ^= (uint32_t)__shfl_xor((int)x, 1);
            x ^= (uint32_t)__shfl_xor((int)x, 2);
            data = x;
        } else if (KIND == 4) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t other = (uint32_t)__shfl((int)idx, (lane & ~3) + j);
                float4 a = table[4 * (size_t)other + (lane & 3)];
                // (a real kernel would transpose here; the xor keeps the dependence on every load)
                x ^= __float_as_uint(a.x) + (uint32_t)j;
            }
            x ^= (uint32_t)__shfl_xor((int)x, 1);
            x ^= (uint32_t)__shfl_xor((int)x, 2);
            data = x ^ (uint32_t)lane;
        } else if (KIND == 8 || KIND == 9) {
#pragma unroll
            for (int j = 0; j < (KIND == 9 ? 3 : 4); ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_global_load_lds(table + 4 * (size_t)idx + j, &s_stage[wave][j][0], 16, 0, 0);
#endif
            }
            uint32_t x = 0;
            if (KIND == 9) {
                const float2 e = *(const float2*)(table + 4 * (size_t)idx + 3);
                x = __float_as_uint(e.x) ^ __float_as_uint(e.y);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float4 a = s_stage[wave][0][lane], b = s_stage[wave][1][lane], c = s_stage[wave][2][lane];
            x ^= __float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z);
            if (KIND == 8) { float4 d = s_stage[wave][3][lane]; x ^= __float_as_uint(d.w); }
            data = x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t other = (uint32_t)__shfl((int)idx, (lane & ~3) + j);
#if defined(__HIP_DEVICE_COMPILE__)   // (the builtin exists in the device pass only)
                __builtin_amdgcn_global_load_lds(table + 4 * (size_t)other + (lane & 3), &s_stage[KIND == 5 ? wave : 0][j][0], 16, 0, 0);
#endif
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the record of lane 4k+j sits in row j at float4 slots 4k .. 4k+3
            const float4* mine = &s_stage[KIND == 5 ? wave : 0][lane & 3][lane & ~3];
            float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
            data = __float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z) ^ __float_as_uint(d.w);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        }
        acc += data;
        idx = next_index(idx, data, n);
    }
    out[gid] = acc ^ idx;
}


template <int KIND>
void run(const float4* table, uint32_t n, uint32_t* out, int cus, int bpc, int iters, const char* what, int per_wave, int active) {
    if (per_wave == 64) per_wave = active;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gk<KIND>, dim3(cus * bpc), dim3(256), 0, 0, table, n, iters, out, active);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double waves = (double)cus * bpc * 4, steps = waves * iters;
    const double recs = steps * per_wave;
    // CU cycles (2.4 GHz nominal) per 64 records fetched
    const double cyc = ms * 1e-3 * 2.4e9 * cus / (recs / 64.0);
    (void)cyc;
    printf("KIND %d %-58s %d blocks/CU: %8.3f ms  %7.2f Grecords/s  %6.1f CU-cycles per 64 records  (%.0f per wave-step)\n", KIND, what, bpc, ms,
           recs / (ms * 1e-3) / 1e9, cyc, ms * 1e-3 * 2.4e9 * cus / steps);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000u;   // records of 64 B
    std::vector<float> h((size_t)n * 16);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; uint32_t b = s >> 3; memcpy(&v, &b, 4); }
    float4* table;
    uint32_t* out;
    hipMalloc(&table, h.size() * 4);
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    const int iters = 2000;
    printf("table: %u records of 64 B = %.1f MB, %d CUs, %d dependent steps per lane\n", n, n * 64e-6, cus, iters);
    const int active = argc > 2 ? atoi(argv[2]) : 64;
    printf("active lanes per wave: %d\n", active);
    for (int bpc : {4, 5, 8}) {
        run<0>(table, n, out, cus, bpc, iters, "lane/record, 4 x dwordx4 (same line)", 64, active);
        run<7>(table, n, out, cus, bpc, iters, "lane/record, 2 x dwordx4", 64, active);
        run<1>(table, n, out, cus, bpc, iters, "lane/record, 1 x dwordx4", 64, active);
        run<6>(table, n, out, cus, bpc, iters, "lane/record, 4 x dword", 64, active);
        run<2>(table, n, out, cus, bpc, iters, "pair/record, 2 x dwordx4 (32 B contiguous per pair)", 32, active);
        run<3>(table, n, out, cus, bpc, iters, "quad/record, 1 x dwordx4 (64 B contiguous per quad)", 16, active);
        run<4>(table, n, out, cus, bpc, iters, "lane/record via 4 quad-cooperative dwordx4 (transposed)", 64, active);
        run<5>(table, n, out, cus, bpc, iters, "lane/record via 4 quad-cooperative LDS-DMA + 4 ds_read_b128", 64, active);
        run<8>(table, n, out, cus, bpc, iters, "lane/record via 4 own-line LDS-DMA + 4 ds_read_b128", 64, active);
        run<9>(table, n, out, cus, bpc, iters, "lane/record via 3 own-line LDS-DMA + dwordx2 + 3 ds_read_b128", 64, active);
    }
    return 0;
}
