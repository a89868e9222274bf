// Developer microbenchmark: VALU issue rate of gfx950 per SIMD for 1..8 resident waves per SIMD.
// Each wave runs N dependent-free v_fma_f32 (8 independent accumulator chains) and stamps s_memtime around them.
// Reported: shader cycles per wave-instruction as seen by one wave, and the aggregate per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            } else if (KIND == 1) {
                a0 = fminf(a0, a1 + c); a1 = fmaxf(a1, a2); a2 = a2 * m; a3 = a3 - c; a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = a6 + c; a7 = a7 * m;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    int cus = 256;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    cus = p.multiProcessorCount;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipMalloc(&cyc, 8 * 4 * cus * 8);
    const int iters = 20000;
    for (int kind = 0; kind < 2; kind++)
        for (int bpc = 1; bpc <= 8; bpc *= 2) {   // blocks of 4 waves per CU = waves per SIMD
            const int blocks = cus * bpc;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 4);
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            for (auto v : h) sum += (double)v;
            const double insts = (double)iters * 64.0;   // wave-instructions per wave
            printf("kind %d  waves/SIMD %d: %.3f ms  cycles per wave-inst seen by a wave %.2f  => per SIMD %.2f cycles/inst  (wall: %.1f Ginst/s chip)\n", kind, bpc,
                   ms, sum / h.size() / insts, sum / h.size() / insts / bpc, insts * blocks * 4 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
