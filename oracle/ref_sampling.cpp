// oracle/ref_sampling.cpp -- TEST INFRASTRUCTURE ONLY (never linked by the product).
//
// Harness around the reference's own sampling.h (FindInterval + Distribution1D, the importance sampler of InfiniteAreaLight,
// lights.cpp:46,54,102), compiled from the header where it lies (/root/reference/sampling.h, `make -C oracle ref`), output
// oracle/_ref/libref_sampling.so.  sampling.h includes only <vector> and calls an unqualified clamp(int, int, int) that its
// includer is expected to have in scope (the reference gets one from template/precomp.h, which cannot be compiled here); this
// harness brings the C++17 standard library's std::clamp into scope -- same value for lo <= hi -- and nothing else.
// tests/golden/make_dist1d_golden.py turns its outputs into tests/golden/dist1d.npz.
#include <algorithm>
using std::clamp;
#include "sampling.h"

extern "C" {

// Distribution1D::Distribution1D (sampling.h:21-33): cdf_out[n + 1], *func_int_out
void ref_dist1d_build(const float* func, int n, float* cdf_out, float* func_int_out) {
    Distribution1D d(func, n);
    for (int i = 0; i <= n; i++) cdf_out[i] = d.cdf[i];
    *func_int_out = d.funcInt;
}

// Distribution1D::SampleContinuous (sampling.h:37-52) and ::SampleDiscrete (:54-61) for k values of u
void ref_dist1d_sample(const float* func, int n, const float* u, int k, float* x_out, float* pdf_out, int* offset_out,
                       int* discrete_out, float* discrete_pdf_out) {
    Distribution1D d(func, n);
    for (int i = 0; i < k; i++) {
        x_out[i] = d.SampleContinuous(u[i], &pdf_out[i], &offset_out[i]);
        discrete_out[i] = d.SampleDiscrete(u[i], &discrete_pdf_out[i]);
    }
}
}
