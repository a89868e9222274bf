// oracle/ref_stb_hdr.cpp -- TEST INFRASTRUCTURE ONLY (never linked by the product).
//
// Harness around the reference's own vendored stb_image (lib/stb_image.h, v2.27), compiled from the header where it lies
// (/root/reference/lib/stb_image.h, `make -C oracle ref`), output oracle/_ref/libref_stb_hdr.so.  It exposes the one call
// HDRTexture's constructor makes (texture.h:44: stbi_loadf(filename, &width, &height, &n, 0)).
// tests/golden/make_hdr_golden.py turns its outputs into tests/golden/hdr_cases.npz.
#define STB_IMAGE_IMPLEMENTATION
#include "lib/stb_image.h"

extern "C" {
float* ref_stbi_loadf(const char* path, int* w, int* h, int* n) { return stbi_loadf(path, w, h, n, 0); }
void ref_stbi_free(float* p) { stbi_image_free(p); }
const char* ref_stbi_failure_reason() { return stbi_failure_reason(); }
}
