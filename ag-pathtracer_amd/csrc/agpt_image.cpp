// agpt_image.cpp -- host-only image writers for the presentation end of the path.  The reference
// shows its Accumulator through an OpenGL blit (myapp.cpp:177, Accumulator::CopyToSurface myapp.h:34-41) and never
// writes a file; a headless GPU node needs one.  PNG: 8-bit RGB from the 0x00RRGGBB words agpt_resolve produces,
// zlib stream of stored (uncompressed) deflate blocks -- no dependency.  PFM: the linear float image (sum / samples),
// bottom row first as the format prescribes (the accumulator keeps the top row first).
// Also the input side of the environment light: the Radiance RGBE (.hdr) reader HDRTexture's constructor gets from
// stbi_loadf (texture.h:41-52) -- see agpt_hdr_parse below.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/agpt.h"

namespace agpt {
int report_error(int code, const std::string& msg);  // agpt_api.hip
}

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24));
    v.push_back(uint8_t(x >> 16));
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}

void put_chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32(out.data() + at, out.size() - at));
}

}  // namespace

extern "C" int agpt_write_png(const char* path, const uint32_t* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) return agpt::report_error(AGPT_ERR_INVALID, "agpt_write_png: bad argument");
    // scanlines: filter byte 0 + RGB
    const size_t stride = 1 + 3 * (size_t)width;
    std::vector<uint8_t> raw(stride * height);
    for (int y = 0; y < height; y++) {
        uint8_t* row = raw.data() + y * stride;
        row[0] = 0;
        for (int x = 0; x < width; x++) {
            const uint32_t c = rgb[(size_t)y * width + x];
            row[1 + 3 * x] = uint8_t(c >> 16);
            row[2 + 3 * x] = uint8_t(c >> 8);
            row[3 + 3 * x] = uint8_t(c);
        }
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;  // adler32
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n));
        z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        off += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8-bit, truecolour, deflate, adaptive filter set, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return agpt::report_error(AGPT_ERR_IO, std::string("agpt_write_png: cannot open ") + path);
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? AGPT_OK : agpt::report_error(AGPT_ERR_IO, std::string("agpt_write_png: short write to ") + path);
}

extern "C" int agpt_write_pfm(const char* path, const float* accum_host, int width, int height, int samples) {
    if (!path || !accum_host || width <= 0 || height <= 0 || samples <= 0)
        return agpt::report_error(AGPT_ERR_INVALID, "agpt_write_pfm: bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return agpt::report_error(AGPT_ERR_IO, std::string("agpt_write_pfm: cannot open ") + path);
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);  // negative scale = little endian
    std::vector<float> row(3 * (size_t)width);
    const float inv = 1.0f / (float)samples;
    bool ok = true;
    for (int y = height - 1; y >= 0 && ok; y--) {  // accumulator row 0 is the TOP image row (myapp.h:24,34-41); PFM starts at the bottom
        const float* src = accum_host + 4 * (size_t)y * width;
        for (int x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) row[3 * x + c] = src[4 * x + c] * inv;
        ok = std::fwrite(row.data(), 4, row.size(), f) == row.size();
    }
    std::fclose(f);
    return ok ? AGPT_OK : agpt::report_error(AGPT_ERR_IO, std::string("agpt_write_pfm: short write to ") + path);
}


// ---- Radiance RGBE (.hdr) reader -----------------------------------------------------------------------------------------
// What HDRTexture's constructor gets from stbi_loadf(filename, &w, &h, &n, 0) (texture.h:41-52) for a Radiance file, restated
// from the behaviour of the stb_image v2.27 the reference vendors (lib/stb_image.h:7005-7215; public domain / MIT, Sean Barrett,
// HDR loader originally by Nicolas Schulz): the signature test, the header tokenizer with its 1023-character lines, the
// "-Y h +X w" layout only, flat pixels for widths < 8 or >= 32768, per-channel run-length scanlines otherwise, a scanline that
// does not start with (2, 2, len < 128) switching the WHOLE image to flat pixels from that point, and
// value = mantissa * 2^(e - 136) with e = 0 meaning black.  Pinned byte for byte against that header compiled where it lies
// (oracle/ref_stb_hdr.cpp, tests/golden/hdr_cases.npz).  Where the reference would read past the end of the file (it then
// converts uninitialised bytes, or spins on zero-length runs) this reader fails with AGPT_ERR_INVALID instead.
namespace {

struct ByteCursor {
    const uint8_t* at;
    const uint8_t* end;
    bool overrun = false;
    bool exhausted() const { return at >= end; }
    int next() {   // a byte, or 0 past the end (as stb's reader returns)
        if (at < end) return *at++;
        overrun = true;
        return 0;
    }
};

// One header line without its '\n', cut at 1023 characters (the rest of the line is skipped).  As in stb's tokenizer the
// character read last is dropped when the data ends with it (stb_image.h:7031-7051); the callers compare C strings, so an
// embedded NUL ends the token for them.
std::string hdr_header_line(ByteCursor& in) {
    std::string line;
    int ch = in.next();
    while (!in.exhausted() && ch != '\n') {
        line.push_back((char)ch);
        if (line.size() == 1023) {
            while (!in.exhausted() && in.next() != '\n') {
            }
            break;
        }
        ch = in.next();
    }
    return line;
}

inline void rgbe_to_rgb(const uint8_t q[4], float* rgb) {
    if (q[3] == 0) {
        rgb[0] = rgb[1] = rgb[2] = 0.f;
        return;
    }
    const float scale = (float)std::ldexp(1.0, (int)q[3] - 136);   // a power of two: the three products are exact
    rgb[0] = (float)q[0] * scale;
    rgb[1] = (float)q[1] * scale;
    rgb[2] = (float)q[2] * scale;
}

int hdr_fail(const std::string& why) { return agpt::report_error(AGPT_ERR_INVALID, "agpt_hdr_parse: " + why); }

// the rest of the image as 4-byte pixels, starting at pixel index `first`
bool hdr_flat_pixels(ByteCursor& in, float* rgb, size_t first, size_t n_pixels) {
    for (size_t i = first; i < n_pixels; i++) {
        if (in.end - in.at < 4) return false;
        rgbe_to_rgb(in.at, rgb + 3 * i);
        in.at += 4;
    }
    return true;
}

}  // namespace

extern "C" int agpt_hdr_parse(const void* bytes, size_t length, int* width_out, int* height_out, float** rgb_out) {
    if (!bytes || !width_out || !height_out || !rgb_out) return agpt::report_error(AGPT_ERR_INVALID, "agpt_hdr_parse: bad argument");
    *rgb_out = nullptr;
    ByteCursor in{(const uint8_t*)bytes, (const uint8_t*)bytes + length};
    // stbi__hdr_test: the signature line must be exactly one of these, '\n' included
    auto starts_with = [&](const char* sig) { size_t n = strlen(sig); return length >= n && memcmp(bytes, sig, n) == 0; };
    if (!starts_with("#?RADIANCE\n") && !starts_with("#?RGBE\n"))
        return hdr_fail("not a Radiance .hdr file (the reference's stbi_loadf would go on to its LDR decoders, which are out of scope)");
    hdr_header_line(in);
    bool rle_rgbe = false;
    for (;;) {
        const std::string line = hdr_header_line(in);
        if (line.c_str()[0] == 0) break;
        if (strcmp(line.c_str(), "FORMAT=32-bit_rle_rgbe") == 0) rle_rgbe = true;
    }
    if (!rle_rgbe) return hdr_fail("unsupported format (no FORMAT=32-bit_rle_rgbe line)");
    const std::string dims = hdr_header_line(in);
    const char* t = dims.c_str();
    if (strncmp(t, "-Y ", 3)) return hdr_fail("unsupported data layout (resolution line must start with -Y)");
    char* rest = nullptr;
    const int height = (int)strtol(t + 3, &rest, 10);
    while (*rest == ' ') ++rest;
    if (strncmp(rest, "+X ", 3)) return hdr_fail("unsupported data layout (+X must follow -Y)");
    const int width = (int)strtol(rest + 3, nullptr, 10);
    if (height > (1 << 24) || width > (1 << 24)) return hdr_fail("very large image (corrupt?)");
    if (width <= 0 || height <= 0) return hdr_fail("empty or negative resolution");
    if ((long long)width * height * 12 > (long long)INT_MAX) return hdr_fail("image is too large (stb_image limits w*h*12 to INT_MAX)");
    const size_t n_pixels = (size_t)width * height;
    float* rgb = (float*)malloc(n_pixels * 3 * sizeof(float));
    if (!rgb) return agpt::report_error(AGPT_ERR_NOMEM, "agpt_hdr_parse: out of memory");
    auto give_up = [&](const char* why) {
        free(rgb);
        return hdr_fail(why);
    };
    if (width < 8 || width >= 32768) {
        if (!hdr_flat_pixels(in, rgb, 0, n_pixels)) return give_up("truncated pixel data");
    } else {
        std::vector<uint8_t> scan((size_t)width * 4);
        for (int row = 0; row < height; row++) {
            const int b0 = in.next(), b1 = in.next(), b2 = in.next();
            if (b0 != 2 || b1 != 2 || (b2 & 0x80)) {
                // not a run-length scanline: these four bytes are pixel 0 and everything after them is flat pixels from pixel 1
                // on -- whichever row this happens in (stb_image.h:7176-7188 restarts at the top)
                const uint8_t q[4] = {(uint8_t)b0, (uint8_t)b1, (uint8_t)b2, (uint8_t)in.next()};
                if (in.overrun) return give_up("truncated pixel data");
                rgbe_to_rgb(q, rgb);
                if (!hdr_flat_pixels(in, rgb, 1, n_pixels)) return give_up("truncated pixel data");
                break;
            }
            const int coded = (b2 << 8) | in.next();
            if (coded != width) return give_up("invalid decoded scanline length");
            for (int ch = 0; ch < 4; ch++) {
                int x = 0;
                while (x < width) {
                    const int left = width - x;
                    int count = in.next();
                    if (count > 128) {   // a run of one value
                        const uint8_t value = (uint8_t)in.next();
                        count -= 128;
                        if (count > left) return give_up("bad RLE data");
                        for (int z = 0; z < count; z++) scan[(size_t)(x++) * 4 + ch] = value;
                    } else {             // literal bytes
                        if (count > left) return give_up("bad RLE data");
                        for (int z = 0; z < count; z++) scan[(size_t)(x++) * 4 + ch] = (uint8_t)in.next();
                    }
                    if (in.overrun) return give_up("truncated pixel data");
                }
            }
            for (int x = 0; x < width; x++) rgbe_to_rgb(&scan[(size_t)x * 4], rgb + 3 * ((size_t)row * width + x));
        }
    }
    *width_out = width;
    *height_out = height;
    *rgb_out = rgb;
    return AGPT_OK;
}

extern "C" int agpt_hdr_load(const char* path, int* width_out, int* height_out, float** rgb_out) {
    if (!path) return agpt::report_error(AGPT_ERR_INVALID, "agpt_hdr_load: bad argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return agpt::report_error(AGPT_ERR_IO, std::string("agpt_hdr_load: cannot open ") + path);
    std::vector<uint8_t> data;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) data.insert(data.end(), chunk, chunk + got);
    std::fclose(f);
    return agpt_hdr_parse(data.data(), data.size(), width_out, height_out, rgb_out);
}

extern "C" void agpt_hdr_free(float* rgb) { free(rgb); }
