// agpt_image.cpp -- host-only image writers for the presentation end of the path.  The reference
// shows its Accumulator through an OpenGL blit (myapp.cpp:177, Accumulator::CopyToSurface myapp.h:34-41) and never
// writes a file; a headless GPU node needs one.  PNG: 8-bit RGB from the 0x00RRGGBB words agpt_resolve produces,
// zlib stream of stored (uncompressed) deflate blocks -- no dependency.  PFM: the linear float image (sum / samples),
// bottom row first as the format prescribes (the accumulator keeps the top row first).
#include <algorithm>
#include <cstdint>
#include <cstdio>
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
