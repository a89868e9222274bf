// agpt_obj.cpp -- OBJ ingestion with the semantics of the reference's TriangleMesh::LoadObj (trianglemesh.cpp:157-230),
// host-side scene prep.
//
// The reference parses with the vendored tinyobjloader 2.0.0 (tiny_obj_loader.h:26, ObjReader with the default
// triangulate=true) and then:  transforms every `v` with mat4::TransformPoint (template/precomp.h:1020-1029), every
// `vn` with TransformVector of Inverted().Transposed() (:208-217; skipped when ignore_normals), keeps `vt` as (u, v),
// and concatenates the index triplets {vertex, normal, texcoord} of all shapes in order (:200-205) -- i.e. faces in file
// order, one material for the whole mesh.  Restated here from tinyobjloader, for the statements a triangle mesh needs:
//   * its line reader (safeGetline, tiny_obj_loader.h:730-757: LF, CRLF and lone CR end a line) and statement dispatch
//     (LoadObj, :2194-2745: leading blanks skipped, `v` / `vn` / `vt` / `f` / `l` / `p` / `g` / `o` need a blank after the keyword);
//   * its own decimal parser tryParseDouble (:836-960) -- NOT strtod: digits accumulate in a double, the fraction through
//     a power table / pow(10, -k), the exponent as ldexp(m * 5^e, e), parsing is greedy and a failed parse leaves the
//     default (0, or 1 for w) -- then a cast to float;
//   * index triples through atoi (parseTriple, :1093-1151) with 1-based and negative (relative to the counts at that
//     line) indices (fixIndex, :765-787); a zero or unparsable index fails the whole load, also on `l` and `p` lines;
//   * faces are kept per group and triangulated when the group is FLUSHED (`g`, `o`, end of file) against every vertex
//     parsed up to then (exportGroupsToShape, :1365-1600): faces with fewer than 3 corners dropped, polygons cut by its
//     2-D ear clipping (projection axes from the first non-degenerate corner, signed area, pnpoly overlap test, bail-out
//     when no ear is found).
// Materials do not reach LoadObj's output; `usemtl` (which flushes the open faces when the material id changes, :2465-2489)
// is ignored, i.e. treated as tinyobjloader treats it when no .mtl file is found.  Pinned byte for byte against the
// reference's own header by tests/golden/obj_cases.npz (oracle/ref_tinyobj.cpp, tests/golden/make_obj_golden.py).
// Unlike the reference (which then indexes out of bounds), a face that references a vertex / normal / texcoord that does
// not exist is an error here.
//
// Third-party notices.  The token parsers and the polygon triangulation below restate algorithms of tinyobjloader 2.0.0 (each
// function says which); tinyobjloader is distributed under the MIT licence:
//
//     The MIT License (MIT)
//     Copyright (c) 2012-2018 Syoyo Fujita and many contributors.
//     Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//     documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//     rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//     persons to whom the Software is furnished to do so, subject to the following conditions:
//     The above copyright notice and this permission notice shall be included in all copies or substantial portions of the
//     Software.
//     THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//     WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR
//     COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//     OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
//
// inside_polygon() is the PNPOLY point-inclusion test that tinyobjloader takes from
// https://wrf.ecse.rpi.edu/Research/Short_Notes/pnpoly.html:
//
//     Copyright (c) 1970-2003, Wm. Randolph Franklin
//     Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//     documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//     rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//     persons to whom the Software is furnished to do so, subject to the following conditions:
//     1. Redistributions of source code must retain the above copyright notice, this list of conditions and the following
//        disclaimers.
//     2. Redistributions in binary form must reproduce the above copyright notice in the documentation and/or other materials
//        provided with the distribution.
//     3. The name of W. Randolph Franklin may not be used to endorse or promote products derived from this Software without
//        specific prior written permission.
//     THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//     WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR
//     COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//     OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
//
// inverse_transpose() evaluates the 4x4 cofactor expansion of Mesa's gluInvertMatrix (SGI Free Software License B), the
// form the reference's mat4::Inverted uses (template/precomp.h:948-990): the operation order decides the rounded normals.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/agpt.h"

namespace {

struct Corner {
    int v, vt, vn;
};

struct Mat4 {
    float c[16];
};

Mat4 identity() {
    Mat4 m;
    std::memset(m.c, 0, sizeof(m.c));
    m.c[0] = m.c[5] = m.c[10] = m.c[15] = 1.f;
    return m;
}

// mat4::Inverted (template/precomp.h:948-990, the MESA cofactor expansion) followed by ::Transposed (3x3 part)
Mat4 inverse_transpose(const Mat4& M) {
    const float* cell = M.c;
    const float inv[16] = {
        cell[5] * cell[10] * cell[15] - cell[5] * cell[11] * cell[14] - cell[9] * cell[6] * cell[15] + cell[9] * cell[7] * cell[14] + cell[13] * cell[6] * cell[11] - cell[13] * cell[7] * cell[10],
        -cell[1] * cell[10] * cell[15] + cell[1] * cell[11] * cell[14] + cell[9] * cell[2] * cell[15] - cell[9] * cell[3] * cell[14] - cell[13] * cell[2] * cell[11] + cell[13] * cell[3] * cell[10],
        cell[1] * cell[6] * cell[15] - cell[1] * cell[7] * cell[14] - cell[5] * cell[2] * cell[15] + cell[5] * cell[3] * cell[14] + cell[13] * cell[2] * cell[7] - cell[13] * cell[3] * cell[6],
        -cell[1] * cell[6] * cell[11] + cell[1] * cell[7] * cell[10] + cell[5] * cell[2] * cell[11] - cell[5] * cell[3] * cell[10] - cell[9] * cell[2] * cell[7] + cell[9] * cell[3] * cell[6],
        -cell[4] * cell[10] * cell[15] + cell[4] * cell[11] * cell[14] + cell[8] * cell[6] * cell[15] - cell[8] * cell[7] * cell[14] - cell[12] * cell[6] * cell[11] + cell[12] * cell[7] * cell[10],
        cell[0] * cell[10] * cell[15] - cell[0] * cell[11] * cell[14] - cell[8] * cell[2] * cell[15] + cell[8] * cell[3] * cell[14] + cell[12] * cell[2] * cell[11] - cell[12] * cell[3] * cell[10],
        -cell[0] * cell[6] * cell[15] + cell[0] * cell[7] * cell[14] + cell[4] * cell[2] * cell[15] - cell[4] * cell[3] * cell[14] - cell[12] * cell[2] * cell[7] + cell[12] * cell[3] * cell[6],
        cell[0] * cell[6] * cell[11] - cell[0] * cell[7] * cell[10] - cell[4] * cell[2] * cell[11] + cell[4] * cell[3] * cell[10] + cell[8] * cell[2] * cell[7] - cell[8] * cell[3] * cell[6],
        cell[4] * cell[9] * cell[15] - cell[4] * cell[11] * cell[13] - cell[8] * cell[5] * cell[15] + cell[8] * cell[7] * cell[13] + cell[12] * cell[5] * cell[11] - cell[12] * cell[7] * cell[9],
        -cell[0] * cell[9] * cell[15] + cell[0] * cell[11] * cell[13] + cell[8] * cell[1] * cell[15] - cell[8] * cell[3] * cell[13] - cell[12] * cell[1] * cell[11] + cell[12] * cell[3] * cell[9],
        cell[0] * cell[5] * cell[15] - cell[0] * cell[7] * cell[13] - cell[4] * cell[1] * cell[15] + cell[4] * cell[3] * cell[13] + cell[12] * cell[1] * cell[7] - cell[12] * cell[3] * cell[5],
        -cell[0] * cell[5] * cell[11] + cell[0] * cell[7] * cell[9] + cell[4] * cell[1] * cell[11] - cell[4] * cell[3] * cell[9] - cell[8] * cell[1] * cell[7] + cell[8] * cell[3] * cell[5],
        -cell[4] * cell[9] * cell[14] + cell[4] * cell[10] * cell[13] + cell[8] * cell[5] * cell[14] - cell[8] * cell[6] * cell[13] - cell[12] * cell[5] * cell[10] + cell[12] * cell[6] * cell[9],
        cell[0] * cell[9] * cell[14] - cell[0] * cell[10] * cell[13] - cell[8] * cell[1] * cell[14] + cell[8] * cell[2] * cell[13] + cell[12] * cell[1] * cell[10] - cell[12] * cell[2] * cell[9],
        -cell[0] * cell[5] * cell[14] + cell[0] * cell[6] * cell[13] + cell[4] * cell[1] * cell[14] - cell[4] * cell[2] * cell[13] - cell[12] * cell[1] * cell[6] + cell[12] * cell[2] * cell[5],
        cell[0] * cell[5] * cell[10] - cell[0] * cell[6] * cell[9] - cell[4] * cell[1] * cell[10] + cell[4] * cell[2] * cell[9] + cell[8] * cell[1] * cell[6] - cell[8] * cell[2] * cell[5]};
    const float det = cell[0] * inv[0] + cell[1] * inv[4] + cell[2] * inv[8] + cell[3] * inv[12];
    Mat4 I = identity();
    if (det != 0) {
        const float invdet = 1.0f / det;
        for (int i = 0; i < 16; i++) I.c[i] = inv[i] * invdet;
    }
    Mat4 T = identity();  // Transposed(): 3x3 part only, the rest stays identity (template/precomp.h:940-947)
    T.c[0] = I.c[0]; T.c[1] = I.c[4]; T.c[2] = I.c[8];
    T.c[4] = I.c[1]; T.c[5] = I.c[5]; T.c[6] = I.c[9];
    T.c[8] = I.c[2]; T.c[9] = I.c[6]; T.c[10] = I.c[10];
    return T;
}

void transform_point(const Mat4& M, const float v[3], float out[3]) {
    const float* c = M.c;
    float rx = c[0] * v[0] + c[1] * v[1] + c[2] * v[2] + c[3];
    float ry = c[4] * v[0] + c[5] * v[1] + c[6] * v[2] + c[7];
    float rz = c[8] * v[0] + c[9] * v[1] + c[10] * v[2] + c[11];
    const float w = c[12] * v[0] + c[13] * v[1] + c[14] * v[2] + c[15];
    if (w == 1) {
        out[0] = rx; out[1] = ry; out[2] = rz;
        return;
    }
    const float iw = 1.f / w;
    out[0] = rx * iw; out[1] = ry * iw; out[2] = rz * iw;
}
void transform_vector(const Mat4& M, const float v[3], float out[3]) {
    const float* c = M.c;
    out[0] = c[0] * v[0] + c[1] * v[1] + c[2] * v[2];
    out[1] = c[4] * v[0] + c[5] * v[1] + c[6] * v[2];
    out[2] = c[8] * v[0] + c[9] * v[1] + c[10] * v[2];
}

// ---- OBJ token parsers ---------------------------------------------------------------------------------------------------------
// The functions from here to the end of triangulate_polygon() restate, in this file's own form, algorithms of tinyobjloader 2.0.0
// (the header the reference vendors as tiny_obj_loader.h; MIT licence, notice at the top of this file): byte parity with what
// the reference parses requires the same arithmetic in the same order, so the NUMERIC steps are tinyobjloader's even where the
// control flow is not.  Per function: which of its functions is restated, and where it sits in the reference's copy.

// 1-based / negative OBJ index -> 0-based (restates tinyobj::fixIndex, tiny_obj_loader.h:765-787): k > 0 counts from the start,
// k < 0 back from the `count` elements seen so far, 0 is malformed.
bool resolve_index(int k, int count, int* zero_based) {
    if (k == 0) return false;
    *zero_based = k > 0 ? k - 1 : count + k;
    return true;
}

// One face corner `v`, `v/vt`, `v//vn` or `v/vt/vn` (restates tinyobj::parseTriple, tiny_obj_loader.h:1093-1151: every number
// through atoi, a field ends at '/', blank or CR).  Absent fields stay -1.
bool read_corner(const char*& cur, int n_v, int n_vn, int n_vt, Corner* corner) {
    Corner c;
    c.v = c.vt = c.vn = -1;
    auto field = [&cur](int count, int* dst) {
        const bool ok = resolve_index(std::atoi(cur), count, dst);
        cur += std::strcspn(cur, "/ \t\r");
        return ok;
    };
    if (!field(n_v, &c.v)) return false;
    if (*cur == '/') {
        ++cur;
        if (*cur == '/') {   // v//vn
            ++cur;
            if (!field(n_vn, &c.vn)) return false;
        } else {             // v/vt[/vn]
            if (!field(n_vt, &c.vt)) return false;
            if (*cur == '/') {
                ++cur;
                if (!field(n_vn, &c.vn)) return false;
            }
        }
    }
    *corner = c;
    return true;
}

inline bool is_digit(char x) { return static_cast<unsigned int>(x - '0') < 10u; }

// 10^-place as tinyobjloader forms it: a literal for the first seven places, pow() beyond
inline double negative_power_of_ten(int place) {
    static const double literal[8] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
    return place < 8 ? literal[place] : std::pow(10.0, -place);
}

// Decimal text -> double, NOT through strtod (restates tinyobj::tryParseDouble, tiny_obj_loader.h:836-960, whose results differ
// from strtod's in the last place): integer digits accumulate as m = m * 10 + d, fraction digits add d * 10^-place, a decimal
// exponent e gives ldexp(m * 5^e, e).  The scan is greedy -- it stops at the first character that cannot continue the number
// and keeps what it has -- and fails (leaving *value alone) on: empty text, a first character that is not a sign, digit or
// point, no integer digit where one is required, an `e` without digits.
bool scan_decimal(const char* first, const char* last, double* value) {
    if (first >= last) return false;
    const char* p = first;
    bool negative = false, point_first = false;
    if (*p == '+' || *p == '-') {
        negative = *p == '-';
        ++p;
        point_first = p != last && *p == '.';
    } else if (*p == '.') {
        point_first = true;
    } else if (!is_digit(*p)) {
        return false;
    }
    double m = 0.0;
    if (!point_first) {
        const char* const digits = p;
        for (; p != last && is_digit(*p); ++p) {
            m *= 10;
            m += static_cast<int>(*p - '0');
        }
        if (p == digits) return false;
    }
    int e10 = 0;
    if (p != last) {
        bool exponent_may_follow = *p == 'e' || *p == 'E';
        if (*p == '.') {
            ++p;
            for (int place = 1; p != last && is_digit(*p); ++p, ++place) m += static_cast<int>(*p - '0') * negative_power_of_ten(place);
            exponent_may_follow = p != last;
        }
        if (exponent_may_follow && (*p == 'e' || *p == 'E')) {
            ++p;
            bool exponent_negative = false;
            if (p != last && (*p == '+' || *p == '-')) {
                exponent_negative = *p == '-';
                ++p;
            } else if (!is_digit(*p)) {   // (at `last` this reads the delimiter behind the token -- a blank, CR or NUL -- like the original)
                return false;
            }
            const char* const digits = p;
            for (; p != last && is_digit(*p); ++p) {
                e10 *= 10;
                e10 += static_cast<int>(*p - '0');
            }
            if (exponent_negative) e10 = -e10;
            if (p == digits) return false;
        }
    }
    *value = (negative ? -1 : 1) * (e10 ? std::ldexp(m * std::pow(5.0, e10), e10) : m);
    return true;
}

// the next blank-delimited number of a statement as float, `fallback` if it does not parse (restates tinyobj::parseReal,
// tiny_obj_loader.h:962-970)
float parse_real(const char*& token, double fallback = 0.0) {
    token += std::strspn(token, " \t");
    const char* const stop = token + std::strcspn(token, " \t\r");
    double val = fallback;
    scan_decimal(token, stop, &val);
    token = stop;
    return static_cast<float>(val);
}

// Point-in-polygon by crossing number: W. Randolph Franklin's PNPOLY (notice at the top of this file), which tinyobjloader
// embeds (tiny_obj_loader.h:1350-1362) and uses as its ear test; the comparison and the interpolation are his expression.
bool inside_polygon(int n, const float* xs, const float* ys, float px, float py) {
    bool inside = false;
    int prev = n - 1;
    for (int i = 0; i < n; prev = i++) {
        const bool straddles = (ys[i] > py) != (ys[prev] > py);
        if (straddles && px < (xs[prev] - xs[i]) * (py - ys[i]) / (ys[prev] - ys[i]) + xs[i]) inside = !inside;
    }
    return inside;
}

// One polygon -> triangles by 2-D ear clipping (restates the polygon branch of tinyobj::exportGroupsToShape, tiny_obj_loader.h:
// 1365-1600): project on the two axes orthogonal-most to the first non-degenerate corner's normal, take the polygon's winding
// from its signed area, then cut ears -- a corner whose turn matches the winding and whose triangle holds no other remaining
// vertex -- starting the search where the last cut was made; give up when a full round finds no ear (the rest is dropped unless
// exactly a triangle remains).  `xyz` = every vertex parsed so far; corners that point past its end are skipped like there.
void triangulate_polygon(const std::vector<Corner>& face, const std::vector<float>& xyz, std::vector<Corner>& out) {
    const size_t n_corners = face.size();
    if (n_corners < 3) return;
    const size_t n_floats = xyz.size();
    auto corner_vertex = [&face, n_corners](size_t k) { return (size_t)face[k % n_corners].v; };
    // projection plane
    size_t ax = 1, ay = 2;
    for (size_t k = 0; k < n_corners; ++k) {
        const size_t a = corner_vertex(k), b = corner_vertex(k + 1), c = corner_vertex(k + 2);
        if (3 * a + 2 >= n_floats || 3 * b + 2 >= n_floats || 3 * c + 2 >= n_floats) continue;
        const float ux = xyz[3 * b] - xyz[3 * a], uy = xyz[3 * b + 1] - xyz[3 * a + 1], uz = xyz[3 * b + 2] - xyz[3 * a + 2];
        const float wx = xyz[3 * c] - xyz[3 * b], wy = xyz[3 * c + 1] - xyz[3 * b + 1], wz = xyz[3 * c + 2] - xyz[3 * b + 2];
        const float nx = std::fabs(uy * wz - uz * wy), ny = std::fabs(uz * wx - ux * wz), nz = std::fabs(ux * wy - uy * wx);
        const float tiny = std::numeric_limits<float>::epsilon();
        if (!(nx > tiny || ny > tiny || nz > tiny)) continue;
        if (!(nx > ny && nx > nz)) {   // x-dominant normal: (y, z); z-dominant: (x, y); otherwise (x, z)
            ax = 0;
            if (nz > nx && nz > ny) ay = 1;
        }
        break;
    }
    // winding
    float signed_area = 0;
    for (size_t k = 0; k < n_corners; ++k) {
        const size_t a = corner_vertex(k), b = corner_vertex(k + 1);
        if (3 * a + ax >= n_floats || 3 * a + ay >= n_floats || 3 * b + ax >= n_floats || 3 * b + ay >= n_floats) continue;
        const float x0 = xyz[3 * a + ax], y0 = xyz[3 * a + ay], x1 = xyz[3 * b + ax], y1 = xyz[3 * b + ay];
        signed_area += (x0 * y1 - y0 * x1) * 0.5f;
    }
    // ears
    std::vector<Corner> ring = face;
    size_t at = 0, tries_left = face.size(), size_seen = ring.size();
    while (ring.size() > 3 && tries_left > 0) {
        const size_t n = ring.size();
        if (at >= n) at -= n;
        if (size_seen != n) {   // an ear was cut since the last look: a fresh round of tries
            size_seen = n;
            tries_left = n;
        } else {
            --tries_left;
        }
        Corner tri[3];
        float tx[3], ty[3];
        for (size_t k = 0; k < 3; ++k) {
            tri[k] = ring[(at + k) % n];
            const size_t v = (size_t)tri[k].v;
            const bool present = 3 * v + ax < n_floats && 3 * v + ay < n_floats;
            tx[k] = present ? xyz[3 * v + ax] : 0.f;
            ty[k] = present ? xyz[3 * v + ay] : 0.f;
        }
        const float ux = tx[1] - tx[0], uy = ty[1] - ty[0], wx = tx[2] - tx[1], wy = ty[2] - ty[1];
        const float turn = ux * wy - uy * wx;
        bool ear = !(turn * signed_area < 0.0f);
        for (size_t other = 3; ear && other < n; ++other) {
            const size_t idx = (at + other) % n;
            if (idx >= ring.size()) continue;
            const size_t v = (size_t)ring[idx].v;
            if (3 * v + ax >= n_floats || 3 * v + ay >= n_floats) continue;
            if (inside_polygon(3, tx, ty, xyz[3 * v + ax], xyz[3 * v + ay])) ear = false;
        }
        if (!ear) {
            ++at;
            continue;
        }
        out.insert(out.end(), tri, tri + 3);
        ring.erase(ring.begin() + (std::ptrdiff_t)((at + 1) % n));   // the ear's tip leaves the ring
    }
    if (ring.size() == 3) out.insert(out.end(), ring.begin(), ring.end());
}

thread_local std::string g_obj_error;

}  // namespace

struct agpt_obj_mesh {
    std::vector<float> vertices, normals, texcoords;
    std::vector<int32_t> indices;  // (v, n, t) triplets
};

extern "C" {

const char* agpt_obj_last_error(void) { return g_obj_error.c_str(); }

int agpt_obj_parse(const char* text, size_t length, const float* transform16, int ignore_normals, agpt_obj_mesh** out) {
    if (!text || !out) {
        g_obj_error = "agpt_obj_parse: NULL argument";
        return AGPT_ERR_INVALID;
    }
    Mat4 M = identity();
    if (transform16) std::memcpy(M.c, transform16, sizeof(M.c));
    const Mat4 N = inverse_transpose(M);
    std::vector<float> v, vn, vt;
    std::vector<Corner> corners;             // triangulated, all shapes concatenated (trianglemesh.cpp:200-205)
    std::vector<std::vector<Corner>> group;  // PrimGroup::faceGroup: faces waiting for the next flush
    auto flush = [&]() {                     // exportGroupsToShape for the open group
        for (const std::vector<Corner>& face : group) triangulate_polygon(face, v, corners);
        group.clear();
    };
    auto is_space = [](char x) { return x == ' ' || x == '\t'; };
    auto is_new_line = [](char x) { return x == '\r' || x == '\n' || x == '\0'; };
    int lineno = 0;
    size_t pos = 0;
    std::string line;
    while (pos < length) {   // safeGetline: LF, CRLF or a lone CR end a line
        line.clear();
        while (pos < length) {
            const char ch = text[pos++];
            if (ch == '\n') break;
            if (ch == '\r') {
                if (pos < length && text[pos] == '\n') pos++;
                break;
            }
            line += ch;
        }
        lineno++;
        if (line.empty()) continue;
        const char* token = line.c_str();
        token += std::strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;
        if (token[0] == 'v' && is_space(token[1])) {   // parseVertexWithColor: x y z (+ optional colour, ignored by LoadObj)
            token += 2;
            const float x = parse_real(token), y = parse_real(token), z = parse_real(token);
            v.push_back(x); v.push_back(y); v.push_back(z);
            continue;
        }
        if (token[0] == 'v' && token[1] == 'n' && is_space(token[2])) {
            token += 3;
            const float x = parse_real(token), y = parse_real(token), z = parse_real(token);
            vn.push_back(x); vn.push_back(y); vn.push_back(z);
            continue;
        }
        if (token[0] == 'v' && token[1] == 't' && is_space(token[2])) {
            token += 3;
            const float x = parse_real(token), y = parse_real(token);
            vt.push_back(x); vt.push_back(y);
            continue;
        }
        const bool is_face = token[0] == 'f' && is_space(token[1]);
        if (is_face || ((token[0] == 'l' || token[0] == 'p') && is_space(token[1]))) {
            const char kind = token[0];
            token += 2;
            if (is_face) token += std::strspn(token, " \t");
            std::vector<Corner> face;
            while (!is_new_line(token[0])) {
                Corner c;
                if (!read_corner(token, (int)v.size() / 3, (int)vn.size() / 3, (int)vt.size() / 2, &c)) {
                    g_obj_error = std::string("agpt_obj_parse: failed to parse `") + kind + "' line " + std::to_string(lineno) +
                                  " (zero or malformed index)";
                    return AGPT_ERR_INVALID;
                }
                face.push_back(c);
                token += std::strspn(token, " \t\r");
            }
            if (is_face) group.push_back(std::move(face));   // lines and points do not reach LoadObj's output
            continue;
        }
        if ((token[0] == 'g' || token[0] == 'o') && is_space(token[1])) {   // a new group / object flushes the open faces
            flush();
            continue;
        }
        // usemtl, mtllib, s, t, vw, free-form geometry: no influence on LoadObj's output (see the file comment for usemtl)
    }
    flush();
    agpt_obj_mesh* m = new agpt_obj_mesh();
    m->vertices.resize(v.size());
    for (size_t i = 0; i + 2 < v.size(); i += 3) transform_point(M, &v[i], &m->vertices[i]);
    if (!ignore_normals) {
        m->normals.resize(vn.size());
        for (size_t i = 0; i + 2 < vn.size(); i += 3) transform_vector(N, &vn[i], &m->normals[i]);
    }
    m->texcoords = vt;
    const int nv = (int)v.size() / 3, nn = (int)m->normals.size() / 3, nt = (int)vt.size() / 2;
    for (const Corner& c : corners) {
        // the reference indexes normals[]/texcoords[] with whatever tinyobj returned (-1 when absent): reject instead
        if (c.v < 0 || c.v >= nv || (nn > 0 && (c.vn < 0 || c.vn >= nn)) || (nt > 0 && (c.vt < 0 || c.vt >= nt))) {
            delete m;
            g_obj_error = "agpt_obj_parse: a face references a missing vertex / normal / texcoord (the reference would read out of bounds)";
            return AGPT_ERR_INVALID;
        }
        m->indices.push_back(c.v);    // tinyobj's index_t as LoadObj copies it (trianglemesh.cpp:200-205): -1 = absent, which
        m->indices.push_back(c.vn);   // only passes the check above when the mesh has no normals / texcoords at all
        m->indices.push_back(c.vt);
    }
    if (m->indices.empty()) {
        delete m;
        g_obj_error = "agpt_obj_parse: no faces";
        return AGPT_ERR_INVALID;
    }
    *out = m;
    return AGPT_OK;
}

int agpt_obj_load(const char* path, const float* transform16, int ignore_normals, agpt_obj_mesh** out) {
    if (!path) {
        g_obj_error = "agpt_obj_load: NULL path";
        return AGPT_ERR_INVALID;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        g_obj_error = std::string("agpt_obj_load: cannot open ") + path;  // the reference calls exit(1) here
        return AGPT_ERR_INVALID;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    return agpt_obj_parse(s.data(), s.size(), transform16, ignore_normals, out);
}

int agpt_obj_counts(const agpt_obj_mesh* m, int* n_vertices, int* n_normals, int* n_texcoords, int* n_indices) {
    if (!m) return AGPT_ERR_INVALID;
    if (n_vertices) *n_vertices = (int)m->vertices.size() / 3;
    if (n_normals) *n_normals = (int)m->normals.size() / 3;
    if (n_texcoords) *n_texcoords = (int)m->texcoords.size() / 2;
    if (n_indices) *n_indices = (int)m->indices.size() / 3;
    return AGPT_OK;
}

int agpt_obj_get(const agpt_obj_mesh* m, float* vertices, float* normals, float* texcoords, int32_t* indices) {
    if (!m) return AGPT_ERR_INVALID;
    if (vertices) std::memcpy(vertices, m->vertices.data(), m->vertices.size() * 4);
    if (normals) std::memcpy(normals, m->normals.data(), m->normals.size() * 4);
    if (texcoords) std::memcpy(texcoords, m->texcoords.data(), m->texcoords.size() * 4);
    if (indices) std::memcpy(indices, m->indices.data(), m->indices.size() * 4);
    return AGPT_OK;
}

void agpt_obj_free(agpt_obj_mesh* m) { delete m; }

}  // extern "C"
