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

bool fix_index(int idx, int n, int* ret) {
    if (idx > 0) { *ret = idx - 1; return true; }
    if (idx == 0) return false;
    *ret = n + idx;
    return true;
}

// atoi as tinyobjloader uses it: optional blanks, optional sign, digits; anything else ends the number (0 if none)
int c_atoi(const char* p) { return std::atoi(p); }

// parseTriple (tiny_obj_loader.h:1093-1151): i, i/j/k, i//k, i/j
bool parse_triple(const char*& token, int vsize, int vnsize, int vtsize, Corner* ret) {
    Corner vi;
    vi.v = vi.vt = vi.vn = -1;
    if (!fix_index(c_atoi(token), vsize, &vi.v)) return false;
    token += std::strcspn(token, "/ \t\r");
    if (token[0] != '/') {
        *ret = vi;
        return true;
    }
    token++;
    if (token[0] == '/') {   // i//k
        token++;
        if (!fix_index(c_atoi(token), vnsize, &vi.vn)) return false;
        token += std::strcspn(token, "/ \t\r");
        *ret = vi;
        return true;
    }
    if (!fix_index(c_atoi(token), vtsize, &vi.vt)) return false;   // i/j/k or i/j
    token += std::strcspn(token, "/ \t\r");
    if (token[0] != '/') {
        *ret = vi;
        return true;
    }
    token++;   // i/j/k
    if (!fix_index(c_atoi(token), vnsize, &vi.vn)) return false;
    token += std::strcspn(token, "/ \t\r");
    *ret = vi;
    return true;
}

inline bool is_digit(char x) { return static_cast<unsigned int>(x - '0') < 10u; }

// tryParseDouble (tiny_obj_loader.h:836-960), operation for operation
bool try_parse_double(const char* s, const char* s_end, double* result) {
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read = 0;
    bool end_not_reached = false, leading_decimal_dots = false;
    if (*curr == '+' || *curr == '-') {
        sign = *curr;
        curr++;
        if ((curr != s_end) && (*curr == '.')) leading_decimal_dots = true;
    } else if (is_digit(*curr)) {
    } else if (*curr == '.') {
        leading_decimal_dots = true;
    } else {
        return false;
    }
    end_not_reached = (curr != s_end);
    if (!leading_decimal_dots) {
        while (end_not_reached && is_digit(*curr)) {
            mantissa *= 10;
            mantissa += static_cast<int>(*curr - 0x30);
            curr++;
            read++;
            end_not_reached = (curr != s_end);
        }
        if (read == 0) return false;
    }
    if (end_not_reached) {
        bool has_exp = false;
        if (*curr == '.') {
            curr++;
            read = 1;
            end_not_reached = (curr != s_end);
            while (end_not_reached && is_digit(*curr)) {
                static const double pow_lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
                const int lut_entries = sizeof pow_lut / sizeof pow_lut[0];
                mantissa += static_cast<int>(*curr - 0x30) * (read < lut_entries ? pow_lut[read] : std::pow(10.0, -read));
                read++;
                curr++;
                end_not_reached = (curr != s_end);
            }
            has_exp = end_not_reached;
        } else if (*curr == 'e' || *curr == 'E') {
            has_exp = true;
        }
        if (has_exp && (*curr == 'e' || *curr == 'E')) {
            curr++;
            end_not_reached = (curr != s_end);
            if (end_not_reached && (*curr == '+' || *curr == '-')) {
                exp_sign = *curr;
                curr++;
            } else if (is_digit(*curr)) {   // (reads *s_end when the token ends in `e': a blank, CR or NUL -- never a digit)
            } else {
                return false;   // empty E is not allowed
            }
            read = 0;
            end_not_reached = (curr != s_end);
            while (end_not_reached && is_digit(*curr)) {
                exponent *= 10;
                exponent += static_cast<int>(*curr - 0x30);
                curr++;
                read++;
                end_not_reached = (curr != s_end);
            }
            exponent *= (exp_sign == '+' ? 1 : -1);
            if (read == 0) return false;
        }
    }
    *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

// parseReal (tiny_obj_loader.h:962-970)
float parse_real(const char*& token, double default_value = 0.0) {
    token += std::strspn(token, " \t");
    const char* end = token + std::strcspn(token, " \t\r");
    double val = default_value;
    try_parse_double(token, end, &val);
    token = end;
    return static_cast<float>(val);
}

int pnpoly(int nvert, const float* vertx, const float* verty, float testx, float testy) {
    int c = 0;
    for (int i = 0, j = nvert - 1; i < nvert; j = i++)
        if (((verty[i] > testy) != (verty[j] > testy)) &&
            (testx < (vertx[j] - vertx[i]) * (testy - verty[i]) / (verty[j] - verty[i]) + vertx[i]))
            c = !c;
    return c;
}

// tinyobjloader 2.0.0 exportGroupsToShape's polygon path for one face
void triangulate(const std::vector<Corner>& face, const std::vector<float>& v, std::vector<Corner>& out) {
    size_t npolys = face.size();
    if (npolys < 3) return;
    size_t axes[2] = {1, 2};
    for (size_t k = 0; k < npolys; ++k) {
        size_t vi0 = (size_t)face[(k + 0) % npolys].v, vi1 = (size_t)face[(k + 1) % npolys].v, vi2 = (size_t)face[(k + 2) % npolys].v;
        if ((3 * vi0 + 2) >= v.size() || (3 * vi1 + 2) >= v.size() || (3 * vi2 + 2) >= v.size()) continue;
        float e0x = v[vi1 * 3] - v[vi0 * 3], e0y = v[vi1 * 3 + 1] - v[vi0 * 3 + 1], e0z = v[vi1 * 3 + 2] - v[vi0 * 3 + 2];
        float e1x = v[vi2 * 3] - v[vi1 * 3], e1y = v[vi2 * 3 + 1] - v[vi1 * 3 + 1], e1z = v[vi2 * 3 + 2] - v[vi1 * 3 + 2];
        float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps) {
            if (cx > cy && cx > cz) {
            } else {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    float area = 0;
    for (size_t k = 0; k < npolys; ++k) {
        size_t vi0 = (size_t)face[(k + 0) % npolys].v, vi1 = (size_t)face[(k + 1) % npolys].v;
        if ((vi0 * 3 + axes[0]) >= v.size() || (vi0 * 3 + axes[1]) >= v.size() || (vi1 * 3 + axes[0]) >= v.size() ||
            (vi1 * 3 + axes[1]) >= v.size())
            continue;
        float v0x = v[vi0 * 3 + axes[0]], v0y = v[vi0 * 3 + axes[1]], v1x = v[vi1 * 3 + axes[0]], v1y = v[vi1 * 3 + axes[1]];
        area += (v0x * v1y - v0y * v1x) * 0.5f;
    }
    std::vector<Corner> rem = face;
    size_t guess = 0;
    size_t remaining_iterations = face.size();
    size_t previous_remaining = rem.size();
    while (rem.size() > 3 && remaining_iterations > 0) {
        npolys = rem.size();
        if (guess >= npolys) guess -= npolys;
        if (previous_remaining != npolys) {
            previous_remaining = npolys;
            remaining_iterations = npolys;
        } else {
            remaining_iterations--;
        }
        Corner ind[3];
        float vx[3], vy[3];
        for (size_t k = 0; k < 3; k++) {
            ind[k] = rem[(guess + k) % npolys];
            size_t vi = (size_t)ind[k].v;
            if ((vi * 3 + axes[0]) >= v.size() || (vi * 3 + axes[1]) >= v.size()) {
                vx[k] = 0.f;
                vy[k] = 0.f;
            } else {
                vx[k] = v[vi * 3 + axes[0]];
                vy[k] = v[vi * 3 + axes[1]];
            }
        }
        float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        float cross = e0x * e1y - e0y * e1x;
        if (cross * area < 0.0f) {
            guess += 1;
            continue;
        }
        bool overlap = false;
        for (size_t other = 3; other < npolys; ++other) {
            size_t idx = (guess + other) % npolys;
            if (idx >= rem.size()) continue;
            size_t ovi = (size_t)rem[idx].v;
            if ((ovi * 3 + axes[0]) >= v.size() || (ovi * 3 + axes[1]) >= v.size()) continue;
            if (pnpoly(3, vx, vy, v[ovi * 3 + axes[0]], v[ovi * 3 + axes[1]])) {
                overlap = true;
                break;
            }
        }
        if (overlap) {
            guess += 1;
            continue;
        }
        out.push_back(ind[0]);
        out.push_back(ind[1]);
        out.push_back(ind[2]);
        size_t removed = (guess + 1) % npolys;
        while (removed + 1 < npolys) {
            rem[removed] = rem[removed + 1];
            removed += 1;
        }
        rem.pop_back();
    }
    if (rem.size() == 3) {
        out.push_back(rem[0]);
        out.push_back(rem[1]);
        out.push_back(rem[2]);
    }
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
        for (const std::vector<Corner>& face : group) triangulate(face, v, corners);
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
                if (!parse_triple(token, (int)v.size() / 3, (int)vn.size() / 3, (int)vt.size() / 2, &c)) {
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
