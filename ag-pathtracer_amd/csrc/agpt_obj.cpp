// agpt_obj.cpp -- OBJ ingestion with the semantics of the reference's TriangleMesh::LoadObj (trianglemesh.cpp:157-230),
// host-side scene prep.
//
// The reference parses with the vendored tinyobjloader 2.0.0 (tiny_obj_loader.h:26, ObjReader with the default
// triangulate=true) and then:  transforms every `v` with mat4::TransformPoint (template/precomp.h:1020-1029), every
// `vn` with TransformVector of Inverted().Transposed() (:208-217; skipped when ignore_normals), keeps `vt` as (u, v),
// and concatenates the index triplets {vertex, normal, texcoord} of all shapes in order (:200-205) -- i.e. faces in file
// order, one material for the whole mesh.  What is restated here from tinyobjloader is its published behaviour for the
// statements a triangle mesh needs: `v`, `vn`, `vt`, `f` with v, v/vt, v//vn, v/vt/vn references, 1-based and negative
// (relative) indices (fixIndex, tiny_obj_loader.h:770-790), faces with fewer than 3 corners dropped, and polygons
// triangulated by its 2-D ear clipping (exportGroupsToShape, tiny_obj_loader.h:1376-1600: projection axes from the
// first non-degenerate corner, signed area, pnpoly overlap test, fallback when no ear is found).  Everything else
// (materials, groups, smoothing, lines, points, free-form geometry) does not influence LoadObj's output and is skipped.
// PARITY UNPINNED: no OBJ fixture or reference output exists to pin this against (the reference cannot run here);
// numbers are parsed with strtod and rounded to float, tinyobjloader uses its own decimal parser.
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

// v, v/vt, v//vn, v/vt/vn
bool parse_corner(const char*& p, int nv, int nvn, int nvt, Corner* c) {
    c->v = c->vt = c->vn = -1;
    char* e;
    long iv = std::strtol(p, &e, 10);
    if (e == p || !fix_index((int)iv, nv, &c->v)) return false;
    p = e;
    if (*p != '/') return true;
    p++;
    if (*p == '/') {  // v//vn
        p++;
        long in = std::strtol(p, &e, 10);
        if (e == p || !fix_index((int)in, nvn, &c->vn)) return false;
        p = e;
        return true;
    }
    long it = std::strtol(p, &e, 10);
    if (e == p || !fix_index((int)it, nvt, &c->vt)) return false;
    p = e;
    if (*p != '/') return true;
    p++;
    long in = std::strtol(p, &e, 10);
    if (e == p || !fix_index((int)in, nvn, &c->vn)) return false;
    p = e;
    return true;
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
    std::vector<Corner> corners;  // triangulated
    std::string buf(text, length);
    std::istringstream in(buf);
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* e;
            p += 2;
            float x = (float)std::strtod(p, &e); p = e;
            float y = (float)std::strtod(p, &e); p = e;
            float z = (float)std::strtod(p, &e);
            v.push_back(x); v.push_back(y); v.push_back(z);
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            char* e;
            p += 3;
            float x = (float)std::strtod(p, &e); p = e;
            float y = (float)std::strtod(p, &e); p = e;
            float z = (float)std::strtod(p, &e);
            vn.push_back(x); vn.push_back(y); vn.push_back(z);
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            char* e;
            p += 3;
            float x = (float)std::strtod(p, &e); p = e;
            float y = (float)std::strtod(p, &e);
            vt.push_back(x); vt.push_back(y);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            std::vector<Corner> face;
            for (;;) {
                while (*p == ' ' || *p == '\t') p++;
                if (*p == '\0' || *p == '\r' || *p == '\n') break;
                Corner c;
                if (!parse_corner(p, (int)v.size() / 3, (int)vn.size() / 3, (int)vt.size() / 2, &c)) {
                    g_obj_error = "agpt_obj_parse: failed to parse `f' line " + std::to_string(lineno) + " (zero or malformed index)";
                    return AGPT_ERR_INVALID;
                }
                face.push_back(c);
                while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') p++;
            }
            triangulate(face, v, corners);
        }
    }
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
        m->indices.push_back(c.v);
        m->indices.push_back(nn > 0 ? c.vn : c.v);
        m->indices.push_back(nt > 0 ? c.vt : c.v);
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
