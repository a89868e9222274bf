// oracle/ref_tinyobj.cpp -- TEST INFRASTRUCTURE ONLY (never linked by the product).
//
// Harness around the reference's own vendored tinyobjloader, compiled from the header where it lies
// (/root/reference/tiny_obj_loader.h, `make -C oracle ref`), output oracle/_ref/libref_tinyobj.so.  It runs exactly what
// TriangleMesh::LoadObj runs before its own post-processing (trianglemesh.cpp:160-178,191-205): ObjReader::ParseFromFile with
// a default ObjReaderConfig (triangulate = true), then the concatenation of every shape's mesh.indices as
// {vertex_index, normal_index, texcoord_index} triplets.  The mat4 transform of LoadObj lives in template/precomp.h, which
// cannot be compiled here, so the harness returns the untransformed attrib arrays; the transform is
// covered by tests/test_obj.py's restatement.  tests/golden/make_obj_golden.py turns its outputs into fixtures.
#define TINYOBJLOADER_IMPLEMENTATION
#define TINYOBJLOADER_USE_MAPBOX_EARCUT   // as myapp.cpp:10 (inert in this version of the header)
#include "tiny_obj_loader.h"

#include <cstring>

extern "C" {

// returns 1 on success, 0 when ParseFromFile fails (the reference then calls exit(1)); message (error or warning) in msg
int ref_tinyobj_load(const char* path, float** vertices, int* n_vertices, float** normals, int* n_normals, float** texcoords,
                     int* n_texcoords, int** indices, int* n_indices, int* n_shapes, char* msg, int msg_len) {
    tinyobj::ObjReaderConfig reader_config;
    tinyobj::ObjReader reader;
    const bool ok = reader.ParseFromFile(path, reader_config);
    std::string m = reader.Error() + reader.Warning();
    if (msg && msg_len > 0) {
        std::strncpy(msg, m.c_str(), (size_t)msg_len - 1);
        msg[msg_len - 1] = 0;
    }
    if (!ok) return 0;
    const tinyobj::attrib_t& attrib = reader.GetAttrib();
    const std::vector<tinyobj::shape_t>& shapes = reader.GetShapes();
    auto copy = [](const std::vector<tinyobj::real_t>& src, float** dst, int* n, int width) {
        *n = (int)(src.size() / (size_t)width);
        *dst = (float*)std::malloc(sizeof(float) * (src.size() + 1));
        for (size_t i = 0; i < src.size(); i++) (*dst)[i] = src[i];
    };
    copy(attrib.vertices, vertices, n_vertices, 3);
    copy(attrib.normals, normals, n_normals, 3);
    copy(attrib.texcoords, texcoords, n_texcoords, 2);
    size_t total = 0;
    for (const auto& s : shapes) total += s.mesh.indices.size();
    *indices = (int*)std::malloc(sizeof(int) * (3 * total + 1));
    size_t k = 0;
    for (const auto& s : shapes)
        for (const auto& idx : s.mesh.indices) {
            (*indices)[k++] = idx.vertex_index;
            (*indices)[k++] = idx.normal_index;
            (*indices)[k++] = idx.texcoord_index;
        }
    *n_indices = (int)total;
    *n_shapes = (int)shapes.size();
    return 1;
}

void ref_tinyobj_free(void* p) { std::free(p); }

}  // extern "C"
