"""ctypes binding of libagpt_hip.so (include/agpt.h) and the host-side mirror of the reference's
Scene / Intersectable / Integrator surface for the path-tracing hot path.

The HIP library is the only compute path: if it cannot be loaded, or no GPU is present, every hot-path call
raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None

MAT_DISNEY, MAT_MIRROR, MAT_DIFFUSE_ONLY = 0, 1, 2

RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("tmax", np.float32)])
HIT_DTYPE = np.dtype([("hit", np.int32), ("prim", np.int32), ("tri", np.int32),
                      ("t", np.float32), ("b1", np.float32), ("b2", np.float32)])
NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("bmax", np.float32, 3), ("first", np.int32), ("count", np.int32)])

# every symbol include/agpt.h declares (tests check the library exports all of them)
EXPORTS = [
    "agpt_last_error", "agpt_version", "agpt_init", "agpt_set_stream", "agpt_destroy", "agpt_scene_create",
    "agpt_scene_destroy", "agpt_scene_add_material", "agpt_scene_add_mesh", "agpt_scene_add_sphere",
    "agpt_scene_add_plane", "agpt_scene_add_area_light", "agpt_scene_add_uniform_infinite_light", "agpt_scene_add_infinite_area_light",
    "agpt_scene_set_camera",
    "agpt_scene_commit", "agpt_mesh_num_nodes", "agpt_mesh_num_prims", "agpt_mesh_get_bvh", "agpt_bvh_build", "agpt_toplevel_build", "agpt_toplevel_pack16", "agpt_create_backdrop",
    "agpt_intersect_batch", "agpt_intersect_device", "agpt_render", "agpt_li_batch", "agpt_resolve", "agpt_device_alloc", "agpt_device_free",
    "agpt_device_memset", "agpt_device_download", "agpt_device_upload", "agpt_kat_bsdf_eval",
    "agpt_kat_bsdf_sample", "agpt_kat_rng", "agpt_kat_distribution1d", "agpt_dbg_li_batch", "agpt_obj_load", "agpt_obj_parse", "agpt_obj_counts", "agpt_obj_get",
    "agpt_obj_free", "agpt_obj_last_error", "agpt_write_png", "agpt_write_pfm", "agpt_hdr_load", "agpt_hdr_parse", "agpt_hdr_free",
    "agpt_comm_unique_id", "agpt_comm_init", "agpt_comm_destroy", "agpt_gather_tiles", "agpt_deinterleave_tiles",
]


class AgptError(RuntimeError):
    pass


class CameraDesc(C.Structure):
    """CameraDesc (camera.h:17-25)."""
    _fields_ = [("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
                ("aspect_ratio", C.c_float), ("vfov", C.c_float), ("aperture", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("anyhit_rays", C.c_uint64), ("interior_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("shaded_vertices", C.c_uint64), ("samples", C.c_uint64),
                ("outliers", C.c_uint64), ("iterations", C.c_uint64), ("trace_ms", C.c_double),
                ("total_ms", C.c_double), ("trace_launches", C.c_uint64), ("root_tests", C.c_uint64),
                ("ext_ms", C.c_double), ("mis_ms", C.c_double), ("shadow_ms", C.c_double), ("answered_rays", C.c_uint64)]

    @property
    def rays(self):
        return int(self.closest_rays + self.anyhit_rays)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("w", C.c_int32), ("h", C.c_int32), ("spp_begin", C.c_int32), ("spp_count", C.c_int32),
                ("seed_base", C.c_uint32), ("max_depth", C.c_int32), ("accum_pitch", C.c_int32),
                ("accum_row0", C.c_int32), ("samples_per_batch", C.c_int32), ("enable_counters", C.c_int32),
                ("enable_timing", C.c_int32), ("interleave_block", C.c_int32), ("interleave_world", C.c_int32),
                ("interleave_rank", C.c_int32), ("trace_all_rays", C.c_int32)]


def library_path():
    return _build.LIB


def lib():
    """Load libagpt_hip.so (building it with hipcc if missing or stale)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    variant = os.environ.get("AGPT_LIB_VARIANT")   # developer A/B builds (tools/build_variant.py), never set in production
    if variant:
        path = os.path.join(os.path.dirname(path), "libagpt_hip_%s.so" % variant)
        if not os.path.exists(path):
            raise AgptError("AGPT_LIB_VARIANT=%s: %s does not exist" % (variant, path))
    elif not os.path.exists(path) or (os.path.exists(_build.CSRC) and _build.needs_build()):
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(path):
                raise AgptError("libagpt_hip.so is missing and could not be built with hipcc: %s" % e)
    L = C.CDLL(path)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    L.agpt_last_error.restype = C.c_char_p
    L.agpt_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.agpt_set_stream.argtypes = [vp, vp]
    L.agpt_destroy.argtypes = [vp]
    L.agpt_destroy.restype = None
    L.agpt_scene_create.argtypes = [vp, C.POINTER(vp)]
    L.agpt_scene_destroy.argtypes = [vp]
    L.agpt_scene_destroy.restype = None
    L.agpt_scene_add_material.argtypes = [vp, C.c_int, fp, C.c_float, C.c_float]
    L.agpt_scene_add_mesh.argtypes = [vp, fp, C.c_int, fp, C.c_int, fp, C.c_int, ip, C.c_int, C.c_int, C.c_int]
    L.agpt_scene_add_sphere.argtypes = [vp, fp, C.c_float, C.c_int]
    L.agpt_scene_add_plane.argtypes = [vp, fp, fp, C.c_int]
    L.agpt_scene_add_area_light.argtypes = [vp, fp, C.c_float, fp]
    L.agpt_scene_add_uniform_infinite_light.argtypes = [vp, fp]
    L.agpt_scene_add_infinite_area_light.argtypes = [vp, fp, C.c_int, C.c_int]
    L.agpt_scene_set_camera.argtypes = [vp, C.POINTER(CameraDesc)]
    L.agpt_scene_commit.argtypes = [vp]
    L.agpt_mesh_num_nodes.argtypes = [vp, C.c_int]
    L.agpt_mesh_num_prims.argtypes = [vp, C.c_int]
    L.agpt_mesh_get_bvh.argtypes = [vp, C.c_int, vp, ip]
    L.agpt_bvh_build.argtypes = [fp, C.c_int, ip, C.c_int, C.c_int, vp, ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.agpt_toplevel_build.argtypes = [fp, C.c_int, fp]
    L.agpt_toplevel_pack16.argtypes = [fp, C.c_int, C.POINTER(C.c_uint32)]
    L.agpt_create_backdrop.argtypes = [fp, fp, C.c_float, C.c_int, fp, fp, fp, ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.agpt_intersect_batch.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.POINTER(Stats)]
    L.agpt_intersect_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.POINTER(Stats)]
    L.agpt_render.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    L.agpt_li_batch.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.c_int, C.c_int, fp, C.POINTER(C.c_uint32), C.POINTER(Stats)]
    L.agpt_resolve.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.agpt_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.agpt_device_free.argtypes = [vp, vp]
    L.agpt_device_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    L.agpt_device_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.agpt_device_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.agpt_kat_bsdf_eval.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp]
    L.agpt_kat_bsdf_sample.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp, fp, ip]
    L.agpt_kat_rng.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, fp, C.POINTER(C.c_uint32)]
    L.agpt_dbg_li_batch.argtypes = [vp, vp, C.c_int, fp]
    L.agpt_kat_distribution1d.argtypes = [vp, fp, C.c_int, fp, C.c_int, fp, fp, fp, fp]
    L.agpt_obj_load.argtypes = [C.c_char_p, fp, C.c_int, C.POINTER(vp)]
    L.agpt_obj_parse.argtypes = [C.c_char_p, C.c_size_t, fp, C.c_int, C.POINTER(vp)]
    L.agpt_obj_counts.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
    L.agpt_obj_get.argtypes = [vp, fp, fp, fp, ip]
    L.agpt_obj_free.argtypes = [vp]
    L.agpt_obj_free.restype = None
    L.agpt_comm_unique_id.argtypes = [vp]
    L.agpt_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.agpt_comm_destroy.argtypes = [vp]
    L.agpt_comm_destroy.restype = None
    L.agpt_gather_tiles.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.agpt_deinterleave_tiles.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.agpt_obj_last_error.restype = C.c_char_p
    L.agpt_hdr_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(fp)]
    L.agpt_hdr_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(fp)]
    L.agpt_hdr_free.argtypes = [fp]
    L.agpt_hdr_free.restype = None
    _LIB = L
    return L


def _check(rc, what=""):
    if rc < 0:
        raise AgptError("%s failed (%d): %s" % (what, rc, lib().agpt_last_error().decode()))
    return rc


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def create_backdrop(origin, size, radius, steps):
    """TriangleMesh::CreateBackdrop (trianglemesh.cpp:232-318) -> verts[n,3], normals[n,3], uvs[n,2], indices[m,3]."""
    nv = 2 * (steps + 5)
    verts = np.zeros((nv, 3), np.float32)
    normals = np.zeros((nv, 3), np.float32)
    uvs = np.zeros((nv, 2), np.float32)
    idx = np.zeros((6 * (steps + 4), 3), np.int32)
    _, po = _f(origin)
    _, ps = _f(size)
    n_v, n_i = C.c_int(0), C.c_int(0)
    fp = C.POINTER(C.c_float)
    _check(lib().agpt_create_backdrop(po, ps, float(radius), int(steps), verts.ctypes.data_as(fp),
                                      normals.ctypes.data_as(fp), uvs.ctypes.data_as(fp),
                                      idx.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_v), C.byref(n_i)),
           "agpt_create_backdrop")
    return verts, normals, uvs, idx


def toplevel_build(boxes):
    """agpt_toplevel_build: boxes[n, 6] (bmin, bmax) -> (lo[m, 3], hi[m, 3], skip[m], leaf[m]) with m = 2n - 1 nodes in depth-first
    order; leaf[i] = index of the box at a leaf, -1 at an interior node; skip[i] = the node after i's subtree."""
    boxes, pb = _f(np.asarray(boxes).reshape(-1, 6))
    n = boxes.shape[0]
    out = np.zeros((max(2 * n - 1, 1), 8), np.float32)
    m = _check(lib().agpt_toplevel_build(pb, n, out.ctypes.data_as(C.POINTER(C.c_float))), "agpt_toplevel_build")
    assert m == 2 * n - 1
    words = out.view(np.uint32)
    return out[:, 0:3].copy(), out[:, 4:7].copy(), words[:, 3].astype(np.int64), words[:, 7].astype(np.int32)


def toplevel_pack16(lo, hi, skip, leaf):
    """agpt_toplevel_pack16: the 16-byte node form of a tree from toplevel_build -> (lo16[m, 3], hi16[m, 3] as float16, skip[m], leaf[m])."""
    m = len(skip)
    nodes = np.zeros((m, 8), np.float32)
    nodes[:, 0:3], nodes[:, 4:7] = lo, hi
    w = nodes.view(np.uint32)
    w[:, 3] = np.asarray(skip, np.uint32)
    w[:, 7] = np.asarray(leaf, np.int64).astype(np.uint32)   # -1 -> 0xFFFFFFFF
    packed = np.zeros((m, 4), np.uint32)
    _check(lib().agpt_toplevel_pack16(nodes.ctypes.data_as(C.POINTER(C.c_float)), m, packed.ctypes.data_as(C.POINTER(C.c_uint32))),
           "agpt_toplevel_pack16")
    halves = np.stack([packed[:, 0] & 0xFFFF, packed[:, 0] >> 16, packed[:, 1] & 0xFFFF, packed[:, 1] >> 16, packed[:, 2] & 0xFFFF,
                       packed[:, 2] >> 16], axis=1).astype(np.uint16).view(np.float16)
    leaf16 = (packed[:, 3] >> 16).astype(np.int32)
    leaf16[leaf16 == 0xFFFF] = -1
    return halves[:, 0:3].copy(), halves[:, 3:6].copy(), (packed[:, 3] & 0xFFFF).astype(np.int64), leaf16


def bvh_build(verts, indices, max_prims_in_node=1):
    """BVHTriMesh's constructor on the host (no GPU needed) -> (nodes[total+1], prim_index[n_tris], max_depth)."""
    v, pv = _f(np.asarray(verts).reshape(-1, 3))
    ix, pi = _i(np.asarray(indices).reshape(-1, 3))
    n_tris = ix.shape[0] // 3
    nodes = np.zeros(2 * n_tris + 2, NODE_DTYPE)
    order = np.zeros(n_tris, np.int32)
    total, depth = C.c_int(0), C.c_int(0)
    _check(lib().agpt_bvh_build(pv, v.shape[0], pi, ix.shape[0], int(max_prims_in_node), nodes.ctypes.data_as(C.c_void_p),
                                order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), C.byref(depth)),
           "agpt_bvh_build")
    return nodes[:total.value + 1].copy(), order, depth.value


def load_obj(path=None, text=None, transform=None, ignore_normals=False):
    """TriangleMesh::LoadObj (trianglemesh.cpp:157-230): returns (verts[n,3], normals[m,3] or None, uvs[k,2] or None,
    indices[3*tris,3]) ready for Scene.add_mesh.  `transform` is a row-major 4x4 (mat4)."""
    L = lib()
    h = C.c_void_p()
    tp = None
    if transform is not None:
        t = np.ascontiguousarray(transform, np.float32).reshape(16)
        tp = t.ctypes.data_as(C.POINTER(C.c_float))
    if text is not None:
        data = text.encode() if isinstance(text, str) else bytes(text)
        rc = L.agpt_obj_parse(data, len(data), tp, int(bool(ignore_normals)), C.byref(h))
    else:
        rc = L.agpt_obj_load(os.fsencode(path), tp, int(bool(ignore_normals)), C.byref(h))
    if rc < 0:
        raise AgptError("load_obj failed (%d): %s" % (rc, L.agpt_obj_last_error().decode()))
    try:
        nv, nn, nt, ni = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.agpt_obj_counts(h, C.byref(nv), C.byref(nn), C.byref(nt), C.byref(ni))
        v = np.zeros((nv.value, 3), np.float32)
        n = np.zeros((nn.value, 3), np.float32)
        t = np.zeros((nt.value, 2), np.float32)
        ix = np.zeros((ni.value, 3), np.int32)
        fp = C.POINTER(C.c_float)
        L.agpt_obj_get(h, v.ctypes.data_as(fp), n.ctypes.data_as(fp), t.ctypes.data_as(fp), ix.ctypes.data_as(C.POINTER(C.c_int32)))
    finally:
        L.agpt_obj_free(h)
    return v, (n if nn.value else None), (t if nt.value else None), ix


class Context:
    """One GPU. `stream` may be a raw hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.agpt_init(int(device), C.byref(h)), "agpt_init")
        self.h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        _check(self.L.agpt_set_stream(self.h, C.c_void_p(int(stream) if stream else 0)), "agpt_set_stream")

    def close(self):
        if self.h:
            self.L.agpt_destroy(self.h)
            self.h = None

    def alloc(self, nbytes):
        p = C.c_void_p()
        _check(self.L.agpt_device_alloc(self.h, nbytes, C.byref(p)), "agpt_device_alloc")
        return p.value

    def free(self, ptr):
        _check(self.L.agpt_device_free(self.h, C.c_void_p(ptr)), "agpt_device_free")

    def memset(self, ptr, value, nbytes):
        _check(self.L.agpt_device_memset(self.h, C.c_void_p(ptr), value, nbytes), "agpt_device_memset")

    def download(self, ptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        _check(self.L.agpt_device_download(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes),
               "agpt_device_download")
        return out

    def upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(self.L.agpt_device_upload(self.h, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes),
               "agpt_device_upload")

    def rng_floats(self, pixel, wh, sample, seed_base, n):
        out = np.zeros(n, np.float32)
        seed = C.c_uint32(0)
        _check(self.L.agpt_kat_rng(self.h, pixel, wh, sample, seed_base, n, out.ctypes.data_as(C.POINTER(C.c_float)),
                                   C.byref(seed)), "agpt_kat_rng")
        return out, seed.value

    def distribution1d(self, func, u):
        """Distribution1D(func) and SampleContinuous(u[i]) (sampling.h:19-52): (cdf[n + 1], funcInt, x[k], pdf[k])."""
        func = np.ascontiguousarray(func, np.float32)
        u = np.ascontiguousarray(u, np.float32)
        cdf = np.zeros(func.size + 1, np.float32)
        x = np.zeros(max(u.size, 1), np.float32)
        pdf = np.zeros(max(u.size, 1), np.float32)
        fi = C.c_float(0)
        P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _check(self.L.agpt_kat_distribution1d(self.h, P(func), func.size, P(u), u.size, P(cdf), C.byref(fi), P(x), P(pdf)),
               "agpt_kat_distribution1d")
        return cdf, np.float32(fi.value), x[:u.size], pdf[:u.size]

    def deinterleave_tiles(self, compact_ptr, W, H, block_rows, world, rank, full_ptr):
        _check(self.L.agpt_deinterleave_tiles(self.h, C.c_void_p(int(compact_ptr)), W, H, block_rows, world, rank, C.c_void_p(int(full_ptr))),
               "agpt_deinterleave_tiles")

    def resolve(self, accum_ptr, n_pixels, samples):
        out = np.zeros(n_pixels, np.uint32)
        _check(self.L.agpt_resolve(self.h, C.c_void_p(accum_ptr), n_pixels, samples,
                                   out.ctypes.data_as(C.POINTER(C.c_uint32))), "agpt_resolve")
        return out


def comm_unique_id():
    """agpt_comm_unique_id: 128 bytes (ncclUniqueId) that rank 0 hands to the other ranks."""
    buf = C.create_string_buffer(128)
    _check(lib().agpt_comm_unique_id(C.cast(buf, C.c_void_p)), "agpt_comm_unique_id")
    return buf.raw


class Comm:
    """agpt_comm: the RCCL communicator of a multi-GPU render (one rank per process / GPU)."""

    def __init__(self, ctx, world=1, rank=0, unique_id=None):
        self.ctx = ctx
        self.L = ctx.L
        h = C.c_void_p()
        idp = C.c_char_p(unique_id) if unique_id is not None else None
        _check(self.L.agpt_comm_init(ctx.h, C.cast(idp, C.c_void_p) if idp else None, int(world), int(rank), C.byref(h)), "agpt_comm_init")
        self.h = h

    def gather_tiles(self, local_ptr, W, H, block_rows, full_ptr):
        _check(self.L.agpt_gather_tiles(self.h, C.c_void_p(int(local_ptr)), W, H, block_rows, C.c_void_p(int(full_ptr) if full_ptr else 0)),
               "agpt_gather_tiles")

    def close(self):
        if self.h:
            self.L.agpt_comm_destroy(self.h)
            self.h = None


def write_png(path, rgb_words, width, height):
    """agpt_write_png: 0x00RRGGBB words (Context.resolve output), top row first."""
    a = np.ascontiguousarray(rgb_words, dtype=np.uint32)
    assert a.size == width * height
    _check(lib().agpt_write_png(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_uint32)), width, height), "agpt_write_png")


def load_hdr(path=None, data=None):
    """HDRTexture's pixels (texture.h:41-52: stbi_loadf of a Radiance .hdr) as float32 [H, W, 3], top row first -- from a file
    or from the file's bytes.  Feed it to Scene.add_infinite_area_light."""
    L = lib()
    w, h = C.c_int(0), C.c_int(0)
    rgb = C.POINTER(C.c_float)()
    if data is not None:
        buf = bytes(data)
        _check(L.agpt_hdr_parse(buf, len(buf), C.byref(w), C.byref(h), C.byref(rgb)), "agpt_hdr_parse")
    else:
        _check(L.agpt_hdr_load(os.fsencode(path), C.byref(w), C.byref(h), C.byref(rgb)), "agpt_hdr_load")
    try:
        return np.ctypeslib.as_array(rgb, (h.value, w.value, 3)).copy()
    finally:
        L.agpt_hdr_free(rgb)


def write_pfm(path, accum_host, samples):
    """agpt_write_pfm: linear float image sum/samples from a host copy of the float4 accumulator [H, W, 4]."""
    a = np.ascontiguousarray(accum_host, dtype=np.float32)
    h, w, _ = a.shape
    _check(lib().agpt_write_pfm(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_float)), w, h, samples), "agpt_write_pfm")


class Scene:
    """GPU-resident Scene (scene.h:3-30): primitives in insertion order, lights, camera."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        h = C.c_void_p()
        _check(self.L.agpt_scene_create(ctx.h, C.byref(h)), "agpt_scene_create")
        self.h = h
        self.n_lights = 0

    def close(self):
        if self.h:
            self.L.agpt_scene_destroy(self.h)
            self.h = None

    def add_material(self, mtype, color, roughness=0.5, metallic=0.0):
        _, p = _f(color)
        return _check(self.L.agpt_scene_add_material(self.h, int(mtype), p, float(roughness), float(metallic)),
                      "agpt_scene_add_material")

    def add_mesh(self, verts, normals, uvs, indices, material, max_prims_in_node=1):
        v, pv = _f(np.asarray(verts).reshape(-1, 3))
        n = np.zeros((0, 3), np.float32) if normals is None else np.asarray(normals).reshape(-1, 3)
        n, pn = _f(n)
        t = np.zeros((0, 2), np.float32) if uvs is None else np.asarray(uvs).reshape(-1, 2)
        t, pt = _f(t)
        ix, pi = _i(np.asarray(indices).reshape(-1, 3))
        return _check(self.L.agpt_scene_add_mesh(self.h, pv, v.shape[0], pn, n.shape[0], pt, t.shape[0], pi,
                                                 ix.shape[0], int(material), int(max_prims_in_node)),
                      "agpt_scene_add_mesh")

    def add_sphere(self, center, radius, material):
        _, p = _f(center)
        return _check(self.L.agpt_scene_add_sphere(self.h, p, float(radius), int(material)), "agpt_scene_add_sphere")

    def add_plane(self, o, size, material):
        """scene->primitives.push_back(make_shared<Plane>(o, size, material)) (intersectable.h:119-157)."""
        _, po = _f(o)
        _, ps = _f(size)
        return _check(self.L.agpt_scene_add_plane(self.h, po, ps, int(material)), "agpt_scene_add_plane")

    def addAreaLight(self, center, radius, L):
        """Scene::addAreaLight(make_shared<Sphere>(center, radius, nullptr), L)."""
        _, p = _f(center)
        _, pl = _f(L)
        self.n_lights += 1
        return _check(self.L.agpt_scene_add_area_light(self.h, p, float(radius), pl), "agpt_scene_add_area_light")

    add_area_light = addAreaLight

    def add_uniform_infinite_light(self, L):
        _, pl = _f(L)
        self.n_lights += 1
        return _check(self.L.agpt_scene_add_uniform_infinite_light(self.h, pl), "agpt_scene_add_uniform_infinite_light")

    def add_infinite_area_light(self, rgb):
        """InfiniteAreaLight from an in-memory HDR image rgb[H, W, 3] (lights.cpp:31-48)."""
        img = np.ascontiguousarray(rgb, np.float32)
        self.n_lights += 1
        return _check(self.L.agpt_scene_add_infinite_area_light(self.h, img.ctypes.data_as(C.POINTER(C.c_float)),
                                                                img.shape[1], img.shape[0]),
                      "agpt_scene_add_infinite_area_light")

    def set_camera(self, lookfrom, lookat, vup, aspect_ratio, vfov=45.0, aperture=0.0):
        d = CameraDesc()
        d.lookfrom[:] = [float(x) for x in np.float32(lookfrom)]
        d.lookat[:] = [float(x) for x in np.float32(lookat)]
        d.vup[:] = [float(x) for x in np.float32(vup)]
        d.aspect_ratio = float(np.float32(aspect_ratio))
        d.vfov = float(vfov)
        d.aperture = float(aperture)
        _check(self.L.agpt_scene_set_camera(self.h, C.byref(d)), "agpt_scene_set_camera")

    def commit(self):
        _check(self.L.agpt_scene_commit(self.h), "agpt_scene_commit")

    def bvh(self, prim):
        nn = _check(self.L.agpt_mesh_num_nodes(self.h, prim), "agpt_mesh_num_nodes")
        npr = _check(self.L.agpt_mesh_num_prims(self.h, prim), "agpt_mesh_num_prims")
        nodes = np.zeros(nn + 1, NODE_DTYPE)
        order = np.zeros(npr, np.int32)
        _check(self.L.agpt_mesh_get_bvh(self.h, prim, nodes.ctypes.data_as(C.c_void_p),
                                        order.ctypes.data_as(C.POINTER(C.c_int32))), "agpt_mesh_get_bvh")
        return nodes, order

    def _intersect(self, rays, any_hit, counters):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        st = Stats()
        _check(self.L.agpt_intersect_batch(self.h, rays.ctypes.data_as(C.c_void_p), rays.shape[0],
                                           out.ctypes.data_as(C.c_void_p), int(any_hit),
                                           C.byref(st) if counters else None),
               "agpt_intersect_batch")
        return out, st

    def intersect_device(self, rays_ptr, n, out_ptr, any_hit=False, counters=False):
        """agpt_intersect_device: rays / hits stay in HBM (pointers from Context.alloc or the host's own allocations)."""
        st = Stats()
        _check(self.L.agpt_intersect_device(self.h, C.c_void_p(rays_ptr), int(n), C.c_void_p(out_ptr), int(any_hit),
                                            C.byref(st) if counters else None), "agpt_intersect_device")
        return st

    def Intersect(self, rays, counters=False):
        """Scene::Intersect for a batch of rays (scene.h:5-13).  counters=True runs the instrumented kernel
        (work counters in the returned stats); the default runs the production kernel."""
        return self._intersect(rays, 0, counters)

    def IntersectP(self, rays, counters=False):
        """Scene::IntersectP for a batch of rays (scene.h:15-19)."""
        return self._intersect(rays, 1, counters)

    def bsdf_eval(self, material, wo, wi):
        wo, pwo = _f(np.asarray(wo).reshape(-1, 3))
        wi, pwi = _f(np.asarray(wi).reshape(-1, 3))
        n = wo.shape[0]
        f = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        fp = C.POINTER(C.c_float)
        _check(self.L.agpt_kat_bsdf_eval(self.h, material, n, pwo, pwi, f.ctypes.data_as(fp), pdf.ctypes.data_as(fp)),
               "agpt_kat_bsdf_eval")
        return f, pdf

    def bsdf_sample(self, material, wo, u):
        wo, pwo = _f(np.asarray(wo).reshape(-1, 3))
        u, pu = _f(np.asarray(u).reshape(-1, 2))
        n = wo.shape[0]
        wi = np.zeros((n, 3), np.float32)
        f = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        spec = np.zeros(n, np.int32)
        fp = C.POINTER(C.c_float)
        _check(self.L.agpt_kat_bsdf_sample(self.h, material, n, pwo, pu, wi.ctypes.data_as(fp), f.ctypes.data_as(fp),
                                           pdf.ctypes.data_as(fp), spec.ctypes.data_as(C.POINTER(C.c_int32))),
               "agpt_kat_bsdf_sample")
        return wi, f, pdf, spec


class PathTracer:
    """PathTracer (integrator.h:120-196) driving the per-pixel loop of MyApp::Tick (myapp.cpp:163-175) on the GPU."""

    def __init__(self, maxDepth=5):
        self.MaxDepth = int(maxDepth)

    def render(self, scene, W, H, spp, accum_ptr, tile=None, spp_begin=0, seed_base=0, accum_pitch=None, accum_row0=0,
               samples_per_batch=0, counters=False, timing=False, want_stats=True, interleave=None, trace_all_rays=False):
        """Adds `spp` samples per pixel of `tile` (x0, y0, w, h) into the DEVICE float4 buffer at accum_ptr.
        interleave=(block_rows, world, rank) renders this rank's row blocks of the whole film into its compact buffer
        (see tiles.py) in one call.  counters: False/0 = off, True/1 = the reference-order instrumented kernel (the
        reference recursion's visit counts), 2 = the production trace kernel counting the records it fetches itself."""
        x0, y0, w, h = tile if tile is not None else (0, 0, W, H)
        il = interleave if interleave is not None else (0, 0, 0)
        rp = RenderParams(W, H, x0, y0, w, h, spp_begin, spp, seed_base & 0xFFFFFFFF, self.MaxDepth,
                          accum_pitch if accum_pitch is not None else W, accum_row0, samples_per_batch,
                          int(counters), 1 if timing else 0, il[0], il[1], il[2], 1 if trace_all_rays else 0)
        st = Stats()
        _check(scene.L.agpt_render(scene.h, C.byref(rp), C.c_void_p(int(accum_ptr)), C.byref(st) if want_stats else None),
               "agpt_render")
        return st

    def Li(self, scene, rays, rng_states):
        """Integrator::Li (integrator.h:28-31, 120-191) for a batch of rays (RAY_DTYPE) with one xorshift32 state each:
        returns (radiance [n, 3] float32 -- unfiltered --, the states after the paths, stats)."""
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        states = np.ascontiguousarray(rng_states, np.uint32)
        n = len(rays)
        assert states.shape == (n,)
        out = np.zeros((n, 3), np.float32)
        after = np.zeros(n, np.uint32)
        st = Stats()
        u32 = C.POINTER(C.c_uint32)
        _check(scene.L.agpt_li_batch(scene.h, rays.ctypes.data_as(C.c_void_p), states.ctypes.data_as(u32), n, self.MaxDepth,
                                     out.ctypes.data_as(C.POINTER(C.c_float)), after.ctypes.data_as(u32), C.byref(st)), "agpt_li_batch")
        return out, after, st

    @staticmethod
    def DbgLi(scene, rays):
        """DbgIntegrator::Li (integrator.h:107-118) for a batch of rays: (u, v, 0) / 5 of the hit's texture coordinates, red where
        u or v is 0, black on a miss; float32 [n, 3]."""
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        out = np.zeros((len(rays), 3), np.float32)
        _check(scene.L.agpt_dbg_li_batch(scene.h, rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.POINTER(C.c_float))),
               "agpt_dbg_li_batch")
        return out

    def render_to_host(self, scene, W, H, spp, **kw):
        """Convenience for tests: allocates a zeroed accumulator, renders, returns (accum[H,W,4], stats)."""
        ctx = scene.ctx
        nbytes = W * H * 16
        ptr = ctx.alloc(nbytes)
        try:
            ctx.memset(ptr, 0, nbytes)
            st = self.render(scene, W, H, spp, ptr, **kw)
            acc = ctx.download(ptr, (H, W, 4))
        finally:
            ctx.free(ptr)
        return acc, st
