"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package (ag-pathtracer_amd/).  See oracle/agpt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAT_DISNEY, MAT_MIRROR, MAT_DIFFUSE_ONLY = 0, 1, 2
RNG_PER_SAMPLE, RNG_GLOBAL = 0, 1


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "closest_rays", "anyhit_rays", "box_tests", "interior_visits", "tri_tests",
        "shaded_vertices", "samples", "outliers")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self):
        return int(self.closest_rays + self.anyhit_rays)


RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("tmax", np.float32)])
HIT_DTYPE = np.dtype([("hit", np.int32), ("prim", np.int32), ("tri", np.int32),
                      ("t", np.float32), ("b1", np.float32), ("b2", np.float32)])
NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("bmax", np.float32, 3), ("first", np.int32), ("count", np.int32)])


def build(force=False):
    """Compile oracle/liboracle.so with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "agpt_oracle.h", "oracle_math.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int32)
    L.oracle_scene_new.restype = C.c_void_p
    L.oracle_scene_free.argtypes = [C.c_void_p]
    L.oracle_add_material.argtypes = [C.c_void_p, C.c_int, fp, C.c_float, C.c_float]
    L.oracle_add_mesh.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_int, fp, C.c_int, ip, C.c_int, C.c_int, C.c_int]
    L.oracle_add_sphere.argtypes = [C.c_void_p, fp, C.c_float, C.c_int]
    L.oracle_add_plane.argtypes = [C.c_void_p, fp, fp, C.c_int]
    L.oracle_add_area_light.argtypes = [C.c_void_p, fp, C.c_float, fp]
    L.oracle_add_uniform_infinite_light.argtypes = [C.c_void_p, fp]
    L.oracle_add_infinite_area_light.argtypes = [C.c_void_p, fp, C.c_int, C.c_int]
    L.oracle_set_camera.argtypes = [C.c_void_p, fp, fp, fp, C.c_float, C.c_float, C.c_float]
    L.oracle_set_max_depth.argtypes = [C.c_void_p, C.c_int]
    L.oracle_mesh_num_nodes.argtypes = [C.c_void_p, C.c_int]
    L.oracle_mesh_num_prims.argtypes = [C.c_void_p, C.c_int]
    L.oracle_mesh_get_bvh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, ip]
    L.oracle_intersect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Stats)]
    L.oracle_render.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_uint32, C.c_int, C.c_int, fp, C.POINTER(Stats)]
    L.oracle_wang_hash.argtypes = [C.c_uint32]
    L.oracle_wang_hash.restype = C.c_uint32
    L.oracle_sample_seed.argtypes = [C.c_uint32] * 4
    L.oracle_sample_seed.restype = C.c_uint32
    L.oracle_rng_floats.argtypes = [C.c_uint32, C.c_int, fp, C.POINTER(C.c_uint32)]
    L.oracle_dist1d_build.argtypes = [fp, C.c_int, fp, fp]
    L.oracle_dist1d_sample.argtypes = [fp, fp, C.c_float, C.c_int, fp, C.c_int, fp, fp, ip]
    L.oracle_bounds_intersect.argtypes = [fp, fp, C.c_void_p, fp]
    L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
    L.oracle_bsdf_sample.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, fp, C.POINTER(C.c_int)]
    L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_uint32), C.c_void_p]
    L.oracle_dbg_li.argtypes = [C.c_void_p, C.c_void_p, fp]
    L.oracle_li.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), fp, C.POINTER(Stats)]
    L.oracle_create_backdrop.argtypes = [fp, fp, C.c_float, C.c_int, fp, fp, fp, ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oracle_pin_heightfield.argtypes = [C.c_int, C.c_float, fp, fp, fp, ip]
    L.oracle_set_trig_mode.argtypes = [C.c_int]
    L.oracle_hex2lin.argtypes = [C.c_int, fp]
    L.oracle_rgb2lin.argtypes = [fp, fp]
    L.oracle_copy_to_surface.argtypes = [fp, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    _LIB = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def hex2lin(h):
    out = np.zeros(3, np.float32)
    lib().oracle_hex2lin(int(h), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def rgb2lin(rgb):
    a, p = _f(rgb)
    out = np.zeros(3, np.float32)
    lib().oracle_rgb2lin(p, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def copy_to_surface(accum, samples):
    """Accumulator::CopyToSurface: accum[..., 4] float sums -> uint32 0x00RRGGBB per pixel (same leading shape)."""
    a, p = _f(accum)
    assert a.shape[-1] == 4
    out = np.zeros(a.shape[:-1], np.uint32)
    lib().oracle_copy_to_surface(p, int(out.size), int(samples), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def create_backdrop(origin, size, radius, steps):
    """TriangleMesh::CreateBackdrop -> (verts[n,3], normals[n,3], uvs[n,2], indices[m,3])."""
    nv = 2 * (steps + 5)
    verts = np.zeros((nv, 3), np.float32)
    normals = np.zeros((nv, 3), np.float32)
    uvs = np.zeros((nv, 2), np.float32)
    idx = np.zeros((6 * (steps + 4), 3), np.int32)
    o, po = _f(origin)
    s, ps = _f(size)
    n_v, n_i = C.c_int(0), C.c_int(0)
    lib().oracle_create_backdrop(po, ps, float(radius), int(steps), verts.ctypes.data_as(C.POINTER(C.c_float)),
                                 normals.ctypes.data_as(C.POINTER(C.c_float)), uvs.ctypes.data_as(C.POINTER(C.c_float)),
                                 idx.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_v), C.byref(n_i))
    assert n_v.value == nv and n_i.value == idx.shape[0]
    return verts, normals, uvs, idx


def pin_heightfield(n, S=2.0):
    """Heightfield of the survey's reference runs -> (verts, normals, uvs, indices)."""
    nv = (n + 1) * (n + 1)
    verts = np.zeros((nv, 3), np.float32)
    normals = np.zeros((nv, 3), np.float32)
    uvs = np.zeros((nv, 2), np.float32)
    idx = np.zeros((6 * n * n, 3), np.int32)
    lib().oracle_pin_heightfield(int(n), float(S), verts.ctypes.data_as(C.POINTER(C.c_float)),
                                 normals.ctypes.data_as(C.POINTER(C.c_float)), uvs.ctypes.data_as(C.POINTER(C.c_float)),
                                 idx.ctypes.data_as(C.POINTER(C.c_int32)))
    return verts, normals, uvs, idx


class OracleScene:
    """Mirror of the reference's Scene-building calls (myapp.cpp:13-114) against the C oracle."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.oracle_scene_new())
        self._keep = []

    def __del__(self):
        try:
            if self.h:
                self.L.oracle_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def add_material(self, mtype, color, roughness=0.5, metallic=0.0):
        c, p = _f(color)
        return self.L.oracle_add_material(self.h, int(mtype), p, float(roughness), float(metallic))

    def add_mesh(self, verts, normals, uvs, indices, material, max_prims_in_node=1):
        v, pv = _f(np.asarray(verts).reshape(-1, 3))
        n = np.zeros((0, 3), np.float32) if normals is None else np.asarray(normals).reshape(-1, 3)
        n, pn = _f(n)
        t = np.zeros((0, 2), np.float32) if uvs is None else np.asarray(uvs).reshape(-1, 2)
        t, pt = _f(t)
        ix, pi = _i(np.asarray(indices).reshape(-1, 3))
        return self.L.oracle_add_mesh(self.h, pv, v.shape[0], pn, n.shape[0], pt, t.shape[0], pi, ix.shape[0],
                                      int(material), int(max_prims_in_node))

    def add_sphere(self, center, radius, material):
        c, p = _f(center)
        return self.L.oracle_add_sphere(self.h, p, float(radius), int(material))

    def add_plane(self, o, size, material):
        a, pa = _f(o)
        b, pb = _f(size)
        return self.L.oracle_add_plane(self.h, pa, pb, int(material))

    def add_area_light(self, center, radius, L):
        c, p = _f(center)
        l, pl = _f(L)
        return self.L.oracle_add_area_light(self.h, p, float(radius), pl)

    def add_uniform_infinite_light(self, L):
        l, pl = _f(L)
        return self.L.oracle_add_uniform_infinite_light(self.h, pl)

    def add_infinite_area_light(self, rgb):
        img = np.ascontiguousarray(rgb, np.float32)
        h, w = img.shape[0], img.shape[1]
        return self.L.oracle_add_infinite_area_light(self.h, img.ctypes.data_as(C.POINTER(C.c_float)), w, h)

    def set_camera(self, lookfrom, lookat, vup, aspect_ratio, vfov=45.0, aperture=0.0):
        a, pa = _f(lookfrom)
        b, pb = _f(lookat)
        c, pc = _f(vup)
        self.L.oracle_set_camera(self.h, pa, pb, pc, float(aspect_ratio), float(vfov), float(aperture))

    def set_max_depth(self, d):
        self.L.oracle_set_max_depth(self.h, int(d))

    def bvh(self, prim):
        nn = self.L.oracle_mesh_num_nodes(self.h, prim)
        npr = self.L.oracle_mesh_num_prims(self.h, prim)
        nodes = np.zeros(nn + 1, NODE_DTYPE)
        order = np.zeros(npr, np.int32)
        self.L.oracle_mesh_get_bvh(self.h, prim, nodes.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.POINTER(C.c_int32)))
        return nodes, order

    def intersect(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        st = Stats()
        self.L.oracle_intersect_batch(self.h, rays.ctypes.data_as(C.c_void_p), rays.shape[0],
                                      out.ctypes.data_as(C.c_void_p), int(bool(any_hit)), C.byref(st))
        return out, st

    def render(self, W, H, spp, tile=None, spp_begin=0, seed_base=0, rng_mode=RNG_PER_SAMPLE, threads=1, accum=None):
        x0, y0, w, h = tile if tile is not None else (0, 0, W, H)
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        st = Stats()
        self.L.oracle_render(self.h, W, H, x0, y0, w, h, spp_begin, spp, C.c_uint32(seed_base), rng_mode, threads,
                             accum.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        return accum, st

    def bsdf_eval(self, material, wo, wi):
        a, pa = _f(wo)
        b, pb = _f(wi)
        f = np.zeros(3, np.float32)
        pdf = C.c_float(0)
        self.L.oracle_bsdf_eval(self.h, material, pa, pb, f.ctypes.data_as(C.POINTER(C.c_float)), C.byref(pdf))
        return f, pdf.value

    def bsdf_sample(self, material, wo, u):
        a, pa = _f(wo)
        b, pb = _f(u)
        wi = np.zeros(3, np.float32)
        f = np.zeros(3, np.float32)
        pdf = C.c_float(0)
        spec = C.c_int(0)
        self.L.oracle_bsdf_sample(self.h, material, pa, pb, wi.ctypes.data_as(C.POINTER(C.c_float)),
                                  f.ctypes.data_as(C.POINTER(C.c_float)), C.byref(pdf), C.byref(spec))
        return wi, f, pdf.value, spec.value

    def camera_ray(self, s, t, rng=0x12345678):
        r = np.zeros(1, RAY_DTYPE)
        state = C.c_uint32(rng)
        self.L.oracle_camera_ray(self.h, float(s), float(t), C.byref(state), r.ctypes.data_as(C.c_void_p))
        return r[0], state.value

    def li(self, ray, rng):
        r = np.zeros(1, RAY_DTYPE)
        r[0] = ray
        state = C.c_uint32(rng)
        out = np.zeros(3, np.float32)
        st = Stats()
        self.L.oracle_li(self.h, r.ctypes.data_as(C.c_void_p), C.byref(state), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        return out, state.value, st

    def dbg_li(self, rays):
        """DbgIntegrator::Li (integrator.h:107-118) for every ray: float32 [n, 3]."""
        r = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros((r.shape[0], 3), np.float32)
        for i in range(r.shape[0]):
            self.L.oracle_dbg_li(self.h, r[i:i + 1].ctypes.data_as(C.c_void_p), out[i].ctypes.data_as(C.POINTER(C.c_float)))
        return out


TRIG_LIBM, TRIG_CORRECTLY_ROUNDED = 0, 1


_TRIG_MODE = TRIG_LIBM


def set_trig_mode(mode):
    global _TRIG_MODE
    lib().oracle_set_trig_mode(int(mode))
    _TRIG_MODE = int(mode)


def get_trig_mode():
    return _TRIG_MODE


def wang_hash(s):
    return int(lib().oracle_wang_hash(C.c_uint32(s & 0xFFFFFFFF)))


def sample_seed(pixel, wh, sample, seed_base=0):
    return int(lib().oracle_sample_seed(C.c_uint32(pixel), C.c_uint32(wh), C.c_uint32(sample), C.c_uint32(seed_base)))


def rng_floats(seed, n):
    f = np.zeros(n, np.float32)
    u = np.zeros(n, np.uint32)
    lib().oracle_rng_floats(C.c_uint32(seed), n, f.ctypes.data_as(C.POINTER(C.c_float)), u.ctypes.data_as(C.POINTER(C.c_uint32)))
    return f, u


def distribution1d(func, u):
    """Distribution1D(func) then SampleContinuous(u[i]) (sampling.h:19-52): (cdf[n + 1], funcInt, x[k], pdf[k], offset[k])."""
    f, pf = _f(func)
    uu, pu = _f(u)
    cdf = np.zeros(f.size + 1, np.float32)
    fi = np.zeros(1, np.float32)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    lib().oracle_dist1d_build(pf, f.size, P(cdf), P(fi))
    k = uu.size
    x = np.zeros(max(k, 1), np.float32)
    pdf = np.zeros(max(k, 1), np.float32)
    off = np.zeros(max(k, 1), np.int32)
    lib().oracle_dist1d_sample(pf, P(cdf), C.c_float(fi[0]), f.size, pu, k, P(x), P(pdf), off.ctypes.data_as(C.POINTER(C.c_int32)))
    return cdf, fi[0], x[:k], pdf[:k], off[:k]


def bounds_intersect(bmin, bmax, ray):
    a, pa = _f(bmin)
    b, pb = _f(bmax)
    r = np.zeros(1, RAY_DTYPE)
    r[0] = ray
    t = C.c_float(0)
    h = lib().oracle_bounds_intersect(pa, pb, r.ctypes.data_as(C.c_void_p), C.byref(t))
    return bool(h), t.value
