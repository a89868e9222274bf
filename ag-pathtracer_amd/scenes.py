"""Scene descriptions for the path-tracing hot path.

`SceneDesc` records the reference's scene-construction calls (myapp.cpp:13-114: materials, BVHTriMesh / Sphere
primitives in insertion order, Scene::addAreaLight, lights.push_back, CameraDesc) as plain numpy data, so that the
same description can be instantiated on the GPU (`binding.Scene`) -- and, in tests, on the CPU oracle.

No scene assets ship with the reference (its scenes name D://models/bunny.obj and an .hdr that are absent), so the
BASELINE.json configurations use the synthetic generators below (fixed seeds).
"""
import numpy as np

from .binding import MAT_DIFFUSE_ONLY, MAT_DISNEY, MAT_MIRROR  # noqa: F401

F = np.float32


def rgb2lin(c):
    """template/common.h:29-31 (pow 2.2), evaluated in fp32."""
    return np.power(np.asarray(c, F), F(2.2)).astype(F)


def hex2lin(h):
    """template/common.h:33-39."""
    rgb = np.array([((h >> 16) & 0xFF), ((h >> 8) & 0xFF), (h & 0xFF)], F) / F(255.0)
    return rgb2lin(rgb)


# the `cute*` palette of SimpleTestScene (myapp.cpp:70-99)
CUTE_PALETTE = [0xf19a91, 0xedd0ca, 0xf3b8a8, 0xf9ece6, 0xf6e7d0, 0xf5deac, 0xeecf74, 0x9ed5d8, 0x9ba6ac, 0xaebdc4,
                0xb9ddf3, 0x87abc5, 0xcbceb1, 0xf7f7f7, 0xc4ac64, 0xe2f4f6, 0xd2e4e6, 0xbfdcda, 0x69bab3, 0x88cabc,
                0xcdd1d4, 0xe6e5ea, 0x33455b, 0x5b6268, 0x778592, 0xe57a82, 0xcd7d88, 0xe3a3b1, 0xf0d1e3, 0xc5b5d2]


class SceneDesc:
    def __init__(self, name="scene"):
        self.name = name
        self.ops = []
        self.n_materials = 0
        self.n_prims = 0
        self.n_lights = 0
        self.n_tris = 0
        self.camera = None

    def add_material(self, mtype, color, roughness=0.5, metallic=0.0):
        self.ops.append(("material", int(mtype), np.asarray(color, F).copy(), float(roughness), float(metallic)))
        self.n_materials += 1
        return self.n_materials - 1

    def add_mesh(self, verts, normals, uvs, indices, material, max_prims_in_node=1):
        verts = np.ascontiguousarray(verts, F).reshape(-1, 3)
        normals = None if normals is None else np.ascontiguousarray(normals, F).reshape(-1, 3)
        uvs = None if uvs is None else np.ascontiguousarray(uvs, F).reshape(-1, 2)
        indices = np.ascontiguousarray(indices, np.int32).reshape(-1, 3)
        self.ops.append(("mesh", verts, normals, uvs, indices, int(material), int(max_prims_in_node)))
        self.n_prims += 1
        self.n_tris += indices.shape[0] // 3
        return self.n_prims - 1

    def add_sphere(self, center, radius, material):
        self.ops.append(("sphere", np.asarray(center, F).copy(), float(radius), int(material)))
        self.n_prims += 1
        return self.n_prims - 1

    def add_plane(self, o, size, material):
        self.ops.append(("plane", np.asarray(o, F).copy(), np.asarray(size, F).copy(), int(material)))
        self.n_prims += 1
        return self.n_prims - 1

    def add_area_light(self, center, radius, L):
        self.ops.append(("area_light", np.asarray(center, F).copy(), float(radius), np.asarray(L, F).copy()))
        self.n_prims += 1
        self.n_lights += 1
        return self.n_prims - 1

    def add_uniform_infinite_light(self, L):
        self.ops.append(("infinite_light", np.asarray(L, F).copy()))
        self.n_lights += 1
        return self.n_lights - 1

    def add_infinite_area_light(self, rgb):
        self.ops.append(("env_light", np.ascontiguousarray(rgb, F).copy()))
        self.n_lights += 1
        return self.n_lights - 1

    def set_camera(self, lookfrom, lookat, vup, aspect_ratio, vfov=45.0, aperture=0.0):
        self.camera = (np.asarray(lookfrom, F).copy(), np.asarray(lookat, F).copy(), np.asarray(vup, F).copy(),
                       F(aspect_ratio), float(vfov), float(aperture))

    def instantiate(self, target):
        """Replay the recorded calls on `target` (binding.Scene, or any object with the same method names)."""
        for op in self.ops:
            k = op[0]
            if k == "material":
                target.add_material(op[1], op[2], op[3], op[4])
            elif k == "mesh":
                target.add_mesh(op[1], op[2], op[3], op[4], op[5], op[6])
            elif k == "sphere":
                target.add_sphere(op[1], op[2], op[3])
            elif k == "plane":
                target.add_plane(op[1], op[2], op[3])
            elif k == "area_light":
                target.add_area_light(op[1], op[2], op[3])
            elif k == "infinite_light":
                target.add_uniform_infinite_light(op[1])
            elif k == "env_light":
                target.add_infinite_area_light(op[1])
        if self.camera is not None:
            target.set_camera(*self.camera)
        if hasattr(target, "commit"):
            target.commit()
        return target


# ---- mesh generators (fp64 maths, rounded once to fp32: the arrays ARE the scene) -----------------------------
def _iota_idx(tris):
    """[m,3] vertex ids -> [3m,3] index_type rows with v = n = t (trianglemesh.h:10)."""
    t = np.asarray(tris, np.int32).reshape(-1)
    return np.stack([t, t, t], axis=1).astype(np.int32)


def grid_mesh(pfun, nu, nv, flip=False):
    """Tessellated parametric patch p(u,v), u,v in [0,1]: (nu x nv quads) -> verts, smooth normals, uvs, indices."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="xy")
    P = pfun(U, V)  # [nv+1, nu+1, 3]
    e = 1e-4
    dU = (pfun(U + e, V) - pfun(U - e, V)) / (2 * e)
    dV = (pfun(U, V + e) - pfun(U, V - e)) / (2 * e)
    N = np.cross(dU, dV)
    if flip:
        N = -N
    ln = np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where(ln > 1e-12, N / np.maximum(ln, 1e-12), np.array([0.0, 1.0, 0.0]))
    a = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    b, c, d = a + 1, a + nu + 1, a + nu + 2
    if flip:
        tris = np.stack([a, b, c, b, d, c], axis=1).reshape(-1, 3)
    else:
        tris = np.stack([a, c, b, b, c, d], axis=1).reshape(-1, 3)
    uv = np.stack([U, V], axis=-1)
    return (P.reshape(-1, 3).astype(F), N.reshape(-1, 3).astype(F), uv.reshape(-1, 2).astype(F), _iota_idx(tris))


def heightfield(n, S=2.0):
    """n x n quads over [-S,S]^2, h = .35 sin3.1x cos2.7z + .15 sin(9.3x+1) sin7.9z (the survey's proxy surface)."""
    def p(U, V):
        x = -S + 2 * S * U
        z = -S + 2 * S * V
        y = 0.35 * np.sin(3.1 * x) * np.cos(2.7 * z) + 0.15 * np.sin(9.3 * x + 1.0) * np.sin(7.9 * z)
        return np.stack([x, y, z], axis=-1)
    return grid_mesh(p, n, n)


def blob_mesh(n_seg, n_ring, center=(0.0, 0.0, 0.0), radius=1.0, seed=0):
    """Closed genus-0 blob: a sphere displaced by 3 sinusoidal octaves, pole fans + quad rings =
    2*n_seg*(n_ring-1) triangles, smooth normals."""
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * np.pi, 6)

    def r(theta, phi):
        return radius * (1.0 + 0.12 * np.sin(3 * phi + ph[0]) * np.sin(2 * theta + ph[1])
                         + 0.06 * np.sin(5 * phi + ph[2]) * np.sin(4 * theta + ph[3])
                         + 0.03 * np.sin(9 * phi + ph[4]) * np.sin(7 * theta + ph[5]) * np.sin(theta))

    def pos(theta, phi):
        rr = r(theta, phi)
        return np.stack([rr * np.sin(theta) * np.cos(phi), rr * np.cos(theta), rr * np.sin(theta) * np.sin(phi)], -1)

    thetas = np.linspace(0, np.pi, n_ring + 1)[1:-1]
    phis = np.linspace(0, 2 * np.pi, n_seg, endpoint=False)
    T, Pm = np.meshgrid(thetas, phis, indexing="ij")
    ring = pos(T, Pm)
    e = 1e-4
    dT = (pos(T + e, Pm) - pos(T - e, Pm)) / (2 * e)
    dP = (pos(T, Pm + e) - pos(T, Pm - e)) / (2 * e)
    Nr = np.cross(dP, dT)
    Nr /= np.linalg.norm(Nr, axis=-1, keepdims=True)
    north = pos(np.array(0.0), np.array(0.0))
    south = pos(np.array(np.pi), np.array(0.0))
    verts = np.concatenate([north[None], ring.reshape(-1, 3), south[None]], 0) + np.asarray(center)
    normals = np.concatenate([np.array([[0.0, 1.0, 0.0]]), Nr.reshape(-1, 3), np.array([[0.0, -1.0, 0.0]])], 0)
    nrm1 = n_ring - 1
    tris = []
    j = np.arange(n_seg)
    jn = (j + 1) % n_seg
    tris.append(np.stack([np.zeros(n_seg, int), 1 + jn, 1 + j], 1))
    for i in range(nrm1 - 1):
        a = 1 + i * n_seg + j
        b = 1 + i * n_seg + jn
        c = 1 + (i + 1) * n_seg + j
        d = 1 + (i + 1) * n_seg + jn
        tris.append(np.stack([a, b, c], 1))
        tris.append(np.stack([b, d, c], 1))
    last = 1 + (nrm1 - 1) * n_seg
    sidx = 1 + nrm1 * n_seg
    tris.append(np.stack([last + j, last + jn, np.full(n_seg, sidx)], 1))
    tris = np.concatenate(tris, 0)
    th = np.concatenate([[0.0], T.reshape(-1), [np.pi]])
    phv = np.concatenate([[0.0], Pm.reshape(-1), [0.0]])
    uv = np.stack([phv / (2 * np.pi), th / np.pi], 1)
    return verts.astype(F), normals.astype(F), uv.astype(F), _iota_idx(tris)


# ---- the BASELINE.json configurations ------------------------------------------------------------------------
KEY_LIGHT = np.array([1, .941, .914], F)


def scene_c1(backdrop=None):
    """C1: SimpleTestScene's geometry (myapp.cpp:101-111) -- backdrop + gold unit sphere -- lit by BunnyScene's key
    light and a uniform sky (myapp.cpp:36-39,51); camera of SimpleTestScene with aperture 0.  256x256, 1 spp."""
    from .binding import create_backdrop
    d = SceneDesc("C1-cornell-scale")
    floor = d.add_material(MAT_DISNEY, hex2lin(0xcbceb1), 1.0, 0.0)
    gold = d.add_material(MAT_DISNEY, [0.944, 0.776, 0.373], 0.5, 1.0)
    v, n, t, idx = backdrop if backdrop is not None else create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32)
    d.add_mesh(v, n, t, idx, floor, 1)
    d.add_sphere([0, 0, 0], 1.0, gold)
    d.add_area_light([0, 25, -20], 1.0, KEY_LIGHT * F(200))
    d.add_uniform_infinite_light([.4, .45, .5])
    d.set_camera([-1.46, 1.16, -4.64], [0, 0, 0], [0, 1, 0], 1.0, 45.0, 0.0)
    return d


def scene_c2(n_seg=187, backdrop=None, aspect=16.0 / 9.0):
    """C2 "bunny-class": closed blob of 2*n_seg*n_seg triangles (69,938 at n_seg=187) with a single DisneyDiffuse lobe
    (albedo .7) on the backdrop, lights and camera of BunnyScene (myapp.cpp:13-52), 1280x720, 16 spp."""
    from .binding import create_backdrop
    d = SceneDesc("C2-bunny-class")
    floor = d.add_material(MAT_DIFFUSE_ONLY, hex2lin(0xcbceb1))
    cute = d.add_material(MAT_DIFFUSE_ONLY, [.7, .7, .7])
    v, n, t, idx = backdrop if backdrop is not None else create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32)
    d.add_mesh(v, n, t, idx, floor, 1)
    bv, bn, bt, bi = blob_mesh(n_seg, n_seg + 1, center=(0.5, 0.05, 0.5), radius=1.0, seed=0xA6A70002 & 0x7FFFFFFF)
    d.add_mesh(bv, bn, bt, bi, cute, 1)
    lf = np.array([1.24, 0.1, -4.84])
    lf = lf / np.linalg.norm(lf) * 5
    d.add_area_light([0, 25, -20], 1.0, KEY_LIGHT * F(200))
    d.add_area_light([10, 25, -20], 1.0, KEY_LIGHT * F(50))
    d.add_area_light([0, 20, 10], 5.0, KEY_LIGHT * F(1))
    d.add_uniform_infinite_light([.4, .45, .5])
    d.set_camera(lf, [.5, 0, .5], [0, 1, 0], aspect, 30.0, 0.0)
    return d


def _merge(parts):
    vs, ns, ts, ixs, base = [], [], [], [], 0
    for (v, n, t, ix) in parts:
        vs.append(v); ns.append(n); ts.append(t); ixs.append(ix + base); base += v.shape[0]
    return np.concatenate(vs), np.concatenate(ns), np.concatenate(ts), np.concatenate(ixs)


def _atrium_furniture(rng, q, emit):
    """Columns, arches, drapes and urns of the C3 atrium; every finished mesh goes to emit((v, n, t, idx))."""
    # two rows of displaced columns
    col_x = np.linspace(-11.0, 11.0, 6)
    for zc in (-3.6, 3.6):
        for xc in col_x:
            ph = rng.uniform(0, 2 * np.pi, 3)
            flutes = int(rng.choice([8, 12, 16]))

            def col(U, V, xc=xc, zc=zc, ph=ph, flutes=flutes):
                a = 2 * np.pi * U
                y = 9.0 * V
                r = 0.55 + 0.04 * np.cos(flutes * a + ph[0]) + 0.10 * np.exp(-((V - 0.02) / 0.05) ** 2) \
                    + 0.12 * np.exp(-((V - 0.98) / 0.04) ** 2) + 0.02 * np.sin(6 * np.pi * V + ph[1])
                return np.stack([xc + r * np.cos(a), y, zc + r * np.sin(a)], -1)
            emit(grid_mesh(col, q(62), q(80), flip=True))

    # arches spanning neighbouring columns (half tori), grouped per row pair
    for i in range(5):
        parts = []
        for zc in (-3.6, 3.6):
            x0, x1 = col_x[i], col_x[i + 1]
            cx, R = 0.5 * (x0 + x1), 0.5 * (x1 - x0)

            def arch(U, V, cx=cx, R=R, zc=zc):
                a = np.pi * U
                b = 2 * np.pi * V
                rr = 0.32 + 0.03 * np.cos(8 * b)
                return np.stack([cx - (R + rr * np.cos(b)) * np.cos(a), 9.0 + (R + rr * np.cos(b)) * np.sin(a) * 0.6,
                                 zc + rr * np.sin(b)], -1)
            parts.append(grid_mesh(arch, q(44), q(30)))
        emit(_merge(parts))

    # drapes: wavy hanging sheets along the side aisles
    for i in range(6):
        xc = -10.0 + 4.0 * i
        zc = -5.6 if i % 2 == 0 else 5.6
        ph = rng.uniform(0, 2 * np.pi, 2)

        def drape(U, V, xc=xc, zc=zc, ph=ph):
            x = xc + 3.0 * (U - 0.5)
            y = 9.5 - 6.5 * V
            z = zc + 0.35 * np.sin(9 * U * np.pi + ph[0]) * (0.3 + V) + 0.1 * np.sin(5 * V * np.pi + ph[1])
            return np.stack([x, y, z], -1)
        emit(grid_mesh(drape, q(60), q(60)))

    # urns on the floor between the columns
    for i in range(8):
        xc = -9.9 + 2.83 * i
        zc = 0.0 if i % 2 == 0 else (-1.2 if i % 4 == 1 else 1.2)
        v, n, t, ix = blob_mesh(q(40), q(40), center=(xc, 0.78, zc), radius=0.7, seed=int(rng.randint(1 << 30)))
        emit((v, n, t, ix))



def scene_c3(scale=1.0, aspect=16.0 / 9.0, seed=0xA6A70003 & 0x7FFFFFFF, all_metal=False):
    """C3 "Sponza-class": a closed atrium (room 30 x 12 x 14) with two rows of displaced columns, arches, drapes and
    urns from tessellated grids, ~260k triangles at scale=1, ~30 Disney materials (roughness U[.2,1], 15 % metals,
    `cute*` palette), two sphere lights inside + a uniform sky through an open roof strip.  1920x1080, 64 spp."""
    rng = np.random.RandomState(seed)
    d = SceneDesc("C3-sponza-class")
    k = float(np.sqrt(scale))

    def q(n):
        return max(2, int(round(n * k)))

    def material():
        color = hex2lin(CUTE_PALETTE[rng.randint(len(CUTE_PALETTE))])
        if all_metal:
            return d.add_material(MAT_DISNEY, color, float(rng.uniform(.1, .6)), 1.0)
        metal = 1.0 if rng.uniform() < 0.15 else 0.0
        return d.add_material(MAT_DISNEY, color, float(rng.uniform(.2, 1.0)), metal)

    X, Y, Z = 15.0, 12.0, 7.0  # half extents in x,z; height

    def plane(o, eu, ev, nu, nv, flip=False):
        o, eu, ev = map(np.asarray, (o, eu, ev))
        return grid_mesh(lambda U, V: o + U[..., None] * eu + V[..., None] * ev, nu, nv, flip)

    merge = _merge

    # floor, ceiling with an open strip (|z| < 1.5 for |x| < 11), walls
    d.add_mesh(*plane([-X, 0, -Z], [2 * X, 0, 0], [0, 0, 2 * Z], q(110), q(52), flip=True), material())
    ceil = [plane([-X, Y, -Z], [2 * X, 0, 0], [0, 0, Z - 1.5], q(100), q(18)),
            plane([-X, Y, 1.5], [2 * X, 0, 0], [0, 0, Z - 1.5], q(100), q(18)),
            plane([-X, Y, -1.5], [4.0, 0, 0], [0, 0, 3.0], q(14), q(10)),
            plane([X - 4.0, Y, -1.5], [4.0, 0, 0], [0, 0, 3.0], q(14), q(10))]
    d.add_mesh(*merge(ceil), material())
    d.add_mesh(*merge([plane([-X, 0, Z], [2 * X, 0, 0], [0, Y, 0], q(104), q(42)),
                       plane([-X, 0, -Z], [2 * X, 0, 0], [0, Y, 0], q(104), q(42), flip=True)]), material())
    d.add_mesh(*merge([plane([-X, 0, -Z], [0, 0, 2 * Z], [0, Y, 0], q(50), q(42)),
                       plane([X, 0, -Z], [0, 0, 2 * Z], [0, Y, 0], q(50), q(42), flip=True)]), material())

    _atrium_furniture(rng, q, lambda m: d.add_mesh(*m, material()))

    d.add_area_light([-6.0, 10.2, 0.0], 0.6, KEY_LIGHT * F(90))
    d.add_area_light([7.0, 9.8, 0.8], 0.5, KEY_LIGHT * F(70))
    d.add_uniform_infinite_light([.4, .45, .5])
    d.set_camera([-13.2, 4.2, 5.4], [6.0, 4.0, -1.5], [0, 1, 0], aspect, 58.0, 0.0)
    return d


def scene_c5(scale=1.0, aspect=16.0 / 9.0, seed=0xA6A70005 & 0x7FFFFFFF):
    """C5 "5M-triangle multi-mesh": a 4 x 4 field of open colonnades -- 16 instances-by-copy of the C3 furniture
    (columns, arches, drapes, urns; every copy re-generated from its own seed, so the geometry is perturbed, and merged
    into ONE mesh) -- standing on an 887 x 887 heightfield terrain; ~5.0 M triangles at scale=1 in 17 meshes, so
    Scene's linear primitive loop (scene.h:5-19) and BVHs far larger than the L2 are exercised.  All-metal GGX
    (metallic 1, roughness U[.1,.6]) + MIS, four sphere lights + uniform sky.  3840x2160, 1024 spp on 8 GPUs."""
    rng = np.random.RandomState(seed)
    d = SceneDesc("C5-5M-multi-mesh")
    k = float(np.sqrt(scale))

    def q(n):
        return max(2, int(round(n * k)))

    def metal():
        color = hex2lin(CUTE_PALETTE[rng.randint(len(CUTE_PALETTE))])
        return d.add_material(MAT_DISNEY, color, float(rng.uniform(.1, .6)), 1.0)

    n_t = q(887)
    ext = 88.0

    def terrain(U, V):
        x = ext * (2 * U - 1)
        z = ext * (2 * V - 1)
        y = -0.35 + 0.30 * np.sin(0.31 * x) * np.cos(0.27 * z) + 0.12 * np.sin(0.93 * x + 1.0) * np.sin(0.79 * z) \
            + 0.02 * np.sin(4.1 * x) * np.sin(3.7 * z)
        return np.stack([x, y, z], -1)
    d.add_mesh(*grid_mesh(terrain, n_t, n_t, flip=True), metal())
    for i in range(16):
        parts = []
        _atrium_furniture(np.random.RandomState((seed + 7919 * (i + 1)) & 0x7FFFFFFF), q, parts.append)
        v, n, t, ix = _merge(parts)
        off = np.array([-51.0 + 34.0 * (i % 4), 0.0, -33.0 + 22.0 * (i // 4)], F)
        v = (v * F(rng.uniform(0.94, 1.06)) + off).astype(F)
        d.add_mesh(v, n, t, ix, metal())
    for (x, z, r, pw) in ((-40.0, -20.0, 3.0, 260), (38.0, 24.0, 2.5, 300), (-10.0, 40.0, 2.0, 340), (20.0, -44.0, 2.0, 340)):
        d.add_area_light([x, 34.0, z], r, KEY_LIGHT * F(pw))
    d.add_uniform_infinite_light([.4, .45, .5])
    d.set_camera([-84.0, 44.0, 62.0], [-4.0, -8.0, -2.0], [0, 1, 0], aspect, 40.0, 0.0)
    return d


def synthetic_hdr(width=128, height=64, seed=0xA6A70009 & 0x7FFFFFFF):
    """Stand-in for small_workshop_1k.hdr (myapp.cpp:113; no .hdr ships): a sky gradient with a warm sun patch, a few
    window-like bright panels and mild noise, rgb[H, W, 3], values up to ~60."""
    rng = np.random.RandomState(seed)
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    U, V = np.meshgrid(u, v)
    sky = np.stack([0.35 + 0.3 * (1 - V), 0.45 + 0.35 * (1 - V), 0.6 + 0.4 * (1 - V)], -1)
    ground = np.stack([0.25 * np.ones_like(V), 0.22 * np.ones_like(V), 0.18 * np.ones_like(V)], -1)
    img = np.where((V < 0.55)[..., None], sky, ground)
    sun = np.exp(-(((U - 0.3) * 2) ** 2 + (V - 0.22) ** 2) / 0.002)
    img = img + sun[..., None] * np.array([60.0, 52.0, 40.0])
    for (cu, cv) in ((0.62, 0.35), (0.78, 0.33), (0.9, 0.36)):
        panel = ((np.abs(U - cu) < 0.03) & (np.abs(V - cv) < 0.06)).astype(np.float64)
        img = img + panel[..., None] * np.array([6.0, 6.5, 7.5])
    img = img * (1.0 + 0.05 * rng.standard_normal(img.shape))
    return np.maximum(img, 0.0).astype(F)


def scene_simple_test(hdr=None, backdrop=None, aperture=0.1):
    """The reference's default scene, SimpleTestScene (myapp.cpp:55-114): backdrop + gold sphere lit ONLY by the HDR
    environment light, thin-lens camera (aperture .1)."""
    from .binding import create_backdrop
    d = SceneDesc("SimpleTestScene")
    floor = d.add_material(MAT_DISNEY, hex2lin(0xcbceb1), 1.0, 0.0)
    gold = d.add_material(MAT_DISNEY, [0.944, 0.776, 0.373], 0.5, 1.0)
    v, n, t, idx = backdrop if backdrop is not None else create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32)
    d.add_mesh(v, n, t, idx, floor, 1)
    d.add_sphere([0, 0, 0], 1.0, gold)
    d.add_infinite_area_light(synthetic_hdr() if hdr is None else hdr)
    d.set_camera([-1.46, 1.16, -4.64], [0, 0, 0], [0, 1, 0], 1.0, 45.0, aperture)
    return d


def scene_heightfield(n, disney=True, W=960, H=540):
    """The survey's proxy scene with this package's own heightfield generator."""
    d = SceneDesc("heightfield-%d" % n)
    mat = d.add_material(MAT_DISNEY, hex2lin(0xc5b5d2), .5, 0.) if disney else d.add_material(MAT_DIFFUSE_ONLY, [.7, .7, .7])
    d.add_mesh(*heightfield(n, 2.0), mat, 1)
    d.add_area_light([0, 6, -3], 1.0, KEY_LIGHT * F(60))
    d.add_uniform_infinite_light([.4, .45, .5])
    d.set_camera([0, 2.2, -4.2], [0, 0, 0], [0, 1, 0], F(W) / F(H), 40.0, 0.0)
    return d
