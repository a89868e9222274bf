"""OBJ ingestion: agpt_obj_parse restates TriangleMesh::LoadObj + tinyobjloader 2.0.0's
triangulation.  PARITY UNPINNED -- no OBJ fixture or reference output exists (the reference cannot run here); these tests
check the loader against an independent Python restatement of the same published behaviour, and that a loaded mesh renders
identically on the GPU and the oracle."""
import numpy as np
import pytest

import ag_pathtracer_amd as ag

F = np.float32

CUBE_AND_NGON = """
# unit cube with quads, v/vt/vn references, negative indices, and a concave 6-gon (an L shape)
v -1 -1 -1
v  1 -1 -1
v  1  1 -1
v -1  1 -1
v -1 -1  1
v  1 -1  1
v  1  1  1
v -1  1  1
vn 0 0 -1
vn 0 0 1
vn 0 -1 0
vn 0 1 0
vn -1 0 0
vn 1 0 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
g cube
f 1/1/1 4/4/1 3/3/1 2/2/1
f 5/1/2 6/2/2 7/3/2 8/4/2
usemtl whatever
f 1/1/3 2/2/3 6/3/3 5/4/3
f 4/1/4 8/4/4 7/3/4 3/2/4
f -8/1/5 -4/2/5 -1/3/5 -5/4/5
f 2/1/6 3/2/6 7/3/6 6/4/6
o ell
v 2 0 0
v 4 0 0
v 4 1 0
v 3 1 0
v 3 3 0
v 2 3 0
f 9/1/2 10/2/2 11/3/2 12/4/2 13/1/2 14/2/2
f 9/1/2 10/2/2
"""


def ref_triangulate(face, v):
    """Independent restatement of tinyobjloader 2.0.0's ear clipping (float32 arithmetic)."""
    n = len(face)
    if n < 3:
        return []
    axes = [1, 2]
    for k in range(n):
        p0, p1, p2 = (v[face[(k + j) % n][0]] for j in range(3))
        e0, e1 = p1 - p0, p2 - p1
        c = np.abs(np.array([e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]], F))
        if (c > np.finfo(F).eps).any():
            if not (c[0] > c[1] and c[0] > c[2]):
                axes[0] = 0
                if c[2] > c[0] and c[2] > c[1]:
                    axes[1] = 1
            break
    area = F(0)
    for k in range(n):
        a, b = v[face[k][0]], v[face[(k + 1) % n][0]]
        area = F(area + F(F(a[axes[0]] * b[axes[1]]) - F(a[axes[1]] * b[axes[0]])) * F(0.5))
    rem = list(face)
    out = []
    guess, iters, prev = 0, len(face), len(face)
    while len(rem) > 3 and iters > 0:
        m = len(rem)
        if guess >= m:
            guess -= m
        if prev != m:
            prev, iters = m, m
        else:
            iters -= 1
        ind = [rem[(guess + k) % m] for k in range(3)]
        vx = [v[i[0]][axes[0]] for i in ind]
        vy = [v[i[0]][axes[1]] for i in ind]
        cross = F(F(vx[1] - vx[0]) * F(vy[2] - vy[1])) - F(F(vy[1] - vy[0]) * F(vx[2] - vx[1]))
        if cross * area < 0:
            guess += 1
            continue
        overlap = False
        for other in range(3, m):
            t = v[rem[(guess + other) % m][0]]
            tx, ty = t[axes[0]], t[axes[1]]
            c = False
            j = 2
            for i in range(3):
                if (vy[i] > ty) != (vy[j] > ty) and tx < F(F(F(vx[j] - vx[i]) * F(ty - vy[i])) / F(vy[j] - vy[i])) + vx[i]:
                    c = not c
                j = i
            if c:
                overlap = True
                break
        if overlap:
            guess += 1
            continue
        out += ind
        del rem[(guess + 1) % m]
    if len(rem) == 3:
        out += rem
    return out


def ref_load_obj(text, M=None):
    v, vn, vt, faces = [], [], [], []
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            v.append([F(x) for x in tok[1:4]])
        elif tok[0] == "vn":
            vn.append([F(x) for x in tok[1:4]])
        elif tok[0] == "vt":
            vt.append([F(x) for x in tok[1:3]])
        elif tok[0] == "f":
            face = []
            for c in tok[1:]:
                parts = c.split("/")
                fix = lambda s, n: (int(s) - 1) if int(s) > 0 else n + int(s)  # noqa: E731
                iv = fix(parts[0], len(v))
                it = fix(parts[1], len(vt)) if len(parts) > 1 and parts[1] else -1
                inn = fix(parts[2], len(vn)) if len(parts) > 2 and parts[2] else -1
                face.append((iv, inn, it))
            faces.append(face)
    va = np.array(v, F)
    idx = []
    for f in faces:
        idx += ref_triangulate(f, va)
    M = np.eye(4, dtype=F) if M is None else np.asarray(M, F)
    pts = np.stack([M[0, 0] * va[:, 0] + M[0, 1] * va[:, 1] + M[0, 2] * va[:, 2] + M[0, 3],
                    M[1, 0] * va[:, 0] + M[1, 1] * va[:, 1] + M[1, 2] * va[:, 2] + M[1, 3],
                    M[2, 0] * va[:, 0] + M[2, 1] * va[:, 1] + M[2, 2] * va[:, 2] + M[2, 3]], 1).astype(F)
    return pts, np.array(vn, F), np.array(vt, F), np.array(idx, np.int32)


def test_obj_triangulation_and_indices_match_independent_restatement():
    v, n, t, ix = ag.load_obj(text=CUBE_AND_NGON)
    rv, rn, rt, rix = ref_load_obj(CUBE_AND_NGON)
    assert ix.shape == (3 * (6 * 2 + 4), 3)          # 6 quads -> 12 triangles, the 6-gon -> 4, the 2-corner face dropped
    assert np.array_equal(ix, rix)
    assert v.tobytes() == rv.tobytes() and t.tobytes() == rt.tobytes()
    assert np.allclose(n, rn)                          # identity transform: inverse transpose is exact
    # the concave corner (vertex 12, zero-based 11) must not produce a triangle outside the L: total area == 4 (2x1 + 1x2)
    tri = ix[-12:, 0].reshape(-1, 3)
    area = sum(0.5 * abs(np.cross(v[b] - v[a], v[c] - v[a])[2]) for a, b, c in tri)
    assert abs(area - 4.0) < 1e-6


def test_obj_transform_and_errors():
    M = np.array([[2, 0, 0, 1], [0, 0, -1, 2], [0, 1, 0, 3], [0, 0, 0, 1]], F)   # scale x, rotate y/z, translate
    v, n, t, ix = ag.load_obj(text=CUBE_AND_NGON, transform=M)
    rv, rn, _, _ = ref_load_obj(CUBE_AND_NGON, M)
    assert v.tobytes() == rv.tobytes()
    Nt = np.linalg.inv(M.astype(np.float64)).T[:3, :3]
    assert np.allclose(n, rn @ Nt.T, atol=1e-6)
    v2, n2, t2, ix2 = ag.load_obj(text=CUBE_AND_NGON, ignore_normals=True)
    assert n2 is None and np.array_equal(ix2[:, 0], ix[:, 0])
    with pytest.raises(ag.AgptError):
        ag.load_obj(text="v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n")          # zero index
    with pytest.raises(ag.AgptError):
        ag.load_obj(text="v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1 2 3\n")  # normals exist but the face has none
    with pytest.raises(ag.AgptError):
        ag.load_obj(path="/nonexistent/bunny.obj")                          # the reference exit(1)s here
    with pytest.raises(ag.AgptError):
        ag.load_obj(text="v 0 0 0\n")


@pytest.mark.gpu
def test_loaded_obj_renders_like_the_oracle(tmp_path):
    from helpers import gpu_scene, oracle_scene
    from oracle import binding as ob
    p = tmp_path / "mesh.obj"
    p.write_text(CUBE_AND_NGON)
    M = np.array([[.6, 0, 0, 0], [0, .6, 0, 0], [0, 0, .6, 0], [0, 0, 0, 1]], F)
    v, n, t, ix = ag.load_obj(path=str(p), transform=M)
    d = ag.scenes.scene_c1()
    d.add_mesh(v, n, t, ix, 0, 1)
    g = gpu_scene(d)
    acc, st = ag.PathTracer(5).render_to_host(g, 96, 96, 2)
    g.close()
    o = oracle_scene(d)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        oacc, ost = o.render(96, 96, 2, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    assert np.mean(np.all(acc[..., :3].view(np.uint32) == oacc[..., :3].view(np.uint32), axis=-1)) == 1.0
    assert st.rays == ost.rays
