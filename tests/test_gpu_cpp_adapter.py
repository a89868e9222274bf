"""The C++ host adapter (include/agpt_host.hpp) end to end: examples/simple_test_scene.cpp -- reference-style
MyApp::Init + 4 MyApp::Tick calls -- compiled with g++ against libagpt_hip.so, run on the GPU, checked against the
oracle's 4-spp render of the same scene."""
import os
import re
import subprocess

import numpy as np
import pytest

import ag_pathtracer_amd as ag
from helpers import oracle_scene
from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path):
    exe = str(tmp_path / "simple_test_scene")
    lib = ag.library_path()
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "simple_test_scene.cpp"), "-o", exe, lib,
                           "-Wl,-rpath," + os.path.dirname(lib)])
    return exe


def test_cpp_example_compiles_and_links(tmp_path):
    build_example(tmp_path)


@pytest.mark.gpu
def test_cpp_example_matches_oracle(tmp_path):
    exe = build_example(tmp_path)
    out = subprocess.check_output([exe, "4"]).decode()
    m = re.search(r"frames=4 rays=(\d+) mean_rgb=([\d.]+) ([\d.]+) ([\d.]+)", out)
    assert m, out
    # same scene built through the Python mirror for the oracle (the example passes gold first, floor second)
    d = ag.SceneDesc("simple")
    gold = d.add_material(ag.MAT_DISNEY, [0.944, 0.776, 0.373], .5, 1.)
    floor = d.add_material(ag.MAT_DISNEY, np.power(np.float32([0xcb, 0xce, 0xb1]) / np.float32(255), np.float32(2.2)), 1., 0.)
    d.add_mesh(*ag.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32), floor, 1)
    d.add_sphere([0, 0, 0], 1.0, gold)
    d.add_area_light([0, 25, -20], 1.0, [200., np.float32(.941) * np.float32(200), np.float32(.914) * np.float32(200)])
    d.add_uniform_infinite_light([.4, .45, .5])
    d.set_camera([-1.46, 1.16, -4.64], [0, 0, 0], [0, 1, 0], 1.0, 45.0, 0.0)
    o = oracle_scene(d)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        acc, st = o.render(256, 256, 4, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    mean = acc[..., :3].astype(np.float64).reshape(-1, 3).sum(0) / (256 * 256) / 4
    got = np.array([float(m.group(i)) for i in (2, 3, 4)])
    # powf(x, 2.2f) of the floor colour is computed by libm in C++ and by numpy here: allow 1e-5 relative
    assert np.allclose(got, mean, rtol=1e-5), (got, mean)
    assert abs(int(m.group(1)) - st.rays) <= 1e-4 * st.rays
    # PathTracer::Li on the pick ray (agpt_li_batch through the adapter), stream seeded like the reference's global RNG
    ml = re.search(r"li: (\S+) (\S+) (\S+) state=0x([0-9a-f]+)", out)
    assert ml, out
    ray = np.zeros(1, ag.RAY_DTYPE)
    ray["o"] = [-1.46, 1.16, -4.64]
    ray["d"] = [1.46, -1.16, 4.64]
    ray["tmax"] = 3.402823466e+38
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        L, state, _ = o.li(ray[0], 0x12345678)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    assert np.allclose(np.float32([float(ml.group(i)) for i in (1, 2, 3)]), L, rtol=2e-5), (ml.groups(), L)
    assert int(ml.group(4), 16) == state
    md = re.search(r"dbg: (\S+) (\S+) (\S+)", out)
    assert md, out
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        want = o.dbg_li(ray)[0]
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    assert np.allclose(np.float32([float(md.group(i)) for i in (1, 2, 3)]), want, rtol=1e-6), (md.groups(), want)
    assert want.any()


@pytest.mark.gpu
def test_cpp_example_lit_by_an_hdr_file_matches_oracle(tmp_path):
    """The reference's default lighting (myapp.cpp:113): agpt::InfiniteAreaLight("file.hdr") = agpt_hdr_load + the library's
    Distribution1D, through the C++ adapter; the oracle gets the pixels the vendored stb_image decoded (tests/golden/hdr_cases.npz)."""
    exe = build_example(tmp_path)
    g = np.load(os.path.join(ROOT, "tests", "golden", "hdr_cases.npz"))
    hdr = tmp_path / "env.hdr"
    hdr.write_bytes(g["rle_64x32_sky/bytes"].tobytes())
    out = subprocess.check_output([exe, "2", str(hdr)]).decode()
    m = re.search(r"frames=2 rays=(\d+) mean_rgb=([\d.]+) ([\d.]+) ([\d.]+)", out)
    assert m, out
    w, h = (int(v) for v in g["rle_64x32_sky/size"])
    d = ag.SceneDesc("simple_hdr")
    gold = d.add_material(ag.MAT_DISNEY, [0.944, 0.776, 0.373], .5, 1.)
    floor = d.add_material(ag.MAT_DISNEY, np.power(np.float32([0xcb, 0xce, 0xb1]) / np.float32(255), np.float32(2.2)), 1., 0.)
    d.add_mesh(*ag.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32), floor, 1)
    d.add_sphere([0, 0, 0], 1.0, gold)
    d.add_area_light([0, 25, -20], 1.0, [200., np.float32(.941) * np.float32(200), np.float32(.914) * np.float32(200)])
    d.add_infinite_area_light(g["rle_64x32_sky/pixels"].view(np.float32).reshape(h, w, 3))
    d.set_camera([-1.46, 1.16, -4.64], [0, 0, 0], [0, 1, 0], 1.0, 45.0, 0.0)
    o = oracle_scene(d)
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    try:
        acc, st = o.render(256, 256, 2, threads=8)
    finally:
        ob.set_trig_mode(ob.TRIG_LIBM)
    mean = acc[..., :3].astype(np.float64).reshape(-1, 3).sum(0) / (256 * 256) / 2
    got = np.array([float(m.group(i)) for i in (2, 3, 4)])
    assert np.allclose(got, mean, rtol=1e-5), (got, mean)
    assert abs(int(m.group(1)) - st.rays) <= 1e-4 * st.rays


def write_obj(path, v, n, idx):
    """Wavefront OBJ with positions, normals and v//vn faces (1-based)."""
    with open(path, "w") as f:
        for p in v:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for p in n:
            f.write("vn %.9g %.9g %.9g\n" % tuple(p))
        tri = idx.reshape(-1, 3, 3)
        for t in tri:
            f.write("f " + " ".join("%d//%d" % (c[0] + 1, c[1] + 1) for c in t) + "\n")


@pytest.mark.gpu
def test_bunny_scene_example_obj_orbit_png(tmp_path):
    """examples/bunny_scene.cpp = the reference's BunnyScene (myapp.cpp:13-52) with an OBJ stand-in: LoadObj with a
    transform, backdrop, three sphere lights + sky, 8 progressive frames, an orbit step, PNG + PFM output.  The PFM is
    compared with the same scene rendered through the Python mirror (same library, same OBJ loader)."""
    exe = str(tmp_path / "bunny_scene")
    lib = ag.library_path()
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bunny_scene.cpp"), "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib)])
    v, n, t, idx = ag.scenes.blob_mesh(40, 30, center=(0.0, 1.0, 0.0), radius=0.8, seed=5)
    obj = str(tmp_path / "blob.obj")
    write_obj(obj, v, n, idx)
    prefix = str(tmp_path / "out")
    spp = 8
    out = subprocess.check_output([exe, obj, prefix, str(spp)]).decode()
    m = re.search(r"view0 spp=8 triangles=(\d+) mean_rgb=([\d.]+) ([\d.]+) ([\d.]+)", out)
    assert m, out
    assert int(m.group(1)) == idx.shape[0] // 3
    # the same scene through the Python mirror
    pi = np.float32(3.14159265358979323846)
    c, s = np.cos(pi, dtype=np.float32), np.sin(pi, dtype=np.float32)
    transform = np.float32([[c, 0, s, .25], [0, 1, 0, -1.05], [-s, 0, c, .5], [0, 0, 0, 1]])
    mv, mn, mt, midx = ag.load_obj(obj, transform=transform)
    d = ag.SceneDesc("bunny")
    lin = lambda h: np.power(np.float32([(h >> 16) & 255, (h >> 8) & 255, h & 255]) / np.float32(255), np.float32(2.2))  # noqa: E731
    cute = d.add_material(ag.MAT_DISNEY, lin(0xc5b5d2), .25, 0.)
    floor = d.add_material(ag.MAT_DISNEY, lin(0xcbceb1), 1., 0.)
    d.add_mesh(*ag.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32), floor, 1)
    d.add_mesh(mv, mn, mt, midx, cute, 1)
    white = np.float32([1., .941, .914])
    d.add_area_light([0, 25, -20], 1.0, white * np.float32(200))
    d.add_area_light([10, 25, -20], 1.0, white * np.float32(50))
    d.add_area_light([0, 20, 10], 5.0, white)
    d.add_uniform_infinite_light([.4, .45, .5])
    nrm = np.sqrt(np.float32(1.24) ** 2 + np.float32(0.1) ** 2 + np.float32(4.84) ** 2)
    d.set_camera(np.float32([1.24, 0.1, -4.84]) / nrm * np.float32(5), [.5, 0, .5], [0, 1, 0], 1.0, 30.0, 0.0)
    from helpers import gpu_scene
    g = gpu_scene(d)
    acc, st = ag.PathTracer(5).render_to_host(g, 320, 320, spp)
    g.close()
    # PFM: bottom row first, sum / spp
    raw = open(prefix + "_view0.pfm", "rb").read()
    head = b"PF\n320 320\n-1.0\n"
    assert raw.startswith(head)
    pfm = np.frombuffer(raw[len(head):], "<f4").reshape(320, 320, 3)[::-1]
    want = acc[..., :3] * np.float32(1.0 / spp)
    # libm powf / cosf in the C++ host vs numpy here may differ in the last bit of a colour or of the transform
    assert np.allclose(pfm.mean((0, 1)), want.mean((0, 1)), rtol=2e-4), (pfm.mean((0, 1)), want.mean((0, 1)))
    assert np.mean(np.all(np.abs(pfm - want) <= 1e-3 * np.abs(want) + 1e-5, axis=-1)) > 0.99
    for view in (0, 1):
        b = open(prefix + "_view%d.png" % view, "rb").read()
        assert b[:8] == b"\x89PNG\r\n\x1a\n" and len(b) > 320 * 320 * 3
    m2 = re.search(r"lookfrom_after_orbit=([-\d.]+) ([-\d.]+) ([-\d.]+)", out)
    lf = np.float32([float(m2.group(i)) for i in (1, 2, 3)])
    lf0 = np.float32([1.24, 0.1, -4.84]) / nrm * np.float32(5)
    lookat = np.float32([.5, 0, .5])
    assert abs(np.linalg.norm(lf - lookat) - np.linalg.norm(lf0 - lookat)) < 1e-3   # the orbit keeps the distance to lookat
    assert np.linalg.norm(lf - lf0) > 1.0                                             # and it did move


@pytest.mark.gpu
def test_cpp_rank_shares_reassemble_the_single_gpu_film(tmp_path):
    """agpt::RankShare / PathTracer::Render(scene, acc, film_height, share, spp): three rank shares rendered one after the
    other on one GPU and de-interleaved on the host equal the single-GPU film bit for bit."""
    src = tmp_path / "shares.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include <vector>
#include "agpt_host.hpp"
using namespace agpt;
int main() {
    const int W = 96, H = 70, spp = 3, world = 3;
    Context ctx(0);
    Scene scene(ctx);
    int gold = DisneyMaterial::Make(scene, float3{0.944f, 0.776f, 0.373f}, .5f, 1.f);
    int floor = DisneyMaterial::Make(scene, float3{.6f, .6f, .5f}, 1.f, 0.f);
    scene.primitives_push_back(TriangleMesh::CreateBackdrop(float3{0, -1, 20}, float3{40, 20, 40}, 7.5f, 32), floor, 1);
    scene.primitives_push_back(Sphere{float3{0, 0, 0}, 1.f}, gold);
    scene.addAreaLight(Sphere{float3{0, 25, -20}, 1.f}, float3{200.f, 188.f, 183.f});
    scene.lights_push_back(UniformInfiniteLight{float3{.4f, .45f, .5f}});
    scene.camera = CameraDesc{{-1.46f, 1.16f, -4.64f}, {0, 0, 0}, {0, 1, 0}, float(W) / H, 45.f, 0.f};
    scene.commit();
    PathTracer pt;
    Accumulator full(ctx, W, H);
    pt.Render(scene, full, spp);
    std::vector<float> ref = full.Download(), film((size_t)W * H * 4, 0.f);
    for (int r = 0; r < world; r++) {
        RankShare share{8, world, r};
        const int rows = share.Rows(H);
        Accumulator acc(ctx, W, rows);
        pt.Render(scene, acc, H, share, spp);
        std::vector<float> part = acc.Download();
        // block j of this rank = film block k = r + j*world; inside a block the rows are flipped like AddSample does
        int local = 0;
        for (int y0 = 0, k = 0; y0 < H; y0 += 8, k++) {
            const int h = H - y0 < 8 ? H - y0 : 8;
            if (k % world != r) continue;
            for (int i = 0; i < h; i++)   // local row (local + i) holds film y = y0 + h - 1 - i, i.e. accumulator row H-1-y
                std::memcpy(&film[(size_t)(H - 1 - (y0 + h - 1 - i)) * W * 4], &part[(size_t)(local + i) * W * 4], (size_t)W * 16);
            local += h;
        }
    }
    // the library's own gather (agpt_gather_tiles through agpt::Comm), world = 1: compact layout -> Accumulator::pixels order
    Comm comm(ctx, {}, 1, 0);
    RankShare one{8, 1, 0};
    Accumulator compact(ctx, W, one.Rows(H)), gathered(ctx, W, H);
    pt.Render(scene, compact, H, one, spp);
    comm.GatherTiles(compact, H, one, &gathered);
    std::vector<float> g = gathered.Download();
    const bool ok = std::memcmp(film.data(), ref.data(), film.size() * 4) == 0 && std::memcmp(g.data(), ref.data(), g.size() * 4) == 0 &&
                    gathered.NumSamples() == spp;
    std::printf("%s\n", ok ? "identical" : "DIFFERENT");
    return 0;
}''')
    exe = str(tmp_path / "shares")
    lib = ag.library_path()
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe, lib,
                           "-Wl,-rpath," + os.path.dirname(lib)])
    assert subprocess.check_output([exe]).decode().strip() == "identical"
