"""Host-only presentation helpers of SURVEY 8(f) rank 4: agpt_write_png / agpt_write_pfm (no GPU needed) and the C++
adapter's RotatingCamera (camera.h:109-162)."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import ag_pathtracer_amd as ag

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks.append((tag, data))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, _, _, _ = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype) == (8, 2)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("w,h", [(1, 1), (37, 11), (640, 360)])   # 640x360 spans several 64 KiB stored blocks
def test_png_round_trip(tmp_path, w, h):
    L = ag.binding.lib()
    rng = np.random.RandomState(w)
    rgb = rng.randint(0, 1 << 24, size=(h, w)).astype(np.uint32)
    path = str(tmp_path / "a.png")
    assert L.agpt_write_png(path.encode(), rgb.ctypes.data_as(C.POINTER(C.c_uint32)), w, h) == 0
    img = read_png(path)
    assert np.array_equal(img[..., 0], (rgb >> 16) & 255) and np.array_equal(img[..., 1], (rgb >> 8) & 255)
    assert np.array_equal(img[..., 2], rgb & 255)


def test_pfm_is_bottom_up_linear_mean(tmp_path):
    L = ag.binding.lib()
    w, h, spp = 5, 3, 4
    acc = np.random.RandomState(1).rand(h, w, 4).astype(np.float32)
    path = str(tmp_path / "a.pfm")
    assert L.agpt_write_pfm(path.encode(), acc.ctypes.data_as(C.POINTER(C.c_float)), w, h, spp) == 0
    b = open(path, "rb").read()
    head = b"PF\n5 3\n-1.0\n"
    assert b.startswith(head)
    data = np.frombuffer(b[len(head):], "<f4").reshape(h, w, 3)
    assert np.array_equal(data, (acc[::-1, :, :3] * np.float32(1.0 / spp)))


def test_writers_report_io_errors(tmp_path):
    L = ag.binding.lib()
    rgb = np.zeros(4, np.uint32)
    rc = L.agpt_write_png(str(tmp_path / "no" / "dir.png").encode(), rgb.ctypes.data_as(C.POINTER(C.c_uint32)), 2, 2)
    assert rc == -5 and b"cannot open" in L.agpt_last_error()
    assert L.agpt_write_png(b"x.png", None, 2, 2) == -1


def test_rotating_camera_matches_the_reference_formula(tmp_path):
    src = tmp_path / "rc.cpp"
    src.write_text(r'''
#include <cstdio>
#include "agpt_host.hpp"
int main() {
    agpt::CameraDesc d{{-1.46f, 1.16f, -4.64f}, {0, 0, 0}, {0, 1, 0}, 1.f, 45.f, 0.1f};   // myapp.cpp:107-111
    agpt::RotatingCamera cam(d);
    std::printf("%.9g %.9g\n", cam.xAngle(), cam.yAngle());
    const float steps[4][2] = {{0.f, 0.f}, {-0.1f, 0.25f}, {0.5f, -1.f}, {-3.f, 0.3f}};
    for (auto& s : steps) {
        const agpt::CameraDesc& n = cam.update(s[0], s[1]);
        std::printf("%.9g %.9g %.9g\n", n.lookfrom[0], n.lookfrom[1], n.lookfrom[2]);
    }
}''')
    exe = tmp_path / "rc"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "ag-pathtracer_amd"), "-lagpt_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "ag-pathtracer_amd")])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    lf = np.array([-1.46, 1.16, -4.64])
    dist = np.linalg.norm(lf)
    n = lf / dist
    xa = -np.arccos(np.hypot(n[0], n[2])) if n[1] > 0 else np.arccos(np.hypot(n[0], n[2]))
    ya = np.arccos(n[2] / np.hypot(n[0], n[2])) * (-1 if n[0] < 0 else 1)
    got = np.array(out[0].split(), float)
    assert np.allclose(got, [xa, ya], atol=1e-6)
    for line, (dx, dy) in zip(out[1:5], [(0, 0), (-0.1, 0.25), (0.5, -1.0), (-3.0, 0.3)]):
        xa = min(0.0, max(-(np.pi / 2 - 1e-4), xa + dx))
        ya = ya + dy
        want = np.array([np.sin(ya) * np.cos(xa), -np.sin(xa), np.cos(ya) * np.cos(xa)]) * dist
        assert np.allclose(np.array(line.split(), float), want, atol=2e-5)
    # the first update with zero deltas reproduces the original lookfrom
    assert np.allclose(np.array(out[1].split(), float), lf, atol=2e-5)
