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
