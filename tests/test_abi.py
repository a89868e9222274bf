"""CPU-only checks of the drop-in boundary: libagpt_hip.so loads without a GPU, exports every function
include/agpt.h declares, and its host-only entry points (BVH build, backdrop) behave like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest

import ag_pathtracer_amd as ag

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "agpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(agpt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = ag.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libagpt_hip.so does not export %s" % n
    assert sorted(ag.EXPORTS) == names
    assert L.agpt_version() >= 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ag.AgptError):
        ag.Context(0)


def test_host_bvh_build_matches_oracle(oracle):
    rng = np.random.RandomState(3)
    cases = []
    cases.append(oracle.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32))
    cases.append(oracle.pin_heightfield(40, 2.0))
    cases.append(ag.scenes.blob_mesh(24, 25))
    # soup with duplicated triangles (identical centroids -> forced multi-prim leaves) and random sizes
    v = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    t = rng.randint(300, size=(500, 3)).astype(np.int32)
    t = np.concatenate([t, t[:40], t[:40]])
    cases.append((v, None, None, np.stack([t.reshape(-1)] * 3, 1)))
    for maxp in (1, 4):
        for (v, n, uv, idx) in cases:
            nodes, order, depth = ag.bvh_build(v, idx, maxp)
            s = oracle.OracleScene()
            m = s.add_material(oracle.MAT_DIFFUSE_ONLY, [.5, .5, .5])
            p = s.add_mesh(v, n, uv, idx, m, maxp)
            on, oo = s.bvh(p)
            assert on.tobytes() == nodes.tobytes()
            assert np.array_equal(oo, order)


def test_host_bvh_pinned_node_counts(oracle):
    """The product's own builder reproduces the reference's recorded tree shapes."""
    v, n, t, idx = oracle.pin_heightfield(187, 2.0)
    nodes, order, depth = ag.bvh_build(v, idx, 1)
    assert (len(nodes) - 1, depth) == (107269, 20)
    v, n, t, idx = oracle.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32)
    nodes, order, depth = ag.bvh_build(v, idx, 1)
    assert len(nodes) - 1 == 71
    v, n, t, idx = oracle.pin_heightfield(361, 2.0)
    nodes, order, depth = ag.bvh_build(v, idx, 1)
    assert (len(nodes) - 1, depth) == (393989, 23)
    v, n, t, idx = oracle.pin_heightfield(1581, 2.0)          # 4,999,122 triangles, multi-threaded build
    nodes, order, depth = ag.bvh_build(v, idx, 1)
    assert (len(nodes) - 1, depth) == (7465695, 28)


def test_backdrop_matches_oracle(oracle):
    a = ag.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32)
    b = oracle.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()


def test_invalid_arguments_return_errors():
    L = ag.lib()
    assert L.agpt_bvh_build(None, 0, None, 0, 1, None, None, None, None) < 0
    assert b"agpt_bvh_build" in L.agpt_last_error()
    with pytest.raises(ag.AgptError):
        ag.bvh_build(np.zeros((3, 3), np.float32), np.array([[0, 0, 0], [1, 1, 1], [7, 7, 7]], np.int32))


def test_header_is_plain_c(tmp_path):
    """include/agpt.h is the drop-in boundary: it must compile as C99 (no C++ or torch types in the signatures)."""
    import subprocess
    src = tmp_path / "c99.c"
    src.write_text('#include "agpt.h"\nint main(void) { agpt_ctx* c = 0; return agpt_init(0, &c); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])
