"""Host builder of the top-level structure over Scene::primitives (scene.h:5-19; agpt_toplevel_build, the tree k_candidates walks
for lists longer than 64 primitives): structure of the depth-first / skip-link layout, and the stackless walk finds exactly
the boxes a per-box test finds.  CPU only; the GPU side is tests/test_gpu_long_lists.py."""
import numpy as np
import pytest

import ag_pathtracer_amd as ag


def _boxes(n, seed, flat=False):
    rng = np.random.RandomState(seed)
    c = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    h = rng.uniform(0.05, 1.5, (n, 3)).astype(np.float32)
    if flat:
        h[rng.uniform(size=n) < 0.3, 1] = 0   # zero-thickness boxes (axis-aligned quads)
    return np.concatenate([c - h, c + h], axis=1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 157, 960])
def test_skip_tree_structure(n):
    boxes = _boxes(n, n, flat=True)
    lo, hi, skip, leaf = ag.toplevel_build(boxes)
    m = 2 * n - 1
    assert lo.shape == (m, 3) and skip.shape == (m,)
    # every box is exactly one leaf, with its own bounds
    leaves = np.nonzero(leaf >= 0)[0]
    assert sorted(leaf[leaves].tolist()) == list(range(n))
    assert np.array_equal(lo[leaves], boxes[leaf[leaves], 0:3]) and np.array_equal(hi[leaves], boxes[leaf[leaves], 3:6])
    depth = np.zeros(m, int)
    for i in range(m):
        assert i < skip[i] <= m
        if leaf[i] >= 0:
            assert skip[i] == i + 1
        else:
            # depth-first layout: children at i+1 and skip[i+1], subtree ends at skip[i]; bounds = union of the children
            a, b = i + 1, skip[i + 1]
            assert b < skip[i] and skip[b] == skip[i]
            assert np.array_equal(lo[i], np.minimum(lo[a], lo[b])) and np.array_equal(hi[i], np.maximum(hi[a], hi[b]))
            depth[a] = depth[b] = depth[i] + 1
    assert depth.max() <= (int(np.ceil(np.log2(n))) if n > 1 else 0)   # median splits: balanced


def _slab_hit(lo, hi, o, inv, tmax):
    t0, t1 = (lo - o) * inv, (hi - o) * inv
    tmin = max(np.minimum(t0, t1).max(), 0.0)
    tmx = min(np.maximum(t0, t1).min(), tmax)
    return tmx * (1 + 4e-6) + 1e-4 >= tmin     # conservative, like the kernel's test


def test_skip_walk_finds_every_box_the_per_box_test_finds():
    n = 200
    boxes = _boxes(n, 3, flat=True)
    lo, hi, skip, leaf = ag.toplevel_build(boxes)
    rng = np.random.RandomState(4)
    visited_total = 0
    for _ in range(300):
        o = rng.uniform(-12, 12, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        inv = 1.0 / d
        tmax = rng.choice([np.float32(3.4e38), rng.uniform(1, 20)])
        want = {k for k in range(n) if _slab_hit(boxes[k, 0:3].astype(np.float64), boxes[k, 3:6].astype(np.float64), o, inv, tmax)}
        got, i = set(), 0
        while i < len(skip):
            visited_total += 1
            if _slab_hit(lo[i].astype(np.float64), hi[i].astype(np.float64), o, inv, tmax):
                if leaf[i] >= 0:
                    got.add(int(leaf[i]))
                i += 1
            else:
                i = int(skip[i])
        assert got == want
    assert visited_total < 0.35 * 300 * n    # far fewer box tests than one per primitive


def test_toplevel_build_rejects_bad_arguments():
    with pytest.raises(ag.AgptError):
        ag.toplevel_build(np.zeros((0, 6), np.float32))


def test_pack16_rounds_boxes_outward():
    """The GPU reads the tree as 16-byte nodes with half-precision boxes: every packed box must contain the fp32 one, be at most one
    half step larger per coordinate, use no half denormals, and keep the links."""
    rng = np.random.RandomState(7)
    n = 300
    boxes = _boxes(n, 11, flat=True)
    # adversarial coordinates: tiny (below the smallest normal half), exactly representable, huge, zero and negative zero
    boxes[0] = [-1e-7, 1e-7, -3e-6, 2e-7, 4e-6, 5e-6]
    boxes[1] = [0.5, -0.25, 1024.0, 0.5, -0.25, 2048.0]
    boxes[2] = [-7e4, -1e5, 6.5e4, 7e4, 1e5, 6.6e4]
    boxes[3] = [0.0, -0.0, 0.0, 0.0, 0.0, -0.0]
    boxes[4] = [-65504.0, 65504.0, -65505.0, -65504.0, 65505.0, 65504.0]
    lo, hi, skip, leaf = ag.toplevel_build(boxes)
    lo16, hi16, skip16, leaf16 = ag.toplevel_pack16(lo, hi, skip, leaf)
    assert np.array_equal(skip16, skip) and np.array_equal(leaf16, leaf)
    lo_f, hi_f = lo16.astype(np.float32), hi16.astype(np.float32)
    assert np.all(lo_f <= lo) and np.all(hi_f >= hi)
    for packed, exact, up in ((lo16, lo, False), (hi16, hi, True)):
        bits = packed.view(np.uint16)
        denormal = ((bits & 0x7C00) == 0) & ((bits & 0x03FF) != 0)
        assert not denormal.any()
        # tight: one step back towards the exact value would cross it (or the value is the denormal snap / an infinity)
        with np.errstate(over="ignore"):
            inward = np.nextafter(packed, np.float16(-np.inf) if up else np.float16(np.inf))
        crosses = (inward.astype(np.float32) < exact) if up else (inward.astype(np.float32) > exact)
        small = np.abs(exact) < 2.0 ** -14
        assert np.all(crosses | small | np.isinf(packed.astype(np.float32)))
    assert hi16[leaf == 2][0][0] == np.float16(np.inf) and lo16[leaf == 2][0][1] == np.float16(-np.inf)
