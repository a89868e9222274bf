"""Accumulator::CopyToSurface (myapp.h:34-41) = lin2rgb + rgb2uint (template/common.h:41-51): the oracle's restatement
against hand-derived values, and the device's agpt_resolve / k_resolve against the oracle.

Parity unpinned against the reference itself (it has no recorded CopyToSurface output).  Device tolerance: the gamma is
powf(c, 1/2.2f), implementation-defined in the reference too (MSVC CRT there, glibc in the oracle, ROCm's ocml on the
device), so a channel may differ by 1 LSB where 256*pow lands within a few ulps of an integer; everything else is exact.
"""
import numpy as np
import pytest

import ag_pathtracer_amd as ag
from helpers import gpu_context, gpu_scene


def channels(words):
    w = np.asarray(words, np.uint32)
    return np.stack([(w >> 16) & 255, (w >> 8) & 255, w & 255], -1).astype(np.int32)


EDGE = np.float32([
    [0, 0, 0, 0],                      # black
    [1, 1, 1, 7],                      # sum / samples = 1 -> clamp .999 -> int(255.744)
    [.5, .25, .125, 0],
    [1e-30, 1e-38, 1e-45, 0],          # tiny and subnormal sums
    [3.0e38, 1e10, 65504, 0],          # huge
    [-0.5, -0.0, -1e-20, 0],           # negative sums: pow -> NaN -> fminf(NaN, .999) = .999 -> 255; -0 -> pow(-0, e) = 0
    [np.nan, np.inf, .999, 0],
    [.21404114, .21586050, .2177376, 0],   # around 0.5^2.2: quantises near 127/128
])


def expected_channel(c, samples):
    """fp64 model of one channel, used only where the result is far from an integer boundary."""
    x = np.float32(c) / np.float32(samples)
    if np.isnan(x) or x < 0:
        return 255
    p = float(x) ** float(np.float32(1) / np.float32(2.2))
    p = min(p, float(np.float32(0.999)))
    return int(256 * p)


def test_oracle_copy_to_surface_edge_values(oracle):
    spp = 1
    out = channels(oracle.copy_to_surface(EDGE, spp))
    for i, px in enumerate(EDGE):
        for k in range(3):
            v = 256 * min(float(abs(px[k])) ** (1 / 2.2), 0.999) if np.isfinite(px[k]) else 0.
            if np.isfinite(px[k]) and px[k] >= 0 and abs(v - round(v)) < 1e-3:
                continue   # too close to an integer boundary for the fp64 model
            assert out[i, k] == expected_channel(px[k], spp), (i, k, px[k], out[i, k])
    assert out[0].tolist() == [0, 0, 0] and out[1].tolist() == [255, 255, 255]
    assert out[5].tolist() == [255, 0, 255]           # negative -> NaN -> 255, -0 -> 0
    assert out[6].tolist()[:2] == [255, 255]
    # division by the sample count happens BEFORE the gamma (myapp.h:37)
    acc = np.float32([[2.0, 1.0, 0.5, 0]])
    assert channels(oracle.copy_to_surface(acc, 4))[0].tolist() == [expected_channel(2.0, 4), expected_channel(1.0, 4),
                                                                   expected_channel(.5, 4)]
    # words are (r << 16) + (g << 8) + b with nothing in the top byte
    assert int(oracle.copy_to_surface(np.float32([[1, 0, 1, 0]]), 1)[0]) == 0x00FF00FF


def test_oracle_copy_to_surface_monotone_ramp(oracle):
    ramp = np.zeros((4096, 4), np.float32)
    ramp[:, 0] = np.linspace(0, 1.2, 4096, dtype=np.float32)
    ramp[:, 1] = ramp[:, 0] * 2
    ramp[:, 2] = ramp[:, 0] * 3
    out = channels(oracle.copy_to_surface(ramp, 3))
    assert (np.diff(out, axis=0) >= 0).all() and out[:, 2].max() == 255 and out[-1, 0] == 168 and not out[0].any()
    assert np.array_equal(out[:, 0], channels(oracle.copy_to_surface(ramp[:, [0, 0, 0, 3]], 3))[:, 1])


@pytest.mark.gpu
def test_gpu_resolve_matches_oracle(oracle):
    ctx = gpu_context()
    # (1) edge values and a dense ramp, uploaded as an accumulator
    ramp = np.zeros((1 << 16, 4), np.float32)
    rng = np.random.RandomState(3)
    ramp[:, :3] = (rng.uniform(0, 1.3, (1 << 16, 3)) ** 3 * 5).astype(np.float32)
    for acc, spp in ((EDGE, 1), (EDGE, 3), (ramp, 5)):
        ptr = ctx.alloc(acc.nbytes)
        try:
            ctx.upload(ptr, acc)
            g = channels(ctx.resolve(ptr, len(acc), spp))
        finally:
            ctx.free(ptr)
        o = channels(oracle.copy_to_surface(acc, spp))
        d = np.abs(g - o)
        assert d.max() <= 1, "agpt_resolve differs from CopyToSurface by more than 1 LSB"
        assert (d == 0).mean() > 0.995
    # (2) a rendered accumulator (BASELINE config 1)
    desc = ag.scenes.scene_c1()
    scene = gpu_scene(desc)
    W = H = 96
    spp = 3
    ptr = ctx.alloc(W * H * 16)
    try:
        ctx.memset(ptr, 0, W * H * 16)
        ag.PathTracer(5).render(scene, W, H, spp, ptr)
        acc = ctx.download(ptr, (H, W, 4))
        g = channels(ctx.resolve(ptr, W * H, spp)).reshape(H, W, 3)
    finally:
        ctx.free(ptr)
        scene.close()
    o = channels(oracle.copy_to_surface(acc, spp)).reshape(H, W, 3)
    d = np.abs(g - o)
    assert d.max() <= 1 and (d == 0).mean() > 0.995
    assert g.std() > 10   # an actual image, not a constant
