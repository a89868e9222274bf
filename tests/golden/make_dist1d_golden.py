"""Generates tests/golden/dist1d.npz from the REFERENCE's own sampling.h (oracle/_ref/libref_sampling.so, built by
`make -C oracle ref` from /root/reference/sampling.h where it lies).  Run in the build container only:

    python tests/golden/make_dist1d_golden.py

Each case is a step function func[n] and k draws u[k]; the fixture stores what Distribution1D's constructor leaves in
cdf / funcInt (sampling.h:21-33) and what SampleContinuous (:37-52, through FindInterval :4-17) and SampleDiscrete
(:54-61) return for every u, all as float32 bit patterns.  tests/test_dist1d.py requires the oracle (CPU) and
agpt_kat_distribution1d (GPU) to reproduce them bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
F = np.float32


def cases():
    r = np.random.RandomState(20260305)
    c = []
    c.append(("single_bin", np.array([2.5], F)))
    c.append(("two_bins", np.array([1, 3], F)))
    c.append(("all_zero", np.zeros(7, F)))                       # funcInt == 0: the uniform ramp, pdf 0
    c.append(("leading_and_trailing_zeros", np.array([0, 0, 1, 4, 0, 2, 0, 0], F)))   # flat cdf segments: du is not divided
    c.append(("one_spike", np.concatenate([np.zeros(40, F), [1e6], np.zeros(23, F)]).astype(F)))
    c.append(("uniform_257", np.ones(257, F)))
    c.append(("random_1000", r.uniform(0, 1, 1000).astype(F)))
    c.append(("wide_range_4096", (10.0 ** r.uniform(-6, 4, 4096)).astype(F)))
    # the shape InfiniteAreaLight builds (lights.cpp:36-46): max(rgb) * sin(theta_row) over a 64 x 32 image with a sun
    w, h = 64, 32
    img = r.uniform(0, 1, (h, w)).astype(F)
    img[5, 17] = F(9000)
    th = ((np.arange(h, dtype=F) + F(.5)) * F(np.pi) / F(h)).astype(F)
    c.append(("env_64x32", (img * np.sin(th).astype(F)[:, None]).astype(F).reshape(-1)))
    c.append(("large_16384", r.gamma(0.3, 2.0, 16384).astype(F)))   # running fp32 sum over 16,384 terms: order matters
    c.append(("denormal_sum", np.full(16, 1e-41, F)))               # func / n underflows towards 0
    return c


def draws(cdf, r):
    """u values: uniform draws, the ends of [0, 1), every exact cdf value and its fp32 neighbours (the <= in FindInterval)."""
    k = min(cdf.size, 300)
    pick = cdf[np.unique(np.linspace(0, cdf.size - 1, k).astype(int))]
    near = np.concatenate([pick, np.nextafter(pick, F(2)), np.nextafter(pick, F(-1))]).astype(F)
    u = np.concatenate([r.uniform(0, 1, 2000).astype(F), [F(0), np.nextafter(F(1), F(0)), F(0.5), F(1)], near]).astype(F)
    # the integrator only ever passes RandomFloat() in [0, 1]; keep u = 1 (it clamps to the last interval) but nothing outside
    return u[(u >= 0) & (u <= 1)]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_sampling.so"))
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.ref_dist1d_build.argtypes = [fp, C.c_int, fp, fp]
    L.ref_dist1d_sample.argtypes = [fp, C.c_int, fp, C.c_int, fp, fp, ip, ip, fp]
    P = lambda a: a.ctypes.data_as(fp)
    out = {}
    names = []
    r = np.random.RandomState(7)
    for name, func in cases():
        func = np.ascontiguousarray(func, F)
        n = func.size
        cdf = np.zeros(n + 1, F)
        fi = np.zeros(1, F)
        L.ref_dist1d_build(P(func), n, P(cdf), P(fi))
        u = draws(cdf, r)
        k = u.size
        x, pdf, dpdf = np.zeros(k, F), np.zeros(k, F), np.zeros(k, F)
        off, disc = np.zeros(k, np.int32), np.zeros(k, np.int32)
        L.ref_dist1d_sample(P(func), n, P(u), k, P(x), P(pdf), off.ctypes.data_as(ip), disc.ctypes.data_as(ip), P(dpdf))
        assert np.array_equal(off, disc)
        names.append(name)
        for key, a in (("func", func), ("cdf", cdf), ("funcInt", fi), ("u", u), ("x", x), ("pdf", pdf), ("discrete_pdf", dpdf)):
            out[name + "/" + key] = a.view(np.uint32)
        out[name + "/offset"] = off
        print("%-28s n=%6d k=%5d funcInt=%g" % (name, n, k, fi[0]))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "dist1d.npz"), **out)


if __name__ == "__main__":
    main()
