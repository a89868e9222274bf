#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the CPU oracle (oracle/oracle.c, itself pinned to the
reference's recorded outputs -- tests/test_oracle_pins.py).  The reference cannot run in this image, so these vectors
are oracle outputs: they freeze today's pinned behaviour so that the oracle and the HIP path cannot drift together.

    python tests/golden/make_golden.py        (rewrites tests/golden/*.npz; commit the result)

Fixtures are data only: seeded inputs and expected outputs.
  c1_render_64x64_3spp.npz   accumulator [64,64,4] of BASELINE config 1 (per-sample RNG, trig mode 1 = correctly rounded),
                             ray counts; the same render in trig mode 0 (libm) for the oracle-only check
  c1_hits.npz                4096 seeded rays -> closest-hit records and any-hit flags
  bsdf_kat.npz               f / Pdf / Sample_f known answers: Disney rough .25/.5/1 x metallic 0/1, diffuse-only, mirror
  rng.npz                    per-(pixel, sample) seeds and first floats
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ag_pathtracer_amd as ag  # noqa: E402
from oracle import binding as ob  # noqa: E402


def c1_desc():
    return ag.scenes.scene_c1(backdrop=ob.create_backdrop([0, -1, 20], [40, 20, 40], 7.5, 32))


def golden_rays(n=4096, seed=7):
    rng = np.random.RandomState(seed)
    rays = np.zeros(n, ag.RAY_DTYPE)
    rays["o"] = (rng.uniform(-4, 4, (n, 3)) + [0, 2, -2]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d[::16] = np.eye(3)[rng.randint(3, size=len(d[::16]))] * rng.choice([-1, 1], size=(len(d[::16]), 1))
    rays["d"] = d.astype(np.float32)
    rays["tmax"] = np.where(rng.uniform(size=n) < 0.25, rng.uniform(0.5, 30, n), 3.402823466e+38).astype(np.float32)
    return rays


def kat_materials(scene):
    mats = []
    for rough in (.25, .5, 1.0):
        for metal in (0.0, 1.0):
            mats.append(scene.add_material(ag.MAT_DISNEY, [0.944, 0.776, 0.373], rough, metal))
    mats.append(scene.add_material(ag.MAT_DIFFUSE_ONLY, [.7, .6, .5]))
    mats.append(scene.add_material(ag.MAT_MIRROR, [.9, .8, .7]))
    return mats


def kat_inputs(n=512, seed=9):
    rng = np.random.RandomState(seed)
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) * np.where(np.arange(n) % 8 == 0, -1, 1)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rng.normal(size=(n, 3))
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    u = rng.uniform(size=(n, 2))
    u[0] = [0.0, 0.0]
    u[1] = [0.5, 0.5]
    u[2] = [0.999999, 0.25]
    wo[3] = [0, 0, 1]
    return wo.astype(np.float32), wi.astype(np.float32), u.astype(np.float32)


def main():
    o = c1_desc().instantiate(ob.OracleScene())
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    acc1, st1 = o.render(64, 64, 3)
    ob.set_trig_mode(ob.TRIG_LIBM)
    acc0, st0 = o.render(64, 64, 3)
    np.savez_compressed(os.path.join(HERE, "c1_render_64x64_3spp.npz"), accum_cr=acc1, accum_libm=acc0,
                        rays_cr=np.int64([st1.closest_rays, st1.anyhit_rays]), rays_libm=np.int64([st0.closest_rays, st0.anyhit_rays]))
    rays = golden_rays()
    hc, _ = o.intersect(rays, any_hit=False)
    ha, _ = o.intersect(rays, any_hit=True)
    np.savez_compressed(os.path.join(HERE, "c1_hits.npz"), rays=rays, closest=hc, anyhit=ha["hit"])

    s = ob.OracleScene()
    mats = kat_materials(s)
    wo, wi, u = kat_inputs()
    ob.set_trig_mode(ob.TRIG_CORRECTLY_ROUNDED)
    out = {}
    for m in mats:
        f = np.zeros((len(wo), 3), np.float32)
        pdf = np.zeros(len(wo), np.float32)
        swi = np.zeros((len(wo), 3), np.float32)
        sf = np.zeros((len(wo), 3), np.float32)
        spdf = np.zeros(len(wo), np.float32)
        spec = np.zeros(len(wo), np.int32)
        for i in range(len(wo)):
            f[i], pdf[i] = s.bsdf_eval(m, wo[i], wi[i])
            swi[i], sf[i], spdf[i], spec[i] = s.bsdf_sample(m, wo[i], u[i])
        out["f_%d" % m], out["pdf_%d" % m] = f, pdf
        out["swi_%d" % m], out["sf_%d" % m], out["spdf_%d" % m], out["spec_%d" % m] = swi, sf, spdf, spec
    ob.set_trig_mode(ob.TRIG_LIBM)
    np.savez_compressed(os.path.join(HERE, "bsdf_kat.npz"), wo=wo, wi=wi, u=u, n_materials=np.int32(len(mats)), **out)

    seeds = np.array([[p, s_, ob.sample_seed(p, 1920 * 1080, s_, sb)] for p in (0, 1, 12345, 2073599) for s_ in (0, 1, 63)
                      for sb in (0, 0x12345678)], np.int64)
    floats = np.stack([ob.rng_floats(int(sd), 8)[0] for sd in seeds[:, 2]])
    np.savez_compressed(os.path.join(HERE, "rng.npz"), seeds=seeds, floats=floats,
                        seed_bases=np.int64([0, 0x12345678] * (len(seeds) // 2)))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
