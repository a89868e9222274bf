"""Long primitive lists (Scene::primitives, scene.h:5-19): the production kernel walks up to 1024 primitives
in list-order chunks of 64.  Parity = the hit records of the reference's linear walk (oracle), ties between primitives
resolved in list order."""
import numpy as np
import pytest

import ag_pathtracer_amd as ag


def _many_prims(n_prims, seed, duplicates=True):
    rng = np.random.RandomState(seed)
    d = ag.SceneDesc("toplevel-%d" % n_prims)
    mats = [d.add_material(ag.MAT_DIFFUSE_ONLY, rng.uniform(.2, .9, 3)) for _ in range(3)]
    blobs = []
    for i in range(n_prims):
        c = rng.uniform(-6, 6, 3)
        if i % 9 == 4:
            d.add_sphere(c, float(rng.uniform(0.1, 0.4)), mats[i % 3])
        elif duplicates and i % 50 == 49 and blobs:
            # the same geometry again, hundreds of list positions (several chunks) later: every ray that hits it ties
            v, n, t, idx = blobs[rng.randint(len(blobs))]
            d.add_mesh(v, n, t, idx, mats[i % 3], 1)
        else:
            v, n, t, idx = ag.scenes.blob_mesh(5, 4, center=tuple(c), radius=float(rng.uniform(0.15, 0.5)), seed=i)
            blobs.append((v, n, t, idx))
            d.add_mesh(v, n, t, idx, mats[i % 3], 1)
    return d


@pytest.mark.gpu
def test_gpu_900_primitives_hit_records_match_the_list_walk():
    """15 chunks (lists beyond 512 used to fall back to the reference-order kernel): production kernel == instrumented
    reference-order kernel == oracle, bit for bit, closest hit and any hit, with coincident meshes in different chunks."""
    from test_gpu_intersect import check_closest, random_rays
    d = _many_prims(900, 21)
    assert d.n_prims == 900
    hits = check_closest(d, random_rays(d, 120000, seed=5))
    assert hits > 15000


@pytest.mark.gpu
def test_gpu_render_700_primitives_against_oracle():
    from test_gpu_render import compare
    d = _many_prims(698, 22)
    d.add_area_light([0.0, 8.0, 0.0], 0.8, [70, 65, 60])
    d.add_plane([0, -6.5, 0], [10, 10], 0)
    d.add_uniform_infinite_light([.25, .3, .35])
    d.set_camera([0, 3, -16], [0, 0, 0], [0, 1, 0], 1.5, 50.0, 0.0)
    assert d.n_prims == 700
    compare(d, 64, 48, 2)
