"""bench.py's bookkeeping that does not need a GPU: which committed PMC profile a run may use for its per-ray constants."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pmc_profile_is_keyed_by_kernel_build_and_workload(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    mk = lambda name, build, wl, extra=None: (prof / name).write_text(json.dumps(dict(kernel_build=build, workload=wl, trace={"valu_insts_per_traversed_ray": 1.0},
                                                                                     **(extra or {}))))
    mk("pmc_a.json", "BUILD", "c3:1920x1080:64spp:world1:scale1")
    mk("pmc_b.json", "OTHER", "c3:1920x1080:64spp:world1:scale1")
    mk("pmc_c.json", "BUILD", "c5:1920x1080:64spp:world1:scale1")
    (prof / "pmc_l1.json").write_text(json.dumps(dict(kernel_build="BUILD", workload="c3:1920x1080:64spp:world1:scale1")))   # no "trace": not a roofline profile
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_build_id", lambda: "BUILD")
    p = bench.load_pmc_profile("c3:1920x1080:64spp:world1:scale1")
    assert p and p["file"].endswith("pmc_a.json") and p["stand_in"] is False
    # a rank's share of the same film: the whole film's per-ray constants stand in, flagged
    p = bench.load_pmc_profile("c3:1920x1080:64spp:world8:scale1")
    assert p and p["file"].endswith("pmc_a.json") and p["stand_in"] is True
    # an own profile of the split wins over the stand-in
    mk("pmc_d.json", "BUILD", "c3:1920x1080:64spp:world8:scale1")
    p = bench.load_pmc_profile("c3:1920x1080:64spp:world8:scale1")
    assert p and p["file"].endswith("pmc_d.json") and p["stand_in"] is False
    # other film, other sample count, other build: nothing
    assert bench.load_pmc_profile("c3:1920x1080:8spp:world2:scale1") is None
    assert bench.load_pmc_profile("c3:3840x2160:64spp:world1:scale1") is None
    monkeypatch.setattr(bench, "kernel_build_id", lambda: "NEWER")
    assert bench.load_pmc_profile("c3:1920x1080:64spp:world1:scale1") is None


def test_kernel_build_id_covers_every_kernel_source():
    """The id profiles are keyed by is a hash over csrc/ and build.py: a new translation unit or header is part of it."""
    import inspect
    src = inspect.getsource(bench.kernel_build_id)
    assert "csrc" in src and "build.py" in src
    a = bench.kernel_build_id()
    assert len(a) == 16 and a == bench.kernel_build_id()
