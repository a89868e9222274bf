#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X path-tracing hot path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] -- the Sponza-class closed atrium (258,672 triangles, 37
primitives, 35 Disney materials, 2 sphere lights + uniform sky), 1920x1080, 64 spp, PathTracer(MaxDepth=5) with
NEE + MIS; synthetic geometry (the reference ships no assets), seeds fixed.  One "step" = one pass of the hot path
over that whole sample budget: jitter -> Camera::GetRay -> PathTracer::Li -> Accumulator::AddSample for every
(pixel, sample), scene already resident in HBM.

N > 1 (weak scaling): the film's 8-row blocks are dealt round-robin to the ranks and the sample budget is 64*N spp,
so every GPU traces the same number of camera samples as at N = 1; the step ends with the RCCL gather of the
per-rank tile buffers to rank 0.  value = rays traced by all ranks / max-over-ranks time.

The printed JSON line carries `roofline` (dominant kernel: the persistent closest-hit trace kernel; achieved =
algorithmic bytes / HIP-event launch time) and `cpu_baseline` (the CPU oracle timed on this
host's cores on a bounded sample of the same workload -- a reported baseline, not the thing measured).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(closest, anyhit, roots, interior, tris):
    """SURVEY.md section 8(d), reference data layout: per ray 28 (ray) + 16 (hit) + 32 per mesh root box tested
    + 64 per interior node whose child pair is fetched + 88 per triangle tested (4 prim index + 36 index_type +
    48 vertices).  Any-hit rays write a 4-byte flag instead of a 16-byte hit."""
    return 28 * (closest + anyhit) + 16 * closest + 4 * anyhit + 32 * roots + 64 * interior + 88 * tris


def effective_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / float(per) + 0.5)))
        except Exception:  # noqa: BLE001
            pass
    return n


def cpu_baseline(desc, W, H, max_depth, budget_s=12.0):
    """Times the CPU oracle (kind 'port': the C restatement of the reference's path, pinned to the reference's
    recorded outputs -- the reference itself cannot travel to this box) on centred tiles of the same film,
    about `budget_s` seconds single-threaded (the reference as shipped: myapp.cpp:163-175) and the same again with
    OpenMP over scanlines on all usable host cores."""
    from oracle import binding as ob
    ob.build()
    cores = effective_cores()
    o = desc.instantiate(ob.OracleScene())
    o.set_max_depth(max_depth)

    def timed(tw, th, threads):
        tile = ((W - tw) // 2, (H - th) // 2, tw, th)
        t0 = time.time()
        _, st = o.render(W, H, 1, tile=tile, seed_base=0, rng_mode=ob.RNG_PER_SAMPLE, threads=threads)
        probe = max(time.time() - t0, 1e-3)
        spp = int(max(1, min(64, budget_s / probe)))
        t0 = time.time()
        _, st = o.render(W, H, spp, tile=tile, seed_base=0, rng_mode=ob.RNG_PER_SAMPLE, threads=threads)
        return st, time.time() - t0, spp

    st1, dt1, spp1 = timed(320, 180, 1)
    stn, dtn, sppn = timed(960, 540, cores)
    return {
        "value": round(stn.rays / dtn / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "centred 960x540 tile of the 1920x1080 film at %d spp (%d rays, %.1f s) on %d threads; "
                  "single thread: centred 320x180 tile at %d spp (%d rays, %.1f s)"
                  % (sppn, stn.rays, dtn, cores, spp1, st1.rays, dt1),
        "single_thread_value": round(st1.rays / dt1 / 1e6, 3),
        "rays_per_sample": round(stn.rays / float(stn.samples), 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel per GPU-share (total = spp * gpus)")
    ap.add_argument("--scene-scale", type=float, default=1.0)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3 = the BASELINE metric's configuration (default); c5 = the 5M-triangle multi-mesh scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--samples-per-batch", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import ag_pathtracer_amd as ag
    from ag_pathtracer_amd import tiles

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one process per GPU; AGPT_BENCH_BACKEND=gloo lets the N > 1 path be rehearsed with several ranks sharing one GPU
    backend = os.environ.get("AGPT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    W, H, MAXD = args.width, args.height, 5
    spp = args.spp * world
    if args.workload == "c5":
        desc = ag.scenes.scene_c5(scale=args.scene_scale, aspect=W / float(H))
        wl_name = "BASELINE configs[4] geometry: 5M-triangle multi-mesh colonnade field, all-metal GGX"
    else:
        desc = ag.scenes.scene_c3(scale=args.scene_scale, aspect=W / float(H))
        wl_name = "BASELINE configs[2]: Sponza-class closed atrium"
    ctx = ag.Context(dev_index, stream=torch.cuda.current_stream().cuda_stream)
    scene = desc.instantiate(ag.Scene(ctx))
    pt = ag.PathTracer(MAXD)

    rows = tiles.max_local_rows(H, world)
    local = torch.zeros((rows, W, 4), dtype=torch.float32, device=dev)
    # one agpt_render call per step and rank: the whole film at N = 1, this rank's interleaved 8-row blocks otherwise
    interleave = None if world == 1 else (tiles.BLOCK_ROWS, world, rank)

    def step(timing):
        local.zero_()
        tot = dict(closest=0, anyhit=0, trace_ms=0.0, ext_ms=0.0, mis_ms=0.0, shadow_ms=0.0, total_ms=0.0, launches=0,
                   iterations=0)
        for _ in (0,):
            st = pt.render(scene, W, H, spp, local.data_ptr(), accum_pitch=W, samples_per_batch=args.samples_per_batch,
                           timing=timing, interleave=interleave)
            tot["closest"] += st.closest_rays
            tot["anyhit"] += st.anyhit_rays
            tot["trace_ms"] += st.trace_ms
            tot["ext_ms"] += st.ext_ms
            tot["mis_ms"] += st.mis_ms
            tot["shadow_ms"] += st.shadow_ms
            tot["total_ms"] += st.total_ms
            tot["launches"] += st.trace_launches
            tot["iterations"] += st.iterations
        full = tiles.gather_to_rank0(local, W, H, rank, world)
        return tot, full

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    acc = None
    sums = None
    for _ in range(args.steps):
        tot, full = step(True)
        if sums is None:
            sums = tot
        else:
            for k in tot:
                sums[k] += tot[k]
        acc = full
    barrier()
    dt = time.perf_counter() - t0

    rays_local = float(sums["closest"] + sums["anyhit"])
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    r_all = torch.tensor([rays_local], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(r_all, op=dist.ReduceOp.SUM)
    dt_max = float(t_all.item())
    rays_total = float(r_all.item())

    if rank == 0:
        # instrumented pass outside the timed region: deterministic per-scene work counters for the roofline
        # (interior child-pair fetches, triangle tests, root-box tests of the SAME rays), on a 1-spp sample
        cst = pt.render(scene, W, H, min(spp, 2), local.data_ptr(), accum_pitch=W, counters=True, interleave=interleave)
        crays = float(cst.rays)
        bytes_per_ray = algorithmic_bytes(cst.closest_rays, cst.anyhit_rays, cst.root_tests, cst.interior_visits,
                                          cst.tri_tests) / crays
        trace_s = sums["trace_ms"] / 1e3
        launches = max(1, int(sums["launches"]))
        rays_rank0 = rays_local
        achieved = bytes_per_ray * rays_rank0 / trace_s / 1e9 if trace_s > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj) and args.workload == "c3":
            try:
                traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        img_mean = acc[..., :3].double().mean(dim=(0, 1)).div(spp).tolist() if acc is not None else None
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1080p/64spp", "value": round(rays_total / dt_max / 1e6, 2),
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s, %d tris / %d primitives / %d Disney materials, "
                            "%dx%d, %d spp (%d spp x %d GPU shares), PathTracer MaxDepth=5, NEE+MIS"
                            % (wl_name, desc.n_tris, desc.n_prims, desc.n_materials, W, H, spp, args.spp, world),
                "triangles": desc.n_tris, "resolution": [W, H], "spp": spp, "sharding": "8-row blocks round-robin, gather to rank 0",
                "rays_per_step": int(rays_total / args.steps), "rays_per_sample": round(rays_total / args.steps / (W * H * spp), 2),
                "msamples_per_s": round(W * H * spp * args.steps / dt_max / 1e6, 2),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "k_trace (persistent closest-/any-hit BVH traversal)",
                "algorithmic_bytes_per_ray": round(bytes_per_ray, 1),
                "interior_visits_per_ray": round(cst.interior_visits / crays, 2), "tri_tests_per_ray": round(cst.tri_tests / crays, 2),
                "root_tests_per_ray": round(cst.root_tests / crays, 2),
                "launches": launches, "avg_launch_ms": round(sums["trace_ms"] / launches, 4),
                "algorithmic_bytes_per_launch": round(bytes_per_ray * rays_rank0 / launches, 0),
                "trace_ms_per_step": round(sums["trace_ms"] / args.steps, 2), "render_ms_per_step": round(sums["total_ms"] / args.steps, 2),
                "trace_split_ms_per_step": {"continuation": round(sums["ext_ms"] / args.steps, 2), "mis": round(sums["mis_ms"] / args.steps, 2),
                                            "shadow": round(sums["shadow_ms"] / args.steps, 2)},
                "trace_only_mrays_s": round(rays_rank0 / trace_s / 1e6, 1) if trace_s > 0 else None,
                # measured HBM rate of the same kernel: PMC traffic per launch (profiles/traffic.json) / live launch time
                "traffic_gbs": round(traffic / (sums["trace_ms"] / launches * 1e-3) / 1e9, 1) if traffic and trace_s > 0 else None,
                "traffic_frac_of_peak": round(traffic / (sums["trace_ms"] / launches * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if traffic and trace_s > 0 else None,
            },
            "image_mean_rgb": [round(x, 6) for x in img_mean] if img_mean else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, W, H, MAXD)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
