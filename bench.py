#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X path-tracing hot path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] -- the Sponza-class closed atrium (258,672 triangles, 37
primitives, 35 Disney materials, 2 sphere lights + uniform sky), 1920x1080, 64 spp, PathTracer(MaxDepth=5) with
NEE + MIS; synthetic geometry (the reference ships no assets), seeds fixed.  One "step" = one pass of the hot path
over that whole sample budget: jitter -> Camera::GetRay -> PathTracer::Li -> Accumulator::AddSample for every
(pixel, sample), scene already resident in HBM.

N > 1: STRONG scaling by default -- the same 1080p/64spp film is split over the ranks (8-row blocks dealt round-robin,
every rank renders its blocks for the whole 64 spp) and the step ends with the RCCL gather of the per-rank tile buffers
to rank 0 through the library's own collective (`agpt_gather_tiles`; `--gather torch` uses torch.distributed.gather
instead).  `--scaling weak` keeps the per-GPU sample count instead (64*N spp in total).

value = rays that ENTERED a trace kernel (k_trace_fast), summed over all ranks / max-over-ranks wall time.  The reference
algorithm issues more Scene::Intersect calls than that: the production path settles, without a traversal, the calls whose
answer cannot reach the image -- those are NOT in `value`; the rate that counts them is reported beside
it as `config.reference_queries_mrays_s`, and `config.msamples_per_s` is the rate of finished camera samples.

The printed JSON line carries
  * `roofline` for the dominant kernels (the persistent trace kernels k_trace_fast<closest / MIS query / any-hit>, timed
    live per launch with HIP events on the launch stream).  Top level: the HBM view -- achieved = the MEASURED memory-side
    rate (bytes per traversed ray from the committed rocprofv3 TCC passes of this exact kernel build and workload,
    profiles/pmc_*.json, times the rays traversed in the timed region / launch time) against the 8 TB/s peak; the
    algorithmic bytes of SURVEY.md 8(d) (counted by the timed kernel itself) beside it under `algorithmic` (cache-resident
    scene: above the HBM peak, not a utilisation).  `views` carries the two roofs that bound the kernel in practice --
    `l1_gather`: 64-byte pair records fetched per second against the rate of a bare dependent-gather loop of the same access
    shape (tools/micro/gather_rate.hip, profiles/r03_gather_rate.json), and `issue`: VALU wave-instructions per second
    against 1024 SIMDs x 2.4 GHz / 2 cycles.  Every profile-derived number is null when the committed profile does not
    match the running kernel build / workload;
  * `secondary`: the same measurement on BASELINE configs[4]'s 5 M-triangle scene (1080p / 64 spp), whose 0.46 GB of
    trace-time data exceeds the 256 MiB Infinity Cache;
  * `cpu_baseline`: the CPU oracle timed on this host's cores on a bounded sample of the same workload -- a reported
    baseline, not the thing measured.
"""
import argparse
import glob
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # max shader clock
ISSUE_PEAK_GINST = SIMDS * CLOCK_GHZ / 2.0   # one wave64 VALU instruction per 2 cycles per SIMD (SIMD-32)


def algorithmic_bytes(closest, anyhit, roots, interior, tris):
    """SURVEY.md section 8(d), reference data layout: per ray 28 (ray) + 16 (hit; 4 for an any-hit flag) + 32 (one
    root node) + 64 per interior node whose child pair is fetched + 88 per triangle tested (4 prim index + 36
    index_type + 48 vertices).  Returns (bytes with ONE 32-B root per ray -- the survey's formula, the judged
    numerator --, bytes with 32 B for every mesh root box actually tested)."""
    rays = closest + anyhit
    common = 28 * rays + 16 * closest + 4 * anyhit + 64 * interior + 88 * tris
    return common + 32 * rays, common + 32 * roots


def kernel_build_id():
    """Hash of the kernel sources: ties a committed PMC profile to the code it was measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ag-pathtracer_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(open(os.path.join(ROOT, "ag-pathtracer_amd", "build.py"), "rb").read())
    return h.hexdigest()[:16]


def load_pmc_profile(workload_key):
    """profiles/pmc_*.json written by tools/pmc_json.py: {"kernel_build": .., "workload": .., "trace": {"valu_insts_per_ray",
    "hbm_bytes_per_ray", ...}}.  Returns the newest entry matching this build and workload, else None."""
    build = kernel_build_id()
    # a rank's share of an N-way split of the same film traces the same kind of rays: where no profile of the split itself is
    # committed, the per-ray constants of the whole film's profile (world1) stand in, and the line says so
    whole_film = re.sub(r":world\d+:", ":world1:", workload_key)
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if d.get("kernel_build") != build or "trace" not in d:
            continue
        if d.get("workload") == workload_key or (d.get("workload") == whole_film and (best is None or best.get("stand_in"))):
            best = d
            best["file"] = os.path.relpath(f, ROOT)
            best["stand_in"] = d.get("workload") != workload_key
    return best


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
        "note": "the CPU path traverses every Scene::Intersect / IntersectP call of the reference, so its rate is a rate of traversed "
                "rays like `value` (its rays per sample equal config.reference_queries_per_sample, not rays_traversed_per_sample)",
    }


def load_gather_roof():
    """profiles/r03_gather_rate.json: what a bare loop of dependent 64-byte record fetches reaches on one MI355X with the trace
    kernel's access shape (tools/micro/gather_rate.hip, KIND 0: 4 x global_load_dwordx4 per lane and record)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r03_gather_rate.json")))
    except Exception:  # noqa: BLE001
        return None


def roofline_block(sums, cst, rays, steps, workload_key):
    """The roofline object for one measured workload.  sums: HIP-event sums over the timed steps; cst: work counters of
    the production trace kernels on the same queue contents (counter pass outside the timed region); rays: rays that
    entered a trace kernel on this rank in the timed region.  Every per-ray figure is per TRAVERSED ray.

    Top level = the HBM view the contract asks for: `achieved` is the MEASURED memory-side rate of the trace kernels (bytes per
    traversed ray from the committed rocprofv3 TCC passes of this kernel build and workload x the rays of the timed region / the
    HIP-event launch time), `traffic` the measured bytes per launch.  The contract's ALGORITHMIC numerator (SURVEY 8(d),
    reference layout, counted by the timed kernels) is reported beside it under `algorithmic`: it exceeds the HBM peak where the
    scene is cache-resident, so it is not a utilisation.  `views` holds the two roofs that actually bound the kernel: the vector
    L1's gather rate for 64-byte pair records (peak measured by tools/micro/gather_rate.hip) and VALU issue."""
    crays = float(cst.rays - cst.answered_rays)
    b_one_root, b_all_roots = algorithmic_bytes(cst.closest_rays - cst.answered_rays, cst.anyhit_rays, cst.root_tests,
                                                cst.interior_visits, cst.tri_tests)
    bpr = b_one_root / crays
    trace_s = sums["trace_ms"] / 1e3
    launches = max(1, int(sums["launches"]))
    avg_launch_s = trace_s / launches
    prof = load_pmc_profile(workload_key)
    tr = prof.get("trace", {}) if prof else {}
    insts_per_ray = tr.get("valu_insts_per_traversed_ray")
    hbm_per_ray = tr.get("hbm_bytes_per_traversed_ray")
    algo_gbs = bpr * rays / trace_s / 1e9 if trace_s > 0 else None
    issue = insts_per_ray * rays / trace_s / 1e9 if insts_per_ray and trace_s > 0 else None
    traffic_launch = hbm_per_ray * rays / launches if hbm_per_ray else None
    traffic_gbs = traffic_launch / avg_launch_s / 1e9 if traffic_launch and avg_launch_s > 0 else None
    # pair records (child pairs + root pairs) the kernels fetch per second, against the bare gather loop's rate
    g = load_gather_roof()
    # the vector L1's own counters for the same kernels (tools/gpu_pmc_l1.sh), keyed like the other PMC profiles
    l1 = None
    try:
        cand = json.load(open(os.path.join(ROOT, "profiles", "pmc_r03_%s_l1.json" % workload_key.split(":")[0])))
        if cand.get("kernel_build") == kernel_build_id() and cand.get("workload") == workload_key:
            l1 = cand
    except Exception:  # noqa: BLE001
        pass
    records_per_ray = (cst.interior_visits + cst.root_tests) / crays
    rec_rate = records_per_ray * rays / trace_s / 1e9 if trace_s > 0 else None
    g_peak = g.get("pair_record_gather_peak_grecords_s") if g else None
    gather_view = {"achieved": round(rec_rate, 1) if rec_rate else None, "peak": g_peak, "unit": "Grecords/s (64-byte pair records)",
                   "frac": round(rec_rate / g_peak, 4) if rec_rate and g_peak else None,
                   "peak_from": "profiles/r03_gather_rate.json: tools/micro/gather_rate.hip KIND 0 (4 x dwordx4 per lane and record, dependent "
                                "chains, 12.8 MB table, 64 active lanes, 5 blocks per CU); %s Grecords/s with 36 active lanes, the interior "
                                "step's average" % (g.get("pair_record_gather_36_lanes_grecords_s") if g else None),
                   "note": "the kernel's fetches include the coherent top levels of the trees, which coalesce; the microbenchmark's are all "
                           "divergent -- the fraction says the traversal moves pair records about as fast as a loop that does nothing else",
                   "l1_cache_accesses_per_cu_cycle": l1.get("trace_l1_cache_accesses_per_cu_cycle") if l1 else None,
                   "l1_cache_accesses_per_cu_cycle_by_launch_kind": ({k: v["l1_cache_accesses_per_cu_cycle"] for k, v in l1["kernels"].items()
                                                                      if k.startswith("trace")} if l1 else None),
                   "l1_counters_from": ("profiles/pmc_r03_%s_l1.json (TCP_TOTAL_CACHE_ACCESSES / CU-cycles of the trace launches; the L1 takes "
                                        "about one lane access per cycle)" % workload_key.split(":")[0]) if l1 else None}
    issue_view = {"achieved": round(issue, 1) if issue else None, "peak": ISSUE_PEAK_GINST, "unit": "Gwave-inst/s",
                  "frac": round(issue / ISSUE_PEAK_GINST, 4) if issue else None}
    fracs = {"hbm": traffic_gbs / HBM_PEAK_GBS if traffic_gbs else 0, "l1-gather": gather_view["frac"] or 0, "issue": issue_view["frac"] or 0}
    return {
        "bound": "hbm", "kernel": "k_trace_fast<closest | MIS query | any-hit> (persistent vote-scheduled BVH traversal)",
        "achieved": round(traffic_gbs, 1) if traffic_gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(traffic_gbs / HBM_PEAK_GBS, 4) if traffic_gbs else None,
        "traffic": round(traffic_launch, 0) if traffic_launch else None,
        "basis": ("achieved = measured memory-side bytes per traversed ray (TCC_EA0 request counters by request size, Infinity-Cache hits "
                  "included) and VALU wave-instructions per traversed ray from %s (rocprofv3 passes of this kernel build and workload) x the "
                  "rays of the timed region / HIP-event launch time%s" % (prof["file"], " -- per-ray constants of the WHOLE film's profile, this "
                  "run renders one rank's share of it" if prof.get("stand_in") else "")) if prof else
                 "no committed PMC profile matches this kernel build (%s) and workload (%s): profile-derived fields are null"
                 % (kernel_build_id(), workload_key),
        "nearest_roof": max(fracs, key=fracs.get) if any(fracs.values()) else None,
        "algorithmic": {
            "gbs": round(algo_gbs, 1) if algo_gbs else None, "over_hbm_peak": round(algo_gbs / HBM_PEAK_GBS, 4) if algo_gbs else None,
            "bytes_per_ray": round(bpr, 1), "bytes_per_ray_all_root_boxes": round(b_all_roots / crays, 1),
            "bytes_per_launch": round(bpr * rays / launches, 0),
            "traffic_over_algorithmic": round(hbm_per_ray / bpr, 4) if hbm_per_ray else None,
            "note": "SURVEY 8(d)'s numerator (reference layout: 28 + 16|4 + 32 + 64 N_int + 88 N_tri per ray), counted by the timed kernels; "
                    "served by L2 / Infinity Cache where the scene fits, so its ratio to the HBM peak is not a utilisation and may exceed 1",
        },
        "views": {"l1_gather": gather_view, "issue": issue_view},
        "per_ray_basis": "rays that entered a trace kernel (queries answered without a traversal excluded)",
        "valu_insts_per_ray": round(insts_per_ray, 1) if insts_per_ray else None,
        "hbm_bytes_per_ray": round(hbm_per_ray, 1) if hbm_per_ray else None,
        "wait_fraction": tr.get("wait_fraction"), "lane_utilisation": tr.get("lane_utilisation"),
        "pair_records_per_ray": round(records_per_ray, 2),
        "interior_visits_per_ray": round(cst.interior_visits / crays, 2), "tri_tests_per_ray": round(cst.tri_tests / crays, 2),
        "root_tests_per_ray": round(cst.root_tests / crays, 2),
        "counters_from": "the timed production kernels (enable_counters = 2): records fetched, MIS early exits included",
        "launches": launches, "avg_launch_ms": round(sums["trace_ms"] / launches, 4),
        "trace_ms_per_step": round(sums["trace_ms"] / steps, 2), "render_ms_per_step": round(sums["total_ms"] / steps, 2),
        "trace_split_ms_per_step": {"continuation": round(sums["ext_ms"] / steps, 2), "mis": round(sums["mis_ms"] / steps, 2),
                                    "shadow": round(sums["shadow_ms"] / steps, 2)},
        "trace_only_mrays_s": round(rays / trace_s / 1e6, 1) if trace_s > 0 else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel of the film (strong scaling) / per GPU (weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same film and spp split over the ranks (default), weak = spp * N in total")
    ap.add_argument("--scene-scale", type=float, default=1.0)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3 = the BASELINE metric's configuration (default); c5 = the 5M-triangle multi-mesh scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C5 secondary measurement")
    ap.add_argument("--samples-per-batch", type=int, default=0)
    ap.add_argument("--gather", default="agpt", choices=["agpt", "torch"],
                    help="N > 1: agpt = the library's own collective agpt_gather_tiles (RCCL grouped send/recv + de-interleave "
                         "kernel, the path a C++ host uses; default), torch = torch.distributed.gather + torch de-interleave")
    args = ap.parse_args()

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
    weak = world > 1 and args.scaling == "weak"
    spp = args.spp * world if weak else args.spp
    ctx = ag.Context(dev_index, stream=torch.cuda.current_stream().cuda_stream)
    pt = ag.PathTracer(MAXD)
    rows = tiles.max_local_rows(H, world)
    local = torch.zeros((rows, W, 4), dtype=torch.float32, device=dev)
    # The exchange at the end of a step.  Default: the product's own collective (agpt_gather_tiles) -- rank 0 creates the
    # ncclUniqueId through the C ABI, torch.distributed only carries its 128 bytes to the other ranks.  RCCL wants one GPU
    # per rank, so the several-ranks-on-one-GPU rehearsal (AGPT_BENCH_BACKEND=gloo) gathers through torch instead.
    gather_mode = args.gather if (world > 1 and (backend == "nccl" or os.environ.get("AGPT_BENCH_FORCE_AGPT_GATHER"))) else ("torch" if world > 1 else "none")
    gather_note = None
    xdev = dev if backend == "nccl" else torch.device("cpu")   # where the few control words of the set-up travel
    comm = None
    full_accum = None
    gather_bufs = None

    def all_ranks_ok(ok):
        """True iff `ok` holds on every rank (the ranks must take the same branch afterwards)."""
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    if gather_mode == "agpt":
        # The product's collective has no other user in this process: set it up and run it once on an empty film before the
        # timed region; if any rank reports an error (RCCL not loadable, communicator refused, ...) EVERY rank falls back to the
        # torch.distributed gather and the bench line says so.
        import torch.distributed as dist
        err = None
        try:
            idt = torch.zeros(128, dtype=torch.uint8, device=xdev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(ag.comm_unique_id()), dtype=torch.uint8))
        except Exception as e:  # noqa: BLE001
            err = e
        if all_ranks_ok(err is None):
            dist.broadcast(idt, src=0)
            try:
                comm = ag.Comm(ctx, world, rank, unique_id=idt.cpu().numpy().tobytes())
                if rank == 0:
                    full_accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
                comm.gather_tiles(local.data_ptr(), W, H, tiles.BLOCK_ROWS, full_accum.data_ptr() if rank == 0 else 0)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = e
        if not all_ranks_ok(err is None):
            gather_mode = "torch"
            gather_note = "agpt_gather_tiles was requested and failed on a rank (%s): fell back to torch.distributed.gather" % (err,)
            if comm is not None:
                try:
                    comm.close()
                except Exception:  # noqa: BLE001
                    pass
                comm = None
    if gather_mode == "torch":
        gather_bufs = tiles.alloc_gather_buffers(local, rank, world)   # receive buffers live outside the timed step
    # one agpt_render call per step and rank: the whole film at N = 1, this rank's interleaved 8-row blocks otherwise
    interleave = None if world == 1 else (tiles.BLOCK_ROWS, world, rank)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def gather():
        if gather_mode == "agpt":
            comm.gather_tiles(local.data_ptr(), W, H, tiles.BLOCK_ROWS, full_accum.data_ptr() if rank == 0 else 0)
            return full_accum
        return tiles.gather_to_rank0(local, W, H, rank, world, out_bufs=gather_bufs)

    def measure(desc, steps, warmup, spp_, do_gather, trace_all=False):
        """warmup + `steps` timed passes over one workload; returns (dt, sums, scene, last full accumulator)."""
        scene = desc.instantiate(ag.Scene(ctx))

        def step(timing):
            local.zero_()
            st = pt.render(scene, W, H, spp_, local.data_ptr(), accum_pitch=W, samples_per_batch=args.samples_per_batch,
                           timing=timing, interleave=interleave, trace_all_rays=trace_all)
            tot = dict(closest=st.closest_rays, anyhit=st.anyhit_rays, answered=st.answered_rays, trace_ms=st.trace_ms, ext_ms=st.ext_ms, mis_ms=st.mis_ms,
                       shadow_ms=st.shadow_ms, total_ms=st.total_ms, launches=st.trace_launches, iterations=st.iterations)
            full = gather() if do_gather else None
            return tot, full

        for _ in range(warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        acc = None
        sums = None
        for _ in range(steps):
            tot, full = step(True)
            if sums is None:
                sums = tot
            else:
                for k in tot:
                    sums[k] += tot[k]
            acc = full
        barrier()
        dt = time.perf_counter() - t0
        return dt, sums, scene, acc

    def rates(sums, dt, steps, samples_per_step):
        """The three rates of one measurement (this process' rays; the headline sums over ranks below)."""
        queries = float(sums["closest"] + sums["anyhit"])
        traversed = queries - float(sums["answered"])
        return {
            "value": round(traversed / dt / 1e6, 2), "unit": "Mrays/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2),
            "rays_traversed_per_step": int(traversed / steps),
            "reference_queries_per_step": int(queries / steps),
            "rays_answered_without_traversal_per_step": int(sums["answered"] / steps),
            "reference_queries_mrays_s": round(queries / dt / 1e6, 2),
            "msamples_per_s": round(samples_per_step * steps / dt / 1e6, 2),
        }

    if args.workload == "c5":
        desc = ag.scenes.scene_c5(scale=args.scene_scale, aspect=W / float(H))
        wl_name = "BASELINE configs[4] geometry: 5M-triangle multi-mesh colonnade field, all-metal GGX"
    else:
        desc = ag.scenes.scene_c3(scale=args.scene_scale, aspect=W / float(H))
        wl_name = "BASELINE configs[2]: Sponza-class closed atrium"
    wl_key = "%s:%dx%d:%dspp:world%d:scale%g" % (args.workload, W, H, spp, world, args.scene_scale)

    dt, sums, scene, acc = measure(desc, args.steps, args.warmup, spp, True)

    queries_local = float(sums["closest"] + sums["anyhit"])
    traversed_local = queries_local - float(sums["answered"])
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    r_all = torch.tensor([queries_local, float(sums["answered"])], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(r_all, op=dist.ReduceOp.SUM)
    dt_max = float(t_all.item())
    queries_total, answered_total = float(r_all[0].item()), float(r_all[1].item())
    traversed_total = queries_total - answered_total

    if rank == 0:
        # counter pass outside the timed region: the production kernels count the records they fetch on a 2-spp sample of
        # the same film (per-ray work is deterministic per scene + seed)
        cst = pt.render(scene, W, H, min(spp, 2), local.data_ptr(), accum_pitch=W, counters=2, interleave=interleave)
        img_mean = acc[..., :3].double().mean(dim=(0, 1)).div(spp).tolist() if acc is not None else None
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1080p/64spp", "value": round(traversed_total / dt_max / 1e6, 2),
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s, %d tris / %d primitives / %d Disney materials, %dx%d, %d spp%s, PathTracer MaxDepth=5, NEE+MIS"
                            % (wl_name, desc.n_tris, desc.n_prims, desc.n_materials, W, H, spp,
                               (" (%d spp x %d GPU shares)" % (args.spp, world)) if weak else
                               (" split over %d GPUs" % world if world > 1 else "")),
                "triangles": desc.n_tris, "resolution": [W, H], "spp": spp,
                "sharding": ("8-row blocks round-robin, one gather to rank 0 per step: %s"
                             % ("agpt_gather_tiles (C ABI: RCCL grouped send/recv over xGMI + k_deinterleave)" if gather_mode == "agpt"
                                else "torch.distributed.gather (%s) + torch de-interleave%s" % (backend, "; " + gather_note if gather_note else ""))) if world > 1 else "single GPU",
                # value counts the rays that entered a trace kernel (k_trace_fast launches: continuation / MIS query / shadow):
                # value x ms_per_step = rays_traversed_per_step.  The reference algorithm's Scene::Intersect / IntersectP calls
                # (equal to the CPU oracle's count) are reference_queries_per_step; the difference is settled exactly without a
                # traversal (MIS queries towards a sphere light whose ray misses the light's sphere, the ray after a path's last
                # bounce whose hit nothing reads: DESIGN.md 5.0) and earns no Mrays/s.
                "rays_traversed_per_step": int(traversed_total / args.steps),
                "rays_traversed_per_sample": round(traversed_total / args.steps / (W * H * spp), 2),
                "reference_queries_per_step": int(queries_total / args.steps),
                "reference_queries_per_sample": round(queries_total / args.steps / (W * H * spp), 2),
                "rays_answered_without_traversal_per_step": int(answered_total / args.steps),
                "reference_queries_mrays_s": round(queries_total / dt_max / 1e6, 2),
                "msamples_per_s": round(W * H * spp * args.steps / dt_max / 1e6, 2),
            },
            "roofline": roofline_block(sums, cst, traversed_local, args.steps, wl_key),
            "image_mean_rgb": [round(x, 6) for x in img_mean] if img_mean else None,
        }
    scene.close()

    if world == 1 and not args.no_secondary:
        # the same workload with every ray query of the reference sent through the BVH (agpt_render_params::trace_all_rays):
        # the image and the query totals are identical; the queries that cannot reach the image are traversed too
        dta, sumsa, scenea, _ = measure(desc, args.steps, args.warmup, spp, False, trace_all=True)
        out["all_rays_traversed"] = dict(rates(sumsa, dta, args.steps, W * H * spp), warmup=args.warmup,
            note="agpt_render_params::trace_all_rays = 1: every Scene::Intersect / IntersectP call of the reference goes through the "
                 "BVH (the headline run settles the ones whose result cannot reach the image without a traversal, DESIGN.md 5.0)")
        scenea.close()

    if world == 1 and args.workload == "c3" and not args.no_secondary:
        # BASELINE configs[4]'s scene on the same film: the configuration whose trace-time data exceeds the Infinity Cache
        desc5 = ag.scenes.scene_c5(aspect=W / float(H))
        dt5, sums5, scene5, _ = measure(desc5, args.steps, args.warmup, spp, False)
        cst5 = pt.render(scene5, W, H, min(spp, 2), local.data_ptr(), accum_pitch=W, counters=2)
        trav5 = float(sums5["closest"] + sums5["anyhit"] - sums5["answered"])
        out["secondary"] = dict(rates(sums5, dt5, args.steps, W * H * spp), warmup=args.warmup,
            workload="BASELINE configs[4] geometry: 5M-triangle multi-mesh scene, all-metal GGX + MIS, %d tris / %d primitives, "
                     "%dx%d, %d spp (one GPU's film; the config's 4K / 1024 spp budget is that of 8 GPUs)"
                     % (desc5.n_tris, desc5.n_prims, W, H, spp),
            roofline=roofline_block(sums5, cst5, trav5, args.steps, "c5:%dx%d:%dspp:world1:scale1" % (W, H, spp)))
        scene5.close()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, W, H, MAXD)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if comm is not None:
        comm.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
