#!/usr/bin/env python3
"""Builds profiles/pmc_<tag>.json from rocprofv3 --pmc passes of one bench.py workload (tools/gpu_pmc.sh).

    python tools/pmc_json.py <dir with *_counter_collection.csv + the bench lines of the passes> <workload key> <out.json>

Per kernel family (trace = k_trace_fast<*>, shade = k_shade): counter sums over all dispatches of the profiled step,
and the per-ray figures bench.py scales by the rays of its timed region:
    valu_insts_per_traversed_ray   SQ_INSTS_VALU / rays            (wave-instructions)
    hbm_bytes_per_traversed_ray    (read + write bytes leaving the L2s) / rays, with
        read  = 32*RDREQ_32B + 128*RDREQ_128B + 64*(RDREQ - RDREQ_32B - RDREQ_128B)   (TCC_EA0 = the L2s' memory-side requests)
        write = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)
      each counter group collected in its own --pmc pass with --kernel-trace only, as MI355X_MICROARCH.md's HBM / rocprofv3
      section prescribes; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, hence the request-size breakdown.
      Infinity-Cache hits are included (memory-side of L2, not DRAM), so this is an upper bound on true HBM bytes.
    wait_fraction        SQ_WAIT_ANY / SQ_WAVE_CYCLES            (both in quad-cycles, summed over waves)
    lane_utilisation     SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
    issue_fraction_profiled  SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * kernel duration * clock) in the profiled run itself
rays = the rays that entered a trace kernel in the profiled step (config.rays_traversed_per_step of the bench line of the same
pass; the reference's ray queries that are answered without a traversal are not in it).  kernel_build = bench.kernel_build_id().
"""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

d, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
def fam(k):
    """kernel family of a dispatch; the COUNT instantiations of the counter pass (k_trace_fast<MODE, DEPTH, CHUNKED, COUNT, SPILL>
    with COUNT = true) are not the timed kernels and are left out"""
    if "k_trace_fast<" in k:
        args = [a.strip() for a in k.split("k_trace_fast<", 1)[1].split(">", 1)[0].split(",")]
        return None if args[3] == "true" else "trace"
    return "shade" if k.startswith("k_shade") else None

agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
launches = collections.defaultdict(int)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"].replace("void ", ""))
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
    for k, v in seen.items():
        launches[k] = max(launches[k], len(v))
    kt = f.replace("_counter_collection.csv", "_kernel_trace.csv")
    if os.path.exists(kt) and "SQ_INSTS_VALU" in open(f).read(200000):
        for r in csv.DictReader(open(kt)):
            k = fam(r["Kernel_Name"].replace("void ", ""))
            if k:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rays = None
for f in sorted(glob.glob(os.path.join(d, "*.log"))):
    try:
        line = json.loads(open(f).read().strip().splitlines()[-1])
        rays = float(line["config"]["rays_traversed_per_step"])
    except Exception:  # noqa: BLE001
        continue
assert rays, "no bench line found next to the counter files"
res = {"kernel_build": bench.kernel_build_id(), "workload": workload, "rays_traversed_profiled": rays,
       "how": "tools/gpu_pmc.sh: rocprofv3 --pmc <one group per pass> --kernel-trace -- python3 bench.py --steps 1 --warmup 0"}
for k, c in agg.items():
    rd = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * (
        c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0) - c.get("TCC_EA0_RDREQ_128B_sum", 0))
    wr = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0))
    e = {"launches": launches[k], "counters": dict(sorted(c.items()))}
    if c.get("SQ_INSTS_VALU"):
        e["valu_insts_per_traversed_ray"] = c["SQ_INSTS_VALU"] / rays
        e["salu_insts_per_traversed_ray"] = c.get("SQ_INSTS_SALU", 0) / rays
        if c.get("SQ_WAVE_CYCLES"):
            e["wait_fraction"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
        if c.get("SQ_ACTIVE_INST_VALU"):
            e["lane_utilisation"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * c["SQ_ACTIVE_INST_VALU"]), 4)
        if dur.get(k):
            e["profiled_duration_ms"] = dur[k] / 1e6
            e["issue_fraction_profiled"] = round(c["SQ_INSTS_VALU"] * 2 / (bench.SIMDS * dur[k] * bench.CLOCK_GHZ), 4)
    if rd or wr:
        e["hbm_read_bytes"] = rd
        e["hbm_write_bytes"] = wr
        e["hbm_bytes_per_traversed_ray"] = (rd + wr) / rays
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in res.items() if isinstance(v, dict)}))
