#!/usr/bin/env python3
"""Builds profiles/pmc_<tag>_l1.json from the passes of tools/gpu_pmc_l1.sh: the vector L1 / texture-address counters of one
bench.py step per kernel family, normalised per CU-cycle (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs).
    python tools/pmc_l1_json.py <dir with *_counter_collection.csv> <workload key> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

d, workload, out_path = sys.argv[1], sys.argv[2], sys.argv[3]


def fam(k):
    k = k.replace("void ", "")
    if "k_trace_fast<" in k:
        a = [x.strip() for x in k.split("k_trace_fast<", 1)[1].split(">", 1)[0].split(",")]
        if a[3] == "true":   # the counter pass outside the timed region
            return None
        return {"0": "trace_closest_hit", "1": "trace_any_hit", "2": "trace_mis_query"}[a[0]]
    return "shade" if k.startswith("k_shade") else None


agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for r in csv.DictReader(open(os.path.join(d, "grbm_kernel_trace.csv"))):
    k = fam(r["Kernel_Name"])
    if k:
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {"kernel_build": bench.kernel_build_id(), "workload": workload,
       "how": "tools/gpu_pmc_l1.sh: rocprofv3 --pmc <two counters per pass> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 "
              "--no-cpu-baseline --no-secondary; sums over all dispatches of the step; CU-cycles = GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs",
       "kernels": {}}
tot_acc = tot_cyc = 0.0
for k in sorted(agg):
    c = agg[k]
    cu = c["GRBM_GUI_ACTIVE"] / 8 * 256
    out["kernels"][k] = {
        "duration_ms": round(dur[k] / 1e6, 2), "clock_ghz": round(c["GRBM_GUI_ACTIVE"] / 8 / (dur[k] / 1e9) / 1e9, 3), "cu_cycles": cu,
        "l1_cache_accesses_per_cu_cycle": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / cu, 4),
        "ta_busy": round(c["TA_TA_BUSY_sum"] / cu, 4), "ta_addr_stalled_by_tcp": round(c["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / cu, 4),
        "tcp_tag_conflict_stall": round(c["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"] / cu, 4),
        "tcp_pending_stall": round(c["TCP_PENDING_STALL_CYCLES_sum"] / cu, 4),
        "l1_miss_requests_per_cu_cycle": round(c["TCP_TCC_READ_REQ_sum"] / cu, 4),
        "l2_read_latency_cycles": round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / max(c["TCP_TCC_READ_REQ_sum"], 1), 1),
        "l1_accesses_per_wave_instruction": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(c["TA_FLAT_READ_WAVEFRONTS_sum"], 1), 2),
        "counters": dict(sorted(c.items()))}
    if k.startswith("trace"):
        tot_acc += c["TCP_TOTAL_CACHE_ACCESSES_sum"]
        tot_cyc += cu
out["trace_l1_cache_accesses_per_cu_cycle"] = round(tot_acc / tot_cyc, 4)
json.dump(out, open(out_path, "w"), indent=1)
print(out["trace_l1_cache_accesses_per_cu_cycle"], {k: v["l1_cache_accesses_per_cu_cycle"] for k, v in out["kernels"].items()})
