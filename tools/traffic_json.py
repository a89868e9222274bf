#!/usr/bin/env python3
"""Builds profiles/traffic.json from rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md's
HBM section prescribes).  HBM-side bytes of the dominant kernel (k_trace_fast, closest + any-hit) per launch:
    read  = 32*RDREQ_32B + 128*RDREQ_128B + 64*(RDREQ - RDREQ_32B - RDREQ_128B)   (TCC_EA0 = L2's memory-side requests)
    write = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)
FETCH_SIZE/WRITE_SIZE (KB) are recorded next to them; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, which is why
the request-size breakdown is used for the byte figure.  Infinity-Cache hits are included in these counters (they are
L2-miss traffic, not DRAM traffic), so this is an upper bound on true HBM bytes."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
out = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(d + "/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, f)].add(r["Dispatch_Id"])
res = {}
tot_bytes = 0.0
tot_launch = 0
for k in sorted(agg):
    if not k.startswith("k_trace_fast"):
        continue
    c = agg[k]
    n = max(len(v) for (kk, f), v in calls.items() if kk == k)
    rd = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * (
        c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0) - c.get("TCC_EA0_RDREQ_128B_sum", 0))
    wr = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0))
    res[k] = {"launches": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
              "FETCH_SIZE_KB_per_launch": c.get("FETCH_SIZE", 0) / n, "WRITE_SIZE_KB_per_launch": c.get("WRITE_SIZE", 0) / n,
              "counters": {kk: vv for kk, vv in c.items()}}
    tot_bytes += rd + wr
    tot_launch += n
json.dump({"workload": "bench.py --steps 1 --warmup 0 (the full 1080p/64spp step, one 132.7 M-path batch)",
           "kernel": "k_trace_fast<closest> + k_trace_fast<any>", "hbm_bytes_per_launch": tot_bytes / max(1, tot_launch),
           "launches": tot_launch, "per_kernel": res}, open(out, "w"), indent=1)
print(json.dumps({"hbm_bytes_per_launch": tot_bytes / max(1, tot_launch), "launches": tot_launch}))
