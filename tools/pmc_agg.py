#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches), plus the summed kernel
duration from the matching *_kernel_trace.csv when it sits next to the counter file."""
import collections
import csv
import os
import sys

for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(float)
    kt = f.replace("_counter_collection.csv", "_kernel_trace.csv")
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            dur[r["Kernel_Name"].split("(")[0][:48]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print("== %s" % f)
    for k, d in sorted(agg.items()):
        if not any(s in k for s in ("k_trace", "k_shade", "k_generate", "k_accumulate")):
            continue
        print("%s  dispatches=%d  duration_ms=%.3f" % (k, len(calls[k]), dur.get(k, 0.0) / 1e6))
        for c, v in sorted(d.items()):
            print("   %-38s %.5g" % (c, v))
