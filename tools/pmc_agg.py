#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches)."""
import collections
import csv
import sys

f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for k, d in sorted(agg.items()):
    if not any(s in k for s in ("k_trace", "k_shade", "k_generate", "k_accumulate")):
        continue
    print("%s  dispatches=%d" % (k, len(calls[k])))
    for c, v in sorted(d.items()):
        print("   %-34s %.5g" % (c, v))
