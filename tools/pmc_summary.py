#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean of each counter over dispatches of kernels matching a filter."""
import collections
import csv
import glob
import sys

d, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "step_kernel")
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if filt in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} n={len(v):5d} mean={sum(v)/len(v):14.1f} min={min(v):12.1f} max={max(v):12.1f}")
