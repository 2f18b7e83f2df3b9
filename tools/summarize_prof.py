#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs tools/profile.sh wrote: kernel stats and per-kernel counter means."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("==", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.relpath(f, out))
        for k, cs in acc.items():
            if "render_kernel" not in k:
                continue
            print(" ", k, {c: sum(v) / len(v) for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
