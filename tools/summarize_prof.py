#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs tools/profile.sh wrote: kernel stats, per-kernel counter means and
the figures derived from them (VALU lane utilisation, VALU issue occupancy, instruction mix)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel-trace --stats:", os.path.relpath(f, out), "(bench.py --steps 3 --warmup 1, 500 spp)")
    for row in csv.DictReader(open(f)):
        if "rtk::" in row["Name"]:
            print("  ", {k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
allc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "render_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in acc.items():
            allc[c] = sum(v) / len(v)
print("== PMC means per render_kernel dispatch (bench.py --steps 1 --warmup 0 --spp 100: 8.1e7 samples per dispatch)")
for c in sorted(allc):
    print(f"   {c:28s} {allc[c]:.6g}")
g = allc.get
if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    print("== derived")
    print(f"   VALU lane utilisation  = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU) = {g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_INSTS_VALU')):.3f}")
    if g("GRBM_GUI_ACTIVE"):
        cyc = g("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
        print(f"   kernel cycles          = GRBM_GUI_ACTIVE / 8 = {cyc:.4g}")
        print(f"   VALU issue occupancy   = 2 cycles * SQ_INSTS_VALU / 1024 SIMDs / kernel cycles = {2 * g('SQ_INSTS_VALU') / 1024 / cyc:.3f}")
    if g("SQ_INSTS_VALU_FMA_F32"):
        tests = g("SQ_INSTS_VALU_FMA_F32") / 6.0
        print(f"   wave-level sphere tests = SQ_INSTS_VALU_FMA_F32 / 6 = {tests:.4g}; VALU instructions per test = {g('SQ_INSTS_VALU') / tests:.2f}; SALU per test = {g('SQ_INSTS_SALU', 0) / tests:.2f}")
    if g("FETCH_SIZE") is not None or g("WRITE_SIZE") is not None:
        print(f"   FETCH_SIZE = {g('FETCH_SIZE', float('nan')):.6g} KB, WRITE_SIZE = {g('WRITE_SIZE', float('nan')):.6g} KB per dispatch (rocprofv3 units: KB)")
json.dump(allc, open(os.path.join(out, "counters.json"), "w"), indent=1)
