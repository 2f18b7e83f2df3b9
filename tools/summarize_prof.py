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
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
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
traffic = {}
for name in ("traffic_fetch", "traffic_write"):
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(f)) if "render_kernel" in row["Kernel_Name"]]
        if vals:
            # per FRAME: the pass runs bench.py --steps 1 --warmup 0, which renders two frames (one counted, one timed); a frame
            # of the ordered mixed mode is several dispatches (one per staging window)
            traffic[name] = sum(vals) / 2.0
            traffic[name + "_dispatches_per_frame"] = len(vals) / 2.0
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
        if g("SQ_ACTIVE_INST_VALU"):
            print(f"   VALUBusy               = 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / kernel cycles = {100 * 4 * g('SQ_ACTIVE_INST_VALU') / 1024 / cyc:.1f} %")
        if g("SQ_INSTS_VALU_FMA_F32"):
            # fused multiply-adds are the packed ones of the scan (4 cycles per wave), the rest issue in 2
            pk, rest = g("SQ_INSTS_VALU_FMA_F32"), g("SQ_INSTS_VALU") - g("SQ_INSTS_VALU_FMA_F32")
            print(f"   VALU issue cycles      = (4 * FMA_F32 + 2 * other VALU) / 1024 SIMDs / kernel cycles = {(4 * pk + 2 * rest) / 1024 / cyc:.3f}")
            print(f"   instruction mix        : FMA_F32 {pk / g('SQ_INSTS_VALU'):.3f} of VALU instructions; SALU per VALU {g('SQ_INSTS_SALU', 0) / g('SQ_INSTS_VALU'):.3f}")
            f64 = sum(g(k, 0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"))
            if f64:
                print(f"   f64 instructions       : {f64 / g('SQ_INSTS_VALU'):.3f} of VALU instructions (FMA {g('SQ_INSTS_VALU_FMA_F64', 0):.4g}, ADD {g('SQ_INSTS_VALU_ADD_F64', 0):.4g}, MUL {g('SQ_INSTS_VALU_MUL_F64', 0):.4g}, TRANS {g('SQ_INSTS_VALU_TRANS_F64', 0):.4g})")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
        print(f"   wave time              : issue-stalled {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}, parked on waitcnt {g('SQ_WAIT_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}, executing {g('SQ_ACTIVE_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}")
if traffic:
    f, w = traffic.get("traffic_fetch", float("nan")), traffic.get("traffic_write", float("nan"))
    print(f"== memory-side traffic of the render kernel over one 500-spp frame (4.05e8 samples, {traffic.get('traffic_write_dispatches_per_frame', 1):.0f} dispatch(es); rocprofv3 reports KiB; FETCH_SIZE doubled as gfx950 tallies 128-B reads at 64 B)")
    what = {"f32": "8-byte integer atomics, three per pixel and work item of 4 samples (tile sums in LDS), width as counted",
            "mixed-fixed": "8-byte integer atomics, six per pixel and work item of 8 samples (tile sums in LDS), width as counted",
            "mixed": "24-byte stores of the staged sample colours"}.get(mode, "")
    print(f"   FETCH_SIZE {f:.6g} KiB -> {2 * f * 1024 / 1e9:.4f} GB read;  WRITE_SIZE {w:.6g} KiB -> {w * 1024 / 1e9:.3f} GB written ({what})")
    allc["TRAFFIC_BYTES_500SPP"] = (2 * f + w) * 1024
json.dump(allc, open(os.path.join(out, "counters.json"), "w"), indent=1)
