#!/usr/bin/env python3
"""Copy the summaries tools/profile.sh left under gpurun_out/prof_<mode>/ into profiles/ (tracked), named per round.
   usage: tools/collect_profiles.py ROUND [mode ...]      e.g. tools/collect_profiles.py 2 f32 mixed"""
import json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # kernel_source_hash: the same hash bench.py compares against
rnd = sys.argv[1]
modes = sys.argv[2:] or ["mixed-fixed", "mixed", "f32"]
# the sources these passes were taken from, and how: bench.py reports `stale` when the kernels have changed since
merged = {"kernel_source_sha16": bench.kernel_source_hash(),
          "command": "tools/profile.sh <mode>: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --single-mode --no-verify --precision <mode> --steps 3 --warmup 1; "
                     "rocprofv3 --pmc <counters> -- python3 bench.py ... --steps 1 --warmup 0 --spp 100 (SQ passes) / --spp 500 (FETCH_SIZE, WRITE_SIZE, one pass each)",
          "TRAFFIC_BYTES_500SPP": {}}
for mode in modes:
    src = os.path.join(root, "gpurun_out", f"prof_{mode}")
    shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(root, "profiles", f"round{rnd}_kernel_stats_{mode}.csv"))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(root, "profiles", f"round{rnd}_summary_{mode}.txt"))
    shutil.copy(os.path.join(src, "stats_bench.json"), os.path.join(root, "profiles", f"round{rnd}_bench_under_rocprof_{mode}.json"))
    c = json.load(open(os.path.join(src, "counters.json")))
    merged["TRAFFIC_BYTES_500SPP"][mode] = c.pop("TRAFFIC_BYTES_500SPP", None)
    merged[mode] = c
json.dump(merged, open(os.path.join(root, "profiles", f"round{rnd}_pmc_counters.json"), "w"), indent=1)
print("collected", modes)
