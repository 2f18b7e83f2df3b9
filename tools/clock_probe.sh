#!/bin/bash
# effective shader clock and VALU issue occupancy of render_kernel for each library given
#   usage: tools/clock_probe.sh lib1.so lib2.so ...   ("default" = in-tree)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  if [ "$v" = default ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$ROOT/$v; fi
  OUT=$ROOT/gpurun_out/clk/$(basename $v .so)
  mkdir -p "$OUT"
  timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT" -o run -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 0 --spp 500 $AB_FLAGS > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "$v failed"; tail -3 "$OUT/err.txt"; continue; }
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, json, sys, os
out, name = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
b = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
ms = b["roofline"]["kernel_ms"]
cyc = m["GRBM_GUI_ACTIVE"] / 8
print(f"{name}: kernel {ms:.2f} ms, {cyc:.4g} cycles -> {cyc / ms / 1e6:.3f} GHz; VALU insts {m['SQ_INSTS_VALU']:.4g} ({2 * m['SQ_INSTS_VALU'] / 1024 / cyc:.3f} issue occupancy), "
      f"MFMA insts {m.get('SQ_INSTS_MFMA', 0):.4g} (matrix pipe busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc:.3f}), SALU {m['SQ_INSTS_SALU']:.4g}, lane util {m['SQ_THREAD_CYCLES_VALU'] / 64 / m['SQ_INSTS_VALU']:.3f}, wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}, wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}, pct {b['pct_fp32_peak']:.2f}")
PY
done
