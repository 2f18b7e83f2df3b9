#!/bin/bash
# A/B timing of library builds on ONE GPU box: each library in turn, ROUNDS times, interleaved.
#   usage: tools/ab.sh ROUNDS lib1.so lib2.so ...   ("default" = the in-tree library)
ROUNDS=$1; shift
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$PWD/$v; fi
    timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $AB_FLAGS | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['pct_fp32_peak'] or 100 * d['roofline']['frac'],2), round(d['roofline']['kernel_ms'],2), round(d['lane_utilization'],3))"
  done
done
