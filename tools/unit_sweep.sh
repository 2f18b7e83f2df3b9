#!/bin/bash
# Kernel time of the headline frame's queue modes against the unit size (samples a lane traces per unit of work).
#   usage: tools/unit_sweep.sh [spp] [unit sizes...]
SPP=${1:-300}; shift
for u in ${@:-2 4 8 16}; do
  for m in f32 mixed mixed-fixed; do
    echo -n "unit $u: "
    RT_MI355X_UNIT_SAMPLES=$u timeout -k 10 100 python tools/iter_cost.py $m $SPP 2>&1 | grep -v amdgpu.ids | cut -c1-150
  done
done
