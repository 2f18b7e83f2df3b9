#!/bin/bash
# A/B of library builds through the hierarchy (LDS and global form), the three queue modes.   usage: tools/bvh_ab.sh ROUNDS lib...
ROUNDS=$1; shift
for r in $(seq 1 "$ROUNDS"); do for f in bvh bvh-global; do for m in f32 mixed mixed-fixed; do for v in "$@"; do
  if [ "$v" = default ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$PWD/$v; fi
  echo -n "$f "; AB_FETCH=$f timeout -k 10 100 python tools/iter_cost.py $m 300 2>&1 | grep -v amdgpu.ids | cut -c1-100
done; done; done; done
