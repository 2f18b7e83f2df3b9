#!/bin/bash
# A/B timing of library builds on ONE GPU box: each library in turn, ROUNDS times, interleaved (boxes differ by a
# percent or two, so variants are only comparable within one call).
#   usage: [AB_MODE=f32|mixed|f64] [AB_SPP=n] tools/ab.sh ROUNDS lib1.so lib2.so ...   ("default" = the in-tree library)
ROUNDS=$1; shift
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$PWD/$v; fi
    timeout -k 10 100 python tools/iter_cost.py ${AB_MODE:-mixed} ${AB_SPP:-300} 2>&1 | grep -v amdgpu.ids | cut -c1-150
  done
done
