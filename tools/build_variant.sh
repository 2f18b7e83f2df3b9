#!/bin/bash
# Builds a variant of the library with extra compiler flags into build/<name>/librt_<name>.so
# (travels to the GPU box; select it with RT_MI355X_LIB).   usage: tools/build_variant.sh NAME -DFLAG ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC_ROOT:-$ROOT}/raytracing-in-one-weekend_amd/csrc   # SRC_ROOT: build another checkout of the sources
OUT=$ROOT/build/$NAME
mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $*"
DEV="--offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS $DEV -c "$SRC/rt_kernel.hip" -o "$OUT/rt_kernel.o" &
/opt/rocm/bin/hipcc $FLAGS $DEV -x hip -c "$SRC/rt_api.cpp" -o "$OUT/rt_api.o" &
g++ $FLAGS -c "$SRC/rt_host.cpp" -o "$OUT/rt_host.o" &
g++ $FLAGS -c "$SRC/rt_bvh.cpp" -o "$OUT/rt_bvh.o" &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/librt_$NAME.so" "$OUT"/rt_kernel.o "$OUT"/rt_api.o "$OUT"/rt_host.o "$OUT"/rt_bvh.o
echo "$OUT/librt_$NAME.so"
