#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box for ONE arithmetic mode; results land in gpurun_out/prof_<mode>/.
#   usage: tools/profile.sh f32|mixed|mixed-fixed|f64 [extra bench.py flags]
set -u
export TMPDIR=/tmp
MODE=${1:-mixed}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$MODE
mkdir -p "$OUT"
cd "$ROOT"
BENCH="bench.py --no-cpu-baseline --single-mode --no-verify --precision $MODE $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $BENCH --steps 3 --warmup 1 > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || { echo "stats pass failed"; tail -5 "$OUT/stats.err"; exit 1; }
SPP=100
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o run -- python3 $BENCH --steps 1 --warmup 0 --spp $SPP > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name failed"; tail -5 "$OUT/$name.err"; exit 1; }
}
pass pmc1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass pmc2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32
pass pmc3 SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE
pass pmc4 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS
# memory-side traffic of the benchmark's own dispatch (500 spp), one pass per counter
SPP=500
pass traffic_fetch FETCH_SIZE
pass traffic_write WRITE_SIZE
python3 tools/summarize_prof.py "$OUT" "$MODE" | tee "$OUT/summary.txt"
