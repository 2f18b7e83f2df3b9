#!/bin/bash
# The bench lines a round's documentation quotes, each into gpurun_out/<prefix>_<name>.json (one GPU).
#   usage: tools/round_bench.sh PREFIX
P=${1:-rN}
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/${P}_$name.json 2> gpurun_out/${P}_$name.err || { echo "$name failed"; tail -5 gpurun_out/${P}_$name.err; exit 1; }; python - "$name" gpurun_out/${P}_$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
r = d["roofline"]
print(f"{sys.argv[1]:28s} {d['ms_per_step']:9.2f} ms  {d['value']:8.1f} Msamples/s  frac {r['frac']:.3f} ({r['bound']})  verified {d.get('verified')}")
for k, v in d.get("modes", {}).items():
    print(f"    {k:12s} {v['ms_per_step']:9.2f} ms  {v['pct_fp32_peak']:.1f} %  verified {v.get('verified')}")
PY
}
run default --steps 20 --warmup 2
run cfg4_mixed_fixed --config 4 --steps 3 --single-mode --no-cpu-baseline
run cfg4_mixed --config 4 --steps 3 --single-mode --no-cpu-baseline --precision mixed
run cfg4_f32 --config 4 --steps 3 --single-mode --no-cpu-baseline --precision f32
run cfg5_mixed_fixed --config 5 --steps 2 --single-mode
run cfg5_mixed --config 5 --steps 2 --single-mode --no-cpu-baseline --precision mixed
run cfg5_f32 --config 5 --steps 2 --single-mode --no-cpu-baseline --precision f32
run bvh_mixed_fixed --fetch bvh --steps 10 --no-cpu-baseline
run bvh_mixed --fetch bvh --steps 10 --no-cpu-baseline --precision mixed
run bvh_f32 --fetch bvh --steps 10 --no-cpu-baseline --precision f32
run bvh_cfg5_mixed_fixed --config 5 --fetch bvh --steps 5 --no-cpu-baseline
run bvh_cfg5_mixed --config 5 --fetch bvh --steps 5 --no-cpu-baseline --precision mixed
run bvh_cfg4_mixed --config 4 --fetch bvh --steps 3 --no-cpu-baseline --precision mixed
