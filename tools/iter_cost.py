#!/usr/bin/env python3
"""Cost of one wave iteration (one ray segment per lane: closest-hit scan + shading) of the list-scan kernel,
in SIMD cycles at 2.4 GHz, for the library RT_MI355X_LIB selects.  With a diagnostic build (RT_DIAG_COUNTERS) also
how often the scan leaves its straight-line code.   usage: tools/iter_cost.py [f32|mixed|mixed-fixed|f64] [spp]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("raytracing-in-one-weekend_amd")
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fetch = {"": rt.FETCH_AUTO, "bvh": rt.FETCH_BVH, "bvh-global": rt.FETCH_BVH_GLOBAL}[os.environ.get("AB_FETCH", "")]  # AB_FETCH=bvh: through the hierarchy
precision = {"f32": rt.PRECISION_F32, "mixed": rt.PRECISION_MIXED, "mixed-fixed": rt.PRECISION_MIXED_FIXED, "f64": rt.PRECISION_F64}[mode]
cs, sph, mat = rt.build_final_scene(1, 1200)
cs.samples_per_pixel = spp
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
world = rt.World(sph, mat)
acc = torch.zeros((h, w, 3, 2) if mode == "mixed-fixed" else (h, w, 3), dtype=torch.int64 if mode in ("f32", "mixed-fixed") else torch.float64, device="cuda")
st = torch.zeros(8, dtype=torch.int64, device="cuda")
best = None
for k in range(4):
    st.zero_()
    world.render_accumulate(cam, w, h, acc.data_ptr(), seed=1, precision=precision, d_stats=st.data_ptr(), sphere_fetch=fetch)
    torch.cuda.synchronize()
    ns = world.last_kernel_ns()
    best = ns if best is None else min(best, ns)
s = st.cpu().numpy()
seg, iters = int(s[0]), int(s[4])
print(f"{os.path.basename(os.environ.get('RT_MI355X_LIB', 'default'))} {mode}: kernel {best / 1e6:.2f} ms, segments/sample {seg / (w * h * spp):.3f}, lane occupancy {seg / max(1, 64 * iters):.3f}, "
      f"SIMD cycles per wave iteration {best * 2.4 * 1024 / max(1, iters):.0f}; counters 5..7 per iteration: {s[5] / max(1, iters):.2f} {s[6] / max(1, iters):.2f} {s[7] / max(1, iters):.2f}")
if os.environ.get("RT_MI355X_LIB", "").endswith("phases.so"):
    # RT_DIAG_PHASES build: wave cycles (s_memtime ticks, 100 MHz) per phase, summed over waves
    ph = {"deal + camera ray": s[5], "scan (fast path)": s[6], "scan slow path (park + resolve)": s[7], "shading of hits": s[1], "background + end of sample": s[2]}
    tot = float(sum(ph.values()))
    print("   wave time by phase: " + ", ".join(f"{k} {v / tot:.3f}" for k, v in ph.items()))
