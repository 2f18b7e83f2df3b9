#!/usr/bin/env python3
"""Cost of one wave iteration of the list-scan kernel under different material mixes of the final scene
(all lambertian, all metal, all glass, ...): how much of the time the divergent material code is.
Measured: 16.0k SIMD cycles per iteration whatever the mix, 14.6k with glass only (no rejection sampling):
sorting rays by material would gain nothing.   usage: tools/phase_probe.py"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
cs, sph, mat = rt.build_final_scene(1, 1200)
cs.samples_per_pixel = 100
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
def run(label, mat):
    img = np.zeros((h, w, 3))
    best = None
    for _ in range(3):
        st = rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, want_stats=True, precision=rt.PRECISION_F32)
        ns, seg, iters = int(st[2]), int(st[0]), int(st[4])
        cyc = ns * 2.4 * 1024 / iters   # SIMD cycles per wave iteration at 2.4 GHz
        best = cyc if best is None else min(best, cyc)
    print(f"{label:28s} segments/sample {seg / (w * h * 100):.3f}  lane occupancy {seg / (64 * iters):.3f}  SIMD cycles per wave iteration {best:.0f}")
run("final scene", mat)
m = mat.copy(); m["type"][:] = 0
run("all lambertian", m)
m = mat.copy(); m["type"][m["type"] == 2] = 0
run("glass -> lambertian", m)
m = mat.copy(); m["type"][m["type"] == 1] = 0
run("metal -> lambertian", m)
m = mat.copy(); m["type"][:] = 1; m["param1"][:] = 0.3
run("all metal", m)
m = mat.copy(); m["type"][:] = 2; m["param1"][:] = 1.5
run("all glass", m)
