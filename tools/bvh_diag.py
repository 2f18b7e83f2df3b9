#!/usr/bin/env python3
"""Traversal statistics of RT_FETCH_BVH from a library built with -DRT_DIAG_BVH (stats[7] = per-wave
traversal-loop trips): boxes per segment (lane mean) against trips per wave iteration (slowest lane).
   usage: RT_MI355X_LIB=build/ab/bvh_diag.so tools/bvh_diag.py [scene width spp]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
scene, width, spp = (sys.argv[1:] + ["final", "1200", "50"])[:3]
width, spp = int(width), int(spp)
cs, sph, mat = {"final": lambda: rt.build_final_scene(1, width), "stress": lambda: rt.build_stress_scene(5, 10000, width), "dev": lambda: rt.build_dev_scene(width)}[scene]()
cs.samples_per_pixel = spp
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
img = np.zeros((h, w, 3))
st = rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, sphere_fetch=rt.FETCH_BVH, want_stats=True, precision=rt.PRECISION_F32)
seg, iters, nodes, tests, trips = (int(st[i]) for i in (0, 4, 5, 6, 7))
print(f"{scene}: segments {seg}, wave iterations {iters}, boxes/segment {nodes / seg:.2f}, sphere tests/segment {tests / seg:.2f}, "
      f"traversal trips per wave iteration {trips / iters:.1f} (slowest lane) vs lane mean {nodes / seg:.1f}: lane efficiency {nodes / (64 * trips):.3f}, kernel {int(st[2]) / 1e6:.1f} ms")
