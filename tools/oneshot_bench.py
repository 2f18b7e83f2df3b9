#!/usr/bin/env python3
"""Wall time of the host-buffer entry point called in a loop, as the reference's benchmark does
(bench_materials, main_test.odin:8-46: 100 renders of the dev scene at 200 px, 100 spp), plus single calls at
the headline size.  The library is the one RT_MI355X_LIB selects.   usage: tools/oneshot_bench.py [rounds]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
name = os.path.basename(os.environ.get("RT_MI355X_LIB", "default"))
cs, sph, mat = rt.build_dev_scene(200)
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
img = np.zeros((h, w, 3))
rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, precision=rt.PRECISION_F32)   # the first call loads the code object
MODES = [(rt.PRECISION_F32, "f32"), (getattr(rt, "PRECISION_MIXED", None), "mixed"), (getattr(rt, "PRECISION_MIXED_FIXED", None), "mixed-fixed")]
for precision, label in MODES:
    if precision is None:
        continue
    rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, precision=precision)
    t0 = time.perf_counter()
    for _ in range(rounds):
        rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, precision=precision)
    dt = time.perf_counter() - t0
    st = rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, precision=precision, want_stats=True)
    print(f"{name} dev {w}x{h} {cs.samples_per_pixel} spp {label}: {rounds} rounds in {dt * 1e3:.1f} ms, {rounds / dt:.1f} rounds/s, {dt / rounds * 1e3:.3f} ms per call (kernels {int(st[2]) / 1e6:.3f} ms)")
cs, sph, mat = rt.build_final_scene(1, 1200)
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
img = np.zeros((h, w, 3))
for precision, label in MODES:
    if precision is None:
        continue
    for k in range(3):
        t0 = time.perf_counter()
        st = rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, want_stats=True, precision=precision)
        dt = time.perf_counter() - t0
        print(f"{name} final {w}x{h} 500 spp {label} call {k}: {dt * 1e3:.1f} ms wall, kernels {int(st[2]) / 1e6:.1f} ms")
t0 = time.perf_counter()
text = rt.render_ppm(w, h, cam, sph, mat, seed=1, precision=rt.PRECISION_F32)
print(f"{name} final render_ppm: {(time.perf_counter() - t0) * 1e3:.1f} ms wall, {len(text)} bytes")
