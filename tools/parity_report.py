#!/usr/bin/env python3
"""f32 throughput mode against the f64 mode (the reference's arithmetic, bitwise equal to the CPU
restatement) on the GPU, same seed and draws: share of pixels within 1e-4 relative in linear
space and share of identical 8-bit pixels, per configuration.
   usage: tools/parity_report.py [scene width spp]..."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")

def report(scene, width, spp, seed=1):
    cs, sph, mat = {"dev": lambda: rt.build_dev_scene(width), "two": lambda: rt.build_two_sphere_scene(width),
                    "final": lambda: rt.build_final_scene(1, width)}[scene]()
    cs.samples_per_pixel = spp
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    out = {}
    for name, precision in (("f32", rt.PRECISION_F32), ("f64", rt.PRECISION_F64)):
        img, lin = np.zeros((h, w, 3)), np.zeros((h, w, 3))
        st = rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=seed, precision=precision, linear_out=lin, want_stats=True)
        out[name] = (img, lin, st)
    (i32, l32, s32), (i64, l64, s64) = out["f32"], out["f64"]
    rel = np.abs(l32 - l64) / np.maximum(l64, 1e-12)
    same8 = np.all(rt.quantize_rgb8(i32) == rt.quantize_rgb8(i64), axis=2)
    d8 = np.abs(rt.quantize_rgb8(i32).astype(int) - rt.quantize_rgb8(i64).astype(int)).max()
    print(f"{scene:5s} {w}x{h} {spp:4d} spp: within 1e-4 rel {100 * np.mean(rel.max(axis=2) <= 1e-4):6.2f}%   identical 8-bit {100 * same8.mean():6.2f}% (max 8-bit step {d8})"
          f"   segments/sample f32 {int(s32[0]) / (w * h * spp):.4f} f64 {int(s64[0]) / (w * h * spp):.4f}   mean f32/f64 {l32.mean() / l64.mean():.6f}"
          f"   kernel ms f32 {int(s32[rt.STAT_KERNEL_NS]) / 1e6:.1f} f64 {int(s64[rt.STAT_KERNEL_NS]) / 1e6:.1f}", flush=True)

if __name__ == "__main__":
    args = sys.argv[1:] or "two 400 100 dev 400 100 final 400 100 final 1200 16 final 1200 100".split()
    for i in range(0, len(args), 3):
        report(args[i], int(args[i + 1]), int(args[i + 2]))
