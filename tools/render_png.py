#!/usr/bin/env python3
"""Render a scene on the GPU and write an 8-bit PNG (quantised exactly like serialize_ppm).
   usage: tools/render_png.py SCENE WIDTH SPP OUT.png"""
import importlib, struct, sys, zlib, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
scene, width, spp, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cs, sph, mat = {"dev": lambda: rt.build_dev_scene(width), "two": lambda: rt.build_two_sphere_scene(width), "final": lambda: rt.build_final_scene(1, width)}[scene]()
cs.samples_per_pixel = spp
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
img = np.zeros((h, w, 3))
rt.render_region(img, (0, 0, w, h), cam, sph, mat, seed=1, precision=rt.PRECISION_F32)
rgb = rt.quantize_rgb8(img)
raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
open(out, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))
print("wrote", out, w, h)
