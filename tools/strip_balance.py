#!/usr/bin/env python3
"""Load balance of the row partition: segments traced and kernel time per rank when the 8-row strips of the 3840x2160 final
scene are dealt round-robin over 2, 4 and 8 ranks (each rank's share rendered in turn on one GPU).   usage: tools/strip_balance.py"""
import importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch
rt = importlib.import_module("raytracing-in-one-weekend_amd")
cs, sph, mat = rt.build_final_scene(1, 3840)
cs.samples_per_pixel = 16
cam = rt.calculate_camera_render_data(cs)
w, h = cs.image_width, cs.image_height
world = rt.World(sph, mat)
for n in (2, 4, 8):
    segs, ms = [], []
    for r in range(n):
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        st = torch.zeros(8, dtype=torch.int64, device="cuda")
        world.render_accumulate(cam, w, h, acc.data_ptr(), seed=1, precision=rt.PRECISION_MIXED, d_stats=st.data_ptr(), strip_rows=8, strip_index=r, strip_count=n)
        torch.cuda.synchronize()
        segs.append(int(st[0])); ms.append(world.last_kernel_ns() / 1e6)
    print(f"N={n}: segments per rank max/mean {max(segs) / (sum(segs) / n):.4f}, kernel ms per rank max/mean {max(ms) / (sum(ms) / n):.4f}  ({', '.join('%.1f' % m for m in ms)})")
