#!/usr/bin/env python3
"""How the slow path of the scan splits between the entries (the per-sphere checks and queueing when a group of four has a
candidate) and the evaluation of the wave queue: needs a library built with -DRT_DIAG_PHASES -DRT_DIAG_FLUSH (tools/build_variant.sh),
selected with RT_MI355X_LIB.  Wave cycles per wave iteration, final scene 1200x675, 100 spp."""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
rt = importlib.import_module("raytracing-in-one-weekend_amd")
for mode in ("mixed", "f32"):
    precision = {"f32": rt.PRECISION_F32, "mixed": rt.PRECISION_MIXED}[mode]
    cs, sph, mat = rt.build_final_scene(1, 1200)
    cs.samples_per_pixel = 100
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    world = rt.World(sph, mat)
    acc = torch.zeros(h, w, 3, dtype=torch.int64 if mode == "f32" else torch.float64, device="cuda")
    st = torch.zeros(8, dtype=torch.int64, device="cuda")
    for k in range(2):
        st.zero_()
        world.render_accumulate(cam, w, h, acc.data_ptr(), seed=1, precision=precision, d_stats=st.data_ptr())
        torch.cuda.synchronize()
    s = st.cpu().numpy()
    tot = float(s[5] + s[6] + s[7] + s[2])  # without the shading time, which this build replaces by the flush time
    print(mode, "iters", s[4], "fresh", s[5] / s[4], "scan fast", s[6] / s[4], "slow", s[7] / s[4], "of which flush", s[1] / s[4], "end", s[2] / s[4], "(ticks of s_memtime per wave iteration)")
