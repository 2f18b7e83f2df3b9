import importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
rt = importlib.import_module("raytracing-in-one-weekend_amd")
cs, sph, mat = rt.build_final_scene(1, 1200); cs.samples_per_pixel = 50
cam = rt.calculate_camera_render_data(cs); w, h = cs.image_width, cs.image_height
world = rt.World(sph, mat)
acc = torch.zeros(h, w, 3, dtype=torch.int64, device='cuda'); st = torch.zeros(8, dtype=torch.int64, device='cuda')
world.render_accumulate(cam, w, h, acc.data_ptr(), seed=1, d_stats=st.data_ptr(), precision=rt.PRECISION_F32); torch.cuda.synchronize()
s = st.cpu().numpy()
it = s[4]; print("segments", s[0], "wave iterations", it, "groups/iter", 488/4)
print("group entries per iteration %.1f (%.1f%% of groups)" % (s[5]/it, 100*s[5]/it/122))
print("wave-level resolve executions per iteration %.2f; parked candidates per lane-segment %.3f" % (s[6]/it, s[7]/s[0]))
