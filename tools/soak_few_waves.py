#!/usr/bin/env python3
"""Soak of the queue kernels' bookkeeping under many work items per wave (RT_MI355X_TEST_RESIDENT_BLOCKS): random worlds,
the f32 and the fixed-point mixed mode, list scan and both forms of the hierarchy; the accumulators of a grid of 1, 2 or 5
workgroups must hold the sums of the full grid, where a wave of so small a frame takes one item at most (the case the
oracle tests pin).   usage: tools/soak_few_waves.py [worlds]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracing-in-one-weekend_amd")
worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")


def sums(acc, precision):
    a = acc.cpu().numpy()
    if precision == rt.PRECISION_MIXED_FIXED:
        return a[..., 0].astype(np.uint64).astype(object) * (1 << 32) + a[..., 1].astype(np.uint64).astype(object)
    return a


def render(world, cam, w, h, precision, fetch, seed):
    acc = torch.full((h, w, 3) + ((2,) if precision == rt.PRECISION_MIXED_FIXED else ()), -1, dtype=torch.int64, device=dev)
    stats = torch.zeros(rt.STATS_COUNT, dtype=torch.int64, device=dev)
    world.render_accumulate(cam, w, h, acc.data_ptr(), seed=seed, precision=precision, sphere_fetch=fetch, d_stats=stats.data_ptr())
    torch.cuda.synchronize()
    return sums(acc, precision), int(stats[rt.STAT_SEGMENTS]), int(stats[rt.STAT_SAMPLES])


for k in range(worlds):
    rng = np.random.default_rng(9000 + k)
    n = int(rng.choice([5, 40, 300, 900]))
    mat = np.zeros(6, dtype=rt.MATERIAL_DTYPE)
    mat["type"] = [0, 1, 2, 2, int(rng.choice([3, 4, 5, 6])), 0]
    mat["albedo"] = rng.uniform(0.1, 0.95, size=(6, 3))
    mat["param1"] = [0.0, rng.uniform(0.0, 1.0), 1.5, 1.0 / 1.5, rng.uniform(0.0, 0.4), 0.0]
    sph = np.zeros(n, dtype=rt.SPHERE_DTYPE)
    sph["center"] = rng.uniform(-6.0, 6.0, size=(n, 3)) * [1.0, 0.15, 1.0]
    sph["radius"] = 10.0 ** rng.uniform(-1.3, -0.2, size=n)
    sph["material_index"] = rng.integers(0, 6, size=n)
    sph["center"][0] = (0.0, -1000.0, 0.0)
    sph["radius"][0] = 999.0
    cs, _, _ = rt.build_final_scene(1, int(rng.choice([64, 96, 120])))
    cs.samples_per_pixel = int(rng.choice([8, 19, 40]))
    cs.max_ray_bounces = int(rng.choice([4, 50]))
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    os.environ.pop("RT_MI355X_TEST_RESIDENT_BLOCKS", None)
    full = rt.World(sph, mat)
    os.environ["RT_MI355X_TEST_RESIDENT_BLOCKS"] = str(int(rng.choice([1, 2, 5])))
    few = rt.World(sph, mat)
    for precision in (rt.PRECISION_F32, rt.PRECISION_MIXED_FIXED):
        want = render(full, cam, w, h, precision, rt.FETCH_AUTO, k)
        for fetch in (rt.FETCH_AUTO, rt.FETCH_BVH, rt.FETCH_BVH_GLOBAL):
            got = render(few, cam, w, h, precision, fetch, k)
            assert got[1:] == want[1:], ("segments / samples", k, precision, fetch, got[1:], want[1:])
            assert np.array_equal(got[0], want[0]), ("sums", k, precision, fetch)
    full.close(); few.close()
    if k % 10 == 9:
        print("worlds done:", k + 1, flush=True)
print("soak ok:", worlds, "worlds")
