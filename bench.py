#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one full render of the workload: every pixel, every sample, through the C ABI
(rt_render_accumulate + rt_finalize), with the world and camera already resident in HBM.
Workload at N=1: the book's final random-spheres scene, 1200x675, 500 spp, depth 50.
N>1 (one process per GPU, launched by torch.distributed.run): sample split -- rank r traces
global sample indices [r*500, (r+1)*500) of every pixel, the Q32.32 accumulators are summed with
an RCCL all-reduce over xGMI and rank 0 finalises a 500*N-spp image (weak scaling: per-GPU
work is fixed).  `--partition rows` instead deals 8-row strips of ONE 500-spp frame over the
ranks and gathers them on rank 0 (strong scaling).

Prints one JSON line on rank 0 with the driver's fields plus `roofline` (FP32 VALU) and
`cpu_baseline` (the f64 CPU restatement of the reference timed on this host's cores).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_TEST = 17                 # algorithmic FLOP per ray/sphere test (3 sub, 5+5 dot products, -r^2, h*h-a*c)
PEAK_FP32_TFLOPS = 157.3    # MI355X vector FP32 peak: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
GATHER_PEAK_G = 917.8       # --fetch bvh: divergent 16-byte gathers per ns from an L2-resident table, measured by tools/gather_peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--scene", default="final", choices=["final", "dev", "stress"])
    ap.add_argument("--partition", default="samples", choices=["samples", "rows"])
    ap.add_argument("--fetch", default="auto", choices=["auto", "scalar", "lds", "bvh", "bvh-global"],
                    help="bvh: closest hit through the bounding-volume hierarchy (not the headline: the roofline then counts the tests carried out)")
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--verify", action="store_true", help="after timing, rank 0 re-renders the frame in one rt_render_region call and compares bitwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def measured_traffic(args):
    """HBM-side bytes of one render_kernel launch of the default workload, from the committed
    rocprofv3 FETCH_SIZE / WRITE_SIZE passes over this same command (tools/profile.sh writes
    TRAFFIC_BYTES_500SPP into the counter summary); None for any other workload."""
    if (args.scene, args.width, args.spp) != ("final", 1200, 500):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "round1_pmc_counters.json")) as f:
            return float(json.load(f)["TRAFFIC_BYTES_500SPP"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(rt, cam, sph, mat, width, height, seconds):
    """The oracle (CPU restatement of the reference, f64, its literal operation order) on a
    bounded sample of the same workload: the full frame at the spp that fits the time budget."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle()
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    t0 = time.perf_counter()
    o.render_region(width, height, (0, 0, width, height), cam, sph, mat, precision=rt.PRECISION_F64, seed=1, sample_count=1, threads=cores)
    est_full_1spp = max(time.perf_counter() - t0, 1e-4)  # the whole frame at 1 spp
    spp = max(1, min(512, int(seconds / est_full_1spp)))
    t0 = time.perf_counter()
    r = o.render_region(width, height, (0, 0, width, height), cam, sph, mat, precision=rt.PRECISION_F64, seed=1, sample_count=spp, threads=cores)
    dt = time.perf_counter() - t0
    samples = width * height * spp
    # one thread, the reference's default (RENDER_MULTITHREADED = false, main.odin:14): every 2nd row of one 1-spp frame
    rows = list(range(0, height, 2))
    t0 = time.perf_counter()
    for v in rows:
        o.render_region(width, height, (0, v, width, v + 1), cam, sph, mat, precision=rt.PRECISION_F64, seed=1, sample_count=1, threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"full {width}x{height} frame at {spp} spp ({samples} samples, {dt:.1f} s), f64, {cores} threads over rows",
            "segments_per_sample": float(r["stats"][0]) / samples,
            "single_thread": {"value": width * len(rows) / dt1 / 1e6, "unit": "Msamples/s", "cores": 1,
                              "sample": f"every 2nd row of the frame at 1 spp ({width * len(rows)} samples, {dt1:.1f} s)"}}


class HostStaged:
    """torch.distributed look-alike whose collectives stage device tensors through the host
    (rehearsal mode only: gloo between ranks that share one GPU)."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def all_reduce(self, t, op):
        h = t.cpu()
        self._d.all_reduce(h, op=op)
        t.copy_(h)

    def reduce(self, t, dst, op):
        h = t.cpu()
        self._d.reduce(h, dst=dst, op=op)
        t.copy_(h)

    def barrier(self):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    multi_gpu = importlib.import_module("raytracing-in-one-weekend_amd.multi_gpu")

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 as: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        # Rehearsal of the N>1 flow on a box with ONE GPU: every rank uses device 0 and the
        # collectives run over gloo on host copies.  The numbers mean nothing; the frame must.
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        if rehearse:
            dist.init_process_group("gloo")
            dist = HostStaged(dist)
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.scene == "final":
        cs, sph, mat = rt.build_final_scene(1, args.width)
    elif args.scene == "dev":
        cs, sph, mat = rt.build_dev_scene(args.width)
    else:
        cs, sph, mat = rt.build_stress_scene(5, 10000, args.width)
    cs.samples_per_pixel = args.spp
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    fetch = {"auto": rt.FETCH_AUTO, "scalar": rt.FETCH_SCALAR, "lds": rt.FETCH_LDS, "bvh": rt.FETCH_BVH, "bvh-global": rt.FETCH_BVH_GLOBAL}[args.fetch]

    world = rt.World(sph, mat, device=local_rank)
    accum = torch.zeros(h, w, 3, dtype=torch.int64, device=dev)  # Q32.32 sums: exact, so the all-reduce is order-independent
    image = torch.zeros(h, w, 3, dtype=torch.float64, device=dev)
    gathered = torch.zeros_like(image)
    stats = torch.zeros(rt.STATS_COUNT, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    by_samples = args.partition == "samples"

    plan = multi_gpu.Plan(args.partition, rank, world_size, args.spp)
    collect = {"stats": False}

    def trace(buf, **kw):
        world.render_accumulate(cam, w, h, buf.data_ptr(), seed=1, sphere_fetch=fetch, sample_chunks=args.chunks,
                                d_stats=stats.data_ptr() if collect["stats"] else 0, stream=stream, **kw)

    def finish(buf, img, total_samples, **kw):
        world.finalize(buf.data_ptr(), w, h, total_samples, img.data_ptr(), stream=stream, **kw)

    def step(collect_stats=False):
        collect["stats"] = collect_stats
        return multi_gpu.render_step(plan, dist, accum, image, gathered, trace, finish)

    def fence():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one counted pass: exact segment count of this rank's share of a step
    step(collect_stats=True)
    torch.cuda.synchronize()
    segments = int(stats[rt.STAT_SEGMENTS].item())
    samples_rank = int(stats[rt.STAT_SAMPLES].item())
    wave_iters = int(stats[rt.STAT_WAVE_ITERATIONS].item())
    bvh_nodes, bvh_tests = int(stats[rt.STAT_BVH_NODES].item()), int(stats[rt.STAT_BVH_TESTS].item())
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    knl = []
    for _ in range(args.steps):
        step()
        # device time of this step's render kernel: HIP events recorded around it on the launch stream
        knl.append(world.last_kernel_ns())
    fence()
    elapsed = time.perf_counter() - t0
    kernel_s = sum(knl) / len(knl) * 1e-9

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    seg_t = torch.tensor([segments, samples_rank], dtype=torch.int64, device=dev)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seg_t, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_spp = plan.total_spp
    samples_step = w * h * total_spp
    value = samples_step * args.steps / elapsed / 1e6

    if rank == 0:
        # list scan: every segment tests every sphere; hierarchy: the tests the traversal carried out (counted by the kernel)
        tests_rank = bvh_tests if args.fetch.startswith("bvh") else segments * len(sph)
        achieved = F_TEST * tests_rank / kernel_s / 1e12
        out = {
            "metric": "Msamples/sec (whole node), final scene 1200x675 500spp; % FP32 VALU peak",
            "value": value, "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if by_samples else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} scene ({len(sph)} spheres, seed 1) {w}x{h}, {args.spp} spp per GPU, depth {cs.max_ray_bounces}",
                       "partition": ("sample split + RCCL all-reduce" if by_samples else "8-row strips + RCCL reduce to rank 0") if world_size > 1 else "single GPU",
                       "total_spp": total_spp, "sphere_fetch": args.fetch, "rng": "counter, deterministic, seed 1"},
            "pct_fp32_peak": 100.0 * achieved / PEAK_FP32_TFLOPS,
            "segments_per_sample": segments / max(1, samples_rank),
            "lane_utilization": segments / max(1, 64 * wave_iters),
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": measured_traffic(args) if not args.fetch.startswith("bvh") else None,
                         "kernel": "render_kernel<float, fma, scalar fetch>", "kernel_ms": kernel_s * 1e3,
                         "flop_per_test": F_TEST, "sphere_tests_per_launch": tests_rank},
        }
        if args.fetch.startswith("bvh"):
            # The traversal is bound by the vector L1's rate of serving per-lane 16-byte gathers that fall into different cache
            # lines: one per box tested, two per sphere tested.  Peak: the L2-resident rate tools/gather_peak measures on this chip.
            gathers = bvh_nodes + 2 * bvh_tests
            out["bvh"] = {"boxes_per_segment": bvh_nodes / max(1, segments), "sphere_tests_per_segment": bvh_tests / max(1, segments)}
            out["roofline"] = {"bound": "vl1d_gather", "achieved": gathers / kernel_s / 1e9, "peak": GATHER_PEAK_G, "unit": "Ggather/s",
                               "frac": gathers / kernel_s / 1e9 / GATHER_PEAK_G, "traffic": None, "kernel": "render_kernel<float, fma, bvh>",
                               "kernel_ms": kernel_s * 1e3, "gathers_per_launch": gathers, "sphere_tests_per_launch": bvh_tests}
            out["pct_fp32_peak"] = None
        if args.verify:
            # the gathered frame against one single-GPU rt_render_region call over all total_spp samples
            import numpy as np
            frame = (image if (by_samples or world_size == 1) else gathered).cpu().numpy()
            cs.samples_per_pixel = total_spp
            whole = np.zeros((h, w, 3))
            rt.render_region(whole, (0, 0, w, h), rt.calculate_camera_render_data(cs), sph, mat, seed=1, device=local_rank)
            out["verified"] = bool(np.array_equal(frame, whole))
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(rt, cam, sph, mat, w, h, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    world.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
