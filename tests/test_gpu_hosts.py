"""Hosts other than this test process, through the C ABI on the GPU:

* tools/rt_cli -- a C++ `main` that does what the reference's `main` does (main.odin:826-852): build the dev scene,
  compute the camera, render, print a P3 PPM to stdout -- run as a child process; its stdout must be, byte for byte,
  serialize_ppm of the f64 oracle's image, and its `--rounds` mode must print the statistics line of the reference's
  benchmark (main_test.odin:72-81).
* bench.py at N = 2 in rehearsal mode (two ranks on the one GPU, collectives over gloo on host copies), launched as the
  driver launches the multi-GPU runs: the JSON line must carry the per-rank diagnostics and a verified frame.

Children are fresh processes started with subprocess; nothing here replaces a process that has touched the GPU."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "rt_cli")


def run_cli(*args):
    assert os.path.exists(CLI), "tools/rt_cli is built by __graft_entry__.build() (make -C tools)"
    return subprocess.run([CLI, *args], capture_output=True, timeout=120)


def test_cli_main_prints_the_reference_image(rt, oracle):
    """`main` (main.odin:826-852) with the library's defaults: dev scene at its native 200 pixels, 100 spp, seed 1"""
    out = run_cli("--scene", "dev", "--width", "200", "--precision", "mixed")
    assert out.returncode == 0, out.stderr.decode()
    cs, sph, mat = rt.build_dev_scene(200)
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    assert (w, h, cs.samples_per_pixel) == (200, 112, 100)
    ref = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_F64, seed=1, threads=8)
    assert out.stdout == oracle.serialize_ppm(ref["image"])
    assert out.stderr.decode().endswith("\rDone.                   \n")        # main.odin:590
    # no --precision at all: the library's default is that same image; the host-side writer gives the same bytes
    assert run_cli("--scene", "dev", "--width", "200").stdout == out.stdout
    assert run_cli("--scene", "dev", "--width", "200", "--host-ppm").stdout == out.stdout
    # the fixed-point sums: the same 8-bit image
    assert run_cli("--scene", "dev", "--width", "200", "--precision", "mixed-fixed").stdout == out.stdout


def test_cli_rounds_is_the_reference_benchmark(rt, oracle):
    """bench_materials (main_test.odin:8-46): N rounds, the last image to stdout, the statistics line of benchmark_print"""
    out = run_cli("--scene", "dev", "--width", "200", "--rounds", "3")
    assert out.returncode == 0, out.stderr.decode()
    err = out.stderr.decode()
    assert "\rRound 3/3" in err and "\rDone.        \n" in err                # main_test.odin:26, :33
    m = re.search(r"\[materials\] 3 rounds, (\d+) bytes processed in (\d+) ns, +(\d+\.\d{3}) rounds/s, +(\d+\.\d{3}) MiB/s\n", err)
    assert m, err
    assert int(m.group(1)) == 3 * 200 * 112                                    # options.bytes = len(image.data): pixels (main_test.odin:24)
    assert abs(float(m.group(3)) - 3 / (int(m.group(2)) * 1e-9)) < 1e-2 * float(m.group(3)) + 1e-3
    cs, sph, mat = rt.build_dev_scene(200)
    cam = rt.calculate_camera_render_data(cs)
    ref = oracle.render_region(200, 112, (0, 0, 200, 112), cam, sph, mat, precision=rt.PRECISION_F64, seed=1, threads=8)
    assert out.stdout == oracle.serialize_ppm(ref["image"])


@pytest.mark.parametrize("partition", ["rows", "samples"])
def test_bench_two_rank_rehearsal(partition):
    env = dict(os.environ, RT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    port = str(29700 + os.getpid() % 200 + (0 if partition == "rows" else 201))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320", "--spp", "16", "--cpu-seconds", "1",
           "--partition", partition]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["verified"] is True
    assert d["config"]["precision"] == "mixed-fixed" and "320x180" in d["config"]["workload"] and d["config"]["total_spp"] == 16
    r = d["ranks"]
    assert r["world_size_reported"] == 2 and r["backend"] == "gloo"
    for key in ("render_kernel_ms", "render_part_ms", "collective_part_ms", "segments", "samples", "step_ms"):
        assert len(r[key]) == 2 and all(v > 0 for v in r[key]), key
    assert sum(r["samples"]) == 320 * 180 * 16
    if partition == "samples":
        assert r["samples"][0] == r["samples"][1]
    assert -1e-9 <= r["imbalance_kernel_ms"] < 1.0 and -1e-9 <= r["imbalance_segments"] < 1.0
    assert d["value"] > 0 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["roofline"]["frac"] > 0 and d["roofline"]["traffic_model"] > 0
