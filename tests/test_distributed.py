"""The N>1 path on the CPU: 2, 3 and 4 gloo ranks run the partition / exchange logic of
raytracing-in-one-weekend_amd/multi_gpu.py with the oracle standing in for the GPU kernels, and
the gathered frame must equal the single-process frame bit for bit.  Image heights are ragged
(22 and 50 rows: last strips of 6 and 2 rows, ranks with unequal shares, a rank with no rows at all)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SPP = 5


def _fixed_words_to_mean(words, total_samples):
    """(.., 2) uint64 accumulator words of RT_PRECISION_MIXED_FIXED -> the sum a * 2^-32 + b * 2^-64 rounded to f64 once, times 1/spp"""
    from fractions import Fraction
    a = words[..., 0].astype(np.uint64).astype(object)
    b = words[..., 1].astype(np.uint64).astype(object)
    total = a * (1 << 32) + b
    return np.vectorize(lambda t: float(Fraction(t, 1 << 64)), otypes=[np.float64])(total) * (1.0 / float(total_samples))


def _worker(rank, world_size, partition, weak, width, port, out_path, mode="f32"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multi_gpu")
    import oracle_lib
    o = oracle_lib.Oracle()
    cs, sph, mat = rt.build_dev_scene(width)
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    plan = mg.Plan(partition, rank, world_size, SPP, weak=weak)
    accum = torch.zeros((h, w, 3, 2) if mode == "mixed-fixed" else (h, w, 3), dtype=torch.int64)
    image = torch.zeros(h, w, 3, dtype=torch.float64)
    gathered = torch.zeros_like(image)

    def trace(buf, strip_rows, strip_index, strip_count, sample_begin, sample_count):
        for v in range(h):
            if (v // strip_rows) % strip_count != strip_index or sample_count == 0:
                continue
            if mode == "mixed-fixed":
                words = o.render_fixed_words(w, h, (0, v, w, v + 1), cam, sph, mat, seed=4, sample_begin=sample_begin, sample_count=sample_count)
                buf[v] = torch.from_numpy(words[v].view(np.int64))
                continue
            r = o.render_region(w, h, (0, v, w, v + 1), cam, sph, mat, precision=rt.PRECISION_F32, seed=4, sample_begin=sample_begin, sample_count=sample_count)
            buf[v] = torch.from_numpy(r["accum"][v].astype(np.int64))

    def finish(buf, img, total_samples, strip_rows, strip_index, strip_count):
        if mode == "mixed-fixed":
            mean = _fixed_words_to_mean(buf.numpy().view(np.uint64), total_samples)
            for v in range(h):
                if (v // strip_rows) % strip_count == strip_index:
                    img[v] = torch.from_numpy(np.sqrt(mean[v]))
            return
        q = buf.numpy().astype(np.uint64)
        as_f64 = (q >> np.uint64(32)).astype(np.float64) * 4294967296.0 + (q & np.uint64(0xFFFFFFFF)).astype(np.float64)
        mean = (as_f64 * (1.0 / 4294967296.0)) * (1.0 / float(total_samples))
        for v in range(h):
            if (v // strip_rows) % strip_count == strip_index:
                img[v] = torch.from_numpy(np.sqrt(mean[v]))

    gather = mg.RowGather(plan, image) if partition == "rows" else None
    for _ in range(2):  # twice: the gather's buffers are reused from frame to frame
        image.fill_(-3.0)
        gathered.fill_(-5.0)
        frame = mg.render_step(plan, dist, accum, image, gathered, trace, finish, gather)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size,width", [(2, 40), (3, 90), (4, 40), (4, 90)])
@pytest.mark.parametrize("partition,weak", [("rows", False), ("samples", False), ("samples", True)])
def test_ranks_equal_one(rt, oracle, tmp_path, partition, weak, world_size, width):
    out = str(tmp_path / "frame.npy")
    port = 29600 + (os.getpid() * 7 + world_size * 31 + width + (0 if partition == "rows" else 97) + (53 if weak else 0)) % 1500
    mp.spawn(_worker, args=(world_size, partition, weak, width, port, out), nprocs=world_size, join=True)
    got = np.load(out)
    cs, sph, mat = rt.build_dev_scene(width)
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    assert h % 8 != 0
    total = SPP * world_size if weak else SPP
    ref = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_F32, seed=4, sample_count=total)
    assert np.array_equal(got, ref["image"])


@pytest.mark.parametrize("world_size,width", [(2, 40), (3, 40)])
@pytest.mark.parametrize("partition,weak", [("rows", False), ("samples", False), ("samples", True)])
def test_ranks_equal_one_with_fixed_point_f64_sums(rt, oracle, tmp_path, partition, weak, world_size, width):
    """RT_PRECISION_MIXED_FIXED: the Q32.64 words of the ranks' shares are added by the all-reduce (integers: exact,
    whatever the split), so the N-way frame is bitwise the one-way frame of all the samples"""
    out = str(tmp_path / "frame.npy")
    port = 31200 + (os.getpid() * 7 + world_size * 31 + width + (0 if partition == "rows" else 97) + (53 if weak else 0)) % 1500
    mp.spawn(_worker, args=(world_size, partition, weak, width, port, out, "mixed-fixed"), nprocs=world_size, join=True)
    got = np.load(out)
    cs, sph, mat = rt.build_dev_scene(width)
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    total = SPP * world_size if weak else SPP
    ref = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_MIXED_FIXED, seed=4, sample_count=total)
    assert np.array_equal(got, ref["image"])


def test_plan_partitions(rt):
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multi_gpu")
    for n in (1, 2, 3, 4, 8):
        rows = [mg.Plan("rows", r, n, 500).owned_rows(675) for r in range(n)]
        assert sorted(sum(rows, [])) == list(range(675))                      # every row exactly once
        assert max(len(r) for r in rows) - min(len(r) for r in rows) <= 8       # within one strip of each other
        assert mg.Plan("rows", 0, n, 500).max_owned_rows(675) == max(len(r) for r in rows)
        weak = [mg.Plan("samples", r, n, 500, weak=True) for r in range(n)]
        assert [p.sample_range for p in weak] == [(r * 500, 500) for r in range(n)] and weak[0].total_spp == 500 * n
        strong = [mg.Plan("samples", r, n, 500) for r in range(n)]
        covered = sum((list(range(b, b + c)) for b, c in (p.sample_range for p in strong)), [])
        assert covered == list(range(500)) and strong[0].total_spp == 500      # a fixed frame: the shares tile [0, 500)
        assert max(c for _, c in (p.sample_range for p in strong)) - min(c for _, c in (p.sample_range for p in strong)) <= 1
        assert mg.Plan("rows", 0, n, 500).total_spp == 500
    with pytest.raises(ValueError):
        mg.Plan("rows", 0, 2, 500, weak=True)
