"""rt_render_accumulate_static -- the entry point that allocates nothing -- against rt_render_accumulate.

Its scratch (queue head, chunk slabs, staged sample colours) is a fixed per-GPU array instead of buffers the
world allocates; the results must be identical, call after call, also when a smaller call follows a larger one
and finds the larger one's leftovers in the array.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rt, width, spp):
    cs, sph, mat = rt.build_final_scene(1, width)
    cs.samples_per_pixel = spp
    return cs, rt.calculate_camera_render_data(cs), sph, mat


@pytest.mark.parametrize("precision,chunks", [(0, 0), (1, 3), (2, 4), (3, 0)])
def test_static_scratch_entry_point_matches_the_allocating_one(rt, precision, chunks):
    import torch
    dev = torch.device("cuda:0")
    dtype = {0: torch.int64, 1: torch.float64, 2: torch.float32, 3: torch.float64}[precision]
    # two calls in a row through the same array: the second smaller in pixels, samples and spheres
    for width, spp, n_spheres in ((256, 12, None), (96, 5, 40)):
        cs, cam, sph, mat = _scene(rt, width, spp)
        if n_spheres is not None:
            sph = sph[:n_spheres]
        w, h = cs.image_width, cs.image_height
        world = rt.World(sph, mat)
        want = torch.full((h, w, 3), -1, dtype=dtype, device=dev)
        got = torch.full((h, w, 3), -1, dtype=dtype, device=dev)
        stats_a = torch.zeros(rt.STATS_COUNT, dtype=torch.int64, device=dev)
        stats_b = torch.zeros(rt.STATS_COUNT, dtype=torch.int64, device=dev)
        world.render_accumulate(cam, w, h, want.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, d_stats=stats_a.data_ptr())
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        world.render_accumulate(cam, w, h, got.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, d_stats=stats_b.data_ptr(),
                                static_scratch=True)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.equal(got, want), (precision, width)
        assert int(stats_a[rt.STAT_SEGMENTS]) == int(stats_b[rt.STAT_SEGMENTS]) > 0
        # a region and strips, as the allocating entry point takes them
        region = (5, 3, w - 7, h - 2)
        want.fill_(-1)
        got.fill_(-1)
        for entry, buf in ((False, want), (True, got)):
            world.render_accumulate(cam, w, h, buf.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, region=region,
                                    strip_rows=8, strip_index=1, strip_count=2, static_scratch=entry)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        world.close()


def test_static_scratch_limits_are_reported(rt):
    import torch
    dev = torch.device("cuda:0")
    cap = rt.lib().rt_static_scratch_bytes()
    assert 16 << 20 <= cap <= 64 << 20
    cs, cam, sph, mat = _scene(rt, 1200, 8)
    w, h = cs.image_width, cs.image_height
    world = rt.World(sph, mat)
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device=dev)
    # f64 with 4 chunks needs 3 slabs of 19.4 MB: refused, with the sizes in the message
    with pytest.raises(rt.RtError) as e:
        world.render_accumulate(cam, w, h, acc.data_ptr(), precision=rt.PRECISION_F64, sample_chunks=4, static_scratch=True)
    assert e.value.code == rt.RT_ERR_UNSUPPORTED and "scratch" in e.value.message
    # one chunk needs no slabs; the mixed mode stages one sample at a time (19.5 MB per sample fits); f32 needs the head only
    want = torch.zeros_like(acc)
    world.render_accumulate(cam, w, h, want.data_ptr(), seed=3, precision=rt.PRECISION_F64, sample_chunks=1)
    world.render_accumulate(cam, w, h, acc.data_ptr(), seed=3, precision=rt.PRECISION_F64, sample_chunks=1, static_scratch=True)
    torch.cuda.synchronize()
    assert torch.equal(acc, want)
    acc.zero_()
    world.render_accumulate(cam, w, h, acc.data_ptr(), seed=3, precision=rt.PRECISION_MIXED, static_scratch=True)
    torch.cuda.synchronize()
    assert torch.equal(acc, want)
    # the hierarchy must have been built by the allocating entry point
    q = torch.zeros(h, w, 3, dtype=torch.int64, device=dev)
    with pytest.raises(rt.RtError):
        world.render_accumulate(cam, w, h, q.data_ptr(), sphere_fetch=rt.FETCH_BVH, static_scratch=True, precision=rt.PRECISION_F32)
    world.close()
    # a frame whose single staged sample exceeds the array: refused
    cs, cam, sph, mat = _scene(rt, 3840, 1)
    w, h = cs.image_width, cs.image_height
    world = rt.World(sph, mat)
    big = torch.zeros(h, w, 3, dtype=torch.float64, device=dev)
    with pytest.raises(rt.RtError) as e:
        world.render_accumulate(cam, w, h, big.data_ptr(), precision=rt.PRECISION_MIXED, static_scratch=True)
    assert e.value.code == rt.RT_ERR_UNSUPPORTED
    world.close()
