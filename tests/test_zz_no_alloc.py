"""rt_render_accumulate_static -- the entry point that allocates nothing -- against rt_render_accumulate.

Its scratch (queue head, chunk slabs, staged sample colours) is a fixed per-GPU array instead of buffers the
world allocates; the results must be identical, call after call, also when a smaller call follows a larger one
and finds the larger one's leftovers in the array.  "Allocates nothing" is checked at the device: the free
memory hipMemGetInfo reports must not move across the call on a world that has never rendered before (so nothing
it needs is there yet), while the allocating entry point, in the same situation, visibly takes memory.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DTYPES = {0: "int64", 1: "float64", 2: "float32", 3: "float64", 4: "int64"}


def _scene(rt, width, spp):
    cs, sph, mat = rt.build_final_scene(1, width)
    cs.samples_per_pixel = spp
    return cs, rt.calculate_camera_render_data(cs), sph, mat


def _accum(torch, precision, h, w, fill):
    shape = (h, w, 3, 2) if precision == 4 else (h, w, 3)
    return torch.full(shape, fill, dtype=getattr(torch, DTYPES[precision]), device="cuda:0")


@pytest.mark.parametrize("precision,chunks", [(0, 0), (1, 3), (2, 4), (3, 0), (4, 0)])
def test_static_scratch_entry_point_matches_the_allocating_one(rt, precision, chunks):
    import torch
    # two calls in a row through the same array: the second smaller in pixels, samples and spheres
    for width, spp, n_spheres in ((256, 12, None), (96, 5, 40)):
        cs, cam, sph, mat = _scene(rt, width, spp)
        if n_spheres is not None:
            sph = sph[:n_spheres]
        w, h = cs.image_width, cs.image_height
        world = rt.World(sph, mat)
        want = _accum(torch, precision, h, w, -1)
        got = _accum(torch, precision, h, w, -1)
        stats_a = torch.zeros(rt.STATS_COUNT, dtype=torch.int64, device="cuda:0")
        stats_b = torch.zeros(rt.STATS_COUNT, dtype=torch.int64, device="cuda:0")
        world.render_accumulate(cam, w, h, want.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, d_stats=stats_a.data_ptr())
        world.render_accumulate(cam, w, h, got.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, d_stats=stats_b.data_ptr(),
                                static_scratch=True)
        torch.cuda.synchronize()
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


@pytest.mark.parametrize("precision,chunks", [(1, 3), (3, 0), (0, 0), (4, 0)])
def test_static_entry_point_takes_no_device_memory(rt, precision, chunks):
    """free device memory (hipMemGetInfo) across the FIRST render call of a world: unchanged through the entry point that
    allocates nothing; lower through the allocating one where it has a buffer to make (chunk slabs, staged colours)"""
    import torch
    cs, cam, sph, mat = _scene(rt, 640, 12)
    w, h = cs.image_width, cs.image_height
    out = _accum(torch, precision, h, w, 0)
    ref = _accum(torch, precision, h, w, 0)
    # the kernels of this precision have run once in this process (code loaded, whatever the runtime sets up at a
    # kernel's first launch done), on a world that is then closed
    warm = rt.World(sph, mat)
    warm.render_accumulate(cam, w, h, ref.data_ptr(), seed=7, precision=precision, sample_chunks=chunks)
    warm.render_accumulate(cam, w, h, out.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, static_scratch=True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    warm.close()
    out.zero_()

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    fresh = rt.World(sph, mat)                 # its first render call is the static one
    before = free_bytes()
    fresh.render_accumulate(cam, w, h, out.data_ptr(), seed=7, precision=precision, sample_chunks=chunks, static_scratch=True)
    after = free_bytes()
    assert after == before, (precision, before - after)
    assert torch.equal(out, ref)
    fresh.close()
    control = rt.World(sph, mat)               # positive control: the same first call through the allocating entry point
    before = free_bytes()
    control.render_accumulate(cam, w, h, out.data_ptr(), seed=7, precision=precision, sample_chunks=chunks)
    after = free_bytes()
    needs_buffer = precision in (1, 3)         # chunk slabs (11 MB) / staged colours (66 MB); the fixed-point modes need none
    assert (before - after >= (8 << 20)) if needs_buffer else (after == before), (precision, before - after)
    control.close()


def test_static_entry_point_under_graph_capture(rt):
    """the caller it exists for: a stream being captured into a HIP graph.  The call records no events and does not touch
    the host; the graph replays to the same accumulators as the allocating entry point's, twice in a row"""
    import torch
    cs, cam, sph, mat = _scene(rt, 160, 6)
    w, h = cs.image_width, cs.image_height
    world = rt.World(sph, mat)
    for precision in (4, 3, 0):
        want = _accum(torch, precision, h, w, -1)
        got = _accum(torch, precision, h, w, -1)
        world.render_accumulate(cam, w, h, want.data_ptr(), seed=9, precision=precision)
        world.render_accumulate(cam, w, h, got.data_ptr(), seed=9, precision=precision, static_scratch=True)   # eager once: nothing is set up lazily inside the capture
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            world.render_accumulate(cam, w, h, got.data_ptr(), seed=9, precision=precision, static_scratch=True,
                                    stream=torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            got.fill_(-1)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(got, want), precision
        # an eager call after the capture still works and still times itself
        got.fill_(-1)
        world.render_accumulate(cam, w, h, got.data_ptr(), seed=9, precision=precision, static_scratch=True)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and world.last_kernel_ns() > 0
        del graph
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
    # a frame whose single staged sample exceeds the array: refused in the ordered mixed mode; the fixed-point mixed
    # mode needs the queue head only and is never refused for size
    cs, cam, sph, mat = _scene(rt, 3840, 1)
    w, h = cs.image_width, cs.image_height
    world = rt.World(sph, mat)
    big = torch.zeros(h, w, 3, 2, dtype=torch.int64, device=dev)
    with pytest.raises(rt.RtError) as e:
        world.render_accumulate(cam, w, h, big.data_ptr(), precision=rt.PRECISION_MIXED, static_scratch=True)
    assert e.value.code == rt.RT_ERR_UNSUPPORTED
    world.render_accumulate(cam, w, h, big.data_ptr(), precision=rt.PRECISION_MIXED_FIXED, static_scratch=True)
    torch.cuda.synchronize()
    assert int((big != 0).sum()) > 0
    world.close()
