"""One process per GPU (torch.distributed; the "nccl" backend is RCCL on ROCm).

The render path shards without any exchange of rays: pixels and samples are independent because
the RNG is keyed by (image pixel, global sample index).  Two partitions, as the reference's own
tile pool is a partition of pixels into disjoint rects (main.odin:657-681):

  rows    : 8-row strips of the image dealt round-robin over the ranks (strip s -> rank s % N);
            each rank finalises its strips, packs them into a strip-major buffer of its own rows
            and rank 0 gathers exactly those rows (height/N rows per rank, no reduction).
  samples : rank r traces its share of the global sample indices of every pixel; the exact
            accumulators (fixed-point integers: Q32.32 of the f32 mode, Q32.64 of the fixed-point mixed
            mode) are summed with one all-reduce, so the result is bitwise the single-GPU image of
            all the samples, and rank 0 finalises.

Scaling: by default the frame is FIXED (strong scaling: N ranks share one frame of `spp`
samples per pixel); `weak=True` with the sample partition gives every rank `spp` samples of
its own, i.e. a frame of N*spp samples per pixel.

`trace` and `finish` are injected so that the partition logic can be exercised on the CPU with
the gloo backend; on a GPU they are World.render_accumulate / World.finalize.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

STRIP_ROWS = 8


@dataclass
class Plan:
    partition: str      # "rows" | "samples"
    rank: int
    world_size: int
    spp: int            # samples per pixel: of the frame (strong scaling) or per rank (weak)
    weak: bool = False  # samples partition only

    def __post_init__(self):
        if self.partition not in ("rows", "samples"):
            raise ValueError(self.partition)
        if self.weak and self.partition != "samples":
            raise ValueError("weak scaling is the sample partition with per-rank spp")

    @property
    def total_spp(self) -> int:
        return self.spp * self.world_size if self.weak else self.spp

    @property
    def sample_range(self):
        """(first global sample index, count) this rank traces"""
        if self.partition == "rows":
            return 0, self.spp
        if self.weak:
            return self.rank * self.spp, self.spp
        lo = self.rank * self.spp // self.world_size
        hi = (self.rank + 1) * self.spp // self.world_size
        return lo, hi - lo

    @property
    def sample_begin(self) -> int:
        return self.sample_range[0]

    @property
    def strips(self):
        """(strip_rows, strip_index, strip_count) of this rank's pixels"""
        return (STRIP_ROWS, self.rank, self.world_size) if self.partition == "rows" else (STRIP_ROWS, 0, 1)

    def owned_rows(self, height: int, rank: int | None = None):
        """image rows rank `rank` (default: this one) renders"""
        if self.partition != "rows":
            return list(range(height))
        index = self.rank if rank is None else rank
        return [v for v in range(height) if (v // STRIP_ROWS) % self.world_size == index]

    def max_owned_rows(self, height: int) -> int:
        return max(len(self.owned_rows(height, r)) for r in range(self.world_size))


class RowGather:
    """Buffers and index tensors of the row partition's gather, made once per (plan, image shape).

    Every rank packs its own rows of the finalised image, in image order, into `send` (padded to
    the largest share so all ranks send the same size); rank 0 receives one such buffer per rank
    and scatters the rows to their places.  Bytes on the wire per rank: its share of the image."""

    def __init__(self, plan: Plan, image):
        import torch
        h = image.shape[0]
        self.plan = plan
        self.rows = [torch.tensor(plan.owned_rows(h, r), dtype=torch.long, device=image.device) for r in range(plan.world_size)]
        n_max = plan.max_owned_rows(h)
        self.send = torch.zeros((n_max,) + tuple(image.shape[1:]), dtype=image.dtype, device=image.device)
        self.recv = [torch.zeros_like(self.send) for _ in range(plan.world_size)] if plan.rank == 0 else None

    def run(self, dist, image, gathered):
        import torch
        mine = self.rows[self.plan.rank]
        if len(mine):
            torch.index_select(image, 0, mine, out=self.send[:len(mine)])
        dist.gather(self.send, gather_list=self.recv, dst=0)
        if self.plan.rank == 0:
            for r, rows in enumerate(self.rows):
                if len(rows):
                    gathered.index_copy_(0, rows, self.recv[r][:len(rows)])
        return gathered


class StepTimers:
    """Device-side times of the two parts of a step on this rank's stream: the render part (trace, and for the row
    partition the finalise) and the collective part (all-reduce + finalise, or gather + scatter of rows).  Events are
    recorded on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self):
        self.marks = []

    def mark(self):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def add(self, begin, middle, end):
        self.marks.append((begin, middle, end))

    def means_ms(self):
        """(render part, collective part) in ms, means over the recorded steps; synchronises"""
        import torch
        torch.cuda.synchronize()
        if not self.marks:
            return 0.0, 0.0
        render = sum(b.elapsed_time(m) for b, m, _ in self.marks) / len(self.marks)
        coll = sum(m.elapsed_time(e) for _, m, e in self.marks) / len(self.marks)
        return render, coll


def render_step(plan: Plan, dist, accum, image, gathered, trace: Callable, finish: Callable, gather: RowGather | None = None,
                timers: StepTimers | None = None):
    """One frame.  accum: the rank's accumulators, image: its finalised image (only its strips are
    written), gathered: rank 0's output buffer.  Returns the buffer holding the frame on rank 0."""
    rows, index, count = plan.strips
    begin, n = plan.sample_range
    t0 = timers.mark() if timers else None
    if plan.partition == "samples":
        trace(accum, strip_rows=rows, strip_index=0, strip_count=1, sample_begin=begin, sample_count=n)
        t1 = timers.mark() if timers else None
        if plan.world_size > 1:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)
        if plan.rank == 0:
            finish(accum, image, total_samples=plan.total_spp, strip_rows=rows, strip_index=0, strip_count=1)
        if timers:
            timers.add(t0, t1, timers.mark())
        return image
    trace(accum, strip_rows=rows, strip_index=index, strip_count=count, sample_begin=begin, sample_count=n)
    finish(accum, image, total_samples=plan.spp, strip_rows=rows, strip_index=index, strip_count=count)
    if plan.world_size == 1:
        return image
    t1 = timers.mark() if timers else None
    if gather is None:
        gather = RowGather(plan, image)
    out = gather.run(dist, image, gathered)
    if timers:
        timers.add(t0, t1, timers.mark())
    return out
