"""One process per GPU (torch.distributed; the "nccl" backend is RCCL on ROCm).

The render path shards without any exchange of rays: pixels and samples are independent because
the RNG is keyed by (image pixel, global sample index).  Two partitions, as the reference's own
tile pool is a partition of pixels (main.odin:657-681):

  rows    : 8-row strips of the image dealt round-robin over the ranks (strip s -> rank s % N);
            each rank finalises its strips and rank 0 gathers the image (a reduce of buffers
            that are zero outside the owner's strips).
  samples : rank r traces global samples [r*spp, (r+1)*spp) of every pixel; the Q32.32
            accumulators are summed with one all-reduce (exact integers, so the result is
            bitwise the single-GPU image of N*spp samples) and rank 0 finalises.

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
    spp: int            # samples per pixel traced by ONE rank per step

    @property
    def total_spp(self) -> int:
        return self.spp * self.world_size if self.partition == "samples" else self.spp

    @property
    def sample_begin(self) -> int:
        return self.rank * self.spp if self.partition == "samples" else 0

    @property
    def strips(self):
        """(strip_rows, strip_index, strip_count) of this rank's pixels"""
        return (STRIP_ROWS, self.rank, self.world_size) if self.partition == "rows" else (STRIP_ROWS, 0, 1)

    def owned_rows(self, height: int):
        """image rows this rank renders"""
        rows, index, count = self.strips
        return [v for v in range(height) if (v // rows) % count == index]


def render_step(plan: Plan, dist, accum, image, gathered, trace: Callable, finish: Callable):
    """One frame.  accum: the rank's accumulators (int64 Q32.32), image: its finalised image
    (zero outside its strips), gathered: rank 0's output buffer.  Returns the buffer holding the
    frame on rank 0."""
    rows, index, count = plan.strips
    if plan.partition == "samples":
        trace(accum, strip_rows=rows, strip_index=0, strip_count=1, sample_begin=plan.sample_begin, sample_count=plan.spp)
        if plan.world_size > 1:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)
        if plan.rank == 0:
            finish(accum, image, total_samples=plan.total_spp, strip_rows=rows, strip_index=0, strip_count=1)
        return image
    trace(accum, strip_rows=rows, strip_index=index, strip_count=count, sample_begin=0, sample_count=plan.spp)
    finish(accum, image, total_samples=plan.spp, strip_rows=rows, strip_index=index, strip_count=count)
    if plan.world_size == 1:
        return image
    gathered.copy_(image)
    dist.reduce(gathered, dst=0, op=dist.ReduceOp.SUM)
    return gathered
