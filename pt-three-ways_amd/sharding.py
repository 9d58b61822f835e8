"""How the DoD radiance path is split across the GPUs of a node (one process per GPU).

The reference's only decomposition is by PASS: one std::async task per full-frame pass, merged
with ArrayOutput::operator+= (src/dod/Scene.cpp:208-246), and offline by seed via raw_to_png.
Two shardings follow from that:

* SEQUENTIAL policy -> shard passes.  Pixels of a pass are serially dependent (one mt19937
  stream per pass), passes are independent.  Rank r renders passes [first_pass, first_pass+n)
  of the full frame; one reduce(sum) of the fp64 RGB sums and the uint32 counts merges them.
* PERPIXEL policy -> shard image rows (tiles).  Every (pass, pixel) sample has its own stream,
  so rank r renders all passes of rows [row_begin, row_end); the other rows of its framebuffer
  stay zero and the same reduce(sum) assembles the frame (a gather expressed as a sum of
  disjoint supports: one collective, no host re-interleave).

Both end in exactly one data-path collective over RCCL (backend "nccl" on ROCm) or gloo (CPU
tests): `reduce_framebuffer`.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def pass_shard(rank: int, world: int, total_passes: int) -> tuple[int, int]:
    """Strong-scaling split of `total_passes` passes: (first_pass, count) for this rank."""
    base, extra = divmod(total_passes, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def weak_pass_shard(rank: int, passes_per_gpu: int) -> tuple[int, int]:
    """Weak-scaling split: every rank renders `passes_per_gpu` passes with distinct seeds."""
    return rank * passes_per_gpu, passes_per_gpu


def row_shard(rank: int, world: int, height: int) -> tuple[int, int]:
    """Contiguous block of image rows [row_begin, row_end) for this rank."""
    base, extra = divmod(height, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_framebuffer(rgb_sum: torch.Tensor, counts: torch.Tensor, dst: int = 0) -> None:
    """output += pass for whole framebuffers: sums every rank's buffers into rank `dst`."""
    if not dist.is_initialized():
        return
    dist.reduce(rgb_sum, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(counts, dst=dst, op=dist.ReduceOp.SUM)
