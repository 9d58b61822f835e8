"""How the DoD radiance path is split across the GPUs of a node (one process per GPU).

The reference's only decomposition is by PASS: one std::async task per full-frame pass, merged
with ArrayOutput::operator+= (src/dod/Scene.cpp:208-246), and offline by seed via raw_to_png
(src/main/raw_to_png.cpp:39-58).  Two shardings follow from that:

* SEQUENTIAL policy -> shard passes.  Pixels of a pass are serially dependent (one mt19937
  stream per pass), passes are independent.  Rank r renders passes [first_pass, first_pass+n)
  of the full frame; one reduce(sum) of the fp64 RGB sums and the uint32 counts merges them.
* PERPIXEL policy -> shard image rows, INTERLEAVED (row y belongs to rank y % world): what a
  pixel costs depends on what it sees, so contiguous blocks would be unbalanced.  Every
  (pass, pixel) sample has its own stream; rank r renders all passes of its rows and one gather
  of the rows (1/world of the frame per rank) assembles the frame on the root.

Both end in exactly one data-path collective.  On GPUs it is the library's own
(`ptw_comm_reduce_framebuffer` / `ptw_comm_gather_rows`: RCCL behind the C ABI, see
csrc/capi_comm.hip) - `FrameComm` below only carries the 128-byte RCCL id from rank 0 to the other
ranks through torch.distributed.  The torch.distributed forms (`reduce_framebuffer`,
`gather_rows`) are the same collectives for CPU tensors under gloo: they exist so that the
sharding arithmetic can be tested without GPUs.
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
    """Contiguous block of image rows [row_begin, row_end) for this rank ((k, k) when empty)."""
    base, extra = divmod(height, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def interleaved_rows(rank: int, world: int) -> dict:
    """ptw_render_params fields of the interleaved-row shard: rows y with y % world == rank."""
    return {"row_stride": world, "row_phase": rank} if world > 1 else {}


def rows_owned(rank: int, world: int, height: int) -> int:
    return len(range(rank, height, world))


# ---- the collectives as torch.distributed calls (CPU tensors / gloo; tests) -------------------
def reduce_framebuffer(rgb_sum: torch.Tensor, counts: torch.Tensor, dst: int = 0) -> None:
    """output += pass for whole framebuffers: sums every rank's buffers into rank `dst`."""
    if not dist.is_initialized():
        return
    dist.reduce(rgb_sum, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(counts, dst=dst, op=dist.ReduceOp.SUM)


def gather_rows(rgb_sum: torch.Tensor, counts: torch.Tensor, dst: int = 0) -> None:
    """Interleaved-row frames: rank r's rows r, r+world, ... are copied into rank `dst`'s buffers
    (same contract as ptw_comm_gather_rows: other ranks' buffers are left alone)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    height = counts.shape[0]
    nmax = rows_owned(0, world, height)

    def packed(t):
        mine = t[rank::world]
        pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype)
        pad[: mine.shape[0]] = mine
        return pad

    for t in (rgb_sum, counts):
        send = packed(t)
        recv = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, recv, dst=dst)
        if rank == dst:
            for r in range(world):
                if r != dst:
                    n = rows_owned(r, world, height)
                    t[r::world] = recv[r][:n]


# ---- the library's own collectives (GPU) ---------------------------------------------------
class FrameComm:
    """This rank's ptw_comm (RCCL behind the C ABI).  torch.distributed only transports the
    128-byte id rank 0 obtained from ptw_comm_unique_id."""

    def __init__(self, pkg, device: int):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        uid = [pkg.Comm.unique_id() if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(uid, src=0)
        self.comm = pkg.Comm.create(uid[0], self.world, self.rank, device)

    def reduce_framebuffer(self, rgb_sum: torch.Tensor, counts: torch.Tensor, dst: int = 0,
                           stream: int = 0) -> None:
        self.comm.reduce_framebuffer(rgb_sum.data_ptr(), counts.data_ptr(), counts.numel(), dst, stream)

    def gather_rows(self, rgb_sum: torch.Tensor, counts: torch.Tensor, dst: int = 0,
                    stream: int = 0) -> None:
        h, w = counts.shape
        self.comm.gather_rows(rgb_sum.data_ptr(), counts.data_ptr(), w, h, dst, stream)

    def wait(self, stream: int = 0, timeout_ms: int = 0) -> None:
        """Completion of the collectives enqueued on `stream` under the library's watchdog
        (ptw_comm_wait): a peer that died after its enqueue is an error, not a hang."""
        self.comm.wait(stream, timeout_ms)

    def describe(self) -> dict:
        """Which wire the collectives use (ptw_comm_describe): HIP's link report, the transport RCCL should
        pick, and what RCCL's own log names when it goes to a file."""
        return self.comm.describe()

    def close(self):
        self.comm.close()
