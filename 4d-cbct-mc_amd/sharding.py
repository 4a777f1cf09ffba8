"""History sharding across the GPUs of one node.

The reference splits histories over MPI ranks and sums the per-rank detector images with one
MPI_Reduce per projection (MC-GPU_v1.3.cu:689-809, :1019).  Here rank r simulates the contiguous
unit range `shard_range(units, r, R)` -- units are history ids (FAST) or RANECU batches (COMPAT) --
and the images are summed with one RCCL reduce (torch.distributed backend "nccl").  Because every
unit owns its RNG stream and tallies are integers, the reduced image equals the single-GPU image
bit for bit, whatever R is.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(units: int, rank: int, world: int) -> Tuple[int, int]:
    """(first, count) of rank's contiguous share of `units` (balanced to within one unit)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    lo = units * rank // world
    hi = units * (rank + 1) // world
    return lo, hi - lo


def reduce_image(image, dst: int = 0, narrow: bool = False) -> int:
    """Sum-reduce a per-rank uint64 tally (held as an int64 torch tensor, any shape) onto rank `dst`, in place there.
    Returns the payload bytes this rank handed to the collective.

    `narrow`: send 32-bit words when the SUM provably fits -- the ranks first agree on the largest word anywhere (one
    8-byte MAX all-reduce, so every rank takes the same branch; a collective with mismatched dtypes would hang), and if
    `max * world < 2^32` the low words are summed modulo 2^32 (two's-complement add = exact unsigned sum) and widened
    again on `dst`.  Halves the bytes on the xGMI links; exact either way."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    world = dist.get_world_size()
    if narrow:
        top = image.max().reshape(1)
        dist.all_reduce(top, op=dist.ReduceOp.MAX)
        if int(top.item()) * world < 2 ** 32:
            small = image.to(torch.int32)  # keeps the low 32 bits
            dist.reduce(small, dst=dst, op=dist.ReduceOp.SUM)
            if dist.get_rank() == dst:
                image.copy_(small.to(torch.int64) & 0xFFFFFFFF)
            return small.numel() * 4
    dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
    return image.numel() * 8
