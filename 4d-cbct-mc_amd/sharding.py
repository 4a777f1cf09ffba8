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


def reduce_image(image, dst: int = 0):
    """Sum-reduce a per-rank uint64 tally (held as an int64 torch tensor) onto rank `dst`."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
    return image
