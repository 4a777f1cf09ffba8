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


def shard_projections(n_projections: int, rank: int, world: int) -> range:
    """The projections rank `rank` owns under PROJECTION sharding (SURVEY.md 8e's fallback mode): number rank, rank + world, ...
    of the simulated ones.  Every rank simulates ALL histories of its projections; nothing is exchanged and no collective runs
    on the data path.  (The engine's `MCGPU_SHARD_PROJECTIONS` / `MC-GPU_v1.3.x --shard projections` applies the same rule.)"""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return range(rank, n_projections, world)


def reduce_image(image, dst: int = 0, narrow: bool = True, algorithm: str = "scatter") -> int:
    """Sum the per-rank uint64 tallies (held as an int64 torch tensor, any shape) onto rank `dst`, in place there.
    Returns the payload bytes this rank handed to the collectives.

    algorithm "scatter" (default) is shaped for xGMI, which is point-to-point -- every GPU has its own link to every other,
    and a chain/ring reduce to one root pushes the whole tally through single links:
      1. the tally is cut into `world` slices and slice k of every rank goes straight to rank k (all_to_all_single: all
         links of the node busy at once, each carrying 1/world of the payload);
      2. every rank sums the `world` slices it received (one elementwise kernel);
      3. the summed slices are gathered on `dst` (again one slice per link).
    Per link that is 2/world of the tally instead of all of it.  algorithm "reduce" is the plain `dist.reduce`.

    `narrow`: 32-bit words on the wire whenever the values provably fit.  The ranks first agree on the largest word anywhere
    (one 8-byte MAX all-reduce, so every rank takes the same branch; a collective with mismatched dtypes would hang): step 1
    sends 32-bit words if max < 2^31, step 3 (sums of `world` words, as low words) if max * world < 2^32.  Integer sums: the result equals the single-GPU tally bit for bit either way."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    world, rank = dist.get_world_size(), dist.get_rank()
    top = 2 ** 62
    if narrow:
        t = image.max().reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        top = int(t.item())
    if algorithm == "reduce":
        if top * world < 2 ** 31:
            small = image.to(torch.int32)
            dist.reduce(small, dst=dst, op=dist.ReduceOp.SUM)
            if rank == dst:
                image.copy_(small)
            return small.numel() * 4
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
        return image.numel() * 8
    flat = image.view(-1)
    n = flat.numel()
    m = (n + world - 1) // world
    # 32-bit words: a rank's own words must stay below 2^31 for step 1 (they are summed as signed numbers), the sums of
    # `world` of them below 2^32 for step 3 (sent as low words)
    wire1 = torch.int32 if top < 2 ** 31 else torch.int64
    wire2 = torch.int32 if top * world < 2 ** 32 else torch.int64  # low words; widened again with a mask on `dst`
    key = (n, world, str(image.device))
    buf = _BUFFERS.get(key)
    if buf is None:  # staging buffers are kept: a bench step must not pay allocations of a few hundred MB
        while len(_BUFFERS) >= _MAX_BUFFER_SETS:  # a few payload shapes stay resident (full groups and a remainder group)
            _BUFFERS.pop(next(iter(_BUFFERS)))
        # sized for 64-bit words and viewed as 32-bit ones when the payload is narrowed: one set serves both wire formats
        buf = _BUFFERS[key] = {"send": torch.empty(world * m, dtype=torch.int64, device=image.device),
                               "recv": torch.empty(world * m, dtype=torch.int64, device=image.device),
                               "full": torch.empty(world * m, dtype=torch.int64, device=image.device) if rank == dst else None}
    as_wire = lambda b, w: b if w == torch.int64 else b.view(torch.int32)[:world * m]
    send, recv = as_wire(buf["send"], wire1), as_wire(buf["recv"], wire1)
    full = as_wire(buf["full"], wire2) if rank == dst else None
    send[:n].copy_(flat)  # narrowing copy
    send[n:].zero_()      # padding of the last slice
    dist.all_to_all_single(recv, send)
    part = recv.view(world, m).sum(dim=0, dtype=torch.int64)
    part_w = part if wire2 == torch.int64 else part.to(torch.int32)  # keeps the low 32 bits
    parts = list(full.view(world, m).unbind(0)) if rank == dst else None
    dist.gather(part_w, parts, dst=dst)
    if rank == dst:
        flat.copy_(full[:n])  # widening copy (sign-extends)
        if wire2 == torch.int32:
            flat.bitwise_and_(0xFFFFFFFF)
    return send.numel() * send.element_size() + part_w.numel() * part_w.element_size()


def connect_exchange(x, dist, group=None):
    """Swap the address cards of a tally exchange (engine.Exchange: IPC memory and event handles of every rank's landing
    buffer) between the ranks of the initialised process group `dist` (torch.distributed, any backend; `group`: the group the
    small control messages travel on, e.g. a gloo group beside an RCCL world) and connect this rank's end to every peer.  Collective: every rank calls it once, with `x = None` if it could not even create its end.
    Returns (ok, error): ok is the SAME on every rank -- False if any rank failed anywhere (no IPC between these devices, an
    interprocess event refused ...), so that all ranks take the same fallback; it never leaves a peer waiting in a collective."""
    world, rank = dist.get_world_size(), dist.get_rank()
    err = None
    card = None
    if x is not None:
        try:
            card = x.card()
        except Exception as e:  # noqa: BLE001 -- reported, and agreed on below
            err = e
    cards = [None] * world
    dist.all_gather_object(cards, card, group=group)
    if x is None or any(c is None for c in cards):
        err = err or RuntimeError("a rank has no exchange end")
    else:
        try:
            for peer in range(world):
                if peer != rank:
                    x.connect(peer, cards[peer])
        except Exception as e:  # noqa: BLE001
            err = e
    flags = [None] * world
    dist.all_gather_object(flags, err is None, group=group)  # also the barrier: nobody starts pushing before everybody has mapped everybody
    if not all(flags):
        return False, err
    try:  # everybody is mapped: can the copy engine of this device reach every peer's landing buffer?
        x.probe()
    except Exception as e:  # noqa: BLE001
        err = e
    dist.all_gather_object(flags, err is None, group=group)
    return all(flags), err


_BUFFERS: dict = {}
_MAX_BUFFER_SETS = 4
