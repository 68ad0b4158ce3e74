"""Multi-GPU driver pieces: reference views are independent units, so they are dealt to the
ranks with no data-path collective; the only exchange is the final gather of (depth, normal, cost) to
the fusing rank.  The reference has no counterpart: it runs one process per view from a shell loop
(reference scripts/courtyard.sh:29-48) and exchanges results through files."""
from __future__ import annotations

from typing import List, Sequence


def shard_views(n_ref_views: int, world: int, rank: int) -> List[int]:
    """Reference-view ids processed by `rank`: round-robin (all views of a scene cost the same)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_ref_views, world))


def owner_of_view(view: int, world: int) -> int:
    return view % world


def alloc_gather_buffers(dist, tensors: Sequence, dst: int = 0):
    """Receive buffers for gather_results on `dst` (None elsewhere), so that a loop gathers into the same memory."""
    import torch
    if dist.get_rank() != dst:
        return None
    return [[torch.empty_like(t) for _ in range(dist.get_world_size())] for t in tensors]


def gather_results(dist, tensors: Sequence, dst: int = 0, out=None, async_op: bool = False):
    """Gather each rank's result tensors to `dst` (RCCL on GPUs, gloo in the CPU tests).
    Returns on dst a list (per tensor) of lists (per rank) — `out` if given (alloc_gather_buffers) —
    elsewhere None.  With async_op=True the collectives are only enqueued (RCCL: on the collective's own
    stream, so they overlap the next view's kernels) and the list of work handles is returned on every
    rank instead: wait() on each before `tensors` are overwritten or `out` is read."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    if async_op and out is None and rank == dst:           # before anything is enqueued: temporaries would be dropped with the gather in flight
        raise ValueError("async gather needs caller-owned receive buffers (alloc_gather_buffers)")
    res, works = [], []
    for k, t in enumerate(tensors):
        if rank == dst:
            bucket = out[k] if out is not None else [torch.empty_like(t) for _ in range(world)]
            wk = dist.gather(t, bucket, dst=dst, async_op=async_op)
            res.append(bucket)
        else:
            wk = dist.gather(t, None, dst=dst, async_op=async_op)
        works.append(wk)
    if async_op:
        return works
    return res if rank == dst else None
