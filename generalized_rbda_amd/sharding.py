"""Batch sharding across the GPUs of one node.

States are independent and the plan is immutable, so a global batch is split into contiguous slabs,
one per rank (one process per GPU); the only exchange step is gathering the result slabs
(RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).  No reduction, no halo.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slab of rank `rank`; slabs differ by at most one state."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_apply(compute: Callable, q, qd, x, group=None, gather: bool = True):
    """Run `compute(q_slab, qd_slab, x_slab) -> out_slab` on this rank's slab of the global batch and,
    if `gather`, all-gather the slabs into the full [B, nv] result on every rank.

    `compute` is Plan.forward_dynamics / Plan.inverse_dynamics in production; q, qd, x are the
    GLOBAL arrays (every rank holds, or can index, its own slab of them)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = q.shape[0]
    lo, hi = shard_range(B, rank, world)
    out = compute(q[lo:hi], qd[lo:hi], x[lo:hi])
    if not gather or world == 1:
        return out
    # slabs may differ by one row: pad to the largest, gather, trim
    n_max = (B + world - 1) // world
    pad = torch.zeros((n_max, out.shape[1]), dtype=out.dtype, device=out.device)
    pad[: hi - lo] = out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    parts = []
    for r in range(world):
        l, h = shard_range(B, r, world)
        parts.append(bufs[r][: h - l])
    return torch.cat(parts, dim=0)
