"""Stream sharding across the GPUs of one node.

Streams are independent (per-stream hx / ring / overlap-add state, shared read-only weights), so
the path shards with NO data-path collective: rank r owns the contiguous block of streams
``shard_range(total, world, r)`` for the streams' lifetime.  The only communication is ingress /
egress of audio when one rank holds all of it: ``scatter_rows`` / ``gather_rows`` (grouped
send/recv -- RCCL over xGMI when the process group is "nccl", gloo in the CPU tests).
Device RNG streams are keyed by the GLOBAL stream id (``stream_id0`` = first id of the shard), so a
sharded run reproduces the single-GPU run bit for bit.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of the streams rank `rank` owns; the first `total % world` ranks get one extra."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_rows(full: torch.Tensor | None, total: int, row_shape, dtype, device, src: int = 0, group=None) -> torch.Tensor:
    """Root `src` holds `full` (total, *row_shape); every rank receives its own block of rows."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(total, world, rank)
    mine = torch.empty((hi - lo,) + tuple(row_shape), dtype=dtype, device=device)
    if rank == src:
        ops = []
        for r in range(world):
            a, b = shard_range(total, world, r)
            if r == src:
                mine.copy_(full[a:b])
            elif b > a:
                ops.append(dist.P2POp(dist.isend, full[a:b].contiguous(), r, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
    else:
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine, src, group)]) if hi > lo else []
    for q in reqs:
        q.wait()
    return mine


def gather_rows(mine: torch.Tensor, total: int, dst: int = 0, group=None) -> torch.Tensor | None:
    """Inverse of scatter_rows: root `dst` gets (total, *row_shape); other ranks get None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == dst:
        full = torch.empty((total,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        ops = []
        for r in range(world):
            a, b = shard_range(total, world, r)
            if r == dst:
                full[a:b].copy_(mine)
            elif b > a:
                ops.append(dist.P2POp(dist.irecv, full[a:b], r, group))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return full
    if mine.shape[0] > 0:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.contiguous(), dst, group)]):
            q.wait()
    return None
