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


def _staged(group, t: torch.Tensor | None) -> bool:
    """gloo cannot send/recv device tensors: rehearsals of the rank plumbing on a GPU box stage the rows through the host."""
    return dist.get_backend(group) == "gloo" and t is not None and t.is_cuda


def scatter_rows(full: torch.Tensor | None, total: int, row_shape, dtype, device, src: int = 0, group=None,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """Root `src` holds `full` (total, *row_shape); every rank receives its own block of rows (into `out` when given:
    the allocation-free form the double-buffered ingress loop of bench.py uses).  With the "nccl" backend (RCCL over
    xGMI: grouped send/recv, one direct link per peer) the waits below order the CURRENT HIP STREAM behind the transfers
    (they do not block the host), so issuing this on a side stream overlaps ingress with compute."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(total, world, rank)
    mine = out if out is not None else torch.empty((hi - lo,) + tuple(row_shape), dtype=dtype, device=device)
    if tuple(mine.shape) != (hi - lo,) + tuple(row_shape):
        raise ValueError(f"out must have shape {(hi - lo,) + tuple(row_shape)}")
    staged = _staged(group, mine)
    if rank == src:
        ops = []
        for r in range(world):
            a, b = shard_range(total, world, r)
            if r == src:
                mine.copy_(full[a:b])
            elif b > a:
                blk = full[a:b].contiguous()
                ops.append(dist.P2POp(dist.isend, blk.cpu() if staged else blk, r, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for q in reqs:
            q.wait()
    elif hi > lo:
        buf = torch.empty(mine.shape, dtype=mine.dtype) if staged else mine
        for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, src, group)]):
            q.wait()
        if staged:
            mine.copy_(buf)
    return mine


def gather_rows(mine: torch.Tensor, total: int, dst: int = 0, group=None, out: torch.Tensor | None = None) -> torch.Tensor | None:
    """Inverse of scatter_rows: root `dst` gets (total, *row_shape) (written into `out` when given); other ranks get None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    staged = _staged(group, mine)
    if rank == dst:
        full = out if out is not None else torch.empty((total,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        if tuple(full.shape) != (total,) + tuple(mine.shape[1:]):
            raise ValueError(f"out must have shape {(total,) + tuple(mine.shape[1:])}")
        ops, bufs = [], []
        for r in range(world):
            a, b = shard_range(total, world, r)
            if r == dst:
                full[a:b].copy_(mine)
            elif b > a:
                buf = torch.empty((b - a,) + tuple(mine.shape[1:]), dtype=mine.dtype) if staged else full[a:b]
                bufs.append((a, b, buf))
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        if staged:
            for a, b, buf in bufs:
                full[a:b].copy_(buf)
        return full
    if mine.shape[0] > 0:
        blk = mine.contiguous()
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, blk.cpu() if staged else blk, dst, group)]):
            q.wait()
    return None
