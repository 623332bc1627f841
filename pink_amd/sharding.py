"""Batch sharding over the GPUs of one node (one process per GPU).

IK instances are independent, so a batch is split into contiguous ranges
(SURVEY.md section 8e): rank ``r`` of ``W`` solves ``[lo, hi)`` on its own
device and nothing is exchanged while solving.  The only collective is the
optional gather of ``dq`` to one rank afterwards (RCCL when the process group
is ``nccl``, gloo on CPU for tests).
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .batch import IKBatch
from .batch_solver import BatchResult


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced range of rank ``rank``: sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def solve_sharded(batch: IKBatch, solver, rank: int, world: int, gather_to: Optional[int] = 0,
                  group=None, device: str = "cpu") -> Optional[BatchResult]:
    """Solve this rank's shard with ``solver`` and gather the results.

    Returns the full :class:`BatchResult` on rank ``gather_to`` (every rank when
    ``gather_to`` is ``None``: all-gather), ``None`` elsewhere.  ``device`` is where
    the collective's buffers live ("cuda" with the nccl/RCCL backend).
    """
    lo, hi = shard_bounds(batch.B, rank, world)
    local = solver.solve(batch.slice(lo, hi))
    if world == 1:
        return local
    import torch
    import torch.distributed as dist

    nv = batch.nv
    sizes = [shard_bounds(batch.B, r, world) for r in range(world)]
    nmax = max(h - l for l, h in sizes)
    # pad to the largest shard so every rank contributes equally sized buffers
    buf = torch.zeros((nmax, nv + 2), dtype=torch.float64, device=device)
    n = hi - lo
    buf[:n, :nv] = torch.from_numpy(local.dq)
    buf[:n, nv] = torch.from_numpy(local.status.astype(np.float64))
    buf[:n, nv + 1] = torch.from_numpy(local.iters.astype(np.float64))
    if gather_to is None:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == gather_to else None
        dist.gather(buf, parts, dst=gather_to, group=group)
        if rank != gather_to:
            return None
    full = np.concatenate([p[: h - l].cpu().numpy() for p, (l, h) in zip(parts, sizes)], axis=0)
    return BatchResult(np.ascontiguousarray(full[:, :nv]), full[:, nv].astype(np.int32), full[:, nv + 1].astype(np.int32))
