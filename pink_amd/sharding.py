"""Batch sharding over the GPUs of one node (one process per GPU) -- no PyTorch.

IK instances are independent, so a batch is split into contiguous ranges
(SURVEY.md section 8e): rank ``r`` of ``W`` solves ``[lo, hi)`` on its own
device and nothing is exchanged while solving.  The only collective is the
optional gather of ``dq`` (+ status, iteration counts) to one rank afterwards:
``ncclGather`` over xGMI through the C ABI (:class:`pink_amd.comm.RcclComm`), from
the device buffers the solve kernel wrote into device buffers of the root, then
one D2H copy.  :class:`pink_amd.comm.HostComm` moves host arrays over the
rendezvous sockets instead (solvers without device memory: the CPU wave emulator
of the tests).
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .batch import IKBatch
from .batch_solver import BatchResult, split_iters
from .comm import RcclComm


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced range of rank ``rank``: sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _assemble(parts, sizes, nv) -> BatchResult:
    """``parts[r] = (dq, status, iters)`` padded to the largest shard -> the full batch."""
    dq = np.concatenate([np.asarray(p[0]).reshape(-1, nv)[: h - l] for p, (l, h) in zip(parts, sizes)], axis=0)
    st = np.concatenate([np.asarray(p[1])[: h - l] for p, (l, h) in zip(parts, sizes)])
    it = np.concatenate([np.asarray(p[2])[: h - l] for p, (l, h) in zip(parts, sizes)])
    it = np.array(it, dtype=np.int32)  # (raw: iteration count + the solver path in the high bits, as the library writes it)
    return BatchResult(np.ascontiguousarray(dq), np.ascontiguousarray(st, dtype=np.int32), it, split_iters(it))


def solve_sharded(batch: IKBatch, solver, comm, gather_to: Optional[int] = 0) -> Optional[BatchResult]:
    """Solve this rank's shard with ``solver`` and gather the results through ``comm``.

    Returns the full :class:`BatchResult` on rank ``gather_to`` (every rank when
    ``gather_to`` is ``None``: all-gather), ``None`` elsewhere.  With an
    :class:`~pink_amd.comm.RcclComm` the shard stays on the device between the
    solve and the collective; any other comm object only needs
    ``rank, world, gather_arrays(arrays, root)``.
    """
    rank, world = comm.rank, comm.world
    lo, hi = shard_bounds(batch.B, rank, world)
    shard = batch.slice(lo, hi)
    if world == 1:
        return solver.solve(shard)
    nv, n = batch.nv, hi - lo
    sizes = [shard_bounds(batch.B, r, world) for r in range(world)]
    nmax = max(h - l for l, h in sizes)  # every rank contributes equally sized (padded) buffers
    receiver = gather_to is None or rank == gather_to
    if isinstance(comm, RcclComm):
        # device path: kernel outputs -> padded device buffers -> ncclGather -> one D2H on the root
        dev = solver.upload(shard)
        solver.solve_device(dev)
        out = None
        try:
            recv = []
            for ptr, item in ((dev.d_dq, 8 * nv), (dev.d_status, 4), (dev.d_iters, 4)):
                nbytes = nmax * item
                send = ptr
                pad = None
                if n < nmax:  # shorter shard: stage into a buffer of the common size
                    pad = solver.alloc(nbytes)
                    solver.put(pad, np.zeros(nbytes, dtype=np.uint8))
                    solver.copy_d2d(pad, ptr, n * item)
                    send = pad
                recv.append((comm.gather_device(send, nbytes, gather_to), nbytes, pad))
            solver.sync()
            if receiver:
                parts = [[None] * 3 for _ in range(world)]
                for k, ((d_recv, nbytes, _), dt) in enumerate(zip(recv, (np.float64, np.int32, np.int32))):
                    host = np.zeros(world * nbytes, dtype=np.uint8)
                    solver.get(host, d_recv)
                    for r in range(world):
                        parts[r][k] = host[r * nbytes:(r + 1) * nbytes].view(dt)
                out = _assemble(parts, sizes, nv)
        finally:
            for d_recv, _, pad in recv:
                solver.release(d_recv)
                solver.release(pad)
            dev.free()
        return out
    local = solver.solve(shard)
    dq = np.zeros((nmax, nv))
    st = np.zeros(nmax, dtype=np.int32)
    it = np.zeros(nmax, dtype=np.int32)
    dq[:n], st[:n], it[:n] = local.dq, local.status, local.iters
    if local.path is not None:  # (travels the way the library encodes it)
        it[:n] |= local.path.astype(np.int32) << 24
    parts = comm.gather_arrays([dq, st, it], gather_to)
    return _assemble(parts, sizes, nv) if receiver else None


class MultiDeviceSolver:
    """Several GPUs driven from ONE process (SURVEY.md 8(b) ``pinkhip_create(h, device_ids, n_devices)``, section 5
    ``device_ids=``): one :class:`BatchSolver` -- handle, stream, staging area -- per device and one host thread per
    device; ``ctypes`` releases the GIL inside the library calls, so the shards copy and compute concurrently.  A
    batch is cut into contiguous ranges (``shard_bounds``) and nothing is exchanged between the devices; the results
    are concatenated on the host.  ``solver_factory(device_id)`` builds the per-device solver (the tests pass the CPU
    wave emulator)."""

    def __init__(self, device_ids, solver_factory=None):
        from concurrent.futures import ThreadPoolExecutor

        from .batch_solver import BatchSolver

        ids = [int(d) for d in device_ids]
        if not ids or len(set(ids)) != len(ids):
            raise ValueError("device_ids must be a non-empty list of distinct device indices")
        make = solver_factory or (lambda d: BatchSolver(device_id=d))
        self.device_ids = ids
        self.solvers = [make(d) for d in ids]
        self._pool = ThreadPoolExecutor(max_workers=len(ids), thread_name_prefix="pinkhip-dev")

    def map(self, fn):
        """``[fn(r, solver_r) for r]``, each on its own thread."""
        return list(self._pool.map(lambda rs: fn(*rs), enumerate(self.solvers)))

    def solve(self, batch: IKBatch, max_iter: int = 0) -> BatchResult:
        world = len(self.solvers)
        bounds = [shard_bounds(batch.B, r, world) for r in range(world)]
        import time

        def one(r, s):
            if bounds[r][1] <= bounds[r][0]:
                return None
            t0 = time.perf_counter()
            out = s.solve(batch.slice(*bounds[r]), max_iter=max_iter)
            return out, 1e3 * (time.perf_counter() - t0)

        t_all = time.perf_counter()
        done = self.map(one)
        # per-device breakdown of the last call (kernel-only and end-to-end scaling reported separately):
        # end-to-end = H2D + kernel + D2H of that device's shard as seen by its host thread; kernel = the HIP-event time of
        # the shard's solve kernel(s) where the per-device solver reports it
        self.last_timing = {
            "wall_ms": 1e3 * (time.perf_counter() - t_all),
            "per_device": [None if d is None else {"device_id": self.device_ids[r], "instances": bounds[r][1] - bounds[r][0],
                                                   "end_to_end_ms": d[1], "kernel_ms": getattr(self.solvers[r], "last_kernel_ms", None)}
                           for r, d in enumerate(done)],
        }
        parts = [d[0] for d in done if d is not None]
        if not parts:
            return self.solvers[0].solve(batch, max_iter=max_iter)
        path = None if any(p.path is None for p in parts) else np.concatenate([p.path for p in parts])
        return BatchResult(np.concatenate([p.dq for p in parts]), np.concatenate([p.status for p in parts]),
                           np.concatenate([p.iters for p in parts]), path)

    def stack(self, batch: IKBatch):
        return self.solvers[0].stack(batch)

    def frame_task_terms(self, *args):
        return self.solvers[0].frame_task_terms(*args)

    def sync(self) -> None:
        for s in self.solvers:
            s.sync()

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        for s in self.solvers:
            s.close()


_POOLS = {}


def device_pool(device_ids) -> MultiDeviceSolver:
    """The process-wide :class:`MultiDeviceSolver` for this tuple of devices (created on first use)."""
    key = tuple(int(d) for d in device_ids)
    if key not in _POOLS:
        _POOLS[key] = MultiDeviceSolver(key)
    return _POOLS[key]

