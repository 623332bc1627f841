"""Process bootstrap and result transport for the batch-sharded multi-GPU path -- no PyTorch.

One process per GPU.  Instances are independent, so the ranks never
exchange data while solving; what a job needs between processes is

* a **control plane**: barrier, broadcast of the 128-byte RCCL unique id, max / sum of a few floats
  (timing).  :class:`HostRendezvous` does this over a TCP star on rank 0, addressed by the
  ``RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT`` variables every launcher (``torch.distributed.run``,
  ``mpirun`` wrappers, a shell loop) exports;
* a **data plane** for the one collective of the workload, the gather of ``dq`` (+ status, iteration
  counts) to one rank: :class:`RcclComm` = ``ncclGather`` / ``ncclAllGather`` over xGMI on the
  handles' streams through the C ABI (``pinkhip_comm_*``), device buffer to device buffer.
  :class:`HostComm` carries host arrays over the rendezvous sockets instead; it exists for solvers
  without device memory (the CPU wave emulator of the test suite) and as a transport of last resort.
"""

from __future__ import annotations

import os
import socket
import struct
import time
from typing import List, Optional, Sequence

import numpy as np

_MAGIC = b"PINKHIP1"
# Upper bound of one control-plane / host-transport message: the largest thing that ever crosses these sockets is the
# all-gather of host result arrays (HostComm, test transport); a length prefix beyond this is a foreign or broken peer.
MAX_MESSAGE_BYTES = int(os.environ.get("PINKHIP_RDZV_MAX_BYTES", str(4 << 30)))


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock: socket.socket, limit: int = 0) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > (limit or MAX_MESSAGE_BYTES):
        raise ConnectionError(f"rendezvous message of {n} bytes exceeds the limit")
    return _recv_exact(sock, n)


def _job_secret(token: int, world: int) -> bytes:
    """What a peer must present to join: ``PINKHIP_RDZV_SECRET`` (set it to a random string in the launcher's
    environment for jobs on a shared network) hashed with the job's (MASTER_PORT, world size).  Without the variable the
    handshake only tells this job's ranks from another job's probing the same port window."""
    import hashlib

    return hashlib.sha256(os.environ.get("PINKHIP_RDZV_SECRET", "").encode() + struct.pack("<qq", int(token), int(world))).digest()


def _is_local(addr: str) -> bool:
    """``addr`` names this host (then rank 0 listens on it alone, not on every interface)."""
    try:
        infos = socket.getaddrinfo(addr, None)
    except OSError:
        return False
    for info in infos:
        ip = info[4][0]
        if ip.startswith("127.") or ip == "::1":
            return True
        try:
            with socket.socket(info[0], socket.SOCK_DGRAM) as probe:
                probe.bind((ip, 0))
                return True
        except OSError:
            continue
    return False


class HostRendezvous:
    """TCP star centred on rank 0: every collective is "all send to 0, 0 answers".

    ``port`` is where rank 0 listens.  :meth:`from_env` derives it from ``MASTER_PORT`` (that port itself
    belongs to the launcher's own store) unless ``PINKHIP_RDZV_PORT`` names one; rank 0 walks a small
    window of ports until one binds, the other ranks probe the same window and recognise rank 0 by a
    handshake carrying the job's (MASTER_PORT, world size).
    """

    WINDOW = 16

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29611, token: int = 0,
                 timeout: float = 120.0):
        if not 0 <= rank < world:
            raise ValueError("rank out of range")
        self.rank, self.world = int(rank), int(world)
        self._peers: List[Optional[socket.socket]] = [None] * world  # rank 0: sockets to 1..world-1
        self._root: Optional[socket.socket] = None
        hello = _MAGIC + _job_secret(token, self.world)
        if world == 1:
            return
        deadline = time.monotonic() + timeout
        if rank == 0:
            srv = None
            for p in range(port, port + self.WINDOW):
                s = socket.socket()
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    # the job's own address when it names this host (the usual single-node case: 127.0.0.1); every
                    # interface only when MASTER_ADDR does not resolve to a local one
                    s.bind((addr if _is_local(addr) else "", p))
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise OSError(f"no free rendezvous port in {port}..{port + self.WINDOW - 1}")
            srv.listen(world)
            srv.settimeout(1.0)
            joined = 0
            while joined < world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: {joined + 1}/{world} ranks after {timeout:.0f} s")
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                c.settimeout(timeout)
                try:
                    msg = _recv(c, limit=256)  # a hello is 48 bytes
                except (ConnectionError, socket.timeout, struct.error):
                    c.close()
                    continue
                if msg[:len(hello)] != hello:  # somebody else's job probing the window
                    c.close()
                    continue
                (r,) = struct.unpack("<q", msg[len(hello):len(hello) + 8])
                if not 0 < r < world or self._peers[r] is not None:
                    c.close()
                    continue
                _send(c, hello)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(None)  # collectives block for as long as the slowest rank takes
                self._peers[r] = c
                joined += 1
            srv.close()
        else:
            while self._root is None:
                for p in range(port, port + self.WINDOW):
                    try:
                        s = socket.create_connection((addr, p), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        s.settimeout(timeout)
                        _send(s, hello + struct.pack("<q", self.rank))
                        if _recv(s, limit=256) == hello:
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            s.settimeout(None)
                            self._root = s
                            break
                    except (OSError, ConnectionError, struct.error):
                        pass
                    s.close()
                if self._root is None:
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rendezvous: rank {rank} found no rank 0 at {addr}:{port}+")
                    time.sleep(0.05)

    @classmethod
    def from_env(cls, timeout: float = 120.0) -> "HostRendezvous":
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        master_port = int(os.environ.get("MASTER_PORT", "29500"))
        # below the Linux ephemeral range (32768-60999): outgoing connections of other processes cannot sit on it
        port = int(os.environ.get("PINKHIP_RDZV_PORT", 20000 + (master_port * 7 + 13) % 12000))
        return cls(rank, world, addr, port, token=master_port, timeout=timeout)

    # -- collectives (all blocking, all ranks must call them in the same order) ----------------
    def gather_bytes(self, data: bytes, root: int = 0) -> Optional[List[bytes]]:
        """Rank ``root`` gets ``[data of rank 0, ..., data of rank world-1]``, the others ``None``."""
        allb = self.allgather_bytes(data) if root != 0 else self._gather0(data)
        return allb if self.rank == root else None

    def _gather0(self, data: bytes) -> Optional[List[bytes]]:
        if self.world == 1:
            return [data]
        if self.rank == 0:
            return [data] + [_recv(self._peers[r]) for r in range(1, self.world)]
        _send(self._root, data)
        return None

    def broadcast_bytes(self, data: Optional[bytes], root: int = 0) -> bytes:
        if self.world == 1:
            return data
        if root != 0:  # route through rank 0
            parts = self._gather0(data if self.rank == root else b"")
            data = parts[root] if self.rank == 0 else None
        if self.rank == 0:
            for r in range(1, self.world):
                _send(self._peers[r], data)
            return data
        return _recv(self._root)

    def allgather_bytes(self, data: bytes) -> List[bytes]:
        parts = self._gather0(data)
        blob = None
        if self.rank == 0:
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
        blob = self.broadcast_bytes(blob)
        out, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            out.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return out

    def barrier(self) -> None:
        self.allgather_bytes(b"")

    def allreduce_max(self, x: float) -> float:
        return max(struct.unpack("<d", p)[0] for p in self.allgather_bytes(struct.pack("<d", float(x))))

    def allreduce_sum(self, x: float) -> float:
        return sum(struct.unpack("<d", p)[0] for p in self.allgather_bytes(struct.pack("<d", float(x))))

    def close(self) -> None:
        for s in self._peers + [self._root]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers = [None] * self.world
        self._root = None


class HostComm:
    """Result transport over the rendezvous sockets (host arrays)."""

    def __init__(self, rdzv: HostRendezvous):
        self.rdzv = rdzv
        self.rank, self.world = rdzv.rank, rdzv.world

    def gather_arrays(self, arrays: Sequence[np.ndarray], root: Optional[int]) -> Optional[List[List[np.ndarray]]]:
        """``arrays`` of every rank on ``root`` (every rank if ``root`` is ``None``): ``out[rank][k]``."""
        blob = b"".join(struct.pack("<Q", a.nbytes) + np.ascontiguousarray(a).tobytes() for a in arrays)
        parts = self.rdzv.allgather_bytes(blob) if root is None else self.rdzv.gather_bytes(blob, root)
        if parts is None:
            return None
        out = []
        for p in parts:
            off, row = 0, []
            for a in arrays:
                (n,) = struct.unpack_from("<Q", p, off)
                row.append(np.frombuffer(p, dtype=a.dtype, count=n // a.dtype.itemsize, offset=off + 8))
                off += 8 + n
            out.append(row)
        return out

    def barrier(self) -> None:
        self.rdzv.barrier()

    def close(self) -> None:
        pass


class RcclComm:
    """``ncclGather`` / ``ncclAllGather`` between the ranks' :class:`~pink_amd.batch_solver.BatchSolver`
    handles (one per GPU), bootstrapped through the rendezvous: rank 0 creates the unique id, everybody joins."""

    def __init__(self, solver, rdzv: HostRendezvous):
        self.solver, self.rdzv = solver, rdzv
        self.rank, self.world = rdzv.rank, rdzv.world
        uid, err = b"", ""
        if self.rank == 0:
            try:
                uid = solver.comm_unique_id()
            except Exception as exc:  # noqa: BLE001  (librccl missing): tell the others instead of leaving them waiting
                err = repr(exc)
        uid = rdzv.broadcast_bytes(uid if self.rank == 0 else None)
        if not uid:
            raise RuntimeError(f"rank 0 could not create an RCCL unique id {err}")
        try:
            solver.comm_init(uid, self.rank, self.world)  # collective
            ok = 1.0
        except Exception as exc:  # noqa: BLE001
            ok, err = 0.0, repr(exc)
        if rdzv.allreduce_sum(ok) != float(self.world):  # every rank takes the same decision
            if ok:
                solver.comm_destroy()
            raise RuntimeError(f"ncclCommInitRank failed on some rank {err}")

    def gather_device(self, d_send: int, nbytes: int, root: Optional[int]) -> Optional[int]:
        """Gather ``nbytes`` from every rank's device buffer; returns the address of a fresh device buffer
        ``[world * nbytes]`` on the receiving rank(s) (the caller releases it with ``solver.release``)."""
        s = self.solver
        if root is None:
            d_recv = s.alloc(self.world * nbytes)
            s.comm_allgather_bytes(d_send, d_recv, nbytes)
            return d_recv
        d_recv = s.alloc(self.world * nbytes) if self.rank == root else None
        s.comm_gather_bytes(d_send, d_recv, nbytes, root)
        return d_recv

    def barrier(self) -> None:
        self.rdzv.barrier()
        self.solver.sync()

    def close(self) -> None:
        self.solver.comm_destroy()
