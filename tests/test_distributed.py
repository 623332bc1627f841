"""N > 1 path on CPU: world_size-2 jobs shard one batch, solve their ranges (kernel source on the wave
emulator) and gather dq; the result must equal the single-process solve bit for bit.  Two transports:
the product's own TCP rendezvous (pink_amd.comm, no PyTorch) and a gloo process group (test-only comm
object) -- the RCCL transport needs GPUs and is covered by tests/test_gpu_parity.py and bench.py."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_partition():
    from pink_amd.sharding import shard_bounds

    for B in (0, 1, 7, 64, 65537):
        for W in (1, 2, 3, 8):
            rs = [shard_bounds(B, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [h - l for l, h in rs]
            assert max(sizes) - min(sizes) <= 1


def _emu_solver():
    import ctypes

    from pink_amd._lib import Desc, Problem, Result
    from tests.conftest import EmuSolver

    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libpinkemu.so"))
    lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_emu_last_error.restype = ctypes.c_char_p
    return EmuSolver(lib)


def _rdzv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pink_amd.comm import HostRendezvous

    r = HostRendezvous(rank, world, "127.0.0.1", port, token=4242, timeout=60)
    r.barrier()
    got = r.broadcast_bytes(b"id-from-0" if rank == 0 else None)
    got2 = r.broadcast_bytes(b"from-2" if rank == 2 else None, root=2)
    g0 = r.gather_bytes(bytes([rank]) * (rank + 1), root=0)
    g1 = r.gather_bytes(bytes([rank]), root=1)
    ag = r.allgather_bytes(str(rank).encode())
    mx = r.allreduce_max(10.0 - rank)
    sm = r.allreduce_sum(float(rank))
    r.barrier()
    r.close()
    q.put((rank, got, got2, g0, g1, ag, mx, sm))


def test_host_rendezvous_collectives():
    """Three processes over the TCP star: barrier, broadcast (root 0 and not 0), gather, all-gather, max, sum."""
    ctx = mp.get_context("spawn")
    port, world = _free_port(), 3
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, got, got2, g0, g1, ag, mx, sm in res:
        assert got == b"id-from-0" and got2 == b"from-2"
        assert g0 == ([b"\x00", b"\x01\x01", b"\x02\x02\x02"] if rank == 0 else None)
        assert g1 == ([b"\x00", b"\x01", b"\x02"] if rank == 1 else None)
        assert ag == [b"0", b"1", b"2"] and mx == 10.0 and sm == 3.0


def _tcp_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    from pink_amd.comm import HostComm, HostRendezvous
    from pink_amd.sharding import solve_sharded
    from tests.cases import config_case

    comm = HostComm(HostRendezvous(rank, world, "127.0.0.1", port, token=7, timeout=60))
    batch, _ = config_case("draco3", "tight", "dense", 9)  # odd size: unequal shards
    res = solve_sharded(batch, _emu_solver(), comm, gather_to=0)
    everyone = solve_sharded(batch, _emu_solver(), comm, gather_to=None)
    assert everyone is not None and everyone.dq.shape == (9, 30)
    if rank == 0:
        np.savez(out_path, dq=res.dq, status=res.status, iters=res.iters, dq_all=everyone.dq)
    else:
        assert res is None
    comm.barrier()
    comm.rdzv.close()


def _check(emu, out):
    from tests.cases import config_case

    got = np.load(out)
    batch, _ = config_case("draco3", "tight", "dense", 9)
    ref = emu.solve(batch)
    assert np.array_equal(got["dq"], ref.dq) and np.array_equal(got["dq_all"], ref.dq)
    assert np.array_equal(got["status"], ref.status) and np.array_equal(got["iters"], ref.iters)


def test_two_rank_tcp_matches_single_process(emu, tmp_path):
    ctx = mp.get_context("spawn")
    port, out = _free_port(), str(tmp_path / "gathered.npz")
    ps = [ctx.Process(target=_tcp_worker, args=(r, 2, port, out)) for r in range(2)]
    [p.start() for p in ps]
    [p.join(timeout=600) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    _check(emu, out)


def _tcp_worker_wide(rank, world, port, out_path, B):
    sys.path.insert(0, ROOT)
    from pink_amd.comm import HostComm, HostRendezvous
    from pink_amd.sharding import shard_bounds, solve_sharded
    from tests.cases import config_case

    comm = HostComm(HostRendezvous(rank, world, "127.0.0.1", port, token=11, timeout=120))
    batch, _ = config_case("draco3", "tight", "dense", B)
    lo, hi = shard_bounds(B, rank, world)
    assert 0 <= lo <= hi <= B and hi - lo in (B // world, B // world + 1)
    res = solve_sharded(batch, _emu_solver(), comm, gather_to=0)
    if rank == 0:
        np.savez(out_path, dq=res.dq, status=res.status, iters=res.iters)
    else:
        assert res is None
    comm.barrier()
    comm.rdzv.close()


def test_eight_rank_tcp_uneven_shards_match_single_process(emu, tmp_path):
    """World 8 (BASELINE configuration 5's rank count) over the host rendezvous, a batch that does not divide by eight
    (131 = 524 288 / 4096 + 3: shards of 17 and 16): rank 0's gathered result is bit for bit the single-process one."""
    from tests.cases import config_case

    B, world = 131, 8
    ctx = mp.get_context("spawn")
    port, out = _free_port(), str(tmp_path / "gathered8.npz")
    ps = [ctx.Process(target=_tcp_worker_wide, args=(r, world, port, out, B)) for r in range(world)]
    [p.start() for p in ps]
    [p.join(timeout=900) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    got = np.load(out)
    batch, _ = config_case("draco3", "tight", "dense", B)
    ref = emu.solve(batch)
    assert np.array_equal(got["dq"], ref.dq) and np.array_equal(got["status"], ref.status) and np.array_equal(got["iters"], ref.iters)


class GlooComm:
    """Test-only transport: the same ``gather_arrays`` contract on a gloo process group."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def gather_arrays(self, arrays, root):
        import torch
        import torch.distributed as dist

        out = [[None] * len(arrays) for _ in range(self.world)]
        for k, a in enumerate(arrays):
            t = torch.from_numpy(np.ascontiguousarray(a))
            parts = [torch.empty_like(t) for _ in range(self.world)]
            if root is None:
                dist.all_gather(parts, t)
            else:
                dist.gather(t, parts if self.rank == root else None, dst=root)
            for r in range(self.world):
                out[r][k] = parts[r].numpy()
        return out if root is None or self.rank == root else None


def _gloo_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from pink_amd.sharding import solve_sharded
    from tests.cases import config_case

    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch, _ = config_case("draco3", "tight", "dense", 9)
    comm = GlooComm(rank, world)
    res = solve_sharded(batch, _emu_solver(), comm, gather_to=0)
    everyone = solve_sharded(batch, _emu_solver(), comm, gather_to=None)
    if rank == 0:
        np.savez(out_path, dq=res.dq, status=res.status, iters=res.iters, dq_all=everyone.dq)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(emu, tmp_path):
    import torch.multiprocessing as tmp_mp

    port, out = _free_port(), str(tmp_path / "gathered.npz")
    tmp_mp.spawn(_gloo_worker, args=(2, port, out), nprocs=2, join=True)
    _check(emu, out)


def test_product_has_no_torch_dependency():
    """north_star: host code is Python calling a thin C-ABI HIP extension, no PyTorch."""
    import re

    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "pink_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(import|from)\s+torch\b", open(os.path.join(base, f)).read(), re.M):
                offenders.append(f)
    if re.search(r"^\s*(import|from)\s+torch\b", open(os.path.join(ROOT, "bench.py")).read(), re.M):
        offenders.append("bench.py")
    assert not offenders, offenders
