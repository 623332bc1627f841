"""N > 1 path on CPU: two gloo ranks shard one batch, solve their ranges (kernel
source on the wave emulator) and gather dq; the result must equal the
single-process solve bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    from pink_amd.sharding import shard_bounds

    for B in (0, 1, 7, 64, 65537):
        for W in (1, 2, 3, 8):
            rs = [shard_bounds(B, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [h - l for l, h in rs]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes

    import torch.distributed as dist

    from pink_amd._lib import Desc, Problem, Result
    from pink_amd.sharding import solve_sharded
    from tests.cases import config_case
    from tests.conftest import EmuSolver

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libpinkemu.so"))
    lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_emu_last_error.restype = ctypes.c_char_p
    batch, _ = config_case("draco3", "tight", "dense", 9)  # odd size: unequal shards
    res = solve_sharded(batch, EmuSolver(lib), rank, world, gather_to=0)
    everyone = solve_sharded(batch, EmuSolver(lib), rank, world, gather_to=None)
    assert everyone is not None and everyone.dq.shape == (9, 30)
    if rank == 0:
        np.savez(out_path, dq=res.dq, status=res.status, iters=res.iters, dq_all=everyone.dq)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(emu, tmp_path):
    import torch.multiprocessing as mp

    from tests.cases import config_case

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    batch, _ = config_case("draco3", "tight", "dense", 9)
    ref = emu.solve(batch)
    assert np.array_equal(got["dq"], ref.dq) and np.array_equal(got["dq_all"], ref.dq)
    assert np.array_equal(got["status"], ref.status) and np.array_equal(got["iters"], ref.iters)
