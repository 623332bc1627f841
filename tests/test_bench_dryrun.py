"""bench.py's N > 1 control flow (process group, barrier, max-over-ranks timing, gather of dq,
one JSON line from rank 0) exercised on CPU: two gloo ranks, the device solver replaced by a
stand-in that runs the kernel source on the wave emulator.  The real run needs MI355X GPUs."""
import ctypes
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import ctypes, os, sys, time
    sys.path.insert(0, %(root)r)
    import numpy as np
    from pink_amd import batch_solver
    from pink_amd._lib import Desc, Problem, Result
    from tests.conftest import EmuSolver

    lib = ctypes.CDLL(os.path.join(%(root)r, "tests", "emu", "libpinkemu.so"))
    lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_emu_stack_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.c_void_p, ctypes.c_void_p]
    lib.pinkhip_emu_last_error.restype = ctypes.c_char_p

    class FakeDev:
        def __init__(self, batch, out_ptrs):
            self.batch, self.out_ptrs = batch, out_ptrs
            self.res = None
        def free(self):
            pass

    class FakeSolver:  # same surface as BatchSolver for what bench.py uses
        def __init__(self, device_id=0):
            self.emu = EmuSolver(lib)
            self.t0 = 0.0
        def device_info(self):
            return {"gcn_arch": "cpu-emulator"}
        def upload(self, batch, max_iter=0, out_ptrs=None):
            return FakeDev(batch, out_ptrs)
        def solve_device(self, dev):
            dev.res = r = self.emu.solve(dev.batch)
            if dev.out_ptrs is not None:
                for ptr, arr in zip(dev.out_ptrs, (r.dq, r.status, r.iters)):
                    ctypes.memmove(ptr, arr.ctypes.data, arr.nbytes)
        def stack_device(self, dev):
            self.emu.stack(dev.batch)
        def download(self, dev):
            return dev.res
        def sync(self):
            pass
        def timer_start(self):
            self.t0 = time.perf_counter()
        def timer_stop(self):
            return (time.perf_counter() - self.t0) * 1e3
        def close(self):
            pass

    batch_solver.BatchSolver = FakeSolver
    import __graft_entry__ as g
    g.build_hip = lambda force=False: None
    sys.argv = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "ur5", "--batch", "6",
                "--cpu-sample", "6"]
    import bench
    bench.main()
    """
)


def test_two_rank_bench_control_flow(built, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PINKHIP_BENCH_DEVICE="cpu", PINKHIP_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 12 and line["value"] > 0
    assert isinstance(line["gather_ms"], float)
    assert line["solver_stats"]["failed"] == 0
    assert line["parity"]["max_abs_dq_err_vs_oracle"] < 1e-10
    for key in ("roofline", "cpu_baseline", "stack_only", "metric", "unit", "dtype", "vs_baseline"):
        assert key in line
