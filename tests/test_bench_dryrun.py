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
    EXTRA = %(extra)r
    import numpy as np
    from pink_amd import batch_solver
    from pink_amd._lib import Desc, Problem, Result
    from tests.conftest import EmuSolver

    lib = ctypes.CDLL(os.path.join(%(root)r, "tests", "emu", "libpinkemu.so"))
    lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_emu_stack_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.c_void_p, ctypes.c_void_p]
    lib.pinkhip_emu_last_error.restype = ctypes.c_char_p

    class FakeDev:
        def __init__(self, batch):
            from pink_amd._lib import PackedArgs
            self.batch, self.args = batch, PackedArgs(batch)
            self.res = None
        def free(self):
            pass

    class FakeSolver:  # same surface as BatchSolver for what bench.py uses (no device, no RCCL binding)
        def __init__(self, device_id=0):
            self.emu = EmuSolver(lib)
            self.t0 = 0.0
        def device_info(self):
            return {"gcn_arch": "cpu-emulator"}
        def upload(self, batch, max_iter=0, out_ptrs=None):
            return FakeDev(batch)
        def solve_device(self, dev):
            dev.res = self.emu.solve(dev.batch)
        def stack_device(self, dev):
            self.emu.stack(dev.batch)
        def download(self, dev):
            return dev.res
        def solve(self, batch, max_iter=0):
            return self.emu.solve(batch)
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
    sys.argv = ["bench.py", "--gpus", "%(world)d", "--steps", "2", "--warmup", "1", "--config", "ur5", "--batch", "6"] + EXTRA
    import bench
    bench._device_count = lambda: %(world)d
    bench.main()
    """
)


def _run(tmp_path, extra, world=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "extra": extra, "world": world})
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PINK_BENCH_DETAIL=str(tmp_path / "detail.json"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert not any(o[0].strip() for o in outs[1:]), "only rank 0 prints"
    head = _strict_headline(outs[0][0])
    assert "bench detail: {" in outs[0][1]  # the full record also goes to stderr
    detail = json.loads((tmp_path / "detail.json").read_text(), parse_constant=_no_constants)
    for key in head:  # the headline is a projection of the detail record
        if key not in ("detail", "also", "config", "roofline", "cpu_baseline", "parity", "gather", "comm_note"):
            assert head[key] == detail[key], key
    return head, detail


def _no_constants(name):
    raise AssertionError(f"{name} in the bench output: not JSON")


def _strict_headline(stdout: str) -> dict:
    """What the driver does with stdout: the LAST line, strict JSON, well inside its 8 KB tail."""
    lines = stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:3]
    assert len(lines[0]) < 4096, len(lines[0])  # (round 4's line was 22.6 KB: the driver could not parse it)
    head = json.loads(lines[0], parse_constant=_no_constants)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "detail"):
        assert key in head, key
    assert head["metric"] == "ik_qp_solves_per_s" and head["unit"] == "solves/s" and head["dtype"] == "f64"
    assert set(head["config"]) >= {"workload", "batch_per_gpu", "global_batch", "nv", "Kd", "K", "md", "parallelism", "solver"}
    assert "model" not in head["config"] and len(head["config"]["solver"]) <= 120
    assert set(head["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_qp"}
    return head


def test_two_rank_bench_control_flow(built, tmp_path):
    head, line = _run(tmp_path, [])
    assert head["n_gpus"] == 2 and head["value"] == line["value"] and head["roofline"]["frac"] == line["roofline"]["frac"]
    assert set(head["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and head["cpu_baseline"]["kind"] == "port"
    assert head["parity"]["instances_compared"] == 6 and head["parity"]["max_abs_err"] < 1e-10 and head["parity"]["tolerance"] == 1e-8
    assert len(head["per_rank_kernel_ms"]) == 2 and head["gather"]["rank0_shard_intact"] is True
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 12 and line["value"] > 0
    assert isinstance(line["gather"]["ms"], float) and line["gather"]["rank0_shard_intact"] is True
    assert line["solver_stats"]["failed"] == 0 and len(line["per_rank_kernel_ms"]) == 2
    assert line["parity"]["max_abs_dq_err_vs_oracle"] < 1e-10
    for key in ("roofline", "roofline_fp64", "cpu_baseline", "stack_only", "configs", "end_to_end", "latency_B1_us",
                "metric", "unit", "dtype", "vs_baseline"):
        assert key in line, key
    base = line["cpu_baseline"]
    for leg in ("B0_numpy_per_call", "B1_c_single_thread", "B2_c_all_cores", "B0prime_reference_build_ik"):
        assert leg in base, leg
    assert base["B1_c_single_thread"]["repeats"] >= 10 and base["kind"] == "port" and base["cores"] >= 1
    assert line["roofline"]["frac"] > 0 and line["roofline_fp64"]["peak"] == 78.6
    for cfg in ("ur5_B4096", "jvrc_B65536"):
        c = line["configs"][cfg]
        assert c["failed"] == 0 and c["parity"]["max_abs_dq_err_vs_oracle"] < 1e-9 and c["stack_only"]["frac"] > 0


def test_two_rank_strong_scaling_splits_the_global_batch(built, tmp_path):
    line, detail = _run(tmp_path, ["--scaling", "strong", "--global-batch", "10", "--headline-only", "--no-cpu-baseline"])
    assert "cpu_baseline" not in line and "parity" not in line and detail["gather"]["bytes_per_rank"] == 8 * 5 * 6
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 10 and line["config"]["batch_per_gpu"] == 5
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["gather"]["bytes_per_rank"] == 8 * 5 * 6


def test_eight_rank_strong_scaling_dry_run(built, tmp_path):
    """BASELINE configuration 5's launch shape on CPU: eight ranks, `--scaling strong`, a global batch that does not divide
    by eight (shards of 9 and 8) -- rendezvous, sharding, the barrier + max-over-ranks timing, the gather of dq and rank 0's
    one JSON line, so that the first real 8-GPU lease does not debug control flow.  (No N > 1 RCCL run exists: the pool's
    boxes have one GPU; the transport here is the host rendezvous.)"""
    line, detail = _run(tmp_path, ["--scaling", "strong", "--global-batch", "67", "--headline-only", "--no-cpu-baseline"], world=8)
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["global_batch"] == 67 and line["value"] > 0
    assert len(detail["per_rank_kernel_ms"]) == 8 and line["config"]["batch_per_gpu"] in (8, 9)
    assert detail["gather"].get("rank0_shard_intact") in (True, None) and "failed" not in detail["gather"]


def test_gpus_n_without_a_launcher_starts_n_ranks(built, tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment launches its two ranks itself (one JSON line,
    n_gpus = 2, per-rank kernel and end-to-end times) instead of running one rank."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "extra": ["--headline-only", "--no-cpu-baseline"], "world": 2})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PINK_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _strict_headline(out.stdout)
    assert line["n_gpus"] == 2 and len(line["per_rank_kernel_ms"]) == 2 and line["config"]["global_batch"] == 12


def test_gpus_n_refuses_when_devices_are_missing(built, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text((WORKER % {"root": ROOT, "extra": [], "world": 2}).replace("bench._device_count = lambda: 2", "bench._device_count = lambda: 1"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode != 0 and "device(s) visible" in out.stderr



@pytest.mark.gpu
def test_two_ranks_on_one_real_device(built, tmp_path):
    """The N > 1 control flow on real HIP: `bench.py --gpus 2` with both ranks on device 0 (the pool's boxes have one
    GPU; PINKHIP_ALLOW_SHARED_DEVICE=1) -- self-launch, TCP rendezvous, two library handles on one device, barrier +
    max-over-ranks timing, gather of dq to rank 0 (over the rendezvous: RCCL refuses duplicate devices), one bounded JSON
    line, clean exit without the watchdog.  The closest thing to an 8-GPU dry run this pool offers."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PINKHIP_ALLOW_SHARED_DEVICE="1", PINK_BENCH_DETAIL=str(tmp_path / "detail.json"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096",
                          "--headline-only", "--no-cpu-baseline"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _strict_headline(out.stdout)
    detail = json.loads((tmp_path / "detail.json").read_text(), parse_constant=_no_constants)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8192 and line["value"] > 0
    assert len(line["per_rank_kernel_ms"]) == 2 and all(ms > 0 for ms in line["per_rank_kernel_ms"])
    assert line["gather"]["rank0_shard_intact"] is True and line["gather"]["bytes_per_rank"] == 8 * 4096 * 30
    assert "share a device" in line["comm_note"] and "no N > 1 RCCL run" in line["comm_note"]
    assert detail["solver_stats"]["failed"] == 0 and detail["device"].startswith("gfx")
    assert "watchdog" not in out.stderr.lower() and "Traceback" not in out.stderr
