#!/usr/bin/env python3
"""Headline benchmark: batched IK-QP solves/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (stack H, c + solve the QP) over one
synthetic batch that is already resident in HBM.  N = 1: BASELINE.json config 3
(Draco3-shaped, nv = 30, 4 FrameTasks + PostureTask + joint/velocity box limits,
B = 65 536).  N > 1: the same batch per GPU (config 5, weak scaling; instances
are independent so ranks never exchange data inside the timed region).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

PyTorch is plumbing only (process group, barrier, device-wide sync, output
tensors for the RCCL gather); the work is libpinkhip.so through ctypes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# test hooks (tests/test_bench_dryrun.py runs the N > 1 control flow on CPU with gloo)
DEVICE = os.environ.get("PINKHIP_BENCH_DEVICE", "cuda")
BACKEND = os.environ.get("PINKHIP_BENCH_BACKEND", "nccl")


def cpu_baseline(terms, sample: int):
    """The C oracle (restated Pink + Goldfarb-Idnani; Pink itself cannot run
    offline) on all host cores, on a bounded sample of the same workload."""
    from oracle import c_oracle
    from pink_amd import synthetic

    cores = os.cpu_count() or 1
    pf = synthetic.pink_form(terms.slice(0, sample))
    c_oracle.solve_ik_batch(**{k: (v[:256] if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[0] == sample else v)
                               for k, v in pf.items()}, nthreads=cores)  # warm-up
    t0 = time.perf_counter()
    ref = c_oracle.solve_ik_batch(**pf, nthreads=cores)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="solves/s", cores=cores, kind="port",
                sample=f"{sample} instances of the same workload, C oracle (stack + Goldfarb-Idnani), OpenMP over the batch",
                seconds=dt), ref, pf


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="draco3", choices=["ur5", "draco3", "jvrc"])
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--bounds", default="tight", choices=["tight", "kinematic"])
    ap.add_argument("--jacobians", default="dense", choices=["dense", "kinematic"])
    ap.add_argument("--cpu-sample", type=int, default=32768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra regimes (profiling runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    on_gpu = DEVICE == "cuda"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local_rank)} if on_gpu else {}
        dist.init_process_group(BACKEND, rank=rank, world_size=world, **kw)

    import __graft_entry__ as g
    from pink_amd import synthetic
    from pink_amd.batch_solver import BatchSolver

    # the library travels prebuilt; if it is stale only rank 0 rebuilds it (hipcc), the others wait
    if rank == 0:
        g.build_hip()
    if world > 1:
        dist.barrier()
    B = args.batch
    # every rank draws its own shard of the global batch (seeded by rank): weak scaling
    seed = synthetic.SEED0 + synthetic.CONFIGS[args.config]["config_id"] + 1000 * rank
    terms = synthetic.make_terms(args.config, B, bounds=args.bounds, jacobians=args.jacobians, seed=seed)
    batch = synthetic.pack(terms)
    nv = batch.nv

    solver = BatchSolver(device_id=local_rank)
    info = solver.device_info()
    dq_t = torch.empty((B, nv), dtype=torch.float64, device=DEVICE)
    st_t = torch.empty((B,), dtype=torch.int32, device=DEVICE)
    it_t = torch.empty((B,), dtype=torch.int32, device=DEVICE)
    dev = solver.upload(batch, out_ptrs=(dq_t.data_ptr(), st_t.data_ptr(), it_t.data_ptr()))

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
        else:
            solver.sync()

    for _ in range(args.warmup):
        solver.solve_device(dev)
    barrier()
    t0 = time.perf_counter()
    solver.timer_start()  # HIP events on the stream the kernel runs on
    for _ in range(args.steps):
        solver.solve_device(dev)
    kernel_ms = solver.timer_stop() / max(args.steps, 1)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=DEVICE)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # results: status, iteration statistics, RCCL gather of dq to rank 0 (untimed leg of
    # the job: the only inter-GPU traffic this workload has)
    status = st_t.cpu().numpy()
    iters = it_t.cpu().numpy()
    n_bad = int((status != 0).sum())
    gather_ms = None
    if world > 1:
        try:
            parts = [torch.empty_like(dq_t) for _ in range(world)] if rank == 0 else None
            barrier()
            tg = time.perf_counter()
            dist.gather(dq_t, parts, dst=0)
            if on_gpu:
                torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - tg) * 1e3
            bad_t = torch.tensor([n_bad], dtype=torch.int64, device=DEVICE)
            dist.all_reduce(bad_t)
            n_bad = int(bad_t.item())
        except Exception as exc:  # noqa: BLE001  report, never lose the bench line
            gather_ms = f"failed: {exc}"

    # other input regimes of the same config, kernel time only (HIP events), rank 0
    regimes = {}
    if rank == 0 and not args.headline_only:
        for label, kw in (("kinematic_bounds", dict(bounds="kinematic", jacobians="kinematic")),
                          ("tracking_small_errors", dict(bounds="kinematic", jacobians="kinematic", error_scale=0.02))):
            t2 = synthetic.make_terms(args.config, B, seed=seed + 7, **kw)
            d2 = solver.upload(synthetic.pack(t2))
            solver.solve_device(d2)
            solver.sync()
            solver.timer_start()
            for _ in range(5):
                solver.solve_device(d2)
            ms2 = solver.timer_stop() / 5
            r2 = solver.download(d2)
            regimes[label] = {"kernel_ms": ms2, "solves_per_s": B / (ms2 * 1e-3), "iters_mean": float(r2.iters.mean()),
                              "failed": int((r2.status != 0).sum())}
            d2.free()

    # stack-only kernel: the HBM-streaming half (build_ik equivalent), same batch
    solver.stack_device(dev)
    solver.sync()
    solver.timer_start()
    for _ in range(max(args.steps, 1)):
        solver.stack_device(dev)
    stack_ms = solver.timer_stop() / max(args.steps, 1)

    if rank == 0:
        dq = dq_t.cpu().numpy()
        total = world * B * args.steps
        value = total / elapsed
        bytes_qp = batch.bytes_per_qp()
        achieved = bytes_qp * B / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("solve_kernel_hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "ik_qp_solves_per_s",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}-shaped stand-in: nv={nv}, {len(terms.dense_tasks)} FrameTask(6 rows)+PostureTask, "
                            f"box limits, md={batch.md} barrier rows, B={B} per GPU, bounds={args.bounds}, jacobians={args.jacobians}",
                "batch_per_gpu": B, "global_batch": world * B, "nv": nv, "Kd": batch.Kd, "K": batch.K, "md": batch.md,
                "parallelism": f"batch-sharded x{world}",
                "solver": "Goldfarb-Idnani dual active set, HIP fp64, 64/W QPs per wavefront (W = 32 lanes per QP at nv = 30)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "ik_solve_packed_kernel", "kernel_ms": kernel_ms, "bytes_per_qp": bytes_qp,
                "note": "fused stack+solve is VALU-issue / LDS / latency bound, not HBM bound (DESIGN.md 3.1); stack_only is the HBM-streaming kernel",
            },
            "stack_only": {
                "kernel": "ik_stack_mfma_kernel", "kernel_ms": stack_ms, "bytes_per_qp": batch.bytes_per_stack(),
                "achieved": batch.bytes_per_stack() * B / (stack_ms * 1e-3) / 1e9, "unit": "GB/s",
                "frac": batch.bytes_per_stack() * B / (stack_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "other_regimes": regimes,
            "solver_stats": {"failed": n_bad, "iters_mean": float(iters.mean()), "iters_max": int(iters.max())},
            "gather_ms": gather_ms,
            "device": info.get("gcn_arch"),
        }
        if not args.no_cpu_baseline:
            sample = min(args.cpu_sample, B)
            base, ref, _ = cpu_baseline(terms, sample)
            line["cpu_baseline"] = base
            # the sample is the head of rank 0's batch: report parity on it
            line["parity"] = {"max_abs_dq_err_vs_oracle": float(np.abs(dq[:sample] - ref["dq"]).max()),
                              "instances_compared": sample, "tolerance": 1e-8}
        print(json.dumps(line))
    dev.free()
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
