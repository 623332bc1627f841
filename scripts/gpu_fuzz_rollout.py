#!/usr/bin/env python3
"""One-off fuzz of the device-resident path on the GPU box: random kinematic chains (fixed and floating base), random
configurations and FrameTask targets, optionally a PositionBarrier and a FloatingBaseVelocityLimit -- the whole-step kernel
(solve_ik_batch(device_kinematics=True): kinematics, rows, limits, QP on chip) against the host-evaluated path (tasks,
limits and barriers evaluated per configuration in NumPy as Pink does, only the QP on the device).
   python scripts/gpu_fuzz_rollout.py [first] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pink_amd  # noqa: E402
from pink_amd import Configuration, FrameTask, PostureTask, build_chain, solve_ik_batch  # noqa: E402
from pink_amd.barriers import PositionBarrier  # noqa: E402
from pink_amd.lie import SE3, exp6  # noqa: E402
from pink_amd.limits import FloatingBaseVelocityLimit  # noqa: E402


def one(sd):
    rng = np.random.default_rng(sd)
    ff = bool(rng.random() < 0.5)
    n = int(rng.integers(3, 25))
    m = build_chain(n, free_flyer=ff, seed=int(rng.integers(0, 1000)), limit=float(rng.uniform(1.5, 3.1)),
                    velocity=float(rng.uniform(0.5, 5.0)))
    frames = ["tool0"] + [f"joint_{int(k)}" for k in sorted(set(rng.integers(1, n + 1, size=int(rng.integers(0, 3)))))]
    B, dt = int(rng.integers(1, 7)), float(rng.choice([1e-3, 5e-3, 2e-2]))
    q = np.tile(m.neutral(), (B, 1))
    for j in m.joints:
        if j.kind == "free_flyer":
            for b in range(B):
                M = exp6(rng.normal(size=6) * 0.5)
                q[b, j.idx_q:j.idx_q + 3] = M.translation
                from pink_amd.configuration import _rot_to_quat
                q[b, j.idx_q + 3:j.idx_q + 7] = _rot_to_quat(M.rotation)
        else:
            q[:, j.idx_q] = rng.uniform(-1.2, 1.2, size=B)
    m.floating_base_velocity_limit = None
    if ff and rng.random() < 0.4:
        root_id = m.joints.index(m.root_joint)
        T = SE3(np.eye(3), rng.normal(size=3) * 0.1) if rng.random() < 0.5 else exp6(rng.normal(size=6) * 0.3)
        m.add_frame("base", root_id, T)
        m.floating_base_velocity_limit = FloatingBaseVelocityLimit(m, "base", float(rng.uniform(0.05, 1.0)), float(rng.uniform(0.05, 1.0)))
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    scale = 10 ** rng.uniform(-3, -0.5)
    tasks = [[] for _ in range(B)]
    for k, f in enumerate(frames):
        pc, oc = float(rng.uniform(0.3, 2.0)), float(rng.choice([0.0, 0.5, 1.0]))
        lm, gain = float(rng.choice([0.0, 1e-3, 1e-1])), float(rng.uniform(0.3, 1.0))
        for b, cfg in enumerate(cfgs):
            t = FrameTask(f, pc, oc, lm_damping=lm, gain=gain)
            t.set_target(cfg.get_transform_frame_to_world(f) * exp6(scale * rng.normal(size=6)))
            tasks[b].append(t)
    if rng.random() < 0.8:
        cost = float(10 ** rng.uniform(-3, -1))
        for b in range(B):
            p = PostureTask(cost=cost)
            p.set_target(q[b] if rng.random() < 0.5 else m.neutral())
            tasks[b].append(p)
    bars = []
    if rng.random() < 0.4:
        p0 = np.array([c.get_transform_frame_to_world(frames[0]).translation for c in cfgs])
        bars.append(PositionBarrier(frames[0], indices=[2], p_max=np.array([p0[:, 2].max() + float(rng.uniform(0.0, 0.05))]),
                                    gain=np.array([float(rng.uniform(5.0, 100.0))]),
                                    safe_displacement_gain=float(rng.choice([0.0, 1.0]))))
    kw = dict(barriers=bars or None)
    try:
        V_host = solve_ik_batch(cfgs, tasks, dt, device_kinematics=False, gpu_frame_tasks=False, **kw)
        host_err = None
    except pink_amd.PinkError as exc:
        V_host, host_err = None, type(exc).__name__
    try:
        V_dev = solve_ik_batch(cfgs, tasks, dt, device_kinematics=True, **kw)
        dev_err = None
    except pink_amd.PinkError as exc:
        V_dev, dev_err = None, type(exc).__name__
    if host_err or dev_err:
        return ("same failure" if host_err == dev_err else f"host {host_err} / device {dev_err}"), 0.0
    # two evaluations of the same QP agree to cond(H) eps: the tolerance follows the conditioning of each instance
    cond = np.array([np.linalg.cond(pink_amd.build_ik(cfgs[b], tasks[b], dt, barriers=bars or None).P) for b in range(B)])
    rel = np.abs(V_dev - V_host).max(axis=1) / np.maximum(1.0, np.abs(V_host).max(axis=1))
    err = float((rel / np.maximum(1.0, 1e4 * cond * np.finfo(float).eps / 1e-8)).max())  # (compared with 1e-8)
    if os.environ.get("FUZZ_VERBOSE"):
        print(dict(ff=ff, n=n, frames=frames, B=B, dt=dt, scale=scale, bars=len(bars), fb=m.floating_base_velocity_limit is not None,
                   ntasks=len(tasks[0])), "per-instance rel err", rel, "cond", cond, "max |V_host|", np.abs(V_host).max(axis=1), "max |V_dev|", np.abs(V_dev).max(axis=1))
    return None, err


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    t0, worst, bad, failures = time.time(), 0.0, [], 0
    for sd in range(first, first + count):
        try:
            note, err = one(sd)
        except Exception as exc:  # noqa: BLE001
            note, err = f"{type(exc).__name__}: {exc}", 0.0
        if note == "same failure":
            failures += 1
        elif note:
            bad.append((sd, note))
            print("  seed", sd, "->", note[:200], flush=True)
        elif err > 1e-8:
            bad.append((sd, err))
            print("  seed", sd, "-> relative difference", err, flush=True)
        worst = max(worst, err)
        pink_amd.clear_device_cache()
    print(f"{count} draws: {len(bad)} disagreements {bad[:10]}, {failures} draws where both paths raise the same error, "
          f"largest relative difference {worst:.2e}; {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
