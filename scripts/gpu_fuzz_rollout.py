#!/usr/bin/env python3
"""One-off fuzz of the device-resident path on the GPU box: random kinematic chains (fixed and floating base), random
configurations and FrameTask targets, optionally a PositionBarrier and a FloatingBaseVelocityLimit, and (FUZZ_EXTRAS=1)
the tasks the kernel forms from tables since round 4: RelativeFrameTasks, JointCouplingTasks, DampingTask,
LowAccelerationTask, JointVelocityTask -- and (FUZZ_EXTRAS=2) round 5's rows: BodySphericalBarriers, barriers on frames without a
task, constraints=[FrameTask] -- the whole-step kernel
(solve_ik_batch(device_kinematics=True): kinematics, rows, limits, QP on chip) against the host-evaluated path (tasks,
limits and barriers evaluated per configuration in NumPy as Pink does, only the QP on the device).
   python scripts/gpu_fuzz_rollout.py [first] [count]        (FUZZ_EMU=1: on the CPU wave emulator of tests/emu)"""
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


def _kw_of(kw, b):
    """The keyword arguments of instance b (per-instance constraint lists)."""
    if kw.get("constraints") is None:
        return kw
    return dict(kw, constraints=kw["constraints"][b])


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
    if os.environ.get("FUZZ_EXTRAS"):  # (a second stream: the draws above are those of the earlier runs)
        from pink_amd import DampingTask
        from pink_amd.tasks import JointCouplingTask, JointVelocityTask, LowAccelerationTask, RelativeFrameTask

        r2 = np.random.default_rng(sd + 10 ** 9)
        names = [f"joint_{k}" for k in range(1, n + 1)] + ["tool0"]
        for _ in range(int(r2.integers(0, 3))):  # relative frame tasks: per-instance objects, own targets
            f, r = (str(v) for v in r2.choice(names, size=2, replace=False))
            pc, oc = float(r2.uniform(0.3, 2.0)), float(r2.choice([0.0, 0.5, 1.0]))
            lm, gain = float(r2.choice([0.0, 1e-3])), float(r2.uniform(0.3, 1.0))
            at = int(r2.integers(0, len(tasks[0]) + 1))
            for b, cfg in enumerate(cfgs):
                t = RelativeFrameTask(f, r, pc, oc, lm_damping=lm, gain=gain)
                t.set_target(cfg.get_transform(f, r) * exp6(scale * r2.normal(size=6)))
                tasks[b].insert(at, t)
        shared = []
        if n >= 3 and r2.random() < 0.5:
            js = [f"joint_{int(k)}" for k in r2.choice(np.arange(1, n + 1), size=int(r2.integers(2, 4)), replace=False)]
            shared.append(JointCouplingTask(js, [float(v) for v in r2.uniform(-2.0, 2.0, size=len(js))], float(10 ** r2.uniform(-1, 2)), cfgs[0],
                                            lm_damping=float(r2.choice([0.0, 5e-7])), gain=float(r2.uniform(0.3, 1.0))))
        if r2.random() < 0.4:
            shared.append(DampingTask(cost=float(10 ** r2.uniform(-3, -1))))
        if r2.random() < 0.3:
            la = LowAccelerationTask(cost=float(10 ** r2.uniform(-3, -1)))
            la.set_last_integration(r2.normal(size=m.nv) * 0.3, dt)
            shared.append(la)
        if r2.random() < 0.3:
            jv = JointVelocityTask(cost=float(10 ** r2.uniform(-3, -1)))
            jv.set_target(r2.normal(size=m.nv - (6 if ff else 0)) * 0.3, dt)
            shared.append(jv)
        for t in shared:
            at = int(r2.integers(0, len(tasks[0]) + 1))
            for b in range(B):
                tasks[b].insert(at, t)
    bars = []
    if rng.random() < 0.4:
        p0 = np.array([c.get_transform_frame_to_world(frames[0]).translation for c in cfgs])
        bars.append(PositionBarrier(frames[0], indices=[2], p_max=np.array([p0[:, 2].max() + float(rng.uniform(0.0, 0.05))]),
                                    gain=np.array([float(rng.uniform(5.0, 100.0))]),
                                    safe_displacement_gain=float(rng.choice([0.0, 1.0]))))
    cons = None
    if os.environ.get("FUZZ_EXTRAS") == "2":
        # round 5 (a fourth stream): BodySphericalBarriers, a PositionBarrier on a frame without a task, constraints=[FrameTask]
        # with per-instance targets within reach of one step -- rows the whole-step kernel forms on chip
        from pink_amd.barriers import BodySphericalBarrier

        r4 = np.random.default_rng(sd + 3 * 10 ** 9)
        names = [f"joint_{k}" for k in range(1, n + 1)] + ["tool0"]
        if n >= 4 and r4.random() < 0.5:
            f1, f2 = (str(v) for v in r4.choice(names, size=2, replace=False))
            d = np.array([np.linalg.norm(c.get_transform_frame_to_world(f1).translation - c.get_transform_frame_to_world(f2).translation) for c in cfgs])
            sb = BodySphericalBarrier((f1, f2), d_min=float(d.min() * r4.uniform(0.5, 0.999)), gain=float(r4.uniform(1.0, 50.0)),
                                      safe_displacement_gain=float(r4.choice([0.0, 1.0, 3.0])))
            # (two frames at a fixed distance -- neighbours on one link -- have a zero barrier Jacobian: the reference divides
            # its safe-displacement gain by |J_h|^2 = 0, pink/barriers/barrier.py:196-198; not a draw to compare)
            if d.min() > 1e-3 and min(np.abs(sb.compute_jacobian(c)).max() for c in cfgs) > 1e-6:
                bars.append(sb)
        if r4.random() < 0.3:
            f = str(r4.choice(names))
            p0 = np.array([c.get_transform_frame_to_world(f).translation for c in cfgs])
            ax = int(r4.integers(0, 3))
            bars.append(PositionBarrier(f, indices=[ax], p_min=np.array([p0[:, ax].min() - float(r4.uniform(0.0, 0.05))]),
                                        gain=np.array([float(r4.uniform(5.0, 100.0))]), safe_displacement_gain=float(r4.choice([0.0, 1.0]))))
        if r4.random() < 0.4:
            k = int(r4.integers(max(1, n - 2), n + 1))  # a frame far down the chain: (almost) all joints upstream
            if k + (6 if ff else 0) >= 7:  # six equations need six coordinates upstream
                f, g_ = f"joint_{k}", float(r4.uniform(0.3, 1.0))
                cons = [[] for _ in range(B)]
                for b, cfg in enumerate(cfgs):
                    t = FrameTask(f, 1.0, 1.0, gain=g_)
                    t.set_target(cfg.get_transform_frame_to_world(f) * exp6(1e-4 * r4.normal(size=6)))
                    cons[b].append(t)
    kw = dict(barriers=bars or None)
    if cons is not None:
        kw["constraints"] = cons
    if os.environ.get("FUZZ_EXTRAS") and m.floating_base_velocity_limit is None:
        r3 = np.random.default_rng(sd + 2 * 10 ** 9)
        if r3.random() < 0.3:  # an explicit limit list with an AccelerationLimit on the joints behind the root
            from pink_amd.limits import AccelerationLimit, ConfigurationLimit, VelocityLimit

            a_max = np.r_[np.full(6 if ff else 0, np.inf), 10 ** r3.uniform(1.0, 3.0, size=n)]
            acc = AccelerationLimit(m, a_max)
            acc.set_last_integration(np.r_[np.zeros(6 if ff else 0), r3.normal(size=n)] * 0.3, dt)
            kw["limits"] = [ConfigurationLimit(m, float(r3.uniform(0.3, 1.0))), VelocityLimit(m), acc]
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
        if host_err == dev_err:
            return "same failure", 0.0
        # one route reports "not positive definite" where the other solves: legitimate only where H is singular to working
        # precision (FrameTasks alone on more coordinates than rows, damping 1e-12: the pivot's sign is round-off; quadprog
        # raises there as well)
        try:
            cmax = max(np.linalg.cond(pink_amd.build_ik(cfgs[b], tasks[b], dt, **_kw_of(kw, b)).P) for b in range(B))
        except Exception:  # noqa: BLE001
            cmax = 0.0
        if cmax > 1e15:
            return "same failure", 0.0
        return f"host {host_err} / device {dev_err} (cond(H) up to {cmax:.1e})", 0.0
    # two evaluations of the same QP agree to cond(H) eps: the tolerance follows the conditioning of each instance
    cond = np.array([np.linalg.cond(pink_amd.build_ik(cfgs[b], tasks[b], dt, **_kw_of(kw, b)).P) for b in range(B)])
    rel = np.abs(V_dev - V_host).max(axis=1) / np.maximum(1.0, np.abs(V_host).max(axis=1))
    err = float((rel / np.maximum(1.0, 1e4 * cond * np.finfo(float).eps / 1e-8)).max())  # (compared with 1e-8)
    if os.environ.get("FUZZ_VERBOSE"):
        print(dict(ff=ff, n=n, frames=frames, B=B, dt=dt, scale=scale, bars=len(bars), fb=m.floating_base_velocity_limit is not None,
                   ntasks=len(tasks[0])), "per-instance rel err", rel, "cond", cond, "max |V_host|", np.abs(V_host).max(axis=1), "max |V_dev|", np.abs(V_dev).max(axis=1))
        w = int(np.argmax(rel))
        np.set_printoptions(precision=4, linewidth=200)
        print("worst instance", w, "tasks", [type(t).__name__ for t in tasks[w]], "limits", [type(l).__name__ for l in (kw.get("limits") or [])])
        print("V_host", V_host[w]); print("V_dev - V_host", V_dev[w] - V_host[w])
        pr = pink_amd.build_ik(cfgs[w], tasks[w], dt, **_kw_of(kw, w))
        for name, V in (("host", V_host[w]), ("dev", V_dev[w])):
            dq = V * dt
            print(name, "objective", 0.5 * dq @ pr.P @ dq + pr.q @ dq, "max G dq - h", None if pr.G is None else float((pr.G @ dq - pr.h).max()),
                  "active rows", None if pr.G is None else np.nonzero(pr.G @ dq - pr.h > -1e-9)[0])
    return None, err


def main():
    if os.environ.get("FUZZ_EMU"):
        import ctypes

        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import EmuSolver  # noqa: E402
        from pink_amd._lib import Desc, Problem, Result
        from pink_amd.runtime import set_default_solver

        lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libpinkemu.so"))
        lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
        lib.pinkhip_emu_stack_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.c_void_p, ctypes.c_void_p]
        lib.pinkhip_emu_last_error.restype = ctypes.c_char_p
        lib.pinkhip_emu_frame_task_host.argtypes = [ctypes.c_longlong, ctypes.c_int] + [ctypes.c_void_p] * 5
        set_default_solver(EmuSolver(lib))
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
