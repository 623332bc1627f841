"""Throughput of the device-resident closed loop (FK -> frame tasks -> limits -> QP -> integrate)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pink_amd import Configuration, build_chain
from pink_amd.batch_solver import BatchSolver
from pink_amd.lie import SE3
from pink_amd.rollout import DeviceRollout, pose12

s = BatchSolver(0)
out = {}
cases = (("arm6", build_chain(6), ["tool0"]),
         ("arm12", build_chain(12, seed=3), ["tool0", "joint_6"]),
         ("floating_base_nv30", build_chain(24, free_flyer=True, seed=2), ["tool0", "joint_8", "joint_16", "joint_20"]))
# BASELINE config 4's shape in closed loop: nv = 50 (free flyer + 44 joints), 4 FrameTasks + PostureTask + 2 PositionBarriers
# (x, y, z lower bounds each: 6 barrier rows, formed on chip)
cases = cases + (("jvrc_shape_nv50_2barriers", build_chain(44, free_flyer=True, seed=4), ["tool0", "joint_10", "joint_20", "joint_30"]),)
only = os.environ.get("ROLLOUT_ONLY")
for label, model, frames in [c for c in cases if not only or c[0] == only]:
    B = 65536
    rng = np.random.default_rng(1)
    q0 = np.tile(model.neutral(), (B, 1))
    for j in model.joints:
        if j.kind != "free_flyer":
            q0[:, j.idx_q] = rng.uniform(-0.8, 0.8, size=B)
    cfg = Configuration(model, q0[0])
    specs = [(f, 1.0, 1.0 if i == 0 else 0.0, 1.0, 1e-3) for i, f in enumerate(frames)]
    T = None
    bars = []
    if "barriers" in label:
        from pink_amd.barriers import PositionBarrier
        for f in frames[:2]:  # floors 2 cm below where the frames start (min over the batch is far below: rarely active)
            p = np.array([Configuration(model, q0[b]).get_transform_frame_to_world(f).translation for b in range(64)])
            bars.append(PositionBarrier(f, p_min=p.min(axis=0) - 0.02, gain=np.array([100.0] * 3), safe_displacement_gain=1.0))
    for mode in (("kernel",) if bars else ("kernel", True)):  # the whole step in one kernel / step kernel + solve
        ro = DeviceRollout(s, model, q0, specs, 5e-3, posture_cost=1e-1, fused=mode, position_barriers=bars)
        if T is None:  # targets: each robot's own initial frame poses displaced by a few centimetres
            ro.step(); s.sync()
            T = ro.frame_poses()
            T[:, :, 9:12] += 0.05 * rng.normal(size=(B, len(frames), 3))
        ro.set_targets(T)
        ro.run(5)
        steps = 50
        s.timer_start()
        for _ in range(steps):
            ro.step()
        ms = s.timer_stop() / steps
        _, st, it = ro.last_step()
        name = label + ("_one_kernel" if ro.fused == "kernel" else "_two_launches")
        print(f"{name}: nv={model.nv} B={B}: {ms:.3f} ms per closed-loop step -> {B/ms/1e3:.1f} M robot-steps/s; failed={(st!=0).sum()} qp iters mean {it.mean():.2f}")
        out[name] = dict(nv=model.nv, B=B, ms_per_step=ms, robot_steps_per_s=B / (ms * 1e-3), qp_iters_mean=float(it.mean()),
                         failed=int((st != 0).sum()), frame_tasks=len(frames), launches_per_step=1 if ro.fused == "kernel" else 2, barrier_rows=ro.md)
        ro.free()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rollout_bench.json"), "w"), indent=1)
