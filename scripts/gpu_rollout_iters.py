"""Distribution of the active-set trips per robot in the closed-loop bench with two position barriers (development).
  PINKHIP_LIBRARY=... python scripts/gpu_rollout_iters.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pink_amd import Configuration, build_chain
from pink_amd.barriers import PositionBarrier
from pink_amd.batch_solver import BatchSolver
from pink_amd.rollout import DeviceRollout

B = int(os.environ.get("B", "65536"))
model = build_chain(24, free_flyer=True, seed=2)
frames = ["tool0", "joint_8", "joint_16", "joint_20"]
rng = np.random.default_rng(1)
q0 = np.tile(model.neutral(), (B, 1))
for j in model.joints:
    if j.kind != "free_flyer":
        q0[:, j.idx_q] = rng.uniform(-0.8, 0.8, size=B)
specs = [(f, 1.0, 1.0 if i == 0 else 0.0, 1.0, 1e-3) for i, f in enumerate(frames)]
bars = []
for f in frames[:2]:
    p = np.array([Configuration(model, q0[b]).get_transform_frame_to_world(f).translation for b in range(64)])
    bars.append(PositionBarrier(f, p_min=p.min(axis=0) - 0.02, gain=np.array([100.0] * 3), safe_displacement_gain=1.0))
s = BatchSolver(0)
ro = DeviceRollout(s, model, q0, specs, 5e-3, posture_cost=1e-1, fused="kernel", position_barriers=bars)
ro.step(); s.sync()
T = ro.frame_poses()
T[:, :, 9:12] += 0.05 * rng.normal(size=(B, len(frames), 3))
ro.set_targets(T)
ro.run(5)
for k in range(31):
    qb = ro.configurations().copy() if k == 30 else None
    ro.step()
s.sync()
_, st, it = ro.last_step()
print("iters mean", it.mean(), "max", it.max(), "hist", np.bincount(np.minimum(it, 40))[:41].tolist())
big = np.nonzero(it > 20)[0]
print("robots with > 20 trips:", big.size, big[:10].tolist(), it[big[:10]].tolist(), "paths", np.bincount(ro.last_path.astype(np.int64), minlength=4).tolist())
if big.size:
    np.savez("gpurun_out/slow_robot.npz", q=qb[big[:4]], T=T[big[:4]], idx=big[:4])
ro.free()
