"""Which robots of the closed-loop bench leave the tableau path, step by step (development).
  PINKHIP_LIBRARY=... python scripts/gpu_rollout_handover.py [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pink_amd import build_chain
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
s = BatchSolver(0)
ro = DeviceRollout(s, model, q0, specs, 5e-3, posture_cost=1e-1, fused="kernel")
ro.step(); s.sync()
T = ro.frame_poses()
T[:, :, 9:12] += 0.05 * rng.normal(size=(B, len(frames), 3))
ro.set_targets(T)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    qb = ro.configurations().copy()
    ro.step(); s.sync()
    _, st, it = ro.last_step()
    path = ro.last_path.astype(np.int64)
    idx = np.nonzero(path)[0]
    bad = np.nonzero(st != 0)[0]
    if bad.size: print("   status != 0:", bad[:8].tolist(), st[bad[:8]].tolist(), it[bad[:8]].tolist())
    print(f"step {k}: iters mean {it.mean():.2f} max {it.max()} paths {np.bincount(path, minlength=4).tolist()} handover robots {idx[:8].tolist()} their iters {it[idx[:8]].tolist()}")
    if idx.size and k >= 6:
        np.savez("gpurun_out/handover_robot.npz", q=qb[idx[:4]], T=T[idx[:4]], idx=idx[:4])
ro.free()
