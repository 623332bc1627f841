"""cProfile of pink_amd.solve_ik_batch(ConfigurationBatch, ...) at the headline shape (GPU box)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pink_amd.batch_solver import BatchSolver
from pink_amd.runtime import set_default_solver

s = BatchSolver(0)
set_default_solver(s)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
print(bench.api_level_arrays(B))
from pink_amd import ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik_batch, Configuration
m = build_chain(24, free_flyer=True, seed=2)
frames = ["tool0", "joint_8", "joint_16", "joint_20"]
rng = np.random.default_rng(1)
q = np.tile(m.neutral(), (B, 1))
for j in m.joints:
    if j.kind != "free_flyer":
        q[:, j.idx_q] = rng.uniform(-0.8, 0.8, size=B)
tasks = []
ref = Configuration(m, q[0])
for k, f in enumerate(frames):
    t = FrameTask(f, 1.0, 1.0 if k == 0 else 0.0, lm_damping=1e-3)
    T0 = ref.get_transform_frame_to_world(f)
    t.set_target_poses(np.broadcast_to(T0.rotation, (B, 3, 3)), T0.translation + 0.05 * rng.normal(size=(B, 3)))
    tasks.append(t)
post = PostureTask(cost=1e-1); post.set_target(m.neutral()); tasks.append(post)
cfgs = ConfigurationBatch(m, q)
solve_ik_batch(cfgs, tasks, 5e-3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    solve_ik_batch(cfgs, tasks, 5e-3)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
