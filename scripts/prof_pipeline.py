"""Where a pipelined solve_ik_batch call on page-locked arrays spends its time (GPU box): the stages of
DeviceRollout.solve_pipelined timed separately -- uploads alone, kernels alone, both, with the results going home -- then
the whole call, and its cProfile."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pink_amd
from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik_batch
from pink_amd.batch_solver import BatchSolver
from pink_amd.runtime import set_default_solver
from pink_amd.sharding import shard_bounds

s = BatchSolver(0)
set_default_solver(s)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = build_chain(24, free_flyer=True, seed=2)
frames = ["tool0", "joint_8", "joint_16", "joint_20"]
rng = np.random.default_rng(1)
q = pink_amd.pinned_empty((B, m.nq)); q[:] = m.neutral()
for j in m.joints:
    if j.kind != "free_flyer":
        q[:, j.idx_q] = rng.uniform(-0.8, 0.8, size=B)
tasks = []
ref = Configuration(m, q[0])
for k, f in enumerate(frames):
    t = FrameTask(f, 1.0, 1.0 if k == 0 else 0.0, lm_damping=1e-3)
    T0 = ref.get_transform_frame_to_world(f)
    t.set_target_poses(np.broadcast_to(T0.rotation, (B, 3, 3)), T0.translation + 0.05 * rng.normal(size=(B, 3)), out=pink_amd.pinned_empty((B, 12)))
    tasks.append(t)
post = PostureTask(cost=1e-1); post.set_target(m.neutral()); tasks.append(post)
cfgs = ConfigurationBatch(m, q)
out = pink_amd.pinned_empty((B, m.nv))
solve_ik_batch(cfgs, tasks, 5e-3, out=out)
ro = next(iter(s._pinkhip_rollouts.values()))
tg = [t.target_poses for t in tasks[:4]]
nq, nv, n_chunks = m.nq, m.nv, 4
st, it = s.pinned_empty((B,), np.int32), s.pinned_empty((B,), np.int32)


def med(fn, n=7):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[len(ts) // 2]


def stage(h2d, kern, d2h, wait=True):
    def run():
        for c in range(n_chunks):
            lo, hi = shard_bounds(B, c, n_chunks)
            if h2d:
                s.put_async(ro.d_q + 8 * nq * lo, q[lo:hi])
                for f, t in enumerate(tg):
                    s.put_async(ro.d_Tt + 8 * 12 * (B * f + lo), t[lo:hi])
                if wait:
                    s.wait_copies()
            if kern:
                ro._one_kernel_step(False, lo, hi)
            if d2h:
                s.get_async(out[lo:hi], ro.d_dq + 8 * nv * lo)
                s.get_async(st[lo:hi], ro.d_status + 4 * lo)
                s.get_async(it[lo:hi], ro.d_iters + 4 * lo)
        s.sync()
    return run


ro.targets_per_frame = True
print(f"B = {B}: uploads alone {med(stage(True, False, False)):.3f} ms   kernels alone {med(stage(False, True, False)):.3f} ms   "
      f"downloads alone {med(stage(False, False, True)):.3f} ms")
print(f"uploads + kernels {med(stage(True, True, False)):.3f} ms   uploads + kernels + downloads {med(stage(True, True, True)):.3f} ms   "
      f"kernels + downloads {med(stage(False, True, True)):.3f} ms")
print(f"limit check on the device {med(lambda: ro._check_limits_device(q, True)):.3f} ms   whole call (page-locked arrays, out=) "
      f"{med(lambda: solve_ik_batch(cfgs, tasks, 5e-3, out=out)):.3f} ms")
qp = np.array(q); tp = [np.array(t) for t in tg]
tasks_p = []
for k, f in enumerate(frames):
    t = FrameTask(f, 1.0, 1.0 if k == 0 else 0.0, lm_damping=1e-3); t.target_poses = tp[k]; tasks_p.append(t)
tasks_p.append(post)
cb = ConfigurationBatch(m, qp)
print(f"whole call (pageable arrays) {med(lambda: solve_ik_batch(cb, tasks_p, 5e-3)):.3f} ms")
pr = cProfile.Profile(); pr.enable()  # (the whole call, Python included)
for _ in range(5):
    solve_ik_batch(cfgs, tasks, 5e-3, out=out)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# the split of the batch into ranges (pink_amd.rollout.PIPELINE_SPLIT): equal ranges against shrinking ones
import pink_amd.rollout as _R

for split in [(1, 1, 1, 1), (1, 1, 1, 1, 1, 1), (30, 27, 22, 13, 8), (40, 30, 20, 10), (18, 18, 18, 10), (1, 1, 1), (12, 18, 18, 16)]:
    _R.PIPELINE_SPLIT = tuple(float(v) for v in split)
    print(f"ranges {split}: whole call (page-locked) {med(lambda: solve_ik_batch(cfgs, tasks, 5e-3, out=out), 9):.3f} ms   "
          f"pageable {med(lambda: solve_ik_batch(cb, tasks_p, 5e-3), 5):.3f} ms")
