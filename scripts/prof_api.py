#!/usr/bin/env python3
"""Where a `solve_ik_batch(configurations, tasks, dt)` call spends its time on the GPU box (4 096 six-dof arms,
device kinematics): per-call wall time and a cProfile listing."""
import time, numpy as np, cProfile, pstats, sys
sys.path.insert(0,'.')
from pink_amd import Configuration, FrameTask, PostureTask, build_chain, solve_ik_batch
from pink_amd.lie import SE3
m = build_chain(6); rng=np.random.default_rng(2); cfgs=[]; tasks=[]
B=4096
for _ in range(B):
    q=m.neutral()
    for j in m.joints: q[j.idx_q]=rng.uniform(-0.9,0.9)
    cfg=Configuration(m,q)
    t=FrameTask("tool0",1.0,1.0,lm_damping=1.0)
    t.set_target(cfg.get_transform_frame_to_world("tool0")*SE3(np.eye(3),0.05*rng.normal(size=3)))
    p=PostureTask(cost=1e-3); p.set_target(m.neutral())
    cfgs.append(cfg); tasks.append([t,p])
dt=1/200.
solve_ik_batch(cfgs,tasks,dt)
ts=[]
for _ in range(5):
    t0=time.perf_counter(); solve_ik_batch(cfgs,tasks,dt); ts.append(time.perf_counter()-t0)
print("ms per call", [round(x*1e3,2) for x in ts])
pr=cProfile.Profile(); pr.enable(); solve_ik_batch(cfgs,tasks,dt); pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(25)
