"""Quick timing of the solve kernel on resident batches (draco3 tight/kinematic, jvrc, ur5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pink_amd import synthetic
from pink_amd.batch_solver import BatchSolver
s = BatchSolver(0)
cases = [("draco3", 65536, "tight"), ("draco3", 65536, "kinematic"), ("jvrc", 32768, "tight"), ("ur5", 65536, "tight")]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
for name, B, bounds in cases:
    t = synthetic.make_terms(name, B, bounds=bounds, jacobians="dense" if bounds == "tight" else "kinematic")
    pk = synthetic.pack(t)
    if os.environ.get("UNBOUNDED"):
        pk.lb[:] = -np.inf; pk.ub[:] = np.inf
    dev = s.upload(pk)
    s.solve_device(dev); s.sync()
    best = 1e9
    for rep in range(3):
        s.timer_start()
        for _ in range(5): s.solve_device(dev)
        best = min(best, s.timer_stop() / 5)
    r = s.download(dev)
    print(f"  {name:7s} {bounds:9s} B={B}: {best:.3f} ms  {B/best/1e3:.2f} M/s  bad={(r.status!=0).sum()} it={r.iters.mean():.1f} chk={np.abs(r.dq).sum():.12e}")
    dev.free()
