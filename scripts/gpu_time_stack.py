"""Timing of the stack-only (build_ik P, q) kernel on resident batches; reports GB/s of algorithmic bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pink_amd import synthetic
from pink_amd.batch_solver import BatchSolver
s = BatchSolver(0)
for name, B in (("draco3", 65536), ("draco3", 524288), ("ur5", 1048576), ("jvrc", 65536)):
    t = synthetic.make_terms(name, B, bounds="tight", jacobians="dense")
    pk = synthetic.pack(t)
    dev = s.upload(pk)
    s.stack_device(dev); s.sync()
    best = 1e9
    for rep in range(3):
        s.timer_start()
        for _ in range(10): s.stack_device(dev)
        best = min(best, s.timer_stop() / 10)
    nv, Kd, K = pk.nv, pk.Kd, pk.K
    nbytes = 8 * (Kd * nv + K + nv * nv + nv) * B
    print(f"  {name:7s} B={B}: {best:.4f} ms  {nbytes/best/1e6:.0f} GB/s  ({B/best/1e3:.1f} M/s)")
    dev.free()
