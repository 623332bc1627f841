#!/usr/bin/env python3
"""Five launches of the solve kernel on the headline shape in one input regime (tight | kinematic | tracking), for PMC
passes:  rocprofv3 --pmc ... -- python scripts/solve_regime_once.py tracking"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pink_amd import synthetic  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402

regime = sys.argv[1] if len(sys.argv) > 1 else "tight"
cfg = sys.argv[2] if len(sys.argv) > 2 else "draco3"
if regime == "tracking":
    t = synthetic.make_terms(cfg, 65536, bounds="kinematic", jacobians="kinematic", error_scale=0.02)
else:
    t = synthetic.make_terms(cfg, 65536, bounds=regime, jacobians="dense" if regime == "tight" else "kinematic")
s = BatchSolver(0)
dev = s.upload(synthetic.pack(t))
for _ in range(5):
    s.solve_device(dev)
s.sync()
dev.free()
s.close()
