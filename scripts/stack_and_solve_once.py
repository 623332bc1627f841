#!/usr/bin/env python3
"""Five launches of the solve kernel and of the stack kernel on the bench's headline batch (for the PMC passes:
the bench itself would add its spin-up launches and other configurations to the per-kernel averages)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pink_amd import synthetic  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402

s = BatchSolver(0)
dev = s.upload(synthetic.pack(synthetic.make_terms("draco3", 65536, bounds="tight")))
for _ in range(5):
    s.solve_device(dev)
for _ in range(5):
    s.stack_device(dev)
s.sync()
dev.free()
s.close()
