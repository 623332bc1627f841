#!/usr/bin/env python3
"""The stack-only kernel alone, warmed: 50 untimed launches, then 200 launches of ik_stack_mfma_kernel on the bench's
headline batch -- for a dedicated `rocprofv3 --kernel-trace --stats` whose MEAN is the figure DESIGN.md / README quote
(round-5 review: the line quoted a warmed HIP-event best case, the all-configs profile an unwarmed mean)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pink_amd import synthetic  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402

s = BatchSolver(0)
dev = s.upload(synthetic.pack(synthetic.make_terms("draco3", 65536, bounds="tight")))
for _ in range(50):
    s.stack_device(dev)
s.sync()
s.timer_start()
for _ in range(200):
    s.stack_device(dev)
ms = s.timer_stop() / 200
print(f"ik_stack_mfma_kernel draco3 B=65536: {ms:.4f} ms per launch (HIP events over 200 launches after 50 warm-up launches)")
dev.free()
s.close()
