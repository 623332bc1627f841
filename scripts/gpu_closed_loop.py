"""bench.py's closed-loop figures alone (development; PINKHIP_LIBRARY picks the library, ONLY= a substring of the labels):
  PINKHIP_LIBRARY=$PWD/pink_amd/csrc/libdev.so python scripts/gpu_closed_loop.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402

s = BatchSolver(0)
out = bench.closed_loop_figures(s, int(os.environ.get("B", "65536")), only=os.environ.get("ONLY"))
for k, v in out.items():
    after = " / ".join(f"{a['ms']:.3f} ({a['qp_iters_mean']:.2f})" for a in v["steps_after_a_target_move"])
    print(f"{k:48s} {v['ms_per_step']:.4f} ms  iters {v['qp_iters_mean']:.3f}  handover {v['handover_frac']:.5f}  after a move: {after}")
