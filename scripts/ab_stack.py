#!/usr/bin/env python3
"""A/B timing of the stack-only kernel for library variants: draco3 / jvrc / ur5 at B = 65 536 (+ ur5 at 4096, 1M);
the (H, c) of every later library are compared with the first one's (same arithmetic order: expected identical)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pink_amd import _lib, synthetic  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402

cases = [("draco3", 65536), ("jvrc", 65536), ("ur5", 65536), ("ur5", 4096), ("ur5", 1 << 20)]
if os.environ.get("AB_STACK_ONLY"):
    cases = [c for c in cases if c[0] in os.environ["AB_STACK_ONLY"].split(",")]
batches = {c: synthetic.pack(synthetic.make_terms(c[0], c[1], bounds="tight")) for c in cases}
first = {}
for path in sys.argv[1:]:
    s = BatchSolver(0, library=_lib.load_library(os.path.abspath(path)))
    for c in cases:
        b = batches[c]
        dev = s.upload(b)
        for _ in range(5):
            s.stack_device(dev)
        s.sync()
        best = 1e9
        for _ in range(5):
            s.timer_start()
            for _ in range(20):
                s.stack_device(dev)
            best = min(best, s.timer_stop() / 20)
        gbs = b.bytes_per_stack() * b.B / (best * 1e-3) / 1e9
        H, cv = s.download_stack(dev)
        n = min(b.B, 4096)
        if c not in first:
            first[c] = (H[:n].copy(), cv[:n].copy())
            diff = ""
        else:
            diff = f"  max|dH| {np.abs(H[:n] - first[c][0]).max():.1e} max|dc| {np.abs(cv[:n] - first[c][1]).max():.1e}"
        del H, cv
        print(f"{os.path.basename(path):22s} {c[0]:7s} B={c[1]:8d}  {best * 1e3:9.2f} us  {gbs:7.0f} GB/s  {gbs / 80:5.1f} % of 8 TB/s" + diff, flush=True)
        dev.free()
    s.close()
