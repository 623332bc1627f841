#!/usr/bin/env python3
"""How much of the dense-row instantiation's time is the dense rows themselves: the headline batch (no dense rows),
the same robot with two barrier rows, and with two barrier rows scaled so that they never become active."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pink_amd import _lib, synthetic  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402


def main():
    B = 65536
    synthetic.CONFIGS["draco3b"] = dict(synthetic.CONFIGS["draco3"], n_barriers=2, config_id=13)
    s = BatchSolver(0, library=_lib.load_library(os.path.abspath(sys.argv[1]))) if len(sys.argv) > 1 else BatchSolver(0)
    cases = []
    cases.append(("draco3 (md = 0)", synthetic.pack(synthetic.make_terms("draco3", B))))
    tb = synthetic.make_terms("draco3b", B)
    cases.append(("draco3 + 2 barriers", synthetic.pack(tb)))
    idle = synthetic.pack(tb)
    idle.hd = idle.hd + 1e3  # rows that never become active
    cases.append(("draco3 + 2 idle rows", idle))
    for name, batch in cases:
        dev = s.upload(batch)
        for _ in range(3):
            s.solve_device(dev)
        s.sync()
        s.timer_start()
        for _ in range(20):
            s.solve_device(dev)
        ms = s.timer_stop() / 20
        r = s.download(dev)
        print(f"{name:24s} {ms:.4f} ms  iters {r.iters.mean():.2f}  us per iteration of the launch {1e3 * ms / r.iters.mean():.2f}  failed {(r.status != 0).sum()}")
        dev.free()


if __name__ == "__main__":
    main()
