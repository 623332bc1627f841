#!/usr/bin/env python3
"""A/B timing of kernel variants: every libdev_*.so given on the command line (make DEV=1 EXTRA=... OUT=...)
runs the headline batch (draco3, B = 65 536, tight bounds); prints kernel ms (min / median of repeats),
iteration mean and the error against the C oracle on a sample."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import c_oracle  # noqa: E402
from pink_amd import _lib, synthetic  # noqa: E402
from pink_amd.batch_solver import BatchSolver  # noqa: E402


def main():
    cfg = os.environ.get("AB_CONFIG", "draco3")
    if cfg == "custom":  # AB_NV / AB_NB: fixed-base robot of AB_NV joints, two FrameTasks + posture, AB_NB barrier rows
        synthetic.CONFIGS["custom"] = dict(config_id=20, nv=int(os.environ.get("AB_NV", "12")), root_nv=0, dt=5e-3, frame_costs=[(1.0, 1.0), (1.0, 1.0)],
                                           frame_lm=0.0, posture_cost=1e-1, n_barriers=int(os.environ.get("AB_NB", "0")))
    B = int(os.environ.get("AB_BATCH", "65536"))
    if os.environ.get("AB_BOUNDS") == "tracking":  # bench.py's tracking_small_errors regime
        terms = synthetic.make_terms(cfg, B, bounds="kinematic", jacobians="kinematic", error_scale=0.02)
    else:
        terms = synthetic.make_terms(cfg, B, bounds=os.environ.get("AB_BOUNDS", "tight"), jacobians="dense")
    batch = synthetic.pack(terms)
    n = 2048
    ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms.slice(0, n)), nthreads=16)
    for path in sys.argv[1:]:
        s = BatchSolver(0, library=_lib.load_library(os.path.abspath(path)))
        dev = s.upload(batch)
        for _ in range(3):
            s.solve_device(dev)
        s.sync()
        ms = []
        for _ in range(5):
            s.timer_start()
            for _ in range(20):
                s.solve_device(dev)
            ms.append(s.timer_stop() / 20)
        r = s.download(dev)
        err = float(np.abs(r.dq[:n] - ref["dq"]).max())
        print(f"{os.path.basename(path):28s} min {min(ms):.4f} ms  median {statistics.median(ms):.4f} ms  "
              f"{B / min(ms) / 1e3:.1f} M/s  iters {r.iters.mean():.2f}  failed {(r.status != 0).sum()}  err {err:.2e}  "
              f"paths {' '.join(f'{k}={v:.3f}' for k, v in r.path_fractions().items() if v)}", flush=True)
        dev.free()
        s.close()


if __name__ == "__main__":
    main()
