"""Which instances of a synthetic batch left the tableau path on the GPU, and with how many trips (development).
  python scripts/gpu_paths.py lib.so config bounds B"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pink_amd import _lib, synthetic
from pink_amd.batch_solver import BatchSolver

lib, name, bounds, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
terms = synthetic.make_terms(name, B, bounds=bounds)
s = BatchSolver(0, library=_lib.load_library(os.path.abspath(lib)))
r = s.solve(synthetic.pack(terms))
idx = np.nonzero(r.path != 0)[0]
print("paths", np.bincount(r.path), "status", np.bincount(r.status))
print("handover idx", idx[:40].tolist())
print("iters of those", r.iters[idx[:40]].tolist())
bad = np.nonzero(r.status != 0)[0]
print("status != 0 idx", bad[:40].tolist())
print("their iters", r.iters[bad[:40]].tolist())
print("iters mean tableau", r.iters[r.path == 0].mean())
