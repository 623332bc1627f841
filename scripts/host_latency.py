"""Host-call latency of small batches through pinkhip_solve_host (UR5 shape, B = 1 .. 256): Python call, raw C-ABI call.
    python scripts/host_latency.py [library ...]      (GPU box; default: the product library)"""
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import numpy as np
from pink_amd import _lib, synthetic
from pink_amd.batch_solver import BatchSolver
from oracle import c_oracle
for path in (sys.argv[1:] or [os.path.join("pink_amd", "csrc", "libpinkhip.so")]):  # default: the product library
    s = BatchSolver(0, library=_lib.load_library(os.path.abspath(path)))
    for B in (1, 16, 64, 256):
        terms = synthetic.make_terms("ur5", B, bounds="kinematic", jacobians="kinematic")
        one = synthetic.pack(terms)
        ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms), nthreads=1)
        for _ in range(20): r = s.solve(one)
        tl = []
        for _ in range(300):
            t0 = time.perf_counter(); r = s.solve(one); tl.append(time.perf_counter() - t0)
        err = float(np.abs(r.dq - ref["dq"]).max())
        import ctypes
        from pink_amd.batch_solver import PackedArgs
        from pink_amd._lib import Result
        a = PackedArgs(one, 0); pr = a.host_problem(); rr = Result()
        rr.dq, rr.status, rr.iters = r.dq.ctypes.data, r.status.ctypes.data, r.iters.ctypes.data
        tc = []
        for _ in range(300):
            t0 = time.perf_counter(); s._lib.pinkhip_solve_host(s._h, ctypes.byref(a.desc), ctypes.byref(pr), ctypes.byref(rr)); tc.append(time.perf_counter() - t0)
        print(f"   raw C call median {statistics.median(tc)*1e6:7.1f} us")
        print(f"{os.path.basename(path):20s} B={B:4d} median {statistics.median(tl)*1e6:7.1f} us  p90 {sorted(tl)[270]*1e6:7.1f} us  err {err:.1e} status {r.status.max()} iters {r.iters.mean():.1f}")
    s.close()
