#!/usr/bin/env python3
"""One-off wide fuzz on the GPU box: tests/parity_suite.fuzz (status equality + velocity error against the C oracle) over
many more seeds than the test suite runs, for both stack + solve kernels.   python scripts/gpu_fuzz.py [first] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_suite as ps  # noqa: E402

from pink_amd.batch_solver import BatchSolver  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
s = BatchSolver(0)
for solver in os.environ.get("PINKHIP_FUZZ_KERNELS", "sweep,sweepx,packed").split(","):  # (sweepx: the kernel with virtual dense rows where it is instantiated)
    os.environ["PINKHIP_SOLVER"] = solver
    t0 = time.time()
    n, bad = 0, []
    for sd in range(first, first + count):
        try:
            n += ps.fuzz(s, [sd])
            if sd % 2 == 0:  # the shapes with virtual dense rows: 25 .. 32 coordinates, up to eight dense rows
                n += ps.fuzz(s, [sd], nv_lo=25, nv_hi=33, md_hi=6)
            if sd % 4 == 0:  # the wide instantiations: up to 60 coordinates, up to 12 dense rows
                n += ps.fuzz(s, [sd], nv_lo=34, nv_hi=61, md_hi=13)
            if sd % 2 == 1:  # weakly regularised objectives
                n += ps.fuzz(s, [sd], ill=True)
            if sd % 4 == 1 and solver == "sweep":  # 33 / 34 coordinates behind unbounded leading ones: the instantiation that eliminates two of them
                n += ps.fuzz(s, [sd], nv_lo=33, nv_hi=35, free_lead=int(2 + sd // 4 % 5))
        except AssertionError as exc:
            bad.append(sd)
            print("  seed", sd, "->", str(exc)[:200], flush=True)
    print(f"  {len(bad)} failing seeds: {bad}", flush=True)
    k = ps.kkt_certificate(s, range(first, first + count // 8))
    print(f"  oracle verdicts \"inconsistent\" refuted by a certified point: {ps.REFUTED}; accepted on their certificate AND the exact-arithmetic anchor instead of on dq: {len(ps.CERTIFIED)} (anchor not computable: {len(ps.EXACT_UNSETTLED)})", flush=True)
    del ps.REFUTED[:], ps.CERTIFIED[:], ps.EXACT_UNSETTLED[:]
    print(f"{solver}: {count} seeds, {n} feasible instances within tolerance, statuses equal (but for the refuted verdicts); "
          f"KKT certificates {k}; {time.time() - t0:.1f} s", flush=True)
