"""Trip-count study on the CPU (round 3): how many factorisations / pivots do block methods need
on the BASELINE box-only batches, against the ~29 single-constraint steps of Goldfarb-Idnani?

  python scripts/trip_count_study.py [B]

Methods (all reach the unique minimiser; the count is what differs):
  gi      -- the C oracle's Goldfarb-Idnani (quadprog's rule): adds + drops
  bpp     -- block principal pivoting (Kostreva; Judice & Pires; Kim & Park's backup rule): every infeasible
             coordinate changes sides at once; one solve with H_FF per iteration; single (largest index) pivot
             after `p` non-improving iterations
  bpp+x0  -- the same, started from the bounds violated by the unconstrained minimum
Costs are reported as iterations (= factorisations of H_FF) per QP and as the maximum over pairs of neighbouring
QPs (the two QPs of a wavefront run in lock step).
"""

from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import c_oracle  # noqa: E402
from pink_amd import synthetic  # noqa: E402


def bpp(H, c, lb, ub, start_from_x0=True, max_iter=200, pbar=3, tol=1e-12):
    """Block principal pivoting for min 1/2 x'Hx + c'x, lb <= x <= ub.  Returns (x, iterations, fallbacks)."""
    n = c.size
    state = np.zeros(n, dtype=int)  # 0 free, -1 at lb, +1 at ub
    if start_from_x0:
        x0 = np.linalg.solve(H, -c)
        state = np.where(x0 < lb, -1, np.where(x0 > ub, 1, 0))
        its = 1
    else:
        its = 0
    ninf_best, p = n + 1, pbar
    fallbacks = 0
    while its < max_iter:
        its += 1
        F = state == 0
        x = np.where(state < 0, lb, np.where(state > 0, ub, 0.0))
        if F.any():
            rhs = -(c[F] + H[np.ix_(F, ~F)] @ x[~F])
            x[F] = np.linalg.solve(H[np.ix_(F, F)], rhs)
        g = H @ x + c  # multiplier: g >= 0 at lb, g <= 0 at ub
        scale = 1.0 + np.abs(np.where(np.isfinite(lb), lb, 0.0)) + np.abs(np.where(np.isfinite(ub), ub, 0.0))
        viol_p_lo = F & (x < lb - tol * scale)
        viol_p_hi = F & (x > ub + tol * scale)
        gs = tol * (1.0 + np.abs(g))
        viol_d = ((state < 0) & (g < -gs)) | ((state > 0) & (g > gs))
        ninf = int(viol_p_lo.sum() + viol_p_hi.sum() + viol_d.sum())
        if ninf == 0:
            return x, its, fallbacks
        if ninf < ninf_best:
            ninf_best, p = ninf, pbar
            full = True
        elif p > 0:
            p -= 1
            full = True
        else:
            full = False
            fallbacks += 1
        if full:
            state = np.where(viol_p_lo, -1, np.where(viol_p_hi, 1, np.where(viol_d, 0, state)))
        else:  # Murty: only the infeasible coordinate of largest index
            k = int(np.max(np.nonzero(viol_p_lo | viol_p_hi | viol_d)[0]))
            state[k] = -1 if viol_p_lo[k] else (1 if viol_p_hi[k] else 0)
    return x, its, fallbacks


def study(name, B, **kw):
    terms = synthetic.make_terms(name, B, **kw)
    batch = synthetic.pack(terms)
    pf = synthetic.pink_form(terms)
    ref = c_oracle.solve_ik_batch(**pf, want_Hc=True, nthreads=8)
    H, c = ref["H"], ref["c"]
    lb, ub = batch.lb, batch.ub
    out = {}
    for label, x0 in (("bpp", False), ("bpp+x0", True)):
        its = np.zeros(B, int)
        fb = np.zeros(B, int)
        err = 0.0
        for b in range(B):
            x, its[b], fb[b] = bpp(H[b], c[b], lb[b], ub[b], start_from_x0=x0)
            err = max(err, float(np.abs(x - ref["dq"][b]).max()))
        out[label] = (its, fb, err)
    gi = ref["iters"]
    nact = ((np.abs(ref["dq"] - lb) < 1e-12) | (np.abs(ref["dq"] - ub) < 1e-12)).sum(axis=1)
    print(f"== {name} {kw}  B={B}  active bounds {nact.mean():.1f}")
    print(f"   gi      steps mean {gi.mean():6.2f}  pair-max mean {np.maximum(gi[0::2], gi[1::2]).mean():6.2f}  max {gi.max()}")
    for label, (its, fb, err) in out.items():
        pm = np.maximum(its[0::2], its[1::2]).mean()
        hist = np.bincount(its)
        print(f"   {label:7s} iters mean {its.mean():6.2f}  pair-max mean {pm:6.2f}  max {its.max()}  fallbacks {fb.sum()}  "
              f"max|x - gi| {err:.1e}  hist {dict((i, int(v)) for i, v in enumerate(hist) if v)}")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    study("draco3", B, bounds="tight")
    study("draco3", B, bounds="kinematic")
    study("draco3", B, bounds="kinematic", error_scale=0.02)
    study("ur5", B, bounds="tight")
