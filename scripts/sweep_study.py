"""Pivot-count study for a *sweep tableau* formulation of the box-constrained IK-QP (round 3).

State: T = SWP_F(H), the symmetric sweep of H on the free set F:
    T_FF = -H_FF^-1,  T_SF = H_SF H_FF^-1,  T_SS = H_SS - H_SF H_FF^-1 H_FS
One pivot (sweep k in / out) is a rank-1 update of T: NV broadcast-FMAs per lane in the row-per-lane mapping, no LDS.
x_F and the multipliers g_S follow from one product T v with v = (c_F, -b_S).

Counted here, per QP: pivots (after the start) for
  gi        dual active set (Goldfarb-Idnani logic on the tableau) from the unconstrained minimum, metric-weighted rule
  bpp       block principal pivoting from the all-free start (Kim & Park backup rule), one pivot per flip
  pg<k>+bpp active-set guess from k projected-gradient steps, then bpp; the start sweeps only the guessed-free set
"""

from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import c_oracle  # noqa: E402
from pink_amd import synthetic  # noqa: E402


class Tableau:
    def __init__(self, H):
        self.T = H.copy()
        self.n = H.shape[0]
        self.free = np.zeros(self.n, bool)
        self.pivots = 0

    def pivot(self, k):
        """Sweep k in (if fixed) or out (if free): same rank-1 form, sign of the column differs."""
        T = self.T
        p = T[k, k]
        col = T[:, k].copy()
        sgn = -1.0 if self.free[k] else 1.0  # sweep: B_ik = +A_ik/p ; reverse: -A_ik/p
        T -= np.outer(col, col) / p
        T[:, k] = sgn * col / p
        T[k, :] = sgn * col / p
        T[k, k] = -1.0 / p
        self.free[k] = ~self.free[k]
        self.pivots += 1

    def solve(self, c, bound):
        """x (free: minimiser given the fixed ones; fixed: bound) and g = Hx + c on the fixed ones (0 on free)."""
        F = self.free
        v = np.where(F, c, -bound)
        Tv = self.T @ v
        x = np.where(F, Tv, bound)
        g = np.where(F, 0.0, c - Tv)
        return x, g


def infeasible_sets(tab, x, g, state, lb, ub, tol=1e-12):
    F = tab.free
    sc = 1.0 + np.abs(np.where(np.isfinite(lb), lb, 0)) + np.abs(np.where(np.isfinite(ub), ub, 0))
    lo = F & (x < lb - tol * sc)
    hi = F & (x > ub + tol * sc)
    gs = tol * (1.0 + np.abs(g))
    du = (~F) & (((state < 0) & (g < -gs)) | ((state > 0) & (g > gs)))
    return lo, hi, du


def run_bpp(H, c, lb, ub, guess_state=None, pbar=3, max_iter=300):
    n = c.size
    tab = Tableau(H)
    state = np.zeros(n, int) if guess_state is None else guess_state.copy()  # 0 free, -1 lb, +1 ub
    start = 0
    for k in range(n):
        if state[k] == 0:
            tab.pivot(k)
            start += 1
    tab.pivots = 0
    ninf_best, p, its, fb = n + 1, pbar, 0, 0
    while its < max_iter:
        its += 1
        bound = np.where(state < 0, lb, np.where(state > 0, ub, 0.0))
        x, g = tab.solve(c, bound)
        lo, hi, du = infeasible_sets(tab, x, g, state, lb, ub)
        ninf = int(lo.sum() + hi.sum() + du.sum())
        if ninf == 0:
            return x, start, tab.pivots, its, fb
        if ninf < ninf_best:
            ninf_best, p = ninf, pbar
            full = True
        elif p > 0:
            p -= 1
            full = True
        else:
            full = False
            fb += 1
        idx = np.nonzero(lo | hi | du)[0]
        if not full:
            idx = idx[-1:]
        for k in idx:
            tab.pivot(k)
            state[k] = -1 if lo[k] else (1 if hi[k] else 0)
    return x, start, tab.pivots, its, fb


def pg_guess(H, c, lb, ub, k, precond=True):
    """k projected (diagonally preconditioned) gradient steps from x = clip(0); returns the active-set guess."""
    n = c.size
    d = np.diag(H).copy()
    # step from a bound on the largest eigenvalue of D^-1/2 H D^-1/2 (a few power iterations would do on the GPU;
    # here: the exact value, to see what the best case buys)
    s = 1.0 / np.sqrt(d)
    lam = np.linalg.eigvalsh(H * s[:, None] * s[None, :])[-1]
    x = np.clip(np.zeros(n), lb, ub)
    for _ in range(k):
        g = H @ x + c
        x = np.clip(x - g / (d * lam), lb, ub)
    g = H @ x + c
    state = np.where((x <= lb) & (g > 0), -1, np.where((x >= ub) & (g < 0), 1, 0))
    return state


def run_gi(H, c, lb, ub, max_iter=500, want_state=False):
    """Dual active set on the tableau from the unconstrained minimum; entering rule: violation / sqrt(Z_ii)."""
    n = c.size
    tab = Tableau(H)
    for k in range(n):
        tab.pivot(k)
    tab.pivots = 0
    state = np.zeros(n, int)
    x = tab.T @ c  # = -H^-1 c
    u = np.zeros(n)
    trips = 0
    pending = None
    tol = 1e-13 * max(1.0, n / 8)
    while trips < max_iter:
        if pending is None:
            zd = -np.diag(tab.T)
            slo, sup = x - lb, ub - x
            klo = np.where(tab.free & (slo < -tol * (1 + np.abs(lb))), slo / np.sqrt(np.where(zd > 0, zd, 1)), 0.0)
            kup = np.where(tab.free & (sup < -tol * (1 + np.abs(ub))), sup / np.sqrt(np.where(zd > 0, zd, 1)), 0.0)
            if min(klo.min(), kup.min()) >= 0:
                if want_state:
                    return x, trips, tab, state
                return x, trips
            if klo.min() <= kup.min():
                i, s = int(np.argmin(klo)), -1
            else:
                i, s = int(np.argmin(kup)), 1
            pending = (i, s)
            uplus = 0.0
        trips += 1
        i, s = pending
        col = tab.T[:, i]
        p = col[i]
        t = col / p  # dx_k = t_k tau (free), dg_j = -t_j tau (fixed)
        tau_full = (ub[i] if s > 0 else lb[i]) - x[i]
        # du_j = -sigma_j dg_j = sigma_j t_j tau
        rate = np.where(~tab.free, state * t * tau_full, 0.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            theta = np.where(rate < 0, u / (-rate), np.inf)
        j = int(np.argmin(theta))
        th = min(1.0, theta[j])
        x = np.where(tab.free, x + th * t * tau_full, x)
        u = np.where(~tab.free, u + th * rate, u)
        uplus += th * (s * tau_full / p)  # -sigma_i dg_i = sigma_i tau / p ... p < 0
        if th >= 1.0:
            tab.pivot(i)
            state[i] = s
            u[i] = uplus
            x[i] = ub[i] if s > 0 else lb[i]
            pending = None
        else:
            tab.pivot(j)
            state[j] = 0
            u[j] = 0.0
    return x, trips


def study(name, B, **kw):
    terms = synthetic.make_terms(name, B, **kw)
    batch = synthetic.pack(terms)
    pf = synthetic.pink_form(terms)
    ref = c_oracle.solve_ik_batch(**pf, want_Hc=True, nthreads=8)
    H, c = ref["H"], ref["c"]
    lb, ub = batch.lb, batch.ub
    n = c.shape[1]
    print(f"== {name} {kw} B={B}: oracle GI steps {ref['iters'].mean():.2f}")
    tr = np.zeros(B, int)
    err = 0.0
    for b in range(B):
        x, tr[b] = run_gi(H[b], c[b], lb[b], ub[b])
        err = max(err, np.abs(x - ref["dq"][b]).max())
    print(f"   gi(tableau)  start {n} sweeps + trips {tr.mean():6.2f}  pair-max {np.maximum(tr[0::2], tr[1::2]).mean():6.2f}  err {err:.1e}")
    for label, k in (("bpp", None), ("pg2+bpp", 2), ("pg5+bpp", 5), ("pg10+bpp", 10), ("pg20+bpp", 20)):
        st = np.zeros(B, int)
        pv = np.zeros(B, int)
        its = np.zeros(B, int)
        fb = np.zeros(B, int)
        err = 0.0
        for b in range(B):
            guess = None if k is None else pg_guess(H[b], c[b], lb[b], ub[b], k)
            x, st[b], pv[b], its[b], fb[b] = run_bpp(H[b], c[b], lb[b], ub[b], guess)
            err = max(err, np.abs(x - ref["dq"][b]).max())
        tot = pv
        print(f"   {label:10s}  start {st.mean():5.1f} sweeps + pivots {pv.mean():6.2f} (pair-max {np.maximum(tot[0::2], tot[1::2]).mean():6.2f}, max {pv.max()})"
              f"  iterations {its.mean():5.2f} (pair-max {np.maximum(its[0::2], its[1::2]).mean():5.2f}, max {its.max()})  fallbacks {fb.sum()}  err {err:.1e}")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    study("draco3", B, bounds="tight")
    study("draco3", B, bounds="kinematic")
    study("draco3", B, bounds="kinematic", error_scale=0.02)
    study("ur5", B, bounds="tight")
