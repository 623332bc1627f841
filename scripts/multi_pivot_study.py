"""Pivot-count study on the CPU (round 6): several pivots per trip on the sweep tableau.

  python scripts/multi_pivot_study.py [B] [config] [bounds] [error_scale]

A NumPy model of ik_sweep.h's box-only iteration (same tableau, same entering rule, same ratio test) and of outer
loops that change SEVERAL indices per trip.  What is counted per QP: pivots (one pivot = column + NT broadcast-FMAs on
the GPU), outer iterations (one selection / feasibility pass each), single-pivot Goldfarb-Idnani trips.  The cost model
at the bottom turns the counts into VALU instructions of the <30,0,32> kernel.
"""

from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import c_oracle  # noqa: E402
from pink_amd import synthetic  # noqa: E402


class Tableau:
    """T swept on the free set; state 0 free, 1 at lb, 2 at ub; x coordinates, u multipliers (>= 0 when dual feasible)."""

    def __init__(self, H, c, lb, ub):
        n = c.size
        self.n, self.H, self.c, self.lb, self.ub = n, H, c, lb, ub
        self.T = -np.linalg.inv(H)
        self.x = self.T @ c
        self.u = np.zeros(n)
        self.state = np.zeros(n, int)
        self.pivots = 0
        tol = 1e-13 * (n * 0.125 if n > 8 else 1.0)
        self.thr_lo = -tol * (1 + np.abs(lb))
        self.thr_up = -tol * (1 + np.abs(ub))

    def pivot(self, p):
        """Sweep / reverse sweep on p (both are T -= col col^T / d with the pivot row/column fixed up)."""
        T = self.T
        d = T[p, p]
        col = T[:, p].copy()
        basic = self.state[p] == 0  # before the change: free = basic -> leaves the basis (reverse sweep)
        T -= np.outer(col, col) / d
        sg = -1.0 if basic else 1.0
        T[:, p] = sg * col / d
        T[p, :] = sg * col / d
        T[p, p] = -1.0 / d
        self.pivots += 1

    # -- exact values for the current partition (what the carried x, u equal up to round-off)
    def recompute(self):
        F = self.state == 0
        x = np.where(self.state == 1, self.lb, np.where(self.state == 2, self.ub, 0.0))
        if F.any():
            x[F] = np.linalg.solve(self.H[np.ix_(F, F)], -(self.c[F] + self.H[np.ix_(F, ~F)] @ x[~F]))
        g = self.H @ x + self.c
        self.x = x
        self.u = np.where(self.state == 1, g, np.where(self.state == 2, -g, 0.0))

    def fix(self, p, kind):
        """Full step onto bound `kind` (0 lb, 1 ub) of the free coordinate p, without a ratio test."""
        b = self.lb[p] if kind == 0 else self.ub[p]
        col = self.T[:, p].copy()
        nu = (b - self.x[p]) / (-col[p])
        free = self.state == 0
        phi = np.where(self.state == 1, -1.0, np.where(self.state == 2, 1.0, 0.0))
        self.x = np.where(free, self.x - col * nu, self.x)
        self.u = self.u - phi * col * nu
        self.pivot(p)
        self.state[p] = kind + 1
        self.x[p] = b
        self.u[p] = abs(nu)

    def free(self, p):
        """Release the fixed coordinate p: its multiplier goes to zero along its column."""
        col = self.T[:, p].copy()
        # multiplier (signed gradient) of p is driven to zero: the free coordinates move by T_Fp * g_p
        g_p = self.u[p] if self.state[p] == 1 else -self.u[p]
        free = self.state == 0
        phi = np.where(self.state == 1, -1.0, np.where(self.state == 2, 1.0, 0.0))
        # coordinate p moves by delta = -g_p / T_pp (T_pp = Schur complement > 0); others follow
        delta = -g_p / col[p]
        self.x = np.where(free, self.x - col * delta, self.x)
        self.u = self.u + phi * col * delta * 0.0 + np.where(free, 0.0, phi * (col * delta))
        self.x[p] += delta
        self.u[p] = 0.0
        self.pivot(p)
        self.state[p] = 0

    def violations(self):
        free = self.state == 0
        vlo = free & (self.x - self.lb < self.thr_lo)
        vup = free & (self.ub - self.x < self.thr_up)
        return vlo, vup

    def dual_violations(self, tol=1e-12):
        return (self.state != 0) & (self.u < -tol * (1 + np.abs(self.u)))


def gi_trips(tb: Tableau, max_iter=2000):
    """ik_sweep.h's loop from a dual-feasible tableau: returns (trips, adds, drops)."""
    trips = adds = drops = 0
    pending = None
    uplus = 0.0
    while trips < max_iter:
        if pending is None:
            vlo, vup = tb.violations()
            if not (vlo.any() | vup.any()):
                return trips, adds, drops
            z = -np.diag(tb.T)
            wz = np.where(z > 1e-30, 1.0 / np.maximum(z, 1e-300), 1e30)
            klo = np.where(vlo, -((tb.x - tb.lb) ** 2) * wz, np.inf)
            kup = np.where(vup, -((tb.ub - tb.x) ** 2) * wz, np.inf)
            if klo.min() <= kup.min():
                pending = (int(klo.argmin()), 0)
            else:
                pending = (int(kup.argmin()), 1)
            uplus = 0.0
        trips += 1
        p, kind = pending
        col = tb.T[:, p].copy()
        pv = col[p]
        num = (tb.lb[p] if kind == 0 else tb.ub[p]) - tb.x[p]
        sgn = 1.0 if num >= 0 else -1.0
        full = -abs(num) / pv
        phi = np.where(tb.state == 1, -1.0, np.where(tb.state == 2, 1.0, 0.0))
        rate = phi * col * sgn
        blocking = rate > 0
        ratio = np.where(blocking, np.maximum(tb.u, 0.0) / np.where(blocking, rate, 1.0), np.inf)
        k1 = ratio.min()
        tstep = min(k1, full)
        nu = sgn * tstep
        free = tb.state == 0
        tb.x = np.where(free, tb.x - col * nu, tb.x)
        tb.u = tb.u - phi * col * nu
        uplus += abs(nu)
        if not (k1 < full):
            tb.pivot(p)
            tb.state[p] = kind + 1
            tb.x[p] = tb.lb[p] if kind == 0 else tb.ub[p]
            tb.u[p] = uplus
            pending = None
            adds += 1
        else:
            kd = int(np.argmax(blocking & (ratio == k1)))
            tb.pivot(kd)
            tb.state[kd] = 0
            tb.u[kd] = 0.0
            drops += 1
    return trips, adds, drops


def pdas(tb: Tableau, max_outer=8, patience=1, frac=0.0, use_recompute=True):
    """Primal-dual active-set outer loop on the tableau.  Per outer iteration: every free coordinate whose weighted
    violation key is within `frac` of the worst one is fixed, every fixed coordinate with a wrong-sign multiplier is
    freed.  Stops (returns False) after `max_outer` iterations or when the number of infeasibilities has not decreased
    `patience` times in a row.  Returns (converged, outer, flips)."""
    outer = flips = 0
    best = tb.n * 2 + 1
    left = patience
    while True:
        vlo, vup = tb.violations()
        dv = tb.dual_violations()
        ninf = int(vlo.sum() + vup.sum() + dv.sum())
        if ninf == 0:
            return True, outer, flips
        if outer >= max_outer:
            return False, outer, flips
        if ninf < best:
            best, left = ninf, patience
        else:
            left -= 1
            if left < 0:
                return False, outer, flips
        outer += 1
        if frac > 0.0 and (vlo.any() or vup.any()):
            z = -np.diag(tb.T)
            key = np.where(vlo, (tb.x - tb.lb) ** 2 / z, np.where(vup, (tb.ub - tb.x) ** 2 / z, 0.0))
            keep = key >= frac * key.max()
            vlo &= keep
            vup &= keep
        # the flips are applied one after the other; x, u of the end point do not depend on the order
        for p in np.nonzero(dv)[0]:
            tb.pivot(p)
            tb.state[p] = 0
            flips += 1
        for p in np.nonzero(vlo | vup)[0]:
            tb.pivot(p)
            tb.state[p] = 1 if vlo[p] else 2
            flips += 1
        tb.recompute()


def restore_dual(tb: Tableau):
    """Free wrong-sign multipliers until none is left (the fixed set only shrinks).  Returns the pivots spent."""
    n0 = tb.pivots
    while True:
        dv = tb.dual_violations()
        if not dv.any():
            return tb.pivots - n0
        for p in np.nonzero(dv)[0]:
            tb.pivot(p)
            tb.state[p] = 0
        tb.recompute()


def run(name, B, bounds, error_scale, variants):
    terms = synthetic.make_terms(name, B, bounds=bounds, error_scale=error_scale)
    batch = synthetic.pack(terms)
    pf = synthetic.pink_form(terms)
    ref = c_oracle.solve_ik_batch(**pf, want_Hc=True, nthreads=8)
    H, c = ref["H"], ref["c"]
    lb, ub = batch.lb, batch.ub
    nact = ((np.abs(ref["dq"] - lb) < 1e-12) | (np.abs(ref["dq"] - ub) < 1e-12)).sum(axis=1)
    print(f"== {name} bounds={bounds} error_scale={error_scale} B={B}: active bounds at the minimiser {nact.mean():.1f}")

    # baseline
    trips = np.zeros(B, int)
    adds = np.zeros(B, int)
    drops = np.zeros(B, int)
    v0 = np.zeros(B, int)
    for b in range(B):
        tb = Tableau(H[b], c[b], lb[b], ub[b])
        vlo, vup = tb.violations()
        v0[b] = vlo.sum() + vup.sum()
        trips[b], adds[b], drops[b] = gi_trips(tb)
        assert np.abs(tb.x - ref["dq"][b]).max() < 1e-8, (b, np.abs(tb.x - ref["dq"][b]).max())
    def pm(a):
        return np.maximum(a[0::2], a[1::2]).mean()
    c_trip = 225.0
    print(f"   violated at x0 {v0.mean():.1f};  GI trips {trips.mean():.2f} (adds {adds.mean():.2f} drops {drops.mean():.2f}) pair-max {pm(trips):.2f}"
          f"  -> VALU/wave {pm(209 * adds + 246 * drops):.0f}")

    for label, kw in variants:
        outer = np.zeros(B, int)
        flips = np.zeros(B, int)
        rest = np.zeros(B, int)
        gt = np.zeros(B, int)
        ga = np.zeros(B, int)
        gd = np.zeros(B, int)
        conv = np.zeros(B, bool)
        for b in range(B):
            tb = Tableau(H[b], c[b], lb[b], ub[b])
            conv[b], outer[b], flips[b] = pdas(tb, **kw)
            if not conv[b]:
                rest[b] = restore_dual(tb)
                gt[b], ga[b], gd[b] = gi_trips(tb)
            err = np.abs(tb.x - ref["dq"][b]).max()
            assert err < 1e-8, (label, b, err)
        # cost model (VALU per wave): a flip = column 24 + x/u update 10 + pivot 52; an outer iteration = 60
        cost = 86 * (flips + rest) + 60 * (outer + 1) + 209 * ga + 246 * gd
        print(f"   {label:28s} converged {conv.mean():.3f}  outer {outer.mean():.2f}  flips {flips.mean():.2f}  restore {rest.mean():.2f}  GI trips after {gt.mean():.2f}"
              f"  total pivots {(flips + rest + gt).mean():.2f}  VALU/QP {cost.mean():.0f}  pair-max {pm(cost):.0f}")




# ---------------------------------------------------------------------------------------------------------------
# Goldfarb-Idnani with a candidate list (partial pricing): ONE selection pass nominates several violated
# constraints; they enter one after the other (any violated constraint is a legitimate entering constraint of the
# dual method, so every invariant of the single-pivot loop holds); candidates that stopped being violated are pruned
# with a compare + ballot.  The ratio test is lazy: the full step is applied, and only when a multiplier went negative
# is the blocking constraint looked for.
def gi_candidates(tb: Tableau, frac=0.25, order="index", kmax=64, max_iter=4000):
    trips = adds = drops = sels = 0
    cand = []  # list of (p, kind)
    pending = None
    uplus = 0.0
    while trips < max_iter:
        if pending is None:
            vlo, vup = tb.violations()
            cand = [(p, k) for (p, k) in cand if (vlo[p] if k == 0 else vup[p])]
            if not cand:
                if not (vlo.any() | vup.any()):
                    return trips, adds, drops, sels
                sels += 1
                z = -np.diag(tb.T)
                wz = np.where(z > 1e-30, 1.0 / np.maximum(z, 1e-300), 1e30)
                key = np.where(vlo, ((tb.x - tb.lb) ** 2) * wz, np.where(vup, ((tb.ub - tb.x) ** 2) * wz, 0.0))
                keep = np.nonzero(key >= frac * key.max())[0] if frac > 0 else np.nonzero(key > 0)[0]
                if order == "key":
                    keep = keep[np.argsort(-key[keep])]
                elif order == "best-first":
                    # the most violated first, the rest in index order
                    best = int(key.argmax())
                    keep = np.array([best] + [p for p in keep if p != best])
                keep = keep[:kmax]
                cand = [(int(p), 0 if vlo[p] else 1) for p in keep]
            pending = cand.pop(0)
            uplus = 0.0
        trips += 1
        p, kind = pending
        col = tb.T[:, p].copy()
        pv = col[p]
        num = (tb.lb[p] if kind == 0 else tb.ub[p]) - tb.x[p]
        sgn = 1.0 if num >= 0 else -1.0
        full = -abs(num) / pv
        phi = np.where(tb.state == 1, -1.0, np.where(tb.state == 2, 1.0, 0.0))
        rate = phi * col * sgn
        blocking = rate > 0
        ratio = np.where(blocking, np.maximum(tb.u, 0.0) / np.where(blocking, rate, 1.0), np.inf)
        k1 = ratio.min()
        tstep = min(k1, full)
        nu = sgn * tstep
        free = tb.state == 0
        tb.x = np.where(free, tb.x - col * nu, tb.x)
        tb.u = tb.u - phi * col * nu
        uplus += abs(nu)
        if not (k1 < full):
            tb.pivot(p)
            tb.state[p] = kind + 1
            tb.x[p] = tb.lb[p] if kind == 0 else tb.ub[p]
            tb.u[p] = uplus
            pending = None
            adds += 1
        else:
            kd = int(np.argmax(blocking & (ratio == k1)))
            tb.pivot(kd)
            tb.state[kd] = 0
            tb.u[kd] = 0.0
            drops += 1
    return trips, adds, drops, sels


def run_candidates(name, B, bounds, error_scale, variants):
    terms = synthetic.make_terms(name, B, bounds=bounds, error_scale=error_scale)
    batch = synthetic.pack(terms)
    pf = synthetic.pink_form(terms)
    ref = c_oracle.solve_ik_batch(**pf, want_Hc=True, nthreads=8)
    H, c = ref["H"], ref["c"]
    lb, ub = batch.lb, batch.ub
    print(f"== candidate lists: {name} bounds={bounds} error_scale={error_scale} B={B}")

    def pm(a):
        return np.maximum(a[0::2], a[1::2]).mean()

    for label, kw in variants:
        t = np.zeros(B, int)
        a = np.zeros(B, int)
        d = np.zeros(B, int)
        s = np.zeros(B, int)
        for b in range(B):
            tb = Tableau(H[b], c[b], lb[b], ub[b])
            t[b], a[b], d[b], s[b] = gi_candidates(tb, **kw)
            err = np.abs(tb.x - ref["dq"][b]).max()
            assert err < 1e-8, (label, b, err)
        # add = column 24 + optimistic step 30 + pivot 52 + bookkeeping 12; a trip with a drop pays the real ratio
        # test (45) and the leaving column (30) on top; a selection pass 53
        cost = 118 * a + 193 * d + 53 * s
        old = 209 * a + 246 * d
        print(f"   {label:30s} trips {t.mean():6.2f} (adds {a.mean():5.2f} drops {d.mean():5.2f}) selections {s.mean():5.2f}  pair-max trips {pm(t):6.2f}"
              f"  VALU/QP {cost.mean():5.0f} pair-max {pm(cost):5.0f}   [today's trip costs: {pm(old):5.0f}]")



# ---------------------------------------------------------------------------------------------------------------
# Single principal pivoting (what ik_sweep.h runs since round 6): every trip exchanges the ONE index whose
# complementarity condition fails with the largest weight -- no ratio test, no pending constraint.
def ppm(tb: Tableau, weight="objective", topk=1, max_iter=400):
    n = trips = 0
    while n < max_iter:
        vlo, vup = tb.violations()
        dv = tb.dual_violations()
        if not (vlo.any() or vup.any() or dv.any()):
            return n, trips
        d = np.abs(np.diag(tb.T))
        v = np.where(vlo, tb.x - tb.lb, np.where(vup, tb.ub - tb.x, np.where(dv, tb.u, 0.0)))
        key = v * v / np.maximum(d, 1e-300) if weight == "objective" else v * v
        trips += 1
        done = 0
        for p in np.argsort(-key)[:topk]:
            if key[p] <= 0:
                break
            if done:
                vlo2, vup2 = tb.violations()
                dv2 = tb.dual_violations()
                if not (vlo2[p] or vup2[p] or dv2[p]):
                    continue
                vlo, vup = vlo2, vup2
            tb.pivot(p)
            tb.state[p] = (1 if vlo[p] else 2) if tb.state[p] == 0 else 0
            tb.recompute()
            n += 1
            done += 1
    return -1, trips


def set_basis(tb: Tableau, st):
    """The tableau of a given partition from scratch (a guessed active set as the start)."""
    n, H = tb.n, tb.H
    tb.state = st.astype(int)
    F = st == 0
    T = np.zeros((n, n))
    if F.any():
        inv = np.linalg.inv(H[np.ix_(F, F)])
        T[np.ix_(F, F)] = -inv
        if (~F).any():
            Hnf = H[np.ix_(~F, F)]
            T[np.ix_(~F, F)] = Hnf @ inv
            T[np.ix_(F, ~F)] = (Hnf @ inv).T
            T[np.ix_(~F, ~F)] = H[np.ix_(~F, ~F)] - Hnf @ inv @ Hnf.T
    else:
        T[:] = H
    tb.T = T
    tb.recompute()
    return int(F.sum())


def guess(tb: Tableau, rule):
    """Which coordinates start fixed: `diag` = the bound that -c_i / H_ii violates (ik_sweep.h); `jacobiK` / `gsK` = K
    projected Jacobi / Gauss-Seidel iterations on top; `c` = every bounded coordinate, side by the sign of c."""
    H, c, lb, ub = tb.H, tb.c, tb.lb, tb.ub
    d = np.diag(H)
    if rule == "c":
        bounded = np.isfinite(lb) & np.isfinite(ub)
        return np.where(bounded, np.where(c > 0, 1, 2), 0)
    x = -c / d
    if rule == "rowsum":  # the damped Jacobi step that cannot overshoot: -c_i / sum_j |H_ij|
        x = -c / np.abs(H).sum(axis=1)
    if rule.startswith("mix"):  # -c_i / (H_ii + theta sum_{j != i} |H_ij|)
        th = float(rule[3:])
        x = -c / (d + th * (np.abs(H).sum(axis=1) - d))
    if rule.startswith("jacobi"):
        for _ in range(int(rule[6:])):
            xc = np.clip(x, lb, ub)
            x = -(c + H @ xc - d * xc) / d
    elif rule.startswith("gs"):
        xc = np.clip(x, lb, ub)
        for _ in range(int(rule[2:])):
            for i in range(len(c)):
                x[i] = -(c[i] + H[i] @ xc - d[i] * xc[i]) / d[i]
                xc[i] = min(max(x[i], lb[i]), ub[i])
    return np.where(x < lb, 1, np.where(x > ub, 2, 0))


def run_ppm(name, B, bounds, error_scale):
    terms = synthetic.make_terms(name, B, bounds=bounds, error_scale=error_scale)
    batch = synthetic.pack(terms)
    ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms), want_Hc=True, nthreads=8)
    H, c, lb, ub = ref["H"], ref["c"], batch.lb, batch.ub

    def pm(a):
        return np.maximum(a[0::2], a[1::2]).mean()

    print(f"== principal pivoting: {name} bounds={bounds} error_scale={error_scale} B={B}; oracle (quadprog's rule) {ref['iters'].mean():.2f} steps")
    for label, kw in (("largest objective change", dict()), ("largest violation", dict(weight="none")), ("two per trip", dict(topk=2)), ("three per trip", dict(topk=3))):
        n = np.zeros(B, int)
        t = np.zeros(B, int)
        for b in range(B):
            tb = Tableau(H[b], c[b], lb[b], ub[b])
            n[b], t[b] = ppm(tb, **kw)
            assert n[b] >= 0 and np.abs(tb.x - ref["dq"][b]).max() < 1e-8, (label, b)
        print(f"   from the unconstrained minimum, {label:26s} pivots {n.mean():6.2f} trips {t.mean():6.2f} (pair-max {pm(t):6.2f}) max {n.max()}")
    for rule in ("diag", "rowsum", "mix0.25", "mix0.5", "c", "jacobi1", "jacobi2", "gs1", "gs2"):
        n = np.zeros(B, int)
        sw = np.zeros(B, int)
        free = []
        for b in range(B):
            tb = Tableau(H[b], c[b], lb[b], ub[b])
            st = guess(tb, rule)
            free.append(st == 0)
            sw[b] = set_basis(tb, st)
            n[b], _ = ppm(tb)
            assert n[b] >= 0 and np.abs(tb.x - ref["dq"][b]).max() < 1e-8, (rule, b)
        union = np.array([(free[2 * i] | free[2 * i + 1]).sum() for i in range(B // 2)])
        print(f"   from the guess `{rule:7s}`: sweeps {sw.mean():5.1f} (union of a wave's two QPs {union.mean():5.1f})  pivots {n.mean():6.2f} (pair-max {pm(n):6.2f}) max {n.max()}"
              f"   ~VALU per QP {40 * union.mean() + 165 * pm(n):.0f}")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    name = sys.argv[2] if len(sys.argv) > 2 else "draco3"
    bounds = sys.argv[3] if len(sys.argv) > 3 else "tight"
    es = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    run(name, B, bounds, es, [
        ("pdas outer<=8 patience 1", dict(max_outer=8, patience=1)),
        ("pdas outer<=20 patience 3", dict(max_outer=20, patience=3)),
        ("pdas frac .25 outer<=12", dict(max_outer=12, patience=2, frac=0.25)),
        ("pdas frac .5 outer<=12", dict(max_outer=12, patience=2, frac=0.5)),
    ])
    run_candidates(name, B, bounds, es, [
        ("single (round 5's rule)", dict(frac=1.0)),
        ("frac .5 index order", dict(frac=0.5)),
        ("frac .25 key order", dict(frac=0.25, order="key")),
        ("all violated key order", dict(frac=0.0, order="key")),
        ("top 2 key order", dict(frac=0.0, order="key", kmax=2)),
        ("top 4 key order", dict(frac=0.0, order="key", kmax=4)),
    ])
    run_ppm(name, B, bounds, es)
