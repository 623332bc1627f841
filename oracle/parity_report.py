"""Whole-batch parity report of a solver's output against the C oracle (TEST INFRASTRUCTURE ONLY).

SURVEY.md section 8(d) "Parity procedure": max over the ENTIRE batch of |dq - dq_ref|_inf (absolute and relative to
|dq_ref|_inf), the status histogram of both sides, the fraction of instances whose final active set equals the
oracle's, and the KKT residuals of the checked solution against the QP the pinned stacking builds (pink/solve_ik.py:270-275
is what the numbers stand in for).  Used by ``tests/`` and by ``bench.py``'s parity leg; never by the product.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from . import c_oracle

ACTIVE_TOL = 1e-9  # a constraint counts as active when its slack is below ACTIVE_TOL (1 + |bound|)


def _slice(pf: dict, lo: int, hi: int, B: int) -> dict:
    out = {}
    for k, v in pf.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k in ("J", "e", "G", "h", "diag_extra", "c_extra"):
            out[k] = v[lo:hi]
        elif isinstance(v, np.ndarray) and k == "cost" and v.ndim == 2:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def active_sets(dq, lb, ub, Gd=None, hd=None):
    """Boolean masks of the constraints that are tight at ``dq``: (at_lb, at_ub, dense_active)."""
    at_lb = np.isfinite(lb) & (dq - lb <= ACTIVE_TOL * (1.0 + np.abs(np.where(np.isfinite(lb), lb, 0.0))))
    at_ub = np.isfinite(ub) & (ub - dq <= ACTIVE_TOL * (1.0 + np.abs(np.where(np.isfinite(ub), ub, 0.0))))
    if Gd is not None and Gd.shape[1]:
        slack = hd - np.einsum("bmj,bj->bm", Gd, dq)
        scale = 1.0 + np.abs(hd) + np.linalg.norm(Gd, axis=2) * np.abs(dq).max(axis=1, keepdims=True)
        act = slack <= ACTIVE_TOL * scale
    else:
        act = np.zeros((dq.shape[0], 0), bool)
    return at_lb, at_ub, act


def kkt_residuals(H, c, lb, ub, dq, Gd=None, hd=None, n_eq: int = 0):
    """Vectorised KKT check of ``dq`` for  min 1/2 x'Hx + c'x, lb <= x <= ub, Gd x <= hd (first n_eq rows: =).
    Returns (stationarity, violation, multiplier_sign_violation), each the maximum over the batch; stationarity and
    the multiplier signs are relative to 1 + |c|_inf of the instance."""
    at_lb, at_ub, act = active_sets(dq, lb, ub, Gd, hd)
    g = np.einsum("bij,bj->bi", H, dq) + c
    scale = 1.0 + np.abs(c).max(axis=1, keepdims=True)
    viol = max(float(np.max(np.where(np.isfinite(lb), lb - dq, -np.inf), initial=-np.inf)),
               float(np.max(np.where(np.isfinite(ub), dq - ub, -np.inf), initial=-np.inf)), 0.0)
    free = ~(at_lb | at_ub)
    sign = 0.0
    if act.shape[1]:
        res = np.einsum("bmj,bj->bm", Gd, dq) - hd
        if n_eq:
            viol = max(viol, float(np.abs(res[:, :n_eq]).max()))
            act[:, :n_eq] = True
        if act.shape[1] > n_eq:
            viol = max(viol, float(res[:, n_eq:].max()))
        # multipliers of the active dense rows by least squares on the free coordinates: g_free + Af^T lam = 0
        Af = Gd * act[:, :, None] * free[:, None, :]
        M = np.einsum("bmj,bnj->bmn", Af, Af)
        tr = np.trace(M, axis1=1, axis2=2)[:, None, None]
        M = M + np.eye(M.shape[1])[None] * (~act)[:, :, None] + 1e-13 * (tr + 1e-300) * np.eye(M.shape[1])[None]
        rhs = -np.einsum("bmj,bj->bm", Af, np.where(free, g, 0.0))
        lam = np.linalg.solve(M, rhs[:, :, None])[:, :, 0] * act
        g = g + np.einsum("bmj,bm->bj", Gd, lam)
        if act.shape[1] > n_eq:
            sign = max(sign, float((-lam[:, n_eq:] / scale).max()))
    stat = float(np.abs(np.where(free, g, 0.0) / scale).max())
    # the gradient pushes outward at an active bound: g >= 0 at lb, g <= 0 at ub (both active: pinned, any sign)
    sign = max(sign, float(np.max(np.where(at_lb & ~at_ub, -g, -np.inf) / scale, initial=-np.inf)),
               float(np.max(np.where(at_ub & ~at_lb, g, -np.inf) / scale, initial=-np.inf)), 0.0)
    return stat, viol, sign


def parity_report(pf, batch, dq: np.ndarray, status: np.ndarray, nthreads: int = 0, chunk: int = 8192,
                  H_gpu: Optional[np.ndarray] = None, dq_ref_out: Optional[np.ndarray] = None) -> dict:
    """Compare EVERY instance of ``(dq, status)`` with the C oracle solving ``pf`` -- the Pink-form arrays of the same
    batch (``synthetic.pink_form`` / ``tests.cases``), or a callable ``pf(lo, hi)`` that builds them for a slice
    (the Pink form of 65 536 JVRC-shaped instances is ~7 GB: every limit as dense ``[P; -P]`` rows); ``batch`` is the
    packed batch (box + dense rows) of the same instances.  Chunked so that neither the Pink form nor the oracle's
    (H, c) of a big batch sit in memory at once."""
    B = dq.shape[0]
    rep = dict(instances_compared=0, max_abs_err=0.0, max_rel_err=0.0, status_mismatch=0, active_set_equal=0,
               kkt_stationarity_max=0.0, kkt_violation_max=0.0, kkt_multiplier_sign_max=0.0, oracle_iters_mean=0.0,
               objective_gap_rel_max=0.0, objective_gap_rel_min=0.0)
    hist_ref = np.zeros(4, int)
    md = batch.Gd.shape[1] if getattr(batch, "Gd", None) is not None else 0
    n_eq = int(getattr(batch, "n_eq", 0))
    it_sum = 0
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        pfc = pf(lo, hi) if callable(pf) else _slice(pf, lo, hi, B)
        ref = c_oracle.solve_ik_batch(**pfc, want_Hc=True, nthreads=nthreads, meq=n_eq)
        del pfc
        hist_ref += np.bincount(ref["status"], minlength=4)[:4]
        if dq_ref_out is not None:  # (the caller wants the oracle's points too: oracle/exact_qp.anchor_report)
            dq_ref_out[lo:hi] = np.where((ref["status"] == 0)[:, None], ref["dq"], np.nan)
        it_sum += int(ref["iters"].sum())
        x, st = dq[lo:hi], status[lo:hi]
        rep["status_mismatch"] += int((st != ref["status"]).sum())
        ok = (st == 0) & (ref["status"] == 0)
        if not ok.any():
            rep["instances_compared"] += hi - lo
            continue
        lb, ub = batch.lb[lo:hi][ok], batch.ub[lo:hi][ok]
        Gd = batch.Gd[lo:hi][ok] if md else None
        hd = batch.hd[lo:hi][ok] if md else None
        err = np.abs(x[ok] - ref["dq"][ok]).max(axis=1)
        xr = np.abs(ref["dq"][ok]).max(axis=1)
        rep["max_abs_err"] = max(rep["max_abs_err"], float(err.max()))
        rep["max_rel_err"] = max(rep["max_rel_err"], float((err / np.maximum(xr, 1e-300))[xr > 1e-12].max(initial=0.0)))
        a_g = active_sets(x[ok], lb, ub, Gd, hd)
        a_r = active_sets(ref["dq"][ok], lb, ub, Gd, hd)
        same = np.ones(int(ok.sum()), bool)
        for m1, m2 in zip(a_g, a_r):
            if m1.shape[1]:
                same &= (m1 == m2).all(axis=1)
        rep["active_set_equal"] += int(same.sum()) + int((~ok & (st == ref["status"])).sum())
        # objective of the checked point against the oracle's, f = 1/2 x'Hx + c'x, relative to 1 + |f_ref|: where the
        # minimiser is only weakly determined (flat directions of a weakly regularised H) dq may differ far more than
        # the north-star tolerance between two correct solvers; the objective and the KKT residuals may not
        Hk, ck = ref["H"][ok], ref["c"][ok]
        f_x = 0.5 * np.einsum("bi,bij,bj->b", x[ok], Hk, x[ok]) + np.einsum("bi,bi->b", ck, x[ok])
        f_r = 0.5 * np.einsum("bi,bij,bj->b", ref["dq"][ok], Hk, ref["dq"][ok]) + np.einsum("bi,bi->b", ck, ref["dq"][ok])
        gap = (f_x - f_r) / (1.0 + np.abs(f_r))
        rep["objective_gap_rel_max"] = max(rep["objective_gap_rel_max"], float(gap.max()))
        rep["objective_gap_rel_min"] = min(rep["objective_gap_rel_min"], float(gap.min()))
        stat, viol, sign = kkt_residuals(Hk, ck, lb, ub, x[ok], Gd, hd, n_eq)
        rep["kkt_stationarity_max"] = max(rep["kkt_stationarity_max"], stat)
        rep["kkt_violation_max"] = max(rep["kkt_violation_max"], viol)
        rep["kkt_multiplier_sign_max"] = max(rep["kkt_multiplier_sign_max"], sign)
        if H_gpu is not None:
            rep["max_H_err_rel"] = max(rep.get("max_H_err_rel", 0.0), float(np.abs(H_gpu[lo:hi] - ref["H"]).max() / max(np.abs(ref["H"]).max(), 1e-300)))
        rep["instances_compared"] += hi - lo
    rep["active_set_equal_frac"] = rep.pop("active_set_equal") / max(B, 1)
    rep["status_hist"] = {str(k): int(v) for k, v in enumerate(np.bincount(status, minlength=4)[:4])}
    rep["status_hist_oracle"] = {str(k): int(v) for k, v in enumerate(hist_ref)}
    rep["oracle_iters_mean"] = it_sum / max(B, 1)
    rep["oracle"] = "C restatement of Goldfarb-Idnani with quadprog's rules (oracle/gi_oracle.c), all instances"
    return rep
