"""Exact-arithmetic anchor for one IK QP (TEST INFRASTRUCTURE ONLY: tests/, bench.py's checker legs).

Where ``H`` is weakly regularised (``examples/humanoid_jvrc.py:69-81,112-114`` as shipped: no posture task,
``damping = 1e-12``, cond(H) ~ 1e13-1e14) two correct fp64 solvers differ by far more than 1e-8 along the flat directions
of ``H``, and comparing the HIP kernel with the fp64 oracle says nothing about which of the two is closer to the QP's
minimiser.  This module computes that minimiser in ``digits``-digit arithmetic (mpmath):

* ``P, q`` are formed from the task rows exactly as ``pink/tasks/task.py:145-167`` and ``pink/solve_ik.py:54-67`` state
  them (``W = diag(cost)``, ``mu = lm ||W alpha e||^2``, ``P = damping I + sum (WJ)^T (WJ) + mu I``, ``q = sum -(W alpha e)^T WJ``,
  barrier terms through ``diag_extra``), in multiprecision from the fp64 inputs -- not from the fp64 ``P, q``;
* the constraints are ``G x <= h`` as ``pink/solve_ik.py:70-122`` stacks them (rows ``+-e_i`` included, as rows);
* from a guess of the active set (the rows a candidate point meets) the KKT system of the equality-constrained QP is
  solved by LU in multiprecision, then rows with a negative multiplier leave and violated rows enter, one at a time, until
  the point is primal feasible with non-negative multipliers: by strict convexity that point is THE minimiser (the
  certificate is checked in the same arithmetic, to ``10^-(digits - 15)``).

Slow (a second or two per instance at nv = 50): meant for tens of instances.
"""

from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def _objective_mp(mp, nv, J, e, cost, gain, lm, rows, damping, diag_extra):
    """``P`` (list of lists of mpf), ``q`` (list) of one instance: pink/tasks/task.py:145-167, pink/solve_ik.py:54-67."""
    P = [[mp.mpf(0)] * nv for _ in range(nv)]
    q = [mp.mpf(0)] * nv
    shift = mp.mpf(float(damping)) + (mp.mpf(float(diag_extra)) if diag_extra is not None else 0)
    for t in range(len(rows) - 1):
        r0, r1 = int(rows[t]), int(rows[t + 1])
        a, l = mp.mpf(float(gain[t])), mp.mpf(float(lm[t]))
        WJ, We = [], []
        for r in range(r0, r1):
            w = mp.mpf(float(cost[r]))
            nz = [(j, w * mp.mpf(float(J[r, j]))) for j in range(nv) if J[r, j] != 0.0]
            WJ.append(nz)
            We.append(-a * w * mp.mpf(float(e[r])))  # task.py:159: weighted_error = W @ (-gain * e)
        shift += l * sum(v * v for v in We)  # task.py:160
        for nz, we in zip(WJ, We):
            for i, vi in nz:
                q[i] -= we * vi  # task.py:166: c = -weighted_error^T weighted_jacobian
                Pi = P[i]
                for j, vj in nz:
                    Pi[j] += vi * vj
    for i in range(nv):
        P[i][i] += shift
    return P, q


def exact_minimiser(J: np.ndarray, e: np.ndarray, cost: np.ndarray, gain: Sequence[float], lm: Sequence[float], rows: Sequence[int],
                    damping: float, G: np.ndarray, h: np.ndarray, x_guess: np.ndarray, diag_extra: Optional[float] = None,
                    c_extra: Optional[np.ndarray] = None, digits: int = 50, max_changes: int = 200, meq: int = 0) -> Tuple[np.ndarray, dict]:
    """Minimiser of ``1/2 x'Px + q'x  s.t.  Gx <= h`` for one instance (``P, q`` from the task rows; the first ``meq`` rows of
    ``G`` are equalities, as ``pink/solve_ik.py:140-149`` hands ``A, b`` to the solver), rounded to fp64, and
    ``{"active": rows, "changes": n, "residual": max KKT residual in multiprecision, "multipliers": of the active rows}``."""
    import mpmath as mp

    nv = J.shape[1]
    with mp.workdps(digits):
        P, q = _objective_mp(mp, nv, J, e, cost, gain, lm, rows, damping, diag_extra)
        if c_extra is not None:
            for i in range(nv):
                q[i] += mp.mpf(float(c_extra[i]))
        keep = [r for r in range(G.shape[0]) if np.isfinite(h[r]) and abs(h[r]) < 1e29]
        Gs = {r: [(j, mp.mpf(float(G[r, j]))) for j in range(nv) if G[r, j] != 0.0] for r in keep}
        hs = {r: mp.mpf(float(h[r])) for r in keep}
        tiny = mp.mpf(10) ** (-(digits - 15))

        def slack(r, x):  # h - g x  (>= 0 inside)
            return hs[r] - sum(v * x[j] for j, v in Gs[r])

        def key(r):  # rows with the same normal (Pink stacks a configuration and a velocity row per coordinate)
            return tuple((j, float(v)) for j, v in Gs[r])

        # a coordinate (or row) pinned from both sides -- g x <= h next to -g x <= -h, Pink's lb = ub -- is ONE equality with a
        # multiplier of either sign: the second row of such a pair leaves the problem, the first joins the equalities
        free = set(range(meq))
        by_key = {key(r): r for r in keep if r >= meq}
        for r in list(keep):
            if r < meq or r not in Gs:
                continue
            opp = by_key.get(tuple((j, -v) for j, v in key(r)))
            if opp is not None and opp != r and opp not in free and r not in free and hs[opp] == -hs[r]:
                free.add(r)
                keep.remove(opp)
                del Gs[opp], hs[opp]
                by_key.pop(key(r), None)
        xg = [mp.mpf(float(v)) for v in x_guess]
        active, seen = sorted(free), {}
        for r in keep:
            if r in free:
                continue
            s = slack(r, xg)
            if abs(s) <= mp.mpf(1e-7) * (1 + abs(hs[r])):
                k = key(r)
                if k not in seen or hs[r] < hs[seen[k]]:
                    if k in seen:
                        active.remove(seen[k])
                    seen[k] = r
                    active.append(r)

        def solve(act):
            n = nv + len(act)
            K = mp.zeros(n, n)
            rhs = mp.zeros(n, 1)
            for i in range(nv):
                Pi = P[i]
                for j in range(nv):
                    K[i, j] = Pi[j]
                rhs[i] = -q[i]
            for a, r in enumerate(act):
                for j, v in Gs[r]:
                    K[nv + a, j] = v
                    K[j, nv + a] = v
                rhs[nv + a] = hs[r]
            sol = mp.lu_solve(K, rhs)
            return [sol[i] for i in range(nv)], [sol[nv + a] for a in range(len(act))]

        changes = 0
        while True:
            x, lam = solve(active)
            worst, at = 0, None
            for a, l in enumerate(lam):
                if active[a] not in free and l < -tiny and (at is None or l < worst):
                    worst, at = l, a
            if at is not None:
                active.pop(at)
            else:
                act_keys = {key(r) for r in active}
                viol, enter = tiny, None
                for r in keep:
                    if r in active:
                        continue
                    s = slack(r, x)
                    if -s > viol:
                        viol, enter = -s, r
                if enter is None:
                    break
                if key(enter) in act_keys:  # the same normal with a tighter bound replaces the active one
                    active = [r for r in active if key(r) != key(enter)]
                active.append(enter)
            changes += 1
            if changes > max_changes:
                raise RuntimeError("exact_minimiser: the active set did not settle")
        # certificate in the same arithmetic: stationarity with the multipliers found, feasibility, signs
        grad = [sum(P[i][j] * x[j] for j in range(nv)) + q[i] for i in range(nv)]
        for a, r in enumerate(active):
            for j, v in Gs[r]:
                grad[j] += lam[a] * v
        res = max([abs(g) for g in grad] + [abs(slack(r, x)) for r in active] + [mp.mpf(0)])
        assert res <= tiny * (1 + max(abs(v) for v in q)), res
        assert all(l >= -tiny for a, l in enumerate(lam) if active[a] not in free) and all(slack(r, x) >= -tiny for r in keep if r not in free)
        order = np.argsort(active)
        return np.array([float(v) for v in x]), {"active": [active[i] for i in order], "changes": changes, "residual": float(res),
                                                 "multipliers": [float(lam[i]) for i in order]}


def anchor_report(form_of, damping: float, dq: np.ndarray, dq_oracle: np.ndarray, n_worst: int = 32, n_first: int = 32,
                  ok: Optional[np.ndarray] = None) -> dict:
    """``|dq - dq_exact|`` and ``|dq_oracle - dq_exact|`` on a sample of a batch: the ``n_worst`` instances where the two
    fp64 solutions differ most (the ones a whole-batch ``max_abs_err`` is made of) and the first ``n_first`` ones.
    ``form_of(lo, hi)``: the arrays of ``pink_amd.synthetic.pink_form`` for instances ``lo .. hi`` of the batch."""
    B = dq.shape[0]
    diff = np.abs(dq - dq_oracle).max(axis=1)
    if ok is not None:
        diff = np.where(ok, diff, -1.0)
    pick = list(np.argsort(-diff, kind="stable")[:n_worst]) + [b for b in range(min(n_first, B))]
    pick = [int(b) for b in dict.fromkeys(pick) if diff[b] >= 0.0]
    eg, eo, changes = [], [], 0
    for b in pick:
        pf = form_of(b, b + 1)
        de = pf.get("diag_extra")
        x, info = exact_minimiser(pf["J"][0], pf["e"][0], pf["cost"], pf["gain"], pf["lm"], pf["rows"], damping, pf["G"][0], pf["h"][0], dq[b],
                                  diag_extra=None if de is None else float(de[0]))
        eg.append(float(np.abs(dq[b] - x).max()))
        eo.append(float(np.abs(dq_oracle[b] - x).max()))
        changes += info["changes"]
    eg, eo = np.array(eg), np.array(eo)
    return {"instances": len(pick), "selection": f"{min(n_worst, B)} largest |dq - dq_oracle| + first {min(n_first, B)}",
            "max_abs_err_vs_exact": float(eg.max(initial=0.0)), "oracle_max_abs_err_vs_exact": float(eo.max(initial=0.0)),
            "median_abs_err_vs_exact": float(np.median(eg)) if len(eg) else 0.0, "oracle_median_abs_err_vs_exact": float(np.median(eo)) if len(eo) else 0.0,
            "worst_fp64_disagreement_in_sample": float(diff[pick].max(initial=0.0)),
            "closer_than_oracle_frac": float((eg <= eo).mean()) if len(eg) else 0.0,
            "active_set_changes_from_the_guess": int(changes), "per_instance": [[int(b), float(a), float(o)] for b, a, o in zip(pick, eg, eo)],
            "exact": "KKT system of the final active set in 50-digit arithmetic (mpmath LU), P, q formed in the same arithmetic from the task rows; "
                     "certificate checked to 1e-35 (oracle/exact_qp.py)"}
