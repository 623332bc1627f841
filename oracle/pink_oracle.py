"""CPU oracle for the differential-IK hot path (TEST INFRASTRUCTURE ONLY).

This module is a plain NumPy restatement of the arithmetic that the reference
(stephane-caron/pink) performs between "per-task (J, e) and per-limit (G, h)
arrays exist" and "the displacement dq exists".  It exists so that the HIP path
can be checked against an independent implementation.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``pink_amd`` never does.

Parity status
-------------
* Stacking half (``task_objective`` ... ``build_qp``): PINNED.  It is checked in
  ``tests/test_oracle_golden.py`` against fixtures produced by importing the
  reference's own ``pink.build_ik`` (``tests/golden/make_golden.py``) and
  against the known-answer tests the reference holds for this boundary
  (``tests/test_frame_task.py:123-141``, ``tests/test_low_acceleration_task.py
  :34-42``, ``tests/test_damping_task.py:34-39``).
* QP-solve half (``goldfarb_idnani``): PARITY UNPINNED by a run of the reference.
  The arithmetic lives in the third-party package ``quadprog`` (reached through
  ``qpsolvers.solve_problem`` at ``pink/solve_ik.py:270``); neither package is
  vendored, pinned (quadprog appears in no lock file) or installable here, and
  the reference's tests contain no golden dq.  The restatement follows the
  published algorithm (D. Goldfarb, A. Idnani, "A numerically stable dual
  method for solving strictly convex quadratic programs", Math. Prog. 27,
  1983), which is what quadprog implements.  It is anchored by (i) the known
  answers those packages publish: the Goldfarb-Idnani worked example documented
  for quadprog's ``solve.QP`` (solution, multipliers, active set) and the example
  of the qpsolvers README (inequalities + an equality), both in
  ``tests/test_published_qp.py``; (ii) uniqueness of the minimiser of a strictly
  convex QP plus the KKT certificate computed by ``kkt_residuals``; (iii)
  ``scipy.optimize.lsq_linear(method="bvls")`` on box-only problems and SLSQP on
  the published examples; (iv) the properties the reference's solve tests assert
  (``tests/test_solve_ik.py:79-102``).

All ``file:line`` citations are relative to the reference checkout.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------
# Stacking: tasks -> (H, c)
# ---------------------------------------------------------------------------


def weight_vector(cost, k: int) -> np.ndarray:
    """Diagonal of the task weight matrix W.

    Follows ``pink/tasks/task.py:148-156``: ``None`` means identity, a scalar is
    repeated over the ``k`` rows, a vector is used as given.  (The reference only
    treats a Python ``float`` as scalar; an ``int`` falls into the vector branch
    and breaks there.  The oracle accepts any real scalar.)
    """
    if cost is None:
        return np.ones(k)
    w = np.asarray(cost, dtype=float)
    if w.ndim == 0:
        return np.full(k, float(w))
    if w.shape != (k,):
        raise ValueError(f"cost has shape {w.shape}, task has {k} rows")
    return w


def task_objective(
    J: np.ndarray,
    e: np.ndarray,
    cost,
    gain: float,
    lm_damping: float,
) -> Tuple[np.ndarray, np.ndarray]:
    """One task's contribution (H_t, c_t) to the QP objective.

    ``pink/tasks/task.py:145-167``:  We = W(-gain e),  WJ = W J,
    mu = lm_damping * We.We,  H_t = WJ^T WJ + mu I,  c_t = -We^T WJ.
    """
    J = np.asarray(J, dtype=float)
    e = np.asarray(e, dtype=float)
    k, nv = J.shape
    w = weight_vector(cost, k)
    WJ = w[:, None] * J
    We = w * (-gain * e)
    mu = lm_damping * float(We @ We)
    H = WJ.T @ WJ + mu * np.eye(nv)
    c = -(We @ WJ)
    return H, c


def barrier_objective(
    J_h: np.ndarray,
    safe_displacement_gain: float,
    nv: int,
    safe_displacement: Optional[np.ndarray] = None,
) -> Tuple[np.ndarray, np.ndarray]:
    """Safe-displacement regulariser of one barrier.

    ``pink/barriers/barrier.py:190-203``: active only when the gain exceeds
    1e-6; weight = gain / ||J_h||_F^2 on the identity, linear term pulls toward
    the safe displacement (zero by default, ``barrier.py:149``).
    """
    H = np.zeros((nv, nv))
    c = np.zeros(nv)
    if safe_displacement_gain > 1e-6:
        rho = safe_displacement_gain / float(np.linalg.norm(J_h) ** 2)
        H += rho * np.eye(nv)
        if safe_displacement is not None:
            c += -rho * np.asarray(safe_displacement, dtype=float)
    return H, c


def qp_objective(
    nv: int,
    tasks: Sequence[Tuple[np.ndarray, np.ndarray, object, float, float]],
    damping: float,
    barriers: Sequence[Tuple[np.ndarray, float, Optional[np.ndarray]]] = (),
) -> Tuple[np.ndarray, np.ndarray]:
    """Sum of task and barrier objectives on top of Tikhonov damping.

    ``pink/solve_ik.py:54-67``.  ``tasks`` holds ``(J, e, cost, gain, lm)``
    tuples, ``barriers`` holds ``(J_h, safe_displacement_gain, dq_safe)``.
    """
    H = damping * np.eye(nv)
    c = np.zeros(nv)
    for J, e, cost, gain, lm in tasks:
        H_t, c_t = task_objective(J, e, cost, gain, lm)
        H += H_t
        c += c_t
    for J_h, r, dq_safe in barriers:
        H_b, c_b = barrier_objective(J_h, r, nv, dq_safe)
        H += H_b
        c += c_b
    return H, c


# ---------------------------------------------------------------------------
# Stacking: limits / barriers -> (G, h)
# ---------------------------------------------------------------------------


def configuration_limit_rows(
    q: np.ndarray,
    q_min: np.ndarray,
    q_max: np.ndarray,
    indices: np.ndarray,
    nv: int,
    config_limit_gain: float = 0.5,
) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """Rows of the joint-position limit for a vector-space model (nq == nv).

    ``pink/limits/configuration_limit.py:107-121``: G = [P; -P],
    h = [gain (q_max - q)[idx]; -gain (q_min - q)[idx]] where P selects the
    bounded tangent coordinates.  On a vector space ``pin.difference(q, q1)``
    is ``q1 - q``.
    """
    indices = np.asarray(indices, dtype=int)
    if indices.size == 0:
        return None
    P = np.eye(nv)[indices]
    p_max = config_limit_gain * (q_max - q)[indices]
    p_min = config_limit_gain * (q_min - q)[indices]
    return np.vstack([P, -P]), np.hstack([p_max, -p_min])


def configuration_limit_indices(q_min: np.ndarray, q_max: np.ndarray) -> np.ndarray:
    """Bounded coordinates: ``pink/limits/configuration_limit.py:50-71``."""
    ok = np.logical_and(q_max < 1e20, q_max > q_min + 1e-10)
    return np.nonzero(ok)[0]


def velocity_limit_indices(v_max: np.ndarray) -> np.ndarray:
    """Velocity-limited coordinates: ``pink/limits/velocity_limit.py:61-73``."""
    ok = np.logical_and(v_max < 1e20, v_max > 1e-10)
    return np.nonzero(ok)[0]


def velocity_limit_rows(
    v_max: np.ndarray, indices: np.ndarray, nv: int, dt: float
) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """``pink/limits/velocity_limit.py:113-121``: G = [P; -P], h = dt v_max twice."""
    indices = np.asarray(indices, dtype=int)
    if indices.size == 0:
        return None
    P = np.eye(nv)[indices]
    hv = dt * np.asarray(v_max, dtype=float)[indices]
    return np.vstack([P, -P]), np.hstack([hv, hv])


def barrier_rows(
    J_h: np.ndarray, h_val: np.ndarray, gain, dt: float, gain_function=None
) -> Tuple[np.ndarray, np.ndarray]:
    """CBF rows: ``pink/barriers/barrier.py:246-254``: G = -J_h/dt,
    h_i = gain_i * alpha(h_i(q)) with alpha the identity by default."""
    J_h = np.asarray(J_h, dtype=float)
    h_val = np.asarray(h_val, dtype=float)
    g = np.asarray(gain, dtype=float)
    if g.ndim == 0:
        g = np.full(h_val.shape, float(g))
    alpha = (lambda v: v) if gain_function is None else gain_function
    return -J_h / dt, np.array([g[i] * alpha(h_val[i]) for i in range(h_val.size)])


def qp_inequalities(
    blocks: Sequence[Optional[Tuple[np.ndarray, np.ndarray]]],
) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
    """``pink/solve_ik.py:107-122``: skip ``None`` blocks, stack the rest,
    ``(None, None)`` when nothing is left."""
    G_list = [b[0] for b in blocks if b is not None]
    h_list = [b[1] for b in blocks if b is not None]
    if not G_list:
        return None, None
    return np.vstack(G_list), np.hstack(h_list)


# ---------------------------------------------------------------------------
# QP solve: Goldfarb-Idnani dual active set (what quadprog implements)
# ---------------------------------------------------------------------------

STATUS_OPTIMAL = 0
STATUS_MAX_ITER = 1
STATUS_INFEASIBLE = 2
STATUS_NOT_PD = 3


@dataclass
class QPResult:
    """Outcome of one QP solve."""

    x: Optional[np.ndarray]
    status: int
    iterations: int
    active: np.ndarray  # indices of active inequality rows
    multipliers: np.ndarray  # length m, >= 0

    @property
    def found(self) -> bool:
        return self.status == STATUS_OPTIMAL


def goldfarb_idnani(
    P: np.ndarray,
    q: np.ndarray,
    G: Optional[np.ndarray] = None,
    h: Optional[np.ndarray] = None,
    max_iter: int = 0,
) -> QPResult:
    """Minimise 1/2 x^T P x + q^T x  subject to  G x <= h.

    Dual active-set method of Goldfarb & Idnani (1983) in its projector form
    (SURVEY.md Appendix B.2).  With N the matrix of active normals (columns
    n_i = -G_i, constraints read n_i^T x >= -h_i):

        z = (P^-1 - P^-1 N (N^T P^-1 N)^-1 N^T P^-1) n+     primal direction
        r = (N^T P^-1 N)^-1 N^T P^-1 n+                      dual direction

    The most violated constraint is chosen with the violation divided by the
    row norm, as quadprog does.  This version refactorises each iteration
    (O(n^3)); it is meant for small reference cases.  The C oracle
    (``oracle/gi_oracle.c``) implements the updating (J, R) form.
    """
    P = np.asarray(P, dtype=float)
    q = np.asarray(q, dtype=float)
    n = q.size
    try:
        L = np.linalg.cholesky(P)
    except np.linalg.LinAlgError:
        return QPResult(None, STATUS_NOT_PD, 0, np.zeros(0, int), np.zeros(0))
    Linv = np.linalg.solve(L, np.eye(n))
    Pinv = Linv.T @ Linv
    x = -Pinv @ q
    if G is None or len(G) == 0:
        return QPResult(x, STATUS_OPTIMAL, 0, np.zeros(0, int), np.zeros(0))
    G = np.asarray(G, dtype=float)
    h = np.asarray(h, dtype=float)
    m = h.size
    N_all = -G.T  # columns are normals n_i
    b_all = -h
    norms = np.linalg.norm(G, axis=1)
    norms[norms == 0.0] = 1.0
    if max_iter <= 0:
        max_iter = 20 * (n + m) + 50

    active: list = []
    u = np.zeros(0)
    iterations = 0
    tol = 1e-13 * max(1.0, n / 8.0)  # grows with the dimension like the round-off of the iterate (gi_oracle.c)

    while True:
        s = N_all.T @ x - b_all
        s_scaled = s / norms
        s_scaled[active] = np.inf
        p = int(np.argmin(s_scaled))
        if s_scaled[p] >= -tol * (1.0 + abs(b_all[p]) / norms[p]):
            lam = np.zeros(m)
            lam[active] = u
            return QPResult(x, STATUS_OPTIMAL, iterations, np.array(active, int), lam)
        n_plus = N_all[:, p]
        u_plus = 0.0
        while True:
            iterations += 1
            if iterations > max_iter:
                return QPResult(x, STATUS_MAX_ITER, iterations, np.array(active, int), np.zeros(m))
            if active:
                N = N_all[:, active]
                PiN = Pinv @ N
                M = N.T @ PiN
                r = np.linalg.solve(M, PiN.T @ n_plus)
                z = Pinv @ n_plus - PiN @ r
            else:
                r = np.zeros(0)
                z = Pinv @ n_plus
            zn = float(z @ n_plus)
            dn = float(n_plus @ Pinv @ n_plus)
            # partial step length: largest t keeping u - t r >= 0
            t1 = np.inf
            drop = -1
            for k in range(len(active)):
                if r[k] > 0.0:
                    cand = u[k] / r[k]
                    if cand < t1:
                        t1 = cand
                        drop = k
            # full step length: t such that the violated constraint becomes active
            s_p = float(n_plus @ x - b_all[p])
            if zn > 1e-24 * dn:
                t2 = -s_p / zn
            else:
                t2 = np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return QPResult(None, STATUS_INFEASIBLE, iterations, np.array(active, int), np.zeros(m))
            if not np.isfinite(t2):
                # step in the dual space only
                u = u - t * r
                u_plus += t
                del active[drop]
                u = np.delete(u, drop)
                continue
            x = x + t * z
            u = u - t * r
            u_plus += t
            if t == t2:
                active.append(p)
                u = np.append(u, u_plus)
                break
            del active[drop]
            u = np.delete(u, drop)


# ---------------------------------------------------------------------------
# Algorithm-independent certificate
# ---------------------------------------------------------------------------


def kkt_residuals(P, q, G, h, x, active_tol: float = 1e-9, A=None, b=None, with_nu: bool = False):
    """KKT certificate of ``x`` for ``min 1/2 x'Px + q'x  s.t. Gx <= h, Ax = b``.

    Returns ``(stationarity, primal violation, lam)``: multipliers ``lam >= 0`` of the inequality
    rows that are active within ``active_tol`` (and free multipliers of the equalities) are fitted by
    bounded least squares, the stationarity residual is ``|Px + q + G'lam + A'nu|_inf`` and the
    violation ``max(0, max(Gx - h), |Ax - b|_inf)``.  A strictly convex QP has a unique minimiser, so a
    small certificate proves ``x`` is the solution whatever solver produced it.  ``with_nu``: also the equality multipliers.
    """
    from scipy.optimize import lsq_linear, nnls

    P = np.asarray(P, float)
    q = np.asarray(q, float)
    x = np.asarray(x, float)
    g = P @ x + q
    n_in = 0 if G is None else len(G)
    n_eq = 0 if A is None else len(A)
    if n_in == 0 and n_eq == 0:
        return (float(np.abs(g).max(initial=0.0)), 0.0, np.zeros(0)) + ((np.zeros(0),) if with_nu else ())
    viol = 0.0
    cols = []
    lam = np.zeros(n_in)
    nu = np.zeros(n_eq)
    act = np.zeros(0, dtype=int)
    if n_in:
        G = np.asarray(G, float)
        h = np.asarray(h, float)
        slack = h - G @ x
        viol = float(max(0.0, -(slack.min())))
        norms = np.linalg.norm(G, axis=1)
        norms[norms == 0] = 1.0
        act = np.nonzero(slack / norms <= active_tol)[0]
        if act.size:
            cols.append((G[act] / norms[act, None]).T)  # scale rows for conditioning
    if n_eq:
        A = np.asarray(A, float)
        viol = max(viol, float(np.abs(A @ x - np.asarray(b, float)).max()))
        an = np.linalg.norm(A, axis=1)
        an[an == 0] = 1.0
        cols.append((A / an[:, None]).T)
    r = g.copy()
    if cols:
        M = np.hstack(cols)
        n_act = int(act.size)
        best = None
        # bounded least squares: multipliers of the active inequality rows >= 0, those of the equalities free.  Two
        # methods, the smaller residual wins (scipy's nnls on "free = difference of two non-negative columns" returned a
        # non-minimising point with exactly dependent columns: gpu_fuzz seed 704011).
        lo = np.r_[np.zeros(n_act), np.full(M.shape[1] - n_act, -np.inf)]
        for method in ("bvls", "trf"):
            try:
                sol = lsq_linear(M, -g, bounds=(lo, np.inf), method=method, tol=1e-15, max_iter=50 * max(M.shape)).x
            except Exception:  # noqa: BLE001 - a failed fit is just not a certificate
                continue
            rr = g + M @ sol
            if best is None or np.abs(rr).max() < np.abs(best[0]).max():
                best = (rr, sol)
            if np.abs(rr).max() <= 1e-13 * max(1.0, np.abs(g).max()):
                break
        if n_eq == 0 and (best is None or np.abs(best[0]).max() > 1e-13 * max(1.0, np.abs(g).max())):
            sol, _ = nnls(M, -g, maxiter=50 * max(M.shape))
            rr = g + M @ sol
            if best is None or np.abs(rr).max() < np.abs(best[0]).max():
                best = (rr, sol)
        r, sol = best
        if n_act:
            lam[act] = sol[:n_act] / norms[act]
        if n_eq:
            nu = sol[n_act:] / an
    stat = float(np.abs(r).max(initial=0.0))
    return (stat, viol, lam, nu) if with_nu else (stat, viol, lam)


# ---------------------------------------------------------------------------
# Whole path for one instance
# ---------------------------------------------------------------------------


def qp_equalities(constraints):
    """``(A, b)`` of the tasks enforced strictly: ``A = J``, ``b = -gain e`` stacked task after task
    (``pink/solve_ik.py:140-149``); ``(None, None)`` without constraints (``:138-139``).
    ``constraints`` holds ``(J, e, gain)`` tuples."""
    if not constraints:
        return None, None
    A = np.vstack([np.asarray(J, dtype=float) for J, _e, _g in constraints])
    b = np.hstack([-float(g) * np.asarray(e, dtype=float) for _J, e, g in constraints])
    return A, b


def build_qp(nv, tasks, damping, limit_blocks=(), barrier_terms=(), dt=None):
    """(P, q, G, h) exactly as ``pink.build_ik`` assembles them
    (``pink/solve_ik.py:198-203``), equalities excluded.

    ``barrier_terms`` holds ``(J_h, h_val, gain, safe_displacement_gain,
    dq_safe)``; rows from limits come first, then barriers
    (``pink/solve_ik.py:109-119``).
    """
    P, q = qp_objective(
        nv,
        tasks,
        damping,
        [(Jh, r, dqs) for (Jh, _hv, _g, r, dqs) in barrier_terms],
    )
    blocks = list(limit_blocks)
    for Jh, hv, g, _r, _dqs in barrier_terms:
        blocks.append(barrier_rows(Jh, hv, g, dt))
    G, h = qp_inequalities(blocks)
    return P, q, G, h


def solve_ik_instance(nv, tasks, dt, damping=1e-12, limit_blocks=(), barrier_terms=()):
    """dq for one instance; ``pink/solve_ik.py:260-275`` without the kinematics."""
    P, q, G, h = build_qp(nv, tasks, damping, limit_blocks, barrier_terms, dt)
    res = goldfarb_idnani(P, q, G, h)
    return res, (P, q, G, h)
