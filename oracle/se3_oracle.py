"""CPU oracle for the batched FrameTask terms (TEST INFRASTRUCTURE ONLY).

Restates ``pink/tasks/frame_task.py:176-193,217-227``: ``e = log6(T_frame^-1 T_target)`` and
``J = -Jlog6(T_target^-1 T_frame) J_body``.  The reference obtains ``log6`` / ``Jlog6`` from
Pinocchio, which is not installable offline, so the SE(3) maps are PARITY UNPINNED against
Pinocchio itself; they are written here from the series definitions (matrix exponential /
logarithm through SciPy, Jacobian by central differences of ``log6``), deliberately *not* from the
closed forms the kernel uses, so that agreement is a real check.
"""

from __future__ import annotations

import numpy as np
from scipy.linalg import expm, logm


def pose_matrix(T12: np.ndarray) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = np.asarray(T12[:9]).reshape(3, 3)
    T[:3, 3] = T12[9:12]
    return T


def log6(T: np.ndarray) -> np.ndarray:
    """Twist [v; w] with expm(hat(v, w)) = T, through the matrix logarithm."""
    L = np.real(logm(T))
    w = np.array([L[2, 1] - L[1, 2], L[0, 2] - L[2, 0], L[1, 0] - L[0, 1]]) * 0.5
    return np.hstack([L[:3, 3], w])


def exp6(xi: np.ndarray) -> np.ndarray:
    v, w = xi[:3], xi[3:]
    X = np.zeros((4, 4))
    X[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    X[:3, 3] = v
    return expm(X)


def jlog6_fd(T: np.ndarray, h: float = 1e-6) -> np.ndarray:
    """d log6(T exp6(delta)) / d delta at 0 by central differences (body perturbation)."""
    J = np.zeros((6, 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        J[:, k] = (log6(T @ exp6(d)) - log6(T @ exp6(-d))) / (2 * h)
    return J


def jlog6_mp(T: np.ndarray, digits: int = 50) -> np.ndarray:
    """The same derivative to ~1e-15: matrix logarithm and exponential in ``digits``-digit arithmetic (mpmath's
    ``logm`` / ``expm``: inverse scaling and squaring, no closed form of SE(3)), central differences with a step of
    1e-15 (truncation 1e-30).  Slow (a second per call): the generator of ``tests/golden/pink_round4.npz`` uses it so
    that the reference's ``build_ik`` output can be held to 1e-10 (``pink/tasks/frame_task.py:217-227``)."""
    import mpmath as mp

    with mp.workdps(digits):
        Tm = mp.matrix(np.asarray(T, dtype=float).tolist())
        h = mp.mpf(10) ** -15

        def log6m(M):
            L = mp.logm(M)
            return [L[0, 3], L[1, 3], L[2, 3], (L[2, 1] - L[1, 2]) / 2, (L[0, 2] - L[2, 0]) / 2, (L[1, 0] - L[0, 1]) / 2]

        def exp6m(k, s):
            X = mp.zeros(4, 4)
            if k < 3:
                X[k, 3] = s
            else:
                a, b = [(2, 1), (0, 2), (1, 0)][k - 3]
                X[a, b], X[b, a] = s, -s
            return mp.expm(X)

        J = np.zeros((6, 6))
        for k in range(6):
            fp, fm = log6m(Tm * exp6m(k, h)), log6m(Tm * exp6m(k, -h))
            for r in range(6):
                J[r, k] = float(mp.re((fp[r] - fm[r]) / (2 * h)))
    return J


def frame_task_terms(T_frame: np.ndarray, T_target: np.ndarray, J_body: np.ndarray):
    """Per-instance loop; returns ``(e [B, 6], J [B, 6, nv])`` (J accurate to ~1e-8: finite differences)."""
    B = J_body.shape[0]
    e = np.zeros((B, 6))
    J = np.zeros_like(J_body)
    for b in range(B):
        Tf, Tt = pose_matrix(T_frame[b]), pose_matrix(T_target[b])
        e[b] = log6(np.linalg.solve(Tf, Tt))
        J[b] = -jlog6_fd(np.linalg.solve(Tt, Tf)) @ J_body[b]
    return e, J
