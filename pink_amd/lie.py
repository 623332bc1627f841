"""SE(3) helpers the task layer needs (host side, NumPy).

The reference gets these from Pinocchio (``pin.SE3``, ``pin.log``, ``pin.Jlog6``;
call sites ``pink/tasks/frame_task.py:181-193,222-227``).  Conventions follow
SURVEY.md appendix B.3: twists are ``[linear; angular]``, ``A.actInv(B) = A^-1 B``,
``Jlog6`` is the right (body) Jacobian of ``log6``.
"""

from __future__ import annotations

import numpy as np


def hat(w: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def exp3(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    K = hat(w)
    if th < 1e-8:
        return np.eye(3) + K + 0.5 * K @ K
    if th < 0.1:  # ((1 - cos th) / th^2 as a series: the closed form is good to 2e-10 only at th = 1e-3)
        t2 = th * th
        return np.eye(3) + (np.sin(th) / th) * K + (0.5 + t2 * (-1.0 / 24.0 + t2 * (1.0 / 720.0 + t2 * (-1.0 / 40320.0 + t2 * (1.0 / 3628800.0))))) * K @ K
    return np.eye(3) + (np.sin(th) / th) * K + ((1.0 - np.cos(th)) / th**2) * K @ K


def log3(R: np.ndarray) -> np.ndarray:
    """Rotation vector of ``R``.  ``theta = atan2(|v|/2, (tr R - 1)/2)`` (well conditioned on the
    whole range, unlike ``acos`` next to ``pi``); close to ``pi`` the axis comes from the symmetric
    part ``c I + (1 - c) a a^T`` as Pinocchio does."""
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    c = 0.5 * (np.trace(R) - 1.0)
    s = 0.5 * float(np.linalg.norm(v))
    th = float(np.arctan2(s, c))
    if th < 1e-8:
        return 0.5 * v
    if np.pi - th < 1e-2:
        k = int(np.argmax(np.diag(R)))
        one_c = 1.0 - c
        axis = np.zeros(3)
        axis[k] = np.sqrt(max((R[k, k] - c) / one_c, 0.0))
        for j in range(3):
            if j != k:
                axis[j] = (R[k, j] + R[j, k]) / (2.0 * one_c * axis[k])
        if axis @ v < 0:
            axis = -axis
        return th * axis
    return (th / (2.0 * s)) * v


class SE3:
    """Rigid transform with the subset of ``pin.SE3``'s interface Pink uses."""

    __slots__ = ("rotation", "translation")

    def __init__(self, rotation=None, translation=None):
        self.rotation = np.eye(3) if rotation is None else np.array(rotation, dtype=float)
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=float)

    @staticmethod
    def Identity() -> "SE3":
        return SE3()

    def copy(self) -> "SE3":
        return SE3(self.rotation.copy(), self.translation.copy())

    def inverse(self) -> "SE3":
        Rt = self.rotation.T
        return SE3(Rt, -Rt @ self.translation)

    def __mul__(self, other: "SE3") -> "SE3":
        return SE3(self.rotation @ other.rotation, self.rotation @ other.translation + self.translation)

    def actInv(self, other: "SE3") -> "SE3":
        return self.inverse() * other

    def act(self, point) -> np.ndarray:
        """Image of a point given in this frame (``pin.SE3.act`` on a 3-vector)."""
        return self.rotation @ np.asarray(point, dtype=float) + self.translation

    @property
    def action(self) -> np.ndarray:
        """6 x 6 adjoint acting on twists ``[linear; angular]`` (``pin.SE3.action``)."""
        R, p = self.rotation, self.translation
        A = np.zeros((6, 6))
        A[:3, :3] = R
        A[:3, 3:] = hat(p) @ R
        A[3:, 3:] = R
        return A

    @property
    def actionInverse(self) -> np.ndarray:
        return self.inverse().action

    @property
    def np(self) -> np.ndarray:
        T = np.eye(4)
        T[:3, :3] = self.rotation
        T[:3, 3] = self.translation
        return T

    def __repr__(self) -> str:
        return f"SE3(R={self.rotation.tolist()}, p={self.translation.tolist()})"


# beta(th) = 1 / th^2 - sin th / (2 th (1 - cos th)) = (1 - (th / 2) cot(th / 2)) / th^2 and beta_dot = beta'(th) / th as
# power series in th^2 (|B_2n| / (2n)! and 2 k times them): the closed forms subtract numbers of order 1 / th^2 and
# 1 / th^4 -- at th = 1e-3 they are wrong by 2e-4 and 1e4 relative (double precision against 60 digits), which a
# tracking controller's small orientation errors turned into 1e-8 relative on dq between two implementations of the same
# formula (scripts/gpu_fuzz_rollout.py, round 4).  Eight terms hold 1e-16 up to th = 0.6; the closed forms take over at
# 0.5 (beta 2e-14, beta_dot 4e-12 there, improving with th).  Same constants in pink_amd/lie_batch.py and
# pink_amd/csrc/ik_frame_task.h.
SERIES_TH = 0.5
BETA_SERIES = (1.0 / 12.0, 1.0 / 720.0, 1.0 / 30240.0, 1.0 / 1209600.0, 1.0 / 47900160.0, 691.0 / 1307674368000.0,
               1.0 / 74724249600.0, 3617.0 / 10670622842880000.0)
BETA_DOT_SERIES = tuple(2.0 * k * c for k, c in enumerate(BETA_SERIES))[1:] + (1.3737699290044551303e-13,)  # (the last: 16 |B_18| / 18!)


def _horner(coeffs, t2):
    acc = 0.0
    for c in reversed(coeffs):
        acc = acc * t2 + c
    return acc


def _alpha_beta(th: float):
    if th < SERIES_TH:
        beta = _horner(BETA_SERIES, th * th)
        return 1.0 - th * th * beta, beta
    s, c = np.sin(th), np.cos(th)
    return th * s / (2.0 * (1.0 - c)), 1.0 / th**2 - s / (2.0 * th * (1.0 - c))


def _beta_dot(th: float) -> float:
    if th < SERIES_TH:
        return _horner(BETA_DOT_SERIES, th * th)
    s, c = np.sin(th), np.cos(th)
    return -2.0 / th**4 + (1.0 + s / th) / (2.0 * th**2 * (1.0 - c))


def log6(M: SE3) -> np.ndarray:
    """Twist ``[v; w]`` with ``exp6([v; w]) = M``."""
    w = log3(M.rotation)
    th = float(np.linalg.norm(w))
    p = M.translation
    alpha, beta = _alpha_beta(th)
    v = alpha * p - 0.5 * np.cross(w, p) + beta * (w @ p) * w
    return np.hstack([v, w])


def exp6(xi: np.ndarray) -> SE3:
    v, w = xi[:3], xi[3:]
    th = float(np.linalg.norm(w))
    R = exp3(w)
    K = hat(w)
    if th < 0.1:  # (series: the closed forms cancel -- 2e-10 / 1e-9 relative at th = 1e-3; same as ik_kinematics.h)
        t2 = th * th
        A = 0.5 + t2 * (-1.0 / 24.0 + t2 * (1.0 / 720.0 + t2 * (-1.0 / 40320.0 + t2 * (1.0 / 3628800.0))))
        Bc = 1.0 / 6.0 + t2 * (-1.0 / 120.0 + t2 * (1.0 / 5040.0 + t2 * (-1.0 / 362880.0 + t2 * (1.0 / 39916800.0))))
        V = np.eye(3) + A * K + Bc * K @ K
    else:
        V = np.eye(3) + ((1 - np.cos(th)) / th**2) * K + ((th - np.sin(th)) / th**3) * K @ K
    return SE3(R, V @ v)


def Jlog3(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    d, a = _alpha_beta(th)
    return a * np.outer(w, w) + d * np.eye(3) + 0.5 * hat(w)


def Jlog6(M: SE3) -> np.ndarray:
    """Right Jacobian of ``log6`` at ``M``: ``d log6(M exp6(delta)) / d delta`` at 0."""
    w = log3(M.rotation)
    th = float(np.linalg.norm(w))
    p = M.translation
    A = Jlog3(w)
    _, beta = _alpha_beta(th)
    beta_dot = _beta_dot(th)
    wTp = float(w @ p)
    v3 = beta_dot * wTp * w - (th**2 * beta_dot + 2.0 * beta) * p
    C = np.outer(v3, w) + beta * np.outer(w, p) + beta * wTp * np.eye(3) + 0.5 * hat(p)
    J = np.zeros((6, 6))
    J[:3, :3] = A
    J[:3, 3:] = C @ A
    J[3:, 3:] = A
    return J
