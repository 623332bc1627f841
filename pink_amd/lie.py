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


def _alpha_beta(th: float):
    if th < 1e-4:
        return 1.0 - th**2 / 12.0, 1.0 / 12.0 + th**2 / 720.0
    s, c = np.sin(th), np.cos(th)
    return th * s / (2.0 * (1.0 - c)), 1.0 / th**2 - s / (2.0 * th * (1.0 - c))


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
    if th < 1e-8:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + ((1 - np.cos(th)) / th**2) * K + ((th - np.sin(th)) / th**3) * K @ K
    return SE3(R, V @ v)


def Jlog3(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    if th < 1e-4:
        a, d = 1.0 / 12.0 + th**2 / 720.0, 1.0 - th**2 / 12.0
    else:
        s, c = np.sin(th), np.cos(th)
        a = 1.0 / th**2 - s / (2.0 * th * (1.0 - c))
        d = 0.5 * th * s / (1.0 - c)
    return a * np.outer(w, w) + d * np.eye(3) + 0.5 * hat(w)


def Jlog6(M: SE3) -> np.ndarray:
    """Right Jacobian of ``log6`` at ``M``: ``d log6(M exp6(delta)) / d delta`` at 0."""
    w = log3(M.rotation)
    th = float(np.linalg.norm(w))
    p = M.translation
    A = Jlog3(w)
    _, beta = _alpha_beta(th)
    if th < 1e-4:
        beta_dot = 1.0 / 360.0
    else:
        s, c = np.sin(th), np.cos(th)
        beta_dot = -2.0 / th**4 + (1.0 + s / th) / (2.0 * th**2 * (1.0 - c))
    wTp = float(w @ p)
    v3 = beta_dot * wTp * w - (th**2 * beta_dot + 2.0 * beta) * p
    C = np.outer(v3, w) + beta * np.outer(w, p) + beta * wTp * np.eye(3) + 0.5 * hat(p)
    J = np.zeros((6, 6))
    J[:3, :3] = A
    J[:3, 3:] = C @ A
    J[3:, 3:] = A
    return J
