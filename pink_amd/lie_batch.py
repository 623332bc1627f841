"""SE(3) helpers over a batch: the functions of :mod:`pink_amd.lie` with a leading batch axis.

The per-instance versions (``lie.py``) restate what Pink asks Pinocchio for one configuration at a time
(``pin.log`` / ``pin.Jlog6``, call sites ``pink/tasks/frame_task.py:181-193,222-227``,
``pink/tasks/relative_frame_task.py:142-231``); these evaluate the same formulas for ``B``
transforms at once, so that the host-evaluated path of :func:`pink_amd.solve_ik_batch` has no per-instance Python.
Rotations are ``[B, 3, 3]``, translations ``[B, 3]``, twists ``[linear; angular]`` as ``[B, 6]``.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np

from .lie import BETA_DOT_SERIES, BETA_SERIES, SERIES_TH


def hat(w: np.ndarray) -> np.ndarray:
    """``[B, 3] -> [B, 3, 3]`` cross-product matrices."""
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -w[..., 2], w[..., 1]
    K[..., 1, 0], K[..., 1, 2] = w[..., 2], -w[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -w[..., 1], w[..., 0]
    return K


def log3(R: np.ndarray) -> np.ndarray:
    """Rotation vectors of ``R [B, 3, 3]`` (same branches as :func:`pink_amd.lie.log3`)."""
    v = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    c = 0.5 * (R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1.0)
    s = 0.5 * np.linalg.norm(v, axis=1)
    th = np.arctan2(s, c)
    small = th < 1e-8
    with np.errstate(divide="ignore", invalid="ignore"):
        out = (th / (2.0 * np.where(small, 1.0, s)))[:, None] * v
    out[small] = 0.5 * v[small]
    near_pi = np.nonzero(np.pi - th < 1e-2)[0]
    if near_pi.size:  # rare: the axis from the symmetric part, instance by instance
        from .lie import log3 as log3_one

        for b in near_pi:
            out[b] = log3_one(R[b])
    return out


def _series(coeffs, t2: np.ndarray) -> np.ndarray:
    acc = np.zeros_like(t2)
    for c in reversed(coeffs):
        acc = acc * t2 + c
    return acc


def _alpha_beta(th: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``(th / 2) cot(th / 2)`` and ``beta``: power series below ``lie.SERIES_TH`` (the closed forms cancel: ``lie.py``)."""
    small = th < SERIES_TH
    ths = np.where(small, 1.0, th)
    s, c = np.sin(ths), np.cos(ths)
    t2 = np.where(small, th * th, 0.0)
    bs = _series(BETA_SERIES, t2)
    alpha = np.where(small, 1.0 - t2 * bs, ths * s / (2.0 * (1.0 - c)))
    beta = np.where(small, bs, 1.0 / ths**2 - s / (2.0 * ths * (1.0 - c)))
    return alpha, beta


def log6(R: np.ndarray, p: np.ndarray) -> np.ndarray:
    """Twists ``[B, 6]`` of the transforms ``(R, p)``."""
    w = log3(R)
    th = np.linalg.norm(w, axis=1)
    alpha, beta = _alpha_beta(th)
    wp = np.einsum("bi,bi->b", w, p)
    v = alpha[:, None] * p - 0.5 * np.cross(w, p) + (beta * wp)[:, None] * w
    return np.concatenate([v, w], axis=1)


def Jlog6(R: np.ndarray, p: np.ndarray) -> np.ndarray:
    """Right Jacobians of ``log6`` at the transforms ``(R, p)``: ``[B, 6, 6]``."""
    w = log3(R)
    th = np.linalg.norm(w, axis=1)
    small = th < SERIES_TH
    ths = np.where(small, 1.0, th)
    s, c = np.sin(ths), np.cos(ths)
    d, a = _alpha_beta(th)
    eye = np.eye(3)
    A = a[:, None, None] * (w[:, :, None] * w[:, None, :]) + d[:, None, None] * eye + 0.5 * hat(w)
    beta = a  # (the same expression)
    beta_dot = np.where(small, _series(BETA_DOT_SERIES, np.where(small, th * th, 0.0)), -2.0 / ths**4 + (1.0 + s / ths) / (2.0 * ths**2 * (1.0 - c)))
    wp = np.einsum("bi,bi->b", w, p)
    v3 = (beta_dot * wp)[:, None] * w - (th**2 * beta_dot + 2.0 * beta)[:, None] * p
    C = (v3[:, :, None] * w[:, None, :] + beta[:, None, None] * (w[:, :, None] * p[:, None, :])
         + (beta * wp)[:, None, None] * eye + 0.5 * hat(p))
    J = np.zeros((R.shape[0], 6, 6))
    J[:, :3, :3] = A
    J[:, :3, 3:] = C @ A
    J[:, 3:, 3:] = A
    return J


def quat_to_rot(q: np.ndarray) -> np.ndarray:
    """``[B, 4]`` quaternions ``(x, y, z, w)`` -> ``[B, 3, 3]``."""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)
    R[:, 1, 0], R[:, 1, 1], R[:, 1, 2] = 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)
    R[:, 2, 0], R[:, 2, 1], R[:, 2, 2] = 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)
    return R


def act_inv(Ra: np.ndarray, pa: np.ndarray, Rb: np.ndarray, pb: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``A^-1 B`` for batches of transforms (``pin.SE3.actInv``)."""
    Rat = np.swapaxes(Ra, -1, -2)
    return Rat @ Rb, np.einsum("...ij,...j->...i", Rat, pb - pa)


def adjoint(R: np.ndarray, p: np.ndarray) -> np.ndarray:
    """6 x 6 actions on twists ``[linear; angular]``: ``[[R, [p]x R], [0, R]]`` per instance."""
    A = np.zeros(R.shape[:-2] + (6, 6))
    A[..., :3, :3] = R
    A[..., :3, 3:] = hat(p) @ R
    A[..., 3:, 3:] = R
    return A
