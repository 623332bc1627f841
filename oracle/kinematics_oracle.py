"""CPU oracle for the device-side kinematics (TEST INFRASTRUCTURE ONLY).

Restates what the reference asks Pinocchio for around the IK step -- frame placements
(``pink/configuration.py:163-164``), LOCAL frame Jacobians (``:230-236``), ``pin.integrate``
(``:283``) -- for the kinematic trees the repository supports.  Pinocchio is not installable
offline (PARITY UNPINNED against it); to stay independent of the kernels' closed forms this
oracle composes 4x4 homogeneous matrices with ``scipy.linalg.expm`` of twist matrices and obtains
Jacobians by central finite differences of ``log6(T^-1 T(q (+) h e_j))`` (Pink's own
``tests/test_jacobians.py:47-75`` validates Jacobians the same way).

The model is passed as plain arrays (the tables of ``pink_amd.rollout.ModelArrays``).
"""

from __future__ import annotations

import numpy as np
from scipy.linalg import expm

from .se3_oracle import exp6, log6


def _pose(T12) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = np.asarray(T12[:9]).reshape(3, 3)
    T[:3, 3] = T12[9:12]
    return T


def _quat_matrix(qv) -> np.ndarray:
    """Rotation of a unit quaternion (x, y, z, w) through the exponential of its rotation vector."""
    x, y, z, w = np.asarray(qv, float) / np.linalg.norm(qv)
    n = np.linalg.norm([x, y, z])
    if n < 1e-300:
        return np.eye(3)
    ang = 2.0 * np.arctan2(n, w)
    a = np.array([x, y, z]) / n * ang
    return expm(np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]))


def joint_matrix(jtype: int, axis, q_j) -> np.ndarray:
    T = np.eye(4)
    if jtype == 0:  # revolute
        xi = np.hstack([np.zeros(3), np.asarray(axis) * q_j[0]])
        return exp6(xi)
    if jtype == 1:  # prismatic
        T[:3, 3] = np.asarray(axis) * q_j[0]
        return T
    T[:3, :3] = _quat_matrix(q_j[3:7])
    T[:3, 3] = q_j[:3]
    return T


def forward_kinematics(arr, q) -> list:
    """World placement of every joint frame."""
    oM = []
    for j in range(len(arr.parent)):
        nqj = 7 if arr.jtype[j] == 2 else 1
        X = _pose(arr.placement[j]) @ joint_matrix(int(arr.jtype[j]), arr.axis[j], q[arr.idx_q[j]:arr.idx_q[j] + nqj])
        oM.append(X if arr.parent[j] < 0 else oM[arr.parent[j]] @ X)
    return oM


def frame_poses(arr, q) -> np.ndarray:
    oM = forward_kinematics(arr, q)
    out = []
    for f in range(len(arr.frames)):
        T = _pose(arr.frame_placement[f])
        out.append(T if arr.frame_joint[f] < 0 else oM[arr.frame_joint[f]] @ T)
    return np.array(out)


def integrate(arr, q, v) -> np.ndarray:
    """q (+) v: vector joints add, the free-flyer right-multiplies by exp6 of its body twist."""
    q2 = np.array(q, dtype=float)
    for j in range(len(arr.parent)):
        iq, iv = arr.idx_q[j], arr.idx_v[j]
        if arr.jtype[j] == 2:
            M = joint_matrix(2, None, q[iq:iq + 7]) @ exp6(v[iv:iv + 6])
            q2[iq:iq + 3] = M[:3, 3]
            R = M[:3, :3]
            w = 0.5 * np.sqrt(max(0.0, 1.0 + np.trace(R)))
            if w > 1e-6:
                q2[iq + 3:iq + 7] = [(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w]
            else:  # rotation by pi: axis from the symmetric part
                a = np.sqrt(np.maximum(0.0, 0.5 * (np.diag(R) + 1.0)))
                k = int(np.argmax(a))
                a = np.array([(R[k, i] + R[i, k]) / (4 * a[k]) if i != k else a[k] for i in range(3)])
                q2[iq + 3:iq + 7] = [a[0], a[1], a[2], 0.0]
        else:
            q2[iq] = q[iq] + v[iv]
    return q2


def frame_jacobians_fd(arr, q, h: float = 1e-6) -> np.ndarray:
    """Body Jacobians ``[nf, 6, nv]`` by central differences along every tangent direction."""
    nv = len(arr.v_max)
    T0 = frame_poses(arr, q)
    J = np.zeros((len(arr.frames), 6, nv))
    for j in range(nv):
        d = np.zeros(nv)
        d[j] = h
        Tp, Tm = frame_poses(arr, integrate(arr, q, d)), frame_poses(arr, integrate(arr, q, -d))
        for f in range(len(arr.frames)):
            J[f, :, j] = (log6(np.linalg.solve(T0[f], Tp[f])) - log6(np.linalg.solve(T0[f], Tm[f]))) / (2 * h)
    return J
