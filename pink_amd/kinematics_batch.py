"""Forward kinematics and frame Jacobians of ``B`` configurations at once (NumPy, no per-instance Python).

:class:`pink_amd.configuration.Configuration` restates, for ONE configuration, what Pink asks Pinocchio for
(``pink/configuration.py:131-164`` forward kinematics, ``:203-254`` frame Jacobians and transforms).  The
host-evaluated path of :func:`pink_amd.solve_ik_batch` -- every task / limit / barrier the device-resident path does
not form on chip -- used to build one such object per instance; :class:`BatchKinematics` evaluates the same
quantities with a leading batch axis, one pass over the joints of the tree for all ``B`` configurations.

Conventions as in ``configuration.py``: twists ``[linear; angular]``, body Jacobians
(``pin.LOCAL``), free flyer ``q = [p, quat(x, y, z, w)]`` with the body twist as its tangent.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from . import lie_batch as lb
from .configuration import Configuration, Model


def _hat1(w: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


class BatchKinematics:
    """Kinematic state of ``B`` configurations ``q [B, nq]`` of one model.

    ``R [B, nj, 3, 3]``, ``p [B, nj, 3]``: joint frames in the world (``Configuration.oMi``);
    :meth:`frame_pose`, :meth:`frame_jacobian`, :meth:`world_linear_jacobian` per frame name, cached;
    :meth:`configuration` builds the per-instance object on demand (tasks / limits / barriers without a batched
    evaluator fall back to it)."""

    def __init__(self, model: Model, q: np.ndarray):
        q = np.ascontiguousarray(q, dtype=np.float64)
        if q.ndim != 2 or q.shape[1] != model.nq:
            raise ValueError(f"q must have shape [B, nq = {model.nq}], got {q.shape}")
        self.model, self.q, self.B = model, q, q.shape[0]
        self._poses: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        self._jacs: Dict[str, np.ndarray] = {}
        self._cfgs: Dict[int, Configuration] = {}
        self._axis_w: Optional[np.ndarray] = None
        self._forward()

    # -- pink/configuration.py:131-164 ------------------------------------------------------------------
    def _forward(self) -> None:
        m, q, B = self.model, self.q, self.B
        nj = len(m.joints)
        R = np.empty((B, nj, 3, 3))
        p = np.empty((B, nj, 3))
        for i, j in enumerate(m.joints):
            Rpl, ppl = j.placement.rotation, j.placement.translation
            if j.kind == "revolute":
                K = _hat1(j.axis)
                th = q[:, j.idx_q]
                # placement * exp3(axis theta) = Rpl + sin(theta) Rpl K + (1 - cos(theta)) Rpl K^2  (unit axis)
                M = Rpl + np.sin(th)[:, None, None] * (Rpl @ K) + (1.0 - np.cos(th))[:, None, None] * (Rpl @ K @ K)
                t = ppl  # [3]
            elif j.kind == "prismatic":
                M = Rpl
                t = ppl + q[:, j.idx_q, None] * (Rpl @ j.axis)
            else:  # free flyer
                M = Rpl @ lb.quat_to_rot(q[:, j.idx_q + 3:j.idx_q + 7])
                t = ppl + q[:, j.idx_q:j.idx_q + 3] @ Rpl.T
            if j.parent < 0:
                R[:, i] = M
                p[:, i] = t
            else:
                Rp = R[:, j.parent]
                R[:, i] = Rp @ M
                p[:, i] = p[:, j.parent] + (Rp @ t[..., None])[..., 0] if np.ndim(t) == 2 else p[:, j.parent] + Rp @ t
        self.R, self.p = R, p

    def configuration(self, b: int) -> Configuration:
        """The per-instance object for instance ``b`` (built on demand, cached)."""
        c = self._cfgs.get(b)
        if c is None:
            c = self._cfgs[b] = Configuration(self.model, q=self.q[b])
        return c

    # -- pink/configuration.py:238-254 ------------------------------------------------------------------
    def frame_pose(self, frame: str) -> Tuple[np.ndarray, np.ndarray]:
        """``(R [B, 3, 3], p [B, 3])`` of ``frame`` in the world (``get_transform_frame_to_world``)."""
        out = self._poses.get(frame)
        if out is None:
            f = self.model.frames[self.model.getFrameId(frame)]
            Rf, pf = f.placement.rotation, f.placement.translation
            if f.joint < 0:
                out = (np.broadcast_to(Rf, (self.B, 3, 3)), np.broadcast_to(pf, (self.B, 3)))
            else:
                Rj = self.R[:, f.joint]
                out = (Rj @ Rf, self.p[:, f.joint] + Rj @ pf)
            self._poses[frame] = out
        return out

    def pose12(self, frame: str) -> np.ndarray:
        """``[B, 12]``: rotation row-major, then translation (the layout of the C ABI)."""
        R, p = self.frame_pose(frame)
        return np.concatenate([R.reshape(self.B, 9), p], axis=1)

    def _chain(self, joint: int) -> List[int]:
        out = []
        while joint >= 0:
            out.append(joint)
            joint = self.model.joints[joint].parent
        return out

    def _axes_world(self) -> np.ndarray:
        if self._axis_w is None:
            ax = np.array([np.zeros(3) if j.axis is None else j.axis for j in self.model.joints])
            self._axis_w = np.einsum("bjik,jk->bji", self.R, ax)
        return self._axis_w

    def _jacobian_at(self, Rf: np.ndarray, pf: np.ndarray, joint: int, rotate: bool) -> np.ndarray:
        """``[B, 6, nv]`` Jacobian of the point ``pf`` / frame carried by ``joint``: columns in the world's axes, or
        (``rotate``) in the axes ``Rf`` of the frame (= the body Jacobian, ``pin.LOCAL``).  All ancestors of one kind
        are formed at once (one cross product / one rotation for the whole chain)."""
        m, B = self.model, self.B
        J = np.zeros((B, 6, m.nv))
        aw = self._axes_world()
        chain = self._chain(joint)
        rev = [a for a in chain if m.joints[a].kind == "revolute"]
        pri = [a for a in chain if m.joints[a].kind == "prismatic"]
        rot = (lambda v: v @ Rf) if rotate else (lambda v: v)  # rows v_i -> (Rf^T v_i)^T
        if rev:
            cols = [m.joints[a].idx_v for a in rev]
            ang = aw[:, rev]  # [B, n, 3]
            lin = np.cross(ang, pf[:, None, :] - self.p[:, rev])  # omega x (p_f - p_a)
            J[:, :3, cols] = np.swapaxes(rot(lin), 1, 2)
            J[:, 3:, cols] = np.swapaxes(rot(ang), 1, 2)
        if pri:
            cols = [m.joints[a].idx_v for a in pri]
            J[:, :3, cols] = np.swapaxes(rot(aw[:, pri]), 1, 2)
        for a in chain:
            jt = m.joints[a]
            if jt.kind == "free_flyer":
                # the free flyer's tangent is its body twist: world columns are [R_a, [p_a - p_f]x R_a; 0, R_a]
                Ra = self.R[:, a]
                Rt = np.swapaxes(Rf, 1, 2) if rotate else None
                top = lb.hat(self.p[:, a] - pf) @ Ra
                J[:, :3, jt.idx_v:jt.idx_v + 3] = Rt @ Ra if rotate else Ra
                J[:, 3:, jt.idx_v + 3:jt.idx_v + 6] = Rt @ Ra if rotate else Ra
                J[:, :3, jt.idx_v + 3:jt.idx_v + 6] = Rt @ top if rotate else top
        return J

    # -- pink/configuration.py:203-236 ------------------------------------------------------------------
    def frame_jacobian(self, frame: str) -> np.ndarray:
        """Body Jacobian of ``frame``, ``[B, 6, nv]`` (``get_frame_jacobian``)."""
        J = self._jacs.get(frame)
        if J is None:
            f = self.model.frames[self.model.getFrameId(frame)]
            Rf, pf = self.frame_pose(frame)
            J = self._jacs[frame] = self._jacobian_at(Rf, pf, f.joint, rotate=True)
        return J

    def world_linear_jacobian(self, frame: str) -> np.ndarray:
        """``R_f J_lin`` of ``frame``: velocity of the frame origin in the world's axes, ``[B, 3, nv]``
        (``pink/barriers/position_barrier.py:136-145``)."""
        Rf, _ = self.frame_pose(frame)
        return Rf @ self.frame_jacobian(frame)[:, :3]

    def joint_jacobian_world_aligned(self, joint: int) -> np.ndarray:
        """``pin.getJointJacobian(..., LOCAL_WORLD_ALIGNED)`` of joint ``joint``: ``[B, 6, nv]``."""
        return self._jacobian_at(self.R[:, joint], self.p[:, joint], joint, rotate=False)

    # -- pin.difference / pin.dDifference over the batch --------------------------------------------------
    def _free_flyer_pose(self, j, q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        q = np.broadcast_to(q, (self.B, self.model.nq)) if q.ndim == 1 else q
        return lb.quat_to_rot(q[:, j.idx_q + 3:j.idx_q + 7]), q[:, j.idx_q:j.idx_q + 3]

    def difference(self, q0: np.ndarray, q1: np.ndarray, from_v: int = 0) -> np.ndarray:
        """``q1 (-) q0`` per instance, ``[B, nv]``; either argument may be one configuration ``[nq]`` for all.
        ``from_v``: the caller only reads the tangent coordinates from this one on (a PostureTask: the joints behind the
        root, ``pink/tasks/posture_task.py:100-107``) -- the logarithm of a free flyer in front of it is not taken
        (its entries stay zero): that was 40 % of the host time of a batch evaluated on the host at nv = 30."""
        m = self.model
        q0, q1 = np.asarray(q0, dtype=float), np.asarray(q1, dtype=float)
        out = np.zeros((self.B, m.nv))
        # vector-space joints: q1 - q0, runs of consecutive coordinates in one slice each
        vj = [j for j in m.joints if j.kind != "free_flyer"]
        k = 0
        while k < len(vj):
            e = k
            while e + 1 < len(vj) and vj[e + 1].idx_q == vj[e].idx_q + 1 and vj[e + 1].idx_v == vj[e].idx_v + 1:
                e += 1
            n = e - k + 1
            out[:, vj[k].idx_v:vj[k].idx_v + n] = q1[..., vj[k].idx_q:vj[k].idx_q + n] - q0[..., vj[k].idx_q:vj[k].idx_q + n]
            k = e + 1
        for j in m.joints:
            if j.kind != "free_flyer" or j.idx_v + 6 <= from_v:
                continue
            a = np.broadcast_to(q0[..., j.idx_q:j.idx_q + 7], (self.B, 7))
            c = np.broadcast_to(q1[..., j.idx_q:j.idx_q + 7], (self.B, 7))
            fin = np.isfinite(a).all(axis=1) & np.isfinite(c).all(axis=1)
            if fin.all():
                out[:, j.idx_v:j.idx_v + 6] = lb.log6(*lb.act_inv(lb.quat_to_rot(a[:, 3:]), a[:, :3], lb.quat_to_rot(c[:, 3:]), c[:, :3]))
                continue
            out[:, j.idx_v:j.idx_v + 6] = np.inf  # "difference to an infinite limit": no limit
            if fin.any():
                R0, p0 = lb.quat_to_rot(a[fin, 3:]), a[fin, :3]
                R1, p1 = lb.quat_to_rot(c[fin, 3:]), c[fin, :3]
                out[fin, j.idx_v:j.idx_v + 6] = lb.log6(*lb.act_inv(R0, p0, R1, p1))
        return out

    def d_difference(self, q0: np.ndarray, q1: np.ndarray) -> Optional[np.ndarray]:
        """Jacobian of ``q1 (-) q0`` in ``q1`` per instance, ``[B, nv, nv]`` -- or ``None`` when it is the identity
        for every instance (no free flyer)."""
        m = self.model
        if not any(j.kind == "free_flyer" for j in m.joints):
            return None
        D = np.broadcast_to(np.eye(m.nv), (self.B, m.nv, m.nv)).copy()
        q0, q1 = np.asarray(q0, dtype=float), np.asarray(q1, dtype=float)
        for j in m.joints:
            if j.kind == "free_flyer":
                R0, p0 = self._free_flyer_pose(j, q0)
                R1, p1 = self._free_flyer_pose(j, q1)
                D[:, j.idx_v:j.idx_v + 6, j.idx_v:j.idx_v + 6] = lb.Jlog6(*lb.act_inv(R0, p0, R1, p1))
        return D
