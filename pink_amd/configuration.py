"""Host-side kinematic state: a small stand-in for Pinocchio + ``pink.Configuration``.

The reference delegates forward kinematics, frame Jacobians and manifold
operations to Pinocchio (C++), which is not available offline.  The IK hot path
does not need it (it consumes ``J``, ``e``, bounds), but the *API* the tasks are
written against does (``pink/configuration.py:131-293``).  This module provides
that API in NumPy for kinematic trees of revolute / prismatic joints with an
optional free-flyer root, plus a reader for simple URDF files.  It is upstream of
the GPU path, never a substitute for it.

Conventions: twists are ``[linear; angular]``; the frame
Jacobian is the *body* Jacobian (``pin.LOCAL``); a free-flyer has
``q = [p, quat(x, y, z, w)]`` and its tangent is the body twist.
"""

from __future__ import annotations

import logging
import xml.etree.ElementTree as ET
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .exceptions import FrameNotFound, NotWithinConfigurationLimits
from .lie import SE3, Jlog6, exp3, exp6, log6
from .utils import VectorSpace


def _quat_to_rot(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def _rot_to_quat(R: np.ndarray) -> np.ndarray:
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def _adjoint(T: SE3) -> np.ndarray:
    """6x6 action of ``T`` on twists ``[v; w]``."""
    R, p = T.rotation, T.translation
    px = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[:3, 3:] = px @ R
    A[3:, 3:] = R
    return A


@dataclass
class Joint:
    name: str
    kind: str  # "revolute" | "prismatic" | "free_flyer"
    parent: int  # index of the parent joint, -1 = world
    placement: SE3  # pose of the joint frame in the parent joint frame at q = 0
    axis: Optional[np.ndarray]
    idx_q: int
    nq: int
    idx_v: int
    nv: int


@dataclass
class Frame:
    name: str
    joint: int  # joint the frame is attached to (-1 = world)
    placement: SE3  # pose of the frame in that joint's frame


class Model:
    """Kinematic tree (the subset of ``pin.Model`` Pink reads)."""

    def __init__(self):
        self.joints: List[Joint] = []
        self.frames: List[Frame] = []
        self.nq = 0
        self.nv = 0
        self._lower: List[float] = []
        self._upper: List[float] = []
        self._vel: List[float] = []

    # -- construction ---------------------------------------------------------
    def add_joint(self, name: str, kind: str, parent: int = -1, placement: Optional[SE3] = None,
                  axis=None, lower: float = -np.inf, upper: float = np.inf, velocity: float = np.inf) -> int:
        placement = SE3() if placement is None else placement
        if kind == "free_flyer":
            nq, nv = 7, 6
            lo, up, vel = [-np.inf] * 7, [np.inf] * 7, [np.inf] * 6
        elif kind in ("revolute", "prismatic"):
            nq, nv = 1, 1
            lo, up, vel = [lower], [upper], [velocity]
            axis = np.asarray(axis, dtype=float)
            axis = axis / np.linalg.norm(axis)
        else:
            raise ValueError(kind)
        self.joints.append(Joint(name, kind, parent, placement, axis, self.nq, nq, self.nv, nv))
        self.nq += nq
        self.nv += nv
        self._lower += lo
        self._upper += up
        self._vel += vel
        idx = len(self.joints) - 1
        self.add_frame(name, idx, SE3())
        return idx

    def add_frame(self, name: str, joint: int, placement: Optional[SE3] = None) -> None:
        self.frames.append(Frame(name, joint, SE3() if placement is None else placement))

    def ensure_limits(self) -> "Model":
        """The model's default limits and tangent space, attached on first use (``pink/configuration.py:101-108``
        caches them on the ``pin.Model`` when the first ``Configuration`` is built; here every entry point that reads
        them -- ``Configuration``, ``ConfigurationBatch``, ``solve_ik_batch`` -- goes through this)."""
        if not hasattr(self, "tangent"):
            from .limits import ConfigurationLimit, VelocityLimit

            self.tangent = VectorSpace(self.nv)
            self.configuration_limit = ConfigurationLimit(self)
            self.velocity_limit = VelocityLimit(self)
            self.floating_base_velocity_limit = None
        return self

    # -- pin.Model-like attributes ---------------------------------------------
    @property
    def lowerPositionLimit(self) -> np.ndarray:
        return np.array(self._lower)

    @property
    def upperPositionLimit(self) -> np.ndarray:
        return np.array(self._upper)

    @property
    def velocityLimit(self) -> np.ndarray:
        return np.array(self._vel)

    @property
    def root_joint(self) -> Optional[Joint]:
        """The joint named ``"root_joint"`` (``pink/utils.py:50-54``)."""
        for j in self.joints:
            if j.name == "root_joint":
                return j
        return None

    def getFrameId(self, name: str) -> int:
        for i, f in enumerate(self.frames):
            if f.name == name:
                return i
        raise FrameNotFound(name, self.frames)

    def getJointId(self, name: str) -> int:
        for i, j in enumerate(self.joints):
            if j.name == name:
                return i
        raise FrameNotFound(name, self.joints)

    def neutral(self) -> np.ndarray:
        q = np.zeros(self.nq)
        for j in self.joints:
            if j.kind == "free_flyer":
                q[j.idx_q + 6] = 1.0
        return q

    def joint_transform(self, j: Joint, q: np.ndarray) -> SE3:
        if j.kind == "revolute":
            return SE3(exp3(j.axis * q[j.idx_q]), np.zeros(3))
        if j.kind == "prismatic":
            return SE3(np.eye(3), j.axis * q[j.idx_q])
        return SE3(_quat_to_rot(q[j.idx_q + 3:j.idx_q + 7]), q[j.idx_q:j.idx_q + 3])

    def difference(self, q0: np.ndarray, q1: np.ndarray) -> np.ndarray:
        """``q1 (-) q0`` in the tangent space at ``q0`` (``pin.difference``)."""
        q0, q1 = np.asarray(q0, float), np.asarray(q1, float)
        out = np.zeros(self.nv)
        for j in self.joints:
            if j.kind == "free_flyer":
                sl = slice(j.idx_q, j.idx_q + 7)
                if not (np.isfinite(q0[sl]).all() and np.isfinite(q1[sl]).all()):
                    out[j.idx_v:j.idx_v + 6] = np.inf  # "difference to an infinite limit": no limit
                    continue
                out[j.idx_v:j.idx_v + 6] = log6(self.joint_transform(j, q0).actInv(self.joint_transform(j, q1)))
            else:
                out[j.idx_v] = q1[j.idx_q] - q0[j.idx_q]
        return out

    def d_difference(self, q0: np.ndarray, q1: np.ndarray) -> np.ndarray:
        """Jacobian of ``q1 (-) q0`` with respect to ``q1`` (``pin.dDifference(..., ARG1)``): identity
        on vector-space joints, ``Jlog6(T_0^-1 T_1)`` on a free flyer."""
        q0, q1 = np.asarray(q0, float), np.asarray(q1, float)
        D = np.eye(self.nv)
        for j in self.joints:
            if j.kind == "free_flyer":
                M = self.joint_transform(j, q0).actInv(self.joint_transform(j, q1))
                D[j.idx_v:j.idx_v + 6, j.idx_v:j.idx_v + 6] = Jlog6(M)
        return D

    def integrate(self, q: np.ndarray, v: np.ndarray) -> np.ndarray:
        """``q (+) v`` (``pin.integrate``)."""
        q, v = np.asarray(q, float), np.asarray(v, float)
        out = q.copy()
        for j in self.joints:
            if j.kind == "free_flyer":
                M = self.joint_transform(j, q) * exp6(v[j.idx_v:j.idx_v + 6])
                out[j.idx_q:j.idx_q + 3] = M.translation
                out[j.idx_q + 3:j.idx_q + 7] = _rot_to_quat(M.rotation)
            else:
                out[j.idx_q] = q[j.idx_q] + v[j.idx_v]
        return out


class Configuration:
    """Kinematic state at ``q`` with the interface of ``pink.Configuration``
    (``pink/configuration.py:26-293``)."""

    def __init__(self, model: Model, data=None, q: Optional[np.ndarray] = None, forward_kinematics: bool = True):
        if q is None and data is not None and not hasattr(data, "__dict__"):
            q, data = data, None  # Configuration(model, q)
        model.ensure_limits()  # attached lazily to the model, configuration.py:101-108
        self.model = model
        self.data = self
        self.tangent = model.tangent
        self.q = model.neutral() if q is None else np.array(q, dtype=float)
        self.q.setflags(write=False)
        self.oMi: List[SE3] = []
        self.oMf: List[SE3] = []
        if forward_kinematics:
            self.update()

    def update(self, q: Optional[np.ndarray] = None) -> None:
        """Forward kinematics of every joint and frame (``configuration.py:131-164``)."""
        if q is not None:
            self.q = np.array(q, dtype=float)
            self.q.setflags(write=False)
        m = self.model
        self.oMi = []
        for j in m.joints:
            parent = SE3() if j.parent < 0 else self.oMi[j.parent]
            self.oMi.append(parent * j.placement * m.joint_transform(j, self.q))
        self.oMf = [(SE3() if f.joint < 0 else self.oMi[f.joint]) * f.placement for f in m.frames]

    def check_limits(self, tol: float = 1e-6, safety_break: bool = True) -> None:
        """``configuration.py:166-201``: raise on the first violated joint limit, or (``safety_break=False``) warn about every one."""
        m = self.model
        lo, up = m.lowerPositionLimit, m.upperPositionLimit
        root = m.root_joint
        start = root.nq if root is not None else 0
        q = self.q
        bad = np.nonzero((up > lo + tol) & ((q < lo - tol) | (q > up + tol)))[0]
        bad = bad[bad >= start]
        for i in (int(k) for k in bad):  # raise on the first, or warn about each (configuration.py:183-201)
            if safety_break:
                raise NotWithinConfigurationLimits(i, q[i], lo[i], up[i])
            logging.warning("Value %f at index %d is out of limits: [%f, %f]", q[i], i, lo[i], up[i])

    def get_frame_jacobian(self, frame: str) -> np.ndarray:
        """Body Jacobian of ``frame`` (6 x nv, ``configuration.py:203-236``)."""
        m = self.model
        f = m.frames[m.getFrameId(frame)]
        J = np.zeros((6, m.nv))
        oMf_inv = self.oMf[m.getFrameId(frame)].inverse()
        j = f.joint
        while j >= 0:
            jt = m.joints[j]
            A = _adjoint(oMf_inv * self.oMi[j])  # joint frame -> frame
            if jt.kind == "revolute":
                J[:, jt.idx_v] = A[:, 3:] @ jt.axis
            elif jt.kind == "prismatic":
                J[:, jt.idx_v] = A[:, :3] @ jt.axis
            else:
                J[:, jt.idx_v:jt.idx_v + 6] = A
            j = jt.parent
        return J

    def get_joint_jacobian_world_aligned(self, joint: int) -> np.ndarray:
        """Jacobian of joint ``joint`` in the frame centred at the joint origin with the world's axes
        (``pin.getJointJacobian(..., LOCAL_WORLD_ALIGNED)``): ``[R 0; 0 R]`` times the body Jacobian."""
        m = self.model
        J = np.zeros((6, m.nv))
        oMj_inv = self.oMi[joint].inverse()
        j = joint
        while j >= 0:
            jt = m.joints[j]
            A = _adjoint(oMj_inv * self.oMi[j])
            if jt.kind == "revolute":
                J[:, jt.idx_v] = A[:, 3:] @ jt.axis
            elif jt.kind == "prismatic":
                J[:, jt.idx_v] = A[:, :3] @ jt.axis
            else:
                J[:, jt.idx_v:jt.idx_v + 6] = A
            j = jt.parent
        R = self.oMi[joint].rotation
        return np.vstack([R @ J[:3], R @ J[3:]])

    def get_transform_frame_to_world(self, frame: str) -> SE3:
        return self.oMf[self.model.getFrameId(frame)].copy()

    def get_transform(self, source: str, dest: str) -> SE3:
        return self.get_transform_frame_to_world(dest).actInv(self.get_transform_frame_to_world(source))

    def integrate(self, velocity: np.ndarray, dt: float) -> np.ndarray:
        return self.model.integrate(self.q, np.asarray(velocity) * dt)

    def integrate_inplace(self, velocity: np.ndarray, dt: float) -> None:
        self.update(self.integrate(velocity, dt))


# ---------------------------------------------------------------------------
# builders
# ---------------------------------------------------------------------------


class ConfigurationBatch:
    """``B`` configurations of one model as ONE array ``q [B, nq]``: what :func:`pink_amd.solve_ik_batch` takes in the
    place of a list of :class:`Configuration` objects when forward kinematics, task rows and limits are evaluated by
    the device kernels -- no per-instance Python object, no per-instance loop.  Indexing yields a
    :class:`Configuration` (built on demand) so that host-evaluated tasks / limits still work on it."""

    def __init__(self, model: Model, q: np.ndarray):
        q = np.ascontiguousarray(q, dtype=np.float64)
        if q.ndim != 2 or q.shape[1] != model.nq:
            raise ValueError(f"q must have shape [B, nq = {model.nq}], got {q.shape}")
        self.model, self.q = model.ensure_limits(), q

    def kinematics(self):
        """Forward kinematics of the whole batch (:class:`pink_amd.kinematics_batch.BatchKinematics`) at the CURRENT
        ``q``, evaluated afresh by every call: ``q`` aliases the caller's array, which is refilled in place between
        two ``solve_ik_batch`` calls (``pinned_empty``), so nothing derived from it is kept here.  The object returned
        works on its own copy of ``q`` (one call sees one state)."""
        from .kinematics_batch import BatchKinematics

        return BatchKinematics(self.model, self.q.copy())

    def __len__(self) -> int:
        return self.q.shape[0]

    def __getitem__(self, b):
        if isinstance(b, slice):
            return ConfigurationBatch(self.model, self.q[b])
        return Configuration(self.model, q=self.q[b])

    def __iter__(self):
        return (self[b] for b in range(len(self)))

    def check_limits(self, tol: float = 1e-6, safety_break: bool = True) -> None:
        """``Configuration.check_limits`` (``pink/configuration.py:166-201``) over the batch, vectorised."""
        m = self.model
        lo, up = m.lowerPositionLimit, m.upperPositionLimit
        start = m.root_joint.nq if m.root_joint is not None else 0
        bad = (up > lo + tol) & ((self.q < lo - tol) | (self.q > up + tol))
        bad[:, :start] = False
        for b, i in zip(*(v.tolist() for v in np.nonzero(bad))):
            if safety_break:
                raise NotWithinConfigurationLimits(i, self.q[b, i], lo[i], up[i], instance=b)
            logging.warning("Value %f at index %d of instance %d is out of limits: [%f, %f]", self.q[b, i], i, b, lo[i], up[i])


def _rpy(r: float, p: float, y: float) -> np.ndarray:
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def load_urdf(path: str, free_flyer: bool = False) -> Model:
    """Build a :class:`Model` from a URDF with revolute / continuous / prismatic / fixed
    joints (enough for ``examples/robots/*.urdf`` of the reference)."""
    root = ET.parse(path).getroot()
    children: Dict[str, List[ET.Element]] = {}
    child_links = set()
    for j in root.findall("joint"):
        children.setdefault(j.find("parent").get("link"), []).append(j)
        child_links.add(j.find("child").get("link"))
    links = [l.get("name") for l in root.findall("link")]
    base = next(l for l in links if l not in child_links)
    model = Model()
    base_joint = model.add_joint("root_joint", "free_flyer") if free_flyer else -1
    model.add_frame(base, base_joint, SE3())

    def walk(link: str, joint_idx: int, offset: SE3) -> None:
        for j in children.get(link, []):
            o = j.find("origin")
            xyz = [float(v) for v in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
            rpy = [float(v) for v in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
            T = offset * SE3(_rpy(*rpy), xyz)
            kind = j.get("type")
            child = j.find("child").get("link")
            if kind == "fixed":
                model.add_frame(child, joint_idx, T)
                walk(child, joint_idx, T)
                continue
            ax = j.find("axis")
            axis = [float(v) for v in (ax.get("xyz") if ax is not None else "1 0 0").split()]
            lim = j.find("limit")
            lo = float(lim.get("lower", "-inf")) if lim is not None and kind != "continuous" else -np.inf
            up = float(lim.get("upper", "inf")) if lim is not None and kind != "continuous" else np.inf
            vel = float(lim.get("velocity", "inf")) if lim is not None else np.inf
            idx = model.add_joint(j.get("name"), "prismatic" if kind == "prismatic" else "revolute", joint_idx, T,
                                  axis, lo, up, vel)
            model.add_frame(child, idx, SE3())
            walk(child, idx, SE3())

    walk(base, base_joint, SE3())
    return model


def build_chain(n: int, link_length: float = 0.3, free_flyer: bool = False, seed: int = 0,
                limit: float = np.pi, velocity: float = 3.15) -> Model:
    """A serial arm of ``n`` revolute joints with varied axes (UR-like when ``n = 6``),
    tool frame ``"tool0"`` at the tip."""
    rng = np.random.default_rng(seed)
    model = Model()
    parent = model.add_joint("root_joint", "free_flyer") if free_flyer else -1
    axes = [[0, 0, 1], [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0]]
    for i in range(n):
        axis = axes[i] if i < len(axes) else rng.normal(size=3)
        off = SE3(np.eye(3), [0.0, 0.0, 0.1] if i == 0 else [link_length * (0.5 + 0.5 * ((i + 1) % 2)), 0.0, 0.05 * (i % 3)])
        parent = model.add_joint(f"joint_{i + 1}", "revolute", parent, off, axis, -limit, limit, velocity)
    model.add_frame("tool0", parent, SE3(np.eye(3), [link_length * 0.5, 0.0, 0.0]))
    return model
