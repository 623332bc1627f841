"""Tasks, limits and barriers evaluated for a whole batch of configurations (host side, vectorised NumPy).

Pink evaluates ``task.compute_error`` / ``compute_jacobian``, ``limit.compute_qp_inequalities`` and
``barrier.compute_qp_inequalities`` once per configuration (``pink/solve_ik.py:54-122``).  The stacks the
device-resident path does not form on chip used to take that route here too -- one ``Configuration`` object and one
round of small NumPy calls per instance.  This module evaluates every class the package ships over a
:class:`~pink_amd.kinematics_batch.BatchKinematics` instead: same arithmetic, a leading batch axis, no per-instance
Python.  A class without a batched evaluator (user subclasses) falls back to its own per-configuration methods.

Each evaluator cites the reference method it restates; the per-instance implementations in ``pink_amd/tasks``,
``limits``, ``barriers`` are the specification the tests hold these to (``tests/test_batch_eval.py``).
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lie_batch as lb
from .batch import BarrierTerm, DenseTaskTerm, DiagonalTaskTerm, split_box_rows
from .exceptions import PinkError, TargetNotSet, TaskDefinitionError, TaskJacobianNotSet
from .kinematics_batch import BatchKinematics
from .utils import get_root_joint_dim


def _pose12_of(T) -> np.ndarray:
    return np.hstack([np.asarray(T.rotation, dtype=float).ravel(), np.asarray(T.translation, dtype=float)])


def _same(col: Sequence) -> bool:
    """One task object for the whole batch?"""
    return getattr(col, "shared", False) or all(t is col[0] for t in col)


def _costs(col: Sequence, k: int):
    """Cost of a task slot: the shared one, or ``[B, k]`` when the instances differ."""
    c0 = col[0].cost
    if _same(col):
        return c0
    costs = [np.asarray(t.cost if t.cost is not None else 1.0, dtype=float) for t in col]
    if all(c.shape == costs[0].shape and np.array_equal(c, costs[0]) for c in costs):
        return c0
    return np.array([np.broadcast_to(c, (k,)) for c in costs])


def _check_slot(col: Sequence) -> None:
    t0 = col[0]
    if not _same(col) and any(type(t) is not type(t0) or t.gain != t0.gain or t.lm_damping != t0.lm_damping for t in col):
        raise PinkError("type / gain / lm_damping of one task slot must be the same for every instance of a batch")


# ---------------------------------------------------------------------------------------------------------------
# tasks
# ---------------------------------------------------------------------------------------------------------------
def _frame_targets(col: Sequence, B: int, attr: str, what: str) -> np.ndarray:
    """``[B, 12]`` target poses of a FrameTask / RelativeFrameTask slot."""
    t0 = col[0]
    if _same(col):
        poses = t0.poses12() if hasattr(t0, "poses12") else getattr(t0, "target_poses", None)
        if poses is not None:
            if poses.shape != (B, 12):
                raise PinkError(f"{type(t0).__name__} {t0.frame!r}: {poses.shape[0]} target poses for {B} configurations")
            return poses
        T = getattr(t0, attr)
        if T is None:
            raise TargetNotSet(what.format(t0))
        return np.broadcast_to(_pose12_of(T), (B, 12))
    out = np.empty((B, 12))
    for b, t in enumerate(col):
        T = getattr(t, attr)
        if T is None:
            raise TargetNotSet(what.format(t))
        out[b, :9] = np.asarray(T.rotation, dtype=float).reshape(9)
        out[b, 9:] = T.translation
    return out


def _frame_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/frame_task.py:176-227``: ``e = log6(T_f^-1 T_t)``, ``J = -Jlog6(T_t^-1 T_f) fJ``."""
    t0, B = col[0], kin.B
    if not _same(col) and any(t.frame != t0.frame for t in col):
        raise PinkError("the frame of one task slot must be the same for every instance of a batch")
    Tt = _frame_targets(col, B, "transform_target_to_world", "no target set for frame '{0.frame}'")
    Jb = kin.frame_jacobian(t0.frame)
    if solver is not None:  # the HIP frame-task kernel (pinkhip_frame_task_host): log6 / Jlog6 / the 6x6 by 6xnv product
        e, J = solver.frame_task_terms(kin.pose12(t0.frame), Tt, Jb)
    else:
        Rf, pf = kin.frame_pose(t0.frame)
        Rt, pt = Tt[:, :9].reshape(B, 3, 3), Tt[:, 9:]
        e = lb.log6(*lb.act_inv(Rf, pf, Rt, pt))
        J = -lb.Jlog6(*lb.act_inv(Rt, pt, Rf, pf)) @ Jb
    return DenseTaskTerm(J=J, e=e, cost=_costs(col, 6), gain=t0.gain, lm_damping=t0.lm_damping)


def _relative_frame_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/relative_frame_task.py:142-231``: ``e = log6(T_rt^-1 T_rf)``,
    ``J = Jlog6(T_tf) (fJ_f - Ad(T_fr) rJ_r)``."""
    t0, B = col[0], kin.B
    if not _same(col) and any(t.frame != t0.frame or t.root != t0.root for t in col):
        raise PinkError("frame / root of one task slot must be the same for every instance of a batch")
    Tt = _frame_targets(col, B, "transform_target_to_root",
                        "target pose of frame '{0.frame}' in frame '{0.root}' is undefined")
    Rf, pf = kin.frame_pose(t0.frame)
    Rr, pr = kin.frame_pose(t0.root)
    R_rf, p_rf = lb.act_inv(Rr, pr, Rf, pf)
    R_tf, p_tf = lb.act_inv(Tt[:, :9].reshape(B, 3, 3), Tt[:, 9:], R_rf, p_rf)
    e = lb.log6(R_tf, p_tf)
    R_fr, p_fr = lb.act_inv(R_rf, p_rf, np.broadcast_to(np.eye(3), (B, 3, 3)), np.zeros((B, 3)))  # T_rf^-1
    J = lb.Jlog6(R_tf, p_tf) @ (kin.frame_jacobian(t0.frame) - lb.adjoint(R_fr, p_fr) @ kin.frame_jacobian(t0.root))
    return DenseTaskTerm(J=J, e=e, cost=_costs(col, 6), gain=t0.gain, lm_damping=t0.lm_damping)


def _posture_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/posture_task.py:100-107``: ``q (-) q*`` on the actuated coordinates."""
    t0, B = col[0], kin.B
    root_nv = get_root_joint_dim(kin.model)[1]
    if _same(col):
        if t0.target_q_batch is not None:
            if t0.target_q_batch.shape != kin.q.shape:
                raise PinkError(f"PostureTask: targets {t0.target_q_batch.shape} for configurations {kin.q.shape}")
            qt = t0.target_q_batch
        elif t0.target_q is not None:
            qt = np.asarray(t0.target_q, dtype=float)
        else:
            raise TargetNotSet("no posture target")
    else:
        if any(t.target_q is None for t in col):
            raise TargetNotSet("no posture target")
        qt = np.stack([t.target_q for t in col])
    e = kin.difference(qt, kin.q, from_v=root_nv)[:, root_nv:]
    return DiagonalTaskTerm(col0=root_nv, e=e, cost=_costs(col, e.shape[1]), gain=t0.gain, lm_damping=t0.lm_damping)


def _damping_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/damping_task.py:24-43``: zero error, identity Jacobian on the actuated coordinates."""
    t0 = col[0]
    root_nv = get_root_joint_dim(kin.model)[1]
    k = kin.model.nv - root_nv
    return DiagonalTaskTerm(col0=root_nv, e=np.zeros((kin.B, k)), cost=_costs(col, k), gain=t0.gain, lm_damping=t0.lm_damping)


def _low_acceleration_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/low_acceleration_task.py:46-84``: ``e = -dt v_prev``, ``J = I``."""
    t0, nv = col[0], kin.model.nv
    if _same(col):
        e = np.zeros((kin.B, nv)) if t0.Delta_q_prev is None else np.broadcast_to(-t0.Delta_q_prev, (kin.B, nv))
    else:
        e = np.stack([np.zeros(nv) if t.Delta_q_prev is None else -t.Delta_q_prev for t in col])
    return DiagonalTaskTerm(col0=0, e=np.ascontiguousarray(e), cost=_costs(col, nv), gain=t0.gain, lm_damping=t0.lm_damping)


def _joint_velocity_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/joint_velocity_task.py:59-110``: ``e = dt v*`` (what the reference's ``compute_error`` returns), ``J = I`` after the root joint."""
    t0 = col[0]
    root = kin.model.root_joint
    r = 0 if root is None else root.nv
    k = kin.model.nv - r
    rows = []
    for t in ([t0] if _same(col) else col):
        if t.target_v is None or t.target_dt is None:
            raise TargetNotSet("no target set for joint velocity task")
        if t.target_v.shape[0] != k:
            raise TaskDefinitionError(f"target velocity has dimension {t.target_v.shape[0]}, expected {k}")
        rows.append(t.target_dt * t.target_v)
    e = np.broadcast_to(rows[0], (kin.B, k)) if _same(col) else np.stack(rows)
    return DiagonalTaskTerm(col0=r, e=np.ascontiguousarray(e), cost=_costs(col, k), gain=t0.gain, lm_damping=t0.lm_damping)


def _linear_holonomic_task(kin: BatchKinematics, col, solver=None):
    """``pink/tasks/linear_holonomic_task.py:103-148`` (and ``joint_coupling_task.py``): ``e = A (q (-) q_0) - b``,
    ``J = A dDifference(q_0, q)``."""
    t0, m = col[0], kin.model
    if not _same(col) and any(t.A is not t0.A and not np.array_equal(t.A, t0.A) for t in col):
        raise PinkError("the matrix A of one LinearHolonomicTask slot must be the same for every instance of a batch")
    if t0.A.shape[1] != m.nv:
        raise TaskJacobianNotSet
    if _same(col):
        q_ref = m.neutral() if t0.q_0 is None else np.asarray(t0.q_0, dtype=float)
        b = t0.b
    else:
        q_ref = np.stack([m.neutral() if t.q_0 is None else np.asarray(t.q_0, dtype=float) for t in col])
        b = np.stack([t.b for t in col])
    e = kin.difference(q_ref, kin.q) @ t0.A.T - b
    D = kin.d_difference(q_ref, kin.q)
    J = np.broadcast_to(t0.A, (kin.B,) + t0.A.shape) if D is None else t0.A @ D
    return DenseTaskTerm(J=np.ascontiguousarray(J), e=e, cost=_costs(col, e.shape[1]), gain=t0.gain, lm_damping=t0.lm_damping)


def _fallback_task(kin: BatchKinematics, col, solver=None):
    """A task class without a batched evaluator: its own ``as_term`` per configuration."""
    t0 = col[0]
    terms = [t.as_term(kin.configuration(b)) for b, t in enumerate(col)]
    e = np.concatenate([t.e for t in terms], axis=0)
    cost = _costs(col, e.shape[1])
    if isinstance(terms[0], DiagonalTaskTerm):
        return DiagonalTaskTerm(col0=terms[0].col0, e=e, cost=cost, gain=t0.gain, lm_damping=t0.lm_damping)
    return DenseTaskTerm(J=np.concatenate([t.J for t in terms], axis=0), e=e, cost=cost, gain=t0.gain, lm_damping=t0.lm_damping)


def _task_evaluators():
    from .tasks.frame_task import FrameTask
    from .tasks.linear_holonomic_task import JointCouplingTask, JointVelocityTask, LinearHolonomicTask
    from .tasks.posture_task import DampingTask, LowAccelerationTask, PostureTask
    from .tasks.relative_frame_task import RelativeFrameTask

    return {FrameTask: _frame_task, RelativeFrameTask: _relative_frame_task, PostureTask: _posture_task,
            DampingTask: _damping_task, LowAccelerationTask: _low_acceleration_task, JointVelocityTask: _joint_velocity_task,
            LinearHolonomicTask: _linear_holonomic_task, JointCouplingTask: _linear_holonomic_task}


_TASKS = None


def task_term(kin: BatchKinematics, col: Sequence, solver=None):
    """One task slot of the stacked QP for the whole batch: ``col`` holds the slot's task object of every instance
    (the same object ``B`` times when the batch shares it).  Classes are matched exactly: a subclass may override
    ``compute_error`` / ``compute_jacobian`` and is then evaluated through them."""
    global _TASKS
    if _TASKS is None:
        _TASKS = _task_evaluators()
    _check_slot(col)
    return _TASKS.get(type(col[0]), _fallback_task)(kin, col, solver)


# ---------------------------------------------------------------------------------------------------------------
# limits: merged box + dense rows
# ---------------------------------------------------------------------------------------------------------------
def _vector_joints_only(limit) -> bool:
    return all(j.kind != "free_flyer" for j in getattr(limit, "joints", ()))


def _index(idx):
    """``idx`` as a slice when it is a run of consecutive integers (the actuated coordinates of a robot usually are):
    basic indexing of ``[B, n]`` arrays instead of a gather and a scatter."""
    idx = np.asarray(idx)
    if idx.size and (np.diff(idx) == 1).all():
        return slice(int(idx[0]), int(idx[-1]) + 1)
    return idx


def _fold(op, box: np.ndarray, idx, values) -> None:
    """``box[:, idx] = op(box[:, idx], values)``: one pass over the view, in place, when ``idx`` is a run of columns."""
    if isinstance(idx, slice):
        view = box[:, idx]
        op(view, values, out=view)
    else:
        box[:, idx] = op(box[:, idx], values)


def limit_rows(kin: BatchKinematics, limit, dt: float, lb_: np.ndarray, ub_: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """Fold ``limit`` into the merged box ``lb_, ub_ [B, nv]`` (in place); returns its dense rows ``(G [B, r, nv],
    h [B, r])`` if it has any.  ``pink/solve_ik.py:107-122`` with every ``+-e_i`` row merged per coordinate."""
    from .limits.acceleration_limit import AccelerationLimit
    from .limits.configuration_limit import ConfigurationLimit
    from .limits.floating_base_velocity_limit import FloatingBaseVelocityLimit
    from .limits.velocity_limit import VelocityLimit

    m, B = kin.model, kin.B
    ty = type(limit)
    if ty is ConfigurationLimit and _vector_joints_only(limit):
        # pink/limits/configuration_limit.py:111-120: gain (q_max (-) q), gain (q_min (-) q) on the bounded joints
        if limit.projection_matrix is None:
            return None
        idx = _index(limit.indices)
        iq = _index([j.idx_q for j in limit.joints])
        g = limit.config_limit_gain
        qi = kin.q[:, iq]
        lo = np.subtract(limit.model.lowerPositionLimit[iq], qi)
        if g != 1.0:
            lo *= g
        _fold(np.maximum, lb_, idx, lo)
        up = np.subtract(limit.model.upperPositionLimit[iq], qi, out=lo)  # (lo is folded in: its storage serves again)
        if g != 1.0:
            up *= g
        _fold(np.minimum, ub_, idx, up)
        return None
    if ty is VelocityLimit:
        # pink/limits/velocity_limit.py:118-120
        if limit.projection_matrix is None:
            return None
        idx = _index(limit.indices)
        v = dt * limit.velocity_limit[idx]
        _fold(np.maximum, lb_, idx, -v)
        _fold(np.minimum, ub_, idx, v)
        return None
    if ty is AccelerationLimit and not any(j.kind == "free_flyer" and j.idx_v in limit.indices for j in m.joints):
        # pink/limits/acceleration_limit.py:158-199
        if limit.projection_matrix is None:
            return None
        idx = limit.indices
        iq = np.array([next(j.idx_q for j in m.joints if j.idx_v == i) for i in idx], dtype=int)
        dq_max = np.where(limit.has_configuration_limit, m.upperPositionLimit[iq] - kin.q[:, iq], np.inf)
        dq_min = np.where(limit.has_configuration_limit, kin.q[:, iq] - m.lowerPositionLimit[iq], np.inf)
        dt_sq = dt * dt
        with np.errstate(invalid="ignore"):
            upper = np.minimum(limit.a_max * dt_sq + limit.Delta_q_prev, dt * np.sqrt(2 * limit.a_max * dq_max))
            lower = np.minimum(limit.a_max * dt_sq - limit.Delta_q_prev, dt * np.sqrt(2 * limit.a_max * dq_min))
        lb_[:, idx] = np.maximum(lb_[:, idx], -lower)
        ub_[:, idx] = np.minimum(ub_[:, idx], upper)
        return None
    if ty is FloatingBaseVelocityLimit:
        # pink/limits/floating_base_velocity_limit.py:104-148: +-J_root dq <= dt twist_max; the Jacobian of a frame
        # attached to the root joint is constant on the root's columns (the adjoint of its inverse placement)
        finite = np.isfinite(limit.twist_max)
        if not finite.any():
            return None
        J = kin.configuration(0).get_frame_jacobian(limit.base_frame)
        J[:, :limit.root_idx_v] = 0.0
        J[:, limit.root_idx_v + limit.root_nv:] = 0.0
        rows = J[finite]
        bounds = dt * limit.twist_max[finite]
        blo, bup, Gd, hd = split_box_rows(np.vstack([rows, -rows]), np.hstack([bounds, bounds]), m.nv)
        np.maximum(lb_, blo, out=lb_)
        np.minimum(ub_, bup, out=ub_)
        if len(hd):
            return np.broadcast_to(Gd, (B,) + Gd.shape), np.broadcast_to(hd, (B,) + hd.shape)
        return None
    # any other limit: its own methods per configuration; the number of dense rows may differ from instance to
    # instance (a row that is axis-aligned at one configuration went into the box there; a limit may return None)
    dense: List[Optional[Tuple[np.ndarray, np.ndarray]]] = []
    for b in range(B):
        cfg = kin.configuration(b)
        box = limit.compute_box(cfg, dt) if hasattr(limit, "compute_box") else None
        if box is not None:
            idx, lo, up = box
            lb_[b, idx] = np.maximum(lb_[b, idx], lo)
            ub_[b, idx] = np.minimum(ub_[b, idx], up)
            dense.append(None)
            continue
        mv = limit.compute_qp_inequalities(cfg, dt)
        if mv is None:
            dense.append(None)
            continue
        blo, bup, Gd, hd = split_box_rows(mv[0], mv[1], m.nv)
        lb_[b], ub_[b] = np.maximum(lb_[b], blo), np.minimum(ub_[b], bup)
        dense.append((Gd, hd) if len(hd) else None)
    r_max = max((len(d[1]) for d in dense if d is not None), default=0)
    if not r_max:
        return None
    G = np.zeros((B, r_max, m.nv))
    h = np.full((B, r_max), PAD_ROW_H)
    for b, d in enumerate(dense):
        if d is not None:
            G[b, :len(d[1])], h[b, :len(d[1])] = d
    return G, h


PAD_ROW_H = 1e30  # right-hand side of the padding rows  0 dq <= PAD


# ---------------------------------------------------------------------------------------------------------------
# barriers
# ---------------------------------------------------------------------------------------------------------------
def _class_k(bar, h: np.ndarray) -> np.ndarray:
    """The barrier's class-K function on ``h [B, dim]`` (``pink/barriers/barrier.py:246-252``)."""
    if bar.identity_gain_function:
        return h
    try:
        out = np.asarray(bar.gain_function(h), dtype=float)
        if out.shape == h.shape:
            return out
    except Exception:  # noqa: BLE001  (a scalar-only function)
        pass
    return np.vectorize(bar.gain_function, otypes=[float])(h)


def _safe_displacement(kin: BatchKinematics, bar) -> Optional[np.ndarray]:
    """``compute_safe_displacement`` per configuration (``barrier.py:134-149,193-201``); ``None`` when it is zero
    everywhere (the base class's) or not used."""
    from .barriers.barrier import Barrier

    if not bar.safe_displacement_gain > 1e-6 or type(bar).compute_safe_displacement is Barrier.compute_safe_displacement:
        return None
    sd = np.stack([np.asarray(bar.compute_safe_displacement(kin.configuration(b)), dtype=float) for b in range(kin.B)])
    return sd if sd.any() else None


def barrier_term(kin: BatchKinematics, bar) -> BarrierTerm:
    """One barrier for the whole batch: ``J_h [B, dim, nv]``, ``alpha(h) [B, dim]``."""
    from .barriers.barrier import Barrier
    from .barriers.body_spherical_barrier import BodySphericalBarrier
    from .barriers.position_barrier import PositionBarrier

    ty = type(bar)
    if ty is PositionBarrier:
        # pink/barriers/position_barrier.py:109-153
        _, pos = kin.frame_pose(bar.frame)
        Jw = kin.world_linear_jacobian(bar.frame)[:, bar.indices]
        hs, Js = [], []
        if bar.p_min is not None:
            hs.append(pos[:, bar.indices] - bar.p_min), Js.append(Jw)
        if bar.p_max is not None:
            hs.append(bar.p_max - pos[:, bar.indices]), Js.append(-Jw)
        h, J = np.concatenate(hs, axis=1), np.concatenate(Js, axis=1)
    elif ty is BodySphericalBarrier:
        # pink/barriers/body_spherical_barrier.py:74-140
        d = kin.frame_pose(bar.frames[0])[1] - kin.frame_pose(bar.frames[1])[1]
        h = (np.einsum("bi,bi->b", d, d) - bar.d_min ** 2)[:, None]
        lin = kin.world_linear_jacobian(bar.frames[0]) - kin.world_linear_jacobian(bar.frames[1])
        J = 2.0 * np.einsum("bi,bij->bj", d, lin)[:, None, :]
    else:
        terms = [bar.as_term(kin.configuration(b)) for b in range(kin.B)]
        t0 = terms[0]
        sd = None
        if any(t.safe_displacement is not None for t in terms):
            nv = kin.model.nv
            sd = np.concatenate([np.zeros((1, nv)) if t.safe_displacement is None else t.safe_displacement for t in terms], axis=0)
        return BarrierTerm(J_h=np.concatenate([t.J_h for t in terms], axis=0), h=np.concatenate([t.h for t in terms], axis=0),
                           gain=t0.gain, safe_displacement_gain=t0.safe_displacement_gain, safe_displacement=sd)
    return BarrierTerm(J_h=np.ascontiguousarray(J), h=_class_k(bar, h), gain=bar.gain,
                       safe_displacement_gain=bar.safe_displacement_gain, safe_displacement=_safe_displacement(kin, bar))
