"""Host-evaluated route with the FrameTask rows formed on the device.

A task stack the whole-step kernel does not form on chip (``pink_amd/solve_ik.py``: anything beyond FrameTasks + one
PostureTask under the default limits) used to be evaluated entirely on the host: forward kinematics, body Jacobians,
``log6`` / ``Jlog6`` and the 6 x 6 by 6 x nv products of every FrameTask for the whole batch in NumPy, then ~6 kB per
instance across PCIe.  When every task with a dense Jacobian is a FrameTask or a RelativeFrameTask -- the other tasks of the stack being the
identity-Jacobian ones (PostureTask, DampingTask, LowAccelerationTask, JointVelocityTask: ``pink/tasks/posture_task.py``,
``damping_task.py``, ``low_acceleration_task.py``, ``joint_velocity_task.py``) -- that part needs nothing but ``q`` and
the targets: ``pinkhip_fk_frame_tasks_device`` writes the rows straight into the packed ``J`` / ``e`` streams in HBM
(``pink/tasks/frame_task.py:176-227``), the host only evaluates what is cheap and irregular -- the errors of the
diagonal tasks, the limits (any list: explicit gains, AccelerationLimit, user subclasses), barriers -- and
``pinkhip_solve_device`` stacks and solves.  ``J`` never exists on the host and never crosses PCIe.
"""

from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import numpy as np

from . import batch_eval as be
from ._lib import Desc, Problem, Result, c_double_p, c_int32_p
from .batch import DiagonalTaskTerm, pack_terms
from .batch_solver import split_iters
from .exceptions import PinkError

MAX_FRAME_TASKS = 16


def plan(kin_model, slots: Sequence, constraints) -> Optional[List[int]]:
    """Indices of the FrameTask slots when the hybrid route serves this stack, else ``None``: at least one FrameTask,
    every other task of a class whose batched evaluator yields a diagonal term; equality constraints (slots of
    ``constraints=``) made of FrameTasks / RelativeFrameTasks."""
    from .tasks.frame_task import FrameTask
    from .tasks.linear_holonomic_task import JointVelocityTask
    from .tasks.posture_task import DampingTask, LowAccelerationTask, PostureTask
    from .tasks.relative_frame_task import RelativeFrameTask

    if not hasattr(kin_model, "joints"):
        return None
    for col in constraints or ():  # equality constraints (pink/solve_ik.py:125-149): frame tasks only, one frame per slot
        c0 = col[0]
        if type(c0) not in (FrameTask, RelativeFrameTask):
            return None
        if not getattr(col, "shared", False) and any(
                type(t) is not type(c0) or t.frame != c0.frame or getattr(t, "root", None) != getattr(c0, "root", None) or t.gain != c0.gain
                for t in col):
            return None
    if constraints and len(constraints) > MAX_FRAME_TASKS:
        return None
    frames = [k for k, col in enumerate(slots) if type(col[0]) in (FrameTask, RelativeFrameTask)]
    for k in frames:  # (one frame -- and root -- per slot: the device model holds them)
        t0 = slots[k][0]
        if getattr(slots[k], "shared", False):  # (one task object for the whole batch: nothing to compare)
            continue
        if any(type(t) is not type(t0) or t.frame != t0.frame or getattr(t, "root", None) != getattr(t0, "root", None) for t in slots[k]):
            return None
    if not frames or len(frames) > MAX_FRAME_TASKS:
        return None
    for k, col in enumerate(slots):
        if k not in frames and type(col[0]) not in (PostureTask, DampingTask, LowAccelerationTask, JointVelocityTask):
            return None
    return frames


class HybridState:
    """Device buffers and model tables of one call shape (kept by :func:`pink_amd.solve_ik.solve_ik_batch` like the
    device-resident states)."""

    def __init__(self, api, model, frames: Sequence[str], B: int):
        from .rollout import ModelArrays

        self.api, self.model, self.B = api, model, int(B)
        self.arrays = ModelArrays(model, list(frames))
        self.dmodel = api.model_create(self.arrays.desc)
        self.bufs = {}
        self.cframes, self.carrays, self.cmodel = None, None, None  # device model of the equality constraints' frames

    def constraint_model(self, frames) -> int:
        frames = tuple(frames)
        if self.cframes != frames:
            from .rollout import ModelArrays

            if self.cmodel is not None:
                self.api.model_destroy(self.cmodel)
            self.carrays = ModelArrays(self.model, list(frames))
            self.cmodel = self.api.model_create(self.carrays.desc)
            self.cframes = frames
        return self.cmodel

    def buf(self, name: str, nbytes: int) -> int:
        have = self.bufs.get(name)
        if have is None or have[1] < nbytes:
            if have is not None:
                self.api.release(have[0])
            have = (self.api.alloc(max(int(nbytes), 8)), int(nbytes))
            self.bufs[name] = have
        return have[0]

    def free(self) -> None:
        for ptr, _ in self.bufs.values():
            self.api.release(ptr)
        self.bufs = {}
        if self.cmodel is not None:
            self.api.model_destroy(self.cmodel)
            self.cmodel = None
        self.api.model_destroy(self.dmodel)


def solve(state: HybridState, kin_q: np.ndarray, kin, slots: Sequence, frame_slots: List[int], limits, barriers, dt: float,
          damping: float, max_iter: int, constraints: Sequence = ()):
    """One batched solve on the hybrid route; returns ``(dq, status, iters, path)``.  ``kin`` is a callable that
    yields the host :class:`~pink_amd.kinematics_batch.BatchKinematics` -- forward kinematics on the host are only run
    if something asks for them (barriers, limits without a batched evaluator)."""
    api, model, B = state.api, state.model, state.B
    nv, nq, nf = model.nv, model.nq, len(frame_slots)
    kq = _QOnly(model, kin_q, kin)
    # diagonal tasks, limits, barriers: host (vectorised, pink_amd/batch_eval.py)
    diag = [be.task_term(kq, col, None) for k, col in enumerate(slots) if k not in frame_slots]
    if any(not isinstance(t, DiagonalTaskTerm) for t in diag):
        raise PinkError("hybrid route: a task outside the FrameTasks produced a dense Jacobian")
    lb = np.full((B, nv), -np.inf)
    ub = np.full((B, nv), np.inf)
    dense_rows = []
    for limit in limits:
        rows = be.limit_rows(kq, limit, dt, lb, ub)
        if rows is not None:
            dense_rows.append(rows)
    barrier_terms = [be.barrier_term(kq, bar) for bar in (barriers or [])]
    # equality constraints made of frame tasks (pink/solve_ik.py:125-149: A = J, b = -gain e): their rows are the leading
    # dense rows of the QP; the device writes them there (below), the host only reserves the space
    ccols = list(constraints or ())
    n_eq = 6 * len(ccols)
    eq = [(np.zeros((B, n_eq, nv)), np.zeros((B, n_eq)))] if n_eq else ()
    b0 = pack_terms(nv, diag, dt, damping, boxes=[(lb, ub)], dense_rows=dense_rows, barriers=barrier_terms, batch_size=B,
                    equality_rows=eq)
    # the FrameTask part of the descriptor: one dense task of six rows per slot, in slot order
    fcols = [slots[k] for k in frame_slots]
    for col in fcols:  # (gain / lm_damping are read off the first task of a slot: the same refusal as the host route)
        be._check_slot(col)
    Kd, K = 6 * nf, 6 * nf + b0.K
    fcost = [be._costs(col, 6) for col in fcols]
    batched = b0.cost.ndim == 2 or any(np.ndim(c) == 2 for c in fcost)
    if batched:
        cost = np.concatenate([np.broadcast_to(np.asarray(c, dtype=float), (B, 6)) for c in fcost]
                              + [np.broadcast_to(b0.cost, (B, b0.K))], axis=1)
    else:
        cost = np.concatenate([np.broadcast_to(np.asarray(c, dtype=float), (6,)) for c in fcost] + [b0.cost])
    cost = np.ascontiguousarray(cost)
    e_full = np.zeros((B, K))  # (the kernel overwrites the FrameTask rows: one contiguous upload instead of a strided one)
    e_full[:, Kd:] = b0.e
    # (a RelativeFrameTask's target lives in its root frame: a relative slot of the device model, include/pinkhip.h)
    targets = np.ascontiguousarray(np.stack([_targets_of(col, B) for col in fcols], axis=1))  # [B, nf, 12]
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)  # noqa: E731
    f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)  # noqa: E731
    task_rows = i32([6 * i for i in range(nf + 1)] + [Kd + int(r) for r in b0.task_rows[1:]])
    task_kind = i32([0] * nf + list(b0.task_kind))
    task_col0 = i32([0] * nf + list(b0.task_col0))
    gain = f64([col[0].gain for col in fcols] + list(b0.gain))
    lm = f64([col[0].lm_damping for col in fcols] + list(b0.lm_damping))
    brow, bsafe = i32(b0.barrier_rows), f64(b0.barrier_safe_gain if b0.barrier_safe_gain.size else [0.0])
    d = Desc()
    d.B, d.nv, d.T, d.Kd, d.K, d.md, d.n_eq = B, nv, nf + len(diag), Kd, K, b0.md, n_eq
    d.task_rows, d.task_kind, d.task_col0 = (a.ctypes.data_as(c_int32_p) for a in (task_rows, task_kind, task_col0))
    d.gain, d.lm_damping = gain.ctypes.data_as(c_double_p), lm.ctypes.data_as(c_double_p)
    d.n_barriers = int(b0.barrier_safe_gain.size)
    d.barrier_rows, d.barrier_safe_gain = brow.ctypes.data_as(c_int32_p), bsafe.ctypes.data_as(c_double_p)
    d.damping, d.dt, d.cost_is_batched, d.max_iter = float(damping), float(dt), int(batched), int(max_iter)
    # device side
    s = state
    d_q, d_Tt = s.buf("q", 8 * B * nq), s.buf("Tt", 8 * B * nf * 12)
    d_J, d_e, d_cost = s.buf("J", 8 * B * Kd * nv), s.buf("e", 8 * B * K), s.buf("cost", cost.nbytes)
    d_lb, d_ub = s.buf("lb", 8 * B * nv), s.buf("ub", 8 * B * nv)
    d_dq, d_st, d_it = s.buf("dq", 8 * B * nv), s.buf("status", 4 * B), s.buf("iters", 4 * B)
    api.put(d_q, np.ascontiguousarray(kin_q))
    api.put(d_Tt, targets)
    api.put(d_e, e_full)
    api.put(d_cost, cost)
    api.put(d_lb, b0.lb)
    api.put(d_ub, b0.ub)
    p = Problem()
    p.J, p.e, p.cost, p.lb, p.ub = d_J, d_e, d_cost, d_lb, d_ub
    p.Gd = p.hd = p.c_extra = None
    if b0.md:
        p.Gd, p.hd = s.buf("Gd", b0.Gd.nbytes), s.buf("hd", b0.hd.nbytes)
        api.put(p.Gd, b0.Gd)
        api.put(p.hd, b0.hd)
    if b0.c_extra is not None:
        p.c_extra = s.buf("c_extra", b0.c_extra.nbytes)
        api.put(p.c_extra, b0.c_extra)
    api.fk_frame_tasks(s.dmodel, B, d_q, d_Tt, None, d_e, K, d_J, Kd * nv)
    if n_eq:
        # the same kernel on the constraints' frames, writing J into the leading rows of Gd and e into those of hd (row
        # pitches md nv and md); hd then becomes -gain e on the host: [B, md] doubles go home and back
        cmodel = s.constraint_model(_frame_name(col[0]) for col in ccols)
        ctargets = np.ascontiguousarray(np.stack([_targets_of(col, B) for col in ccols], axis=1))
        d_Tc = s.buf("Ttc", ctargets.nbytes)
        api.put(d_Tc, ctargets)
        api.fk_frame_tasks(cmodel, B, d_q, d_Tc, None, p.hd, b0.md, p.Gd, b0.md * nv)
        api.sync()
        hd = np.empty((B, b0.md))
        api.get(hd, p.hd)
        for k, col in enumerate(ccols):
            hd[:, 6 * k:6 * k + 6] *= -float(col[0].gain)
        api.put(p.hd, hd)
    r = Result()
    r.dq, r.status, r.iters = d_dq, d_st, d_it
    api.solve_raw(d, p, r)
    api.sync()
    dq, st, it = np.empty((B, nv)), np.empty(B, np.int32), np.empty(B, np.int32)
    api.get(dq, d_dq)
    api.get(st, d_st)
    api.get(it, d_it)
    return dq, st, it, split_iters(it)


def _frame_name(task):
    """Entry of a device model's frame list: the frame, or ``(frame, root)`` for a relative slot."""
    return (task.frame, task.root) if hasattr(task, "root") else task.frame


def _targets_of(col, B: int) -> np.ndarray:
    """``[B, 12]`` targets of a frame-task slot (a RelativeFrameTask's lives in its root frame: a relative slot)."""
    if hasattr(col[0], "root"):
        return be._frame_targets(col, B, "transform_target_to_root", "target pose of frame '{0.frame}' in frame '{0.root}' is undefined")
    return be._frame_targets(col, B, "transform_target_to_world", "no target set for frame '{0.frame}'")


class _QOnly:
    """What the batched evaluators read of a :class:`~pink_amd.kinematics_batch.BatchKinematics` when the stack
    needs no forward kinematics on the host (diagonal tasks, joint-space limits): ``q``, the model and the
    manifold difference.  Anything else builds the real thing on first use."""

    def __init__(self, model, q, make):
        self.model, self.q, self.B, self._make, self._kin = model, q, q.shape[0], make, None

    def _full(self):
        if self._kin is None:
            self._kin = self._make()
        return self._kin

    def difference(self, q0, q1, from_v: int = 0):
        from .kinematics_batch import BatchKinematics

        return BatchKinematics.difference(self, q0, q1, from_v)  # (reads the model and q only: no forward kinematics)

    def __getattr__(self, name):
        return getattr(self._full(), name)
