"""Closed-loop differential IK on the device.

The reference iterates ``solve_ik`` + ``configuration.integrate_inplace`` on the host, one
robot at a time (``examples/inverse_kinematics_ur10.py:75-91``, ``tests/test_solve_ik.py
:160-210``).  ``DeviceRollout`` keeps ``B`` robots resident in HBM and runs every stage of a
step as a HIP kernel, with no host round trip between steps:

    q <- q (+) dq of the previous step, forward kinematics, FrameTask e and J,
    box limits, PostureTask error            pinkhip_step_device (ONE launch)
    stack + QP solve                         pinkhip_solve_device

i.e. two launches per control step; with ``fused="kernel"`` the two become ONE
(``pinkhip_rollout_step_device``: the task Jacobians are formed on chip while the objective is stacked and never
reach HBM; the integration closes the same launch).  ``fused=False`` keeps the five separate launches: pinkhip_fk_device, one
pinkhip_frame_task_strided_device per task, pinkhip_limits_posture_device, pinkhip_solve_device,
pinkhip_integrate_checked_device -- kept for A/B runs and as a cross-check of the fusion).

The task stack is the one of Pink's humanoid / arm examples: any number of FrameTasks plus an
optional PostureTask, the model's configuration and velocity limits.
"""

from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from ._lib import Desc, Problem, Result, RolloutStep, Step, c_double_p, c_int32_p
from .configuration import Model
from .exceptions import NoSolutionFound, NotWithinConfigurationLimits, PinkError
from .utils import get_root_joint_dim

# Relative sizes of the ranges one pipelined solve is cut into (DeviceRollout.solve_pipelined).  Four equal ranges:
# ranges that shrink towards the end (less to wait for after the last byte went up) or more of them measured 3-5 %
# SLOWER at B = 65 536 (profiles/prof_pipeline_r04.txt: every range costs five uploads, a launch and three downloads of
# fixed overhead, and a short range fills a third of the chip).  PINKHIP_PIPELINE_SPLIT="30,27,22,13,8" overrides.
import os as _os

PIPELINE_SPLIT = tuple(float(v) for v in _os.environ.get("PINKHIP_PIPELINE_SPLIT", "1,1,1,1").split(","))

_JT = {"revolute": 0, "prismatic": 1, "free_flyer": 2}


class NoWholeStepKernel(PinkError):
    """No instantiation of the whole-step kernel holds this model with its dense rows (``dispatch.h``:
    ``PINKHIP_ROLLOUT_DENSE_TABLE``); ``solve_ik_batch`` then evaluates the batch on the host-evaluated path."""


class ModelDesc(ctypes.Structure):
    """``pinkhip_model_desc``."""

    _fields_ = [
        ("nj", ctypes.c_int32), ("nq", ctypes.c_int32), ("nv", ctypes.c_int32), ("nf", ctypes.c_int32),
        ("root_nv", ctypes.c_int32),
        ("parent", c_int32_p), ("jtype", c_int32_p), ("idx_q", c_int32_p), ("idx_v", c_int32_p),
        ("placement", c_double_p), ("axis", c_double_p), ("frame_joint", c_int32_p),
        ("frame_placement", c_double_p), ("q_min", c_double_p), ("q_max", c_double_p), ("v_max", c_double_p),
        ("frame_root_joint", c_int32_p), ("frame_root_placement", c_double_p),
    ]


def pose12(T) -> np.ndarray:
    """12-double pose: rotation row-major, then translation."""
    return np.hstack([np.asarray(T.rotation, dtype=float).ravel(), np.asarray(T.translation, dtype=float)])


class ModelArrays:
    """NumPy tables of a :class:`pink_amd.configuration.Model` for ``pinkhip_model_create``;
    ``frames`` lists the frame names the rollout needs (their order is the frame index).  An entry ``(frame, root)``
    is a relative slot: the pose of ``frame`` in ``root`` (``pink/tasks/relative_frame_task.py``)."""

    def __init__(self, model: Model, frames: Sequence, velocity_limit: Optional[np.ndarray] = None):
        self.model = model
        self.roots = [f[1] if isinstance(f, tuple) else None for f in frames]
        self.frames = [f[0] if isinstance(f, tuple) else f for f in frames]
        js = model.joints
        i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)  # noqa: E731
        self.parent = i32([j.parent for j in js])
        self.jtype = i32([_JT[j.kind] for j in js])
        self.idx_q = i32([j.idx_q for j in js])
        self.idx_v = i32([j.idx_v for j in js])
        self.placement = np.ascontiguousarray([pose12(j.placement) for j in js], dtype=np.float64)
        self.axis = np.ascontiguousarray([np.zeros(3) if j.axis is None else j.axis for j in js], dtype=np.float64)
        fr = [model.frames[model.getFrameId(n)] for n in self.frames]
        self.frame_joint = i32([f.joint for f in fr]) if fr else np.zeros(1, np.int32)
        self.frame_placement = (np.ascontiguousarray([pose12(f.placement) for f in fr], dtype=np.float64)
                                if fr else np.zeros((1, 12)))
        self.q_min = np.ascontiguousarray(model.lowerPositionLimit, dtype=np.float64)
        self.q_max = np.ascontiguousarray(model.upperPositionLimit, dtype=np.float64)
        # (a VelocityLimit with its own vector, pink/limits/velocity_limit.py:46-58: the table the kernels read)
        self.v_max = np.ascontiguousarray(model.velocityLimit if velocity_limit is None else velocity_limit, dtype=np.float64)
        if self.v_max.shape != (model.nv,):
            raise ValueError(f"velocity_limit must have shape ({model.nv},)")
        d = ModelDesc()
        d.nj, d.nq, d.nv, d.nf = len(js), model.nq, model.nv, len(fr)
        d.root_nv = get_root_joint_dim(model)[1]
        d.parent = self.parent.ctypes.data_as(c_int32_p)
        d.jtype = self.jtype.ctypes.data_as(c_int32_p)
        d.idx_q = self.idx_q.ctypes.data_as(c_int32_p)
        d.idx_v = self.idx_v.ctypes.data_as(c_int32_p)
        d.placement = self.placement.ctypes.data_as(c_double_p)
        d.axis = self.axis.ctypes.data_as(c_double_p)
        d.frame_joint = self.frame_joint.ctypes.data_as(c_int32_p)
        d.frame_placement = self.frame_placement.ctypes.data_as(c_double_p)
        d.q_min = self.q_min.ctypes.data_as(c_double_p)
        d.q_max = self.q_max.ctypes.data_as(c_double_p)
        d.v_max = self.v_max.ctypes.data_as(c_double_p)
        if any(r is not None for r in self.roots):
            rf = [None if r is None else model.frames[model.getFrameId(r)] for r in self.roots]
            self.frame_root_joint = i32([-2 if f is None else f.joint for f in rf])
            self.frame_root_placement = np.ascontiguousarray([np.zeros(12) if f is None else pose12(f.placement) for f in rf], dtype=np.float64)
            d.frame_root_joint = self.frame_root_joint.ctypes.data_as(c_int32_p)
            d.frame_root_placement = self.frame_root_placement.ctypes.data_as(c_double_p)
        self.desc = d


def _floating_base_rows(model, limit, dt: float):
    """``(root_box [12], rows [n, 6], h [n])`` of a :class:`~pink_amd.limits.FloatingBaseVelocityLimit`
    (``pink/limits/floating_base_velocity_limit.py:104-148``): ``+-A dq_root <= dt twist_max`` with ``A`` the adjoint
    of the inverse placement of the limit's frame on the root joint -- the body Jacobian of that frame on the root's
    tangent coordinates, whatever the configuration.  ``(None, ...)`` when no component is bounded."""
    from .batch import split_box_rows
    from .configuration import _adjoint

    finite = np.isfinite(limit.twist_max)
    if not finite.any():
        return None, np.zeros((0, 6)), np.zeros(0)
    if limit.root_nv != 6:
        raise ValueError("the floating-base velocity limit on the device needs a free-flyer root joint")
    frame = model.frames[model.getFrameId(limit.base_frame)]
    A = _adjoint(frame.placement.inverse())[finite]
    bounds = float(dt) * limit.twist_max[finite]
    lb, ub, G, h = split_box_rows(np.vstack([A, -A]), np.hstack([bounds, bounds]), 6)
    return np.ascontiguousarray(np.hstack([lb, ub])), np.ascontiguousarray(G), np.ascontiguousarray(h)


class DeviceRollout:
    """``B`` robots iterating differential IK on the device.

    ``api`` is a :class:`pink_amd.batch_solver.BatchSolver` (or anything with the same raw-pointer
    methods: the test suite passes the CPU wave emulator).  ``frame_tasks`` is a list of
    ``(frame_name, position_cost, orientation_cost, gain, lm_damping)``; ``posture_cost`` enables a
    PostureTask toward ``q_posture`` (default: the initial configuration of each robot).

    A robot whose QP fails at some step (``status != 0``) is frozen: the failed ``dq`` is never integrated
    (the reference raises ``NoSolutionFound`` before integrating, ``pink/solve_ik.py:271-275``) and the first
    failure is remembered per robot on the device; :meth:`run` raises once the steps are done
    (``raise_on_failure=False``: inspect :meth:`failures`).  ``safety_break`` checks the initial
    configurations against the joint limits like ``solve_ik`` does (``pink/solve_ik.py:260``); later
    configurations stay inside them by construction (the QP's box is ``gain * (q_limit - q)``).
    """

    _BUFFERS = ("d_q", "d_T", "d_Jb", "d_Tt", "d_J", "d_e", "d_cost", "d_lb", "d_ub", "d_dq", "d_status", "d_iters", "d_qt",
                "d_fail", "d_Tq")

    def __init__(self, api, model: Model, q0: np.ndarray, frame_tasks: Sequence[tuple], dt: float,
                 posture_cost: Optional[float] = None, posture_gain: float = 1.0, damping: float = 1e-12,
                 config_limit_gain: float = 0.5, q_posture: Optional[np.ndarray] = None, max_iter: int = 0,
                 fused: bool = True, safety_break: bool = True, posture_lm_damping: float = 0.0,
                 position_barriers: Sequence = (), floating_base_limit=None, const_tasks: Sequence = (),
                 diag_tasks: Sequence = (), acceleration_limit: Optional[np.ndarray] = None,
                 velocity_limit: Optional[np.ndarray] = None, constraint_slots: Sequence = ()):
        """``const_tasks``: dense tasks with a constant Jacobian, ``(A [k, nv], b [k], q_0 [nq], cost, gain, lm_damping)``
        each (LinearHolonomicTask / JointCouplingTask on vector-space joints); ``diag_tasks``: identity-Jacobian tasks
        with batch-constant errors, ``(col0, e [k], cost, gain, lm_damping)`` each (DampingTask, LowAccelerationTask,
        JointVelocityTask).  ``acceleration_limit``: ``[3, nv]`` -- ``a_max`` (0: no bound on that coordinate),
        ``Delta_q_prev``, ``has_configuration_limit`` per tangent coordinate of an AccelerationLimit on the joints behind
        the root (``pink/limits/acceleration_limit.py:158-199``), folded into the box on chip; ``velocity_limit``: the
        vector of a VelocityLimit built with its own numbers (the device model then carries it).  The first three need the
        whole-step kernel (``fused="kernel"``).  ``position_barriers`` takes PositionBarriers and BodySphericalBarriers
        (``pink/barriers/body_spherical_barrier.py:73-143``) in Pink's order; their frames are slots of ``frame_tasks``
        (zero costs for a frame that carries no task).  ``constraint_slots``: equality constraints made of frame tasks,
        ``(slot, gain)`` each (``pink/solve_ik.py:125-149``: ``A = J``, ``b = -gain e`` of the FrameTask of that slot of
        ``frame_tasks``, whose costs are zero when it is a constraint only); at most two.  The tables are the same for every robot and stay what they are over
        :meth:`run`: state that follows the previous step of each robot (``LowAccelerationTask.set_last_integration``,
        ``AccelerationLimit.set_last_integration``) is the caller's to refresh between steps
        (:meth:`set_diag_errors`, :meth:`set_acceleration_limit`) -- :func:`pink_amd.solve_ik_batch` does so per call."""
        self.api, self.model, self.dt = api, model, float(dt)
        # "kernel": the whole step in one launch; True: step kernel + solve; False: five separate launches
        self.fused = fused if fused == "kernel" else bool(fused)
        self.B = B = int(q0.shape[0])
        self.nv, self.nq = model.nv, model.nq
        # (a frame task (frame, ...) regulates the frame in the world; ((frame, root), ...) the pose of frame in root --
        # a RelativeFrameTask, pink/tasks/relative_frame_task.py: a relative slot of the device model)
        self.arrays = ModelArrays(model, [ft[0] for ft in frame_tasks], velocity_limit=velocity_limit)
        self.frames = self.arrays.frames
        if any(r is not None for r in self.arrays.roots) and self.fused != "kernel":
            raise ValueError('relative frame tasks need the whole-step kernel: fused="kernel"')
        self.dmodel = api.model_create(self.arrays.desc)
        nf, nv, nq = len(self.frames), self.nv, self.nq
        root_nv = get_root_joint_dim(model)[1]
        if (const_tasks or diag_tasks or acceleration_limit is not None) and self.fused != "kernel":
            raise ValueError('constant-row tasks, extra diagonal tasks and an acceleration limit need the whole-step kernel: fused="kernel"')
        self.d_acc = None
        if acceleration_limit is not None:
            self._acc = np.ascontiguousarray(acceleration_limit, dtype=np.float64)
            if self._acc.shape != (3, nv):
                raise ValueError(f"acceleration_limit must be [3, {nv}]")
            self.d_acc = api.alloc(self._acc.nbytes)
            api.put(self.d_acc, self._acc)
        const_tasks = [(np.atleast_2d(np.asarray(A, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64)),
                        np.asarray(q_0, dtype=np.float64), c, g, l) for A, b, q_0, c, g, l in const_tasks]
        diag_tasks = [(int(c0), np.atleast_1d(np.asarray(e, dtype=np.float64)), c, g, l) for c0, e, c, g, l in diag_tasks]
        self.n_crow = sum(A.shape[0] for A, *_ in const_tasks)
        if any(A.shape[1] != nv or q_0.shape != (nq,) for A, _, q_0, *_ in const_tasks):
            raise ValueError("constant-row tasks: A must be [k, nv] and q_0 [nq]")
        if self.n_crow and any(not np.array_equal(q_0, const_tasks[0][2]) for _, _, q_0, *_ in const_tasks):
            raise ValueError("constant-row tasks must share one reference configuration q_0")
        self.Kd = 6 * nf + self.n_crow
        n_post = nv - root_nv if posture_cost is not None else 0
        self.K = self.Kd + n_post + sum(e.shape[0] for _, e, *_ in diag_tasks)
        # task tables of the QP (include/pinkhip.h: frame tasks, constant-row tasks, then the diagonal ones: the posture first)
        rows, kind, col0, gain, lm, cost = [0], [], [], [], [], []
        for ft in frame_tasks:
            rows.append(rows[-1] + 6), kind.append(0), col0.append(0), gain.append(ft[3]), lm.append(ft[4])
            cost += list(np.broadcast_to(np.asarray(ft[1], float), (3,))) + list(np.broadcast_to(np.asarray(ft[2], float), (3,)))
        for A, b, _, c, g, l in const_tasks:
            k = A.shape[0]
            rows.append(rows[-1] + k), kind.append(0), col0.append(0), gain.append(g), lm.append(l)
            cost += list(np.broadcast_to(np.ones(k) if c is None else np.asarray(c, float), (k,)))
        if n_post:
            rows.append(rows[-1] + n_post), kind.append(1), col0.append(root_nv), gain.append(posture_gain), lm.append(posture_lm_damping)
            cost += [float(posture_cost)] * n_post
        for c0, e, c, g, l in diag_tasks:
            k = e.shape[0]
            if c0 < 0 or c0 + k > nv:
                raise ValueError("diagonal task exceeds the tangent space")
            rows.append(rows[-1] + k), kind.append(1), col0.append(c0), gain.append(g), lm.append(l)
            cost += list(np.broadcast_to(np.ones(k) if c is None else np.asarray(c, float), (k,)))
        T = len(kind)
        self.task_rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.task_kind = np.ascontiguousarray(kind if kind else [0], dtype=np.int32)
        self.task_col0 = np.ascontiguousarray(col0 if col0 else [0], dtype=np.int32)
        self.gain = np.ascontiguousarray(gain if gain else [0.0], dtype=np.float64)
        self.lm = np.ascontiguousarray(lm if lm else [0.0], dtype=np.float64)
        self.cost = np.ascontiguousarray(cost if cost else [0.0], dtype=np.float64)
        self.posture_task = 0 if n_post else -1
        # errors of the diagonal rows behind the dense block (the posture's rows are formed on chip: left at zero)
        self._diag_e = np.zeros(max(self.K - self.Kd, 1))
        off = n_post
        for _, e, *_ in diag_tasks:
            self._diag_e[off:off + e.shape[0]] = e
            off += e.shape[0]
        self._const = None
        if self.n_crow:
            self._const = (np.ascontiguousarray(np.vstack([A for A, *_ in const_tasks])),
                           np.ascontiguousarray(const_tasks[0][2]), np.ascontiguousarray(np.hstack([b for _, b, *_ in const_tasks])))
        self._extra_tasks = bool(const_tasks or diag_tasks)
        # position barriers (pink/barriers/position_barrier.py): one dense row per (index, side), rows in Pink's order
        # (p_min rows, then p_max rows); formed on chip by the whole-step kernel (fused="kernel" only)
        # FloatingBaseVelocityLimit (pink/limits/floating_base_velocity_limit.py:104-148): the Jacobian of a frame attached
        # to the root joint, on the root's tangent coordinates, is the constant adjoint of the frame's placement -- its
        # axis-aligned rows become a box on those coordinates, the others the first dense rows (constant too)
        self.root_box, self.lim_rows, self.lim_h = None, np.zeros((0, 6)), np.zeros(0)
        if floating_base_limit is not None:
            self.root_box, self.lim_rows, self.lim_h = _floating_base_rows(model, floating_base_limit, self.dt)
        # a velocity vector with finite entries on the free-flyer's tangent coordinates (a VelocityLimit built with its own
        # vector, pink/limits/velocity_limit.py:46-73,118-121: rows +-e_i dq <= dt v_i like any joint) joins the same box
        # (a joint is velocity-limited only when ALL its tangent coordinates are, velocity_limit.py:66-74)
        v_root = self.arrays.v_max[:6] if root_nv == 6 else np.zeros(0)
        if root_nv == 6 and bool(((v_root < 1e20) & (v_root > 1e-10)).all()):
            box = np.hstack([-self.dt * v_root, self.dt * v_root])
            if self.root_box is None:
                self.root_box = np.ascontiguousarray(box)
            else:
                self.root_box = np.ascontiguousarray(np.hstack([np.maximum(self.root_box[:6], box[:6]), np.minimum(self.root_box[6:], box[6:])]))
        if self.fused is False and self.root_box is not None:
            raise ValueError("a box on the floating base's coordinates needs fused=True or fused=\"kernel\"")
        n_lim = len(self.lim_h)
        # equality constraints made of frame tasks: the leading dense rows (six per constraint)
        self.cons = [(int(s_), float(g_)) for s_, g_ in constraint_slots]
        if len(self.cons) > 2 or any(s_ < 0 or s_ >= nf for s_, _ in self.cons):
            raise ValueError("at most two constraint slots, each a slot of frame_tasks")
        if self.cons and self.fused != "kernel":
            raise ValueError('equality constraints need the whole-step kernel: fused="kernel"')
        n_eq = 6 * len(self.cons)
        self.n_eq = n_eq

        def plain_slot(name, what):
            # (the frame's world pose and Jacobian: an ordinary slot -- a relative slot carries a signed ancestor table)
            plain = [i for i, (n, r) in enumerate(zip(self.frames, self.arrays.roots)) if n == name and r is None]
            if not plain:
                raise ValueError(f"{what} on frame {name!r}: the frame must be a slot of frame_tasks")
            return plain[0]

        bf, ba, bs, bb, bg, bf2, brow, bsafe = [], [], [], [], [], [], [n_eq + n_lim], []
        for bar in position_barriers:
            if hasattr(bar, "frames"):  # BodySphericalBarrier: one row, axis 3, bound d_min^2 (class-K function h / (1 + |h|))
                f1, f2 = (plain_slot(n, "spherical barrier") for n in bar.frames)
                bf.append(f1), bf2.append(f2), ba.append(3), bs.append(1.0), bb.append(float(bar.d_min) ** 2)
                bg.append(float(np.asarray(bar.gain, dtype=float).ravel()[0]))
                brow.append(n_eq + n_lim + len(bf))
                bsafe.append(float(bar.safe_displacement_gain))
                continue
            if not getattr(bar, "identity_gain_function", False):
                raise ValueError("position barriers on the device use the identity class-K function (the default)")
            f = plain_slot(bar.frame, "position barrier")
            gains = np.asarray(bar.gain, dtype=float)
            k = 0
            for sign, bound in ((1.0, bar.p_min), (-1.0, bar.p_max)):
                if bound is None:
                    continue
                for i, idx in enumerate(bar.indices):
                    bf.append(f), bf2.append(f), ba.append(int(idx)), bs.append(sign), bb.append(float(np.asarray(bound)[i])), bg.append(float(gains[k]))
                    k += 1
            brow.append(n_eq + n_lim + len(bf))
            bsafe.append(float(bar.safe_displacement_gain))
        self.md = n_eq + n_lim + len(bf)
        if self.md and self.fused != "kernel":
            raise ValueError('position barriers and dense floating-base limit rows need the whole-step kernel: fused="kernel"')
        self.brow = np.ascontiguousarray(brow, dtype=np.int32)
        self.bsafe = np.ascontiguousarray(bsafe if bsafe else [0.0], dtype=np.float64)
        self._bar_host = [np.ascontiguousarray(v if v else [0], dtype=t) for v, t in
                          ((bf, np.int32), (ba, np.int32), (bs, np.float64), (bb, np.float64), (bg, np.float64), (bf2, np.int32))]
        self._n_bar_rows = len(bf)
        d = Desc()
        d.B, d.nv, d.T, d.Kd, d.K, d.md, d.n_eq = B, nv, T, self.Kd, self.K, self.md, n_eq
        d.task_rows = self.task_rows.ctypes.data_as(c_int32_p)
        d.task_kind = self.task_kind.ctypes.data_as(c_int32_p)
        d.task_col0 = self.task_col0.ctypes.data_as(c_int32_p)
        d.gain = self.gain.ctypes.data_as(c_double_p)
        d.lm_damping = self.lm.ctypes.data_as(c_double_p)
        d.n_barriers = len(position_barriers)
        d.barrier_rows = self.brow.ctypes.data_as(c_int32_p)
        d.barrier_safe_gain = self.bsafe.ctypes.data_as(c_double_p)
        d.damping, d.dt, d.cost_is_batched, d.max_iter = float(damping), self.dt, 0, int(max_iter)
        self.desc = d
        self.config_limit_gain = float(config_limit_gain)
        self.n_post = n_post
        a = api
        f8 = lambda *shape: a.alloc(8 * max(int(np.prod(shape)), 1))  # noqa: E731
        self.d_q = f8(B, nq)
        self.d_T = f8(B, max(nf, 1), 12)
        self.d_Jb = f8(B, max(nf, 1), 6, nv)
        self.d_Tt = f8(B, max(nf, 1), 12)
        self.d_J = f8(B, max(self.Kd, 1), nv)
        self.d_e = f8(B, max(self.K, 1))
        self.d_cost = f8(max(self.K, 1))
        self.d_lb, self.d_ub, self.d_dq = f8(B, nv), f8(B, nv), f8(B, nv)
        self.d_status, self.d_iters = a.alloc(4 * B), a.alloc(4 * B)
        self.d_Tq = None  # staging of [B, 7] translation + quaternion targets, on first use (_pq_stage)
        self.d_fail = a.alloc(4 * B)  # per robot: status | (step << 8) of its first failing step, 0 = none
        a.put(self.d_fail, np.zeros(B, dtype=np.int32))
        self._fail_dirty = False
        self.d_qt = f8(B, nq)
        self.d_bar, self.d_lim, self.d_extra = [], [], []
        if self._extra_tasks:  # tables of the constant-row tasks and the batch-constant errors of the extra diagonal tasks
            for arr in ([self._diag_e] + (list(self._const) if self._const else [])):
                ptr = a.alloc(max(arr.nbytes, 8))
                a.put(ptr, arr)
                self.d_extra.append(ptr)
        for arr in ([] if self.root_box is None else [self.root_box, np.ascontiguousarray(self.lim_rows), self.lim_h]):
            ptr = a.alloc(max(arr.nbytes, 8))
            if arr.nbytes:
                a.put(ptr, arr)
            self.d_lim.append(ptr)
        if self._n_bar_rows:
            for arr in self._bar_host:
                ptr = a.alloc(max(arr.nbytes, 8))
                a.put(ptr, arr)
                self.d_bar.append(ptr)
        self.d_cons = []
        if self.cons:
            for arr in (np.ascontiguousarray([s_ for s_, _ in self.cons], dtype=np.int32), np.ascontiguousarray([g_ for _, g_ in self.cons], dtype=np.float64)):
                ptr = a.alloc(max(arr.nbytes, 8))
                a.put(ptr, arr)
                self.d_cons.append(ptr)
        self._resident = {}  # frame slot -> token of the frozen target array d_Tt holds for it (FrameTask.freeze_targets)
        self.qt_batched = 1  # d_qt holds [B, nq] (one posture target per robot) or [nq] (one for all)
        self.targets_per_frame = False  # d_Tt holds [B, nf, 12], or one [B, 12] array per frame
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        a.put(self.d_q, q0)
        self._put_posture(q0, q_posture)
        a.put(self.d_cost, self.cost)
        try:
            self._check_limits_device(q0, safety_break)
        except BaseException:
            self.free()
            raise
        p = Problem()
        p.J, p.e, p.cost, p.lb, p.ub = self.d_J, self.d_e, self.d_cost, self.d_lb, self.d_ub
        p.Gd, p.hd, p.c_extra = None, None, None
        self.problem = p
        r = Result()
        r.dq, r.status, r.iters = self.d_dq, self.d_status, self.d_iters
        self.result = r
        self.steps_done = 0
        self._pending = False  # a solved dq waits to be integrated by the next whole-step launch

    def reset(self, q0: np.ndarray, q_posture: Optional[np.ndarray] = None, safety_break: bool = True) -> None:
        """New initial configurations (and posture targets) for the same robots / task stack: buffers, model tables
        and descriptors are kept (``solve_ik_batch`` re-uses one rollout per call shape)."""
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        if q0.shape != (self.B, self.nq):
            raise ValueError(f"q0 must have shape {(self.B, self.nq)}, got {q0.shape}")
        a = self.api
        a.put(self.d_q, q0)
        self._check_limits_device(q0, safety_break)
        if self.n_post:
            self._put_posture(q0, q_posture)
        if self._fail_dirty:
            a.put(self.d_fail, np.zeros(self.B, dtype=np.int32))
            self._fail_dirty = False
        self.steps_done = 0
        self._pending = False

    def _put_posture(self, q0: np.ndarray, q_posture: Optional[np.ndarray], asyn: bool = False) -> None:
        """Posture target(s): ``None`` = each robot's initial configuration, ``[nq]`` = one for all robots (uploaded
        as it is: the kernels take either form, ``target_batched``), ``[B, nq]`` = one per robot.  ``asyn``: one target
        for all robots goes up on the copy stream from a page-locked mirror, without a synchronisation of its own."""
        if asyn and q_posture is not None and np.ndim(q_posture) == 1 and hasattr(self.api, "pinned_empty"):
            if getattr(self, "_h_qt", None) is None:
                self._h_qt = self.api.pinned_empty((self.nq,), np.float64)
            self._h_qt[:] = q_posture
            self.api.put_async(self.d_qt, self._h_qt)
            self.qt_batched = 0
            return
        if q_posture is None:
            self.api.put(self.d_qt, q0)
            self.qt_batched = 1
            return
        qp = np.asarray(q_posture, dtype=np.float64)
        if qp.ndim == 2 and qp.strides[0] == 0:  # a broadcast view of one vector
            qp = qp[0]
        if qp.ndim == 1:
            self.api.put(self.d_qt, np.ascontiguousarray(qp))
            self.qt_batched = 0
        else:
            self.api.put(self.d_qt, np.ascontiguousarray(np.broadcast_to(qp, (self.B, self.nq))))
            self.qt_batched = 1

    def _check_limits_device(self, q0: np.ndarray, safety_break: bool) -> None:
        """``Configuration.check_limits`` (``pink/configuration.py:166-201``) on the uploaded batch, by a device kernel
        (a vectorised host check of 65 536 x 37 entries costs 2 ms per call); the host loop only runs to report."""
        if not hasattr(self.api, "check_limits"):
            return self._check_limits(self.model, q0, safety_break)
        if self.api.check_limits(self.dmodel, self.B, self.d_q) >= 0:
            self._check_limits(self.model, q0, safety_break)

    def set_targets(self, targets) -> None:
        """Frame targets: ``[B, n_frame_tasks, 12]`` poses (rotation row-major, translation), or a list with one
        ``[B, 12]`` array per frame task (uploaded one after the other, no host-side stacking; the whole-step kernel
        addresses them through strides)."""
        nf = len(self.frames)
        if isinstance(targets, (list, tuple)):
            if len(targets) != nf:
                raise ValueError(f"{nf} frame tasks, {len(targets)} target arrays")
            if self.fused == "kernel":
                if not self.targets_per_frame:
                    self._resident = {}
                for f, t in enumerate(targets):
                    tok = getattr(t, "frozen_token", None)  # (FrameTask.freeze_targets: uploaded once per device state)
                    if tok is not None and self._resident.get(f) == tok:
                        continue
                    if np.ndim(t) == 2 and t.shape[1] == 7:  # translation + quaternion: expanded by a device kernel
                        t = np.ascontiguousarray(t, dtype=np.float64)
                        if t.shape[0] != self.B:
                            raise ValueError(f"{t.shape[0]} targets for {self.B} robots")
                        self.api.put(self._pq_stage(f), t)
                        self.api.pose_targets(self.B, self._pq_stage(f), self.d_Tt + 8 * 12 * self.B * f)
                    else:
                        t = np.ascontiguousarray(np.broadcast_to(t, (self.B, 12)), dtype=np.float64)
                        self.api.put(self.d_Tt + 8 * 12 * self.B * f, t)
                    self._resident[f] = tok
                self.targets_per_frame = True
                return
            targets = np.stack([np.broadcast_to(t, (self.B, 12)) for t in targets], axis=1)
        t = np.ascontiguousarray(targets, dtype=np.float64).reshape(self.B, nf, 12)
        self.api.put(self.d_Tt, t)
        self.targets_per_frame = False
        self._resident = {}

    def _pq_stage(self, f: int, lo: int = 0) -> int:
        """Device address of robot ``lo``'s entry in the staging area of frame slot ``f``'s ``[B, 7]`` translation + quaternion
        targets (``FrameTask.set_target_poses_quat``; allocated on first use, ``pinkhip_pose_targets_device`` writes the
        poses the kernels read from it)."""
        if getattr(self, "d_Tq", None) is None:
            if not hasattr(self.api, "pose_targets"):
                raise RuntimeError("this solver has no pinkhip_pose_targets_device: translation + quaternion targets need it")
            self.d_Tq = self.api.alloc(8 * 7 * self.B * max(len(self.frames), 1))
        return self.d_Tq + 8 * 7 * (self.B * f + lo)

    def step(self, integrate: bool = True) -> None:
        """Enqueue one IK step for every robot (asynchronous).  ``integrate=False`` only solves (dq, status and
        iteration counts of ``last_step`` are those of the current configurations, which stay as they are)."""
        a, B, nv, nf = self.api, self.B, self.nv, len(self.frames)
        self._pipelined = None
        self.scaled = False
        if integrate:
            self._fail_dirty = True
        if self.fused == "kernel" and not self._one_kernel_step(integrate):
            self.scaled = False  # (no whole-step kernel for this model: the launches below write dq unscaled)
            if self._extra_tasks or self.d_acc is not None:
                raise NoWholeStepKernel("no whole-step kernel instantiation fits this model: constant-row / extra diagonal tasks / the acceleration limit need it")
            if self.md:
                raise NoWholeStepKernel("no whole-step kernel instantiation with barrier rows fits this model (nv, rows, joints)")
            self.fused = True  # no instantiation for this model: two launches from now on
            if self.targets_per_frame:  # those kernels read [B, nf, 12]: restack what was uploaded frame by frame
                t = np.zeros((nf, B, 12))
                a.get(t, self.d_Tt)
                a.put(self.d_Tt, np.ascontiguousarray(t.transpose(1, 0, 2)))
                self.targets_per_frame = False
        if self.fused == "kernel":
            pass
        elif self.fused:
            # one launch applies the previous step's dq (status-checked), then FK, frame-task rows, limits, posture
            st = Step()
            st.q = self.d_q
            st.dq_prev = self.d_dq if self._pending else None
            st.status, st.first_failure, st.step = self.d_status, self.d_fail, max(self.steps_done - 1, 0)
            st.target_batched = self.qt_batched
            st.T_target, st.T_frames = self.d_Tt, self.d_T
            st.e, st.sE, st.J, st.sJ = self.d_e, self.K, self.d_J, self.Kd * nv
            st.dt, st.config_limit_gain = self.dt, self.config_limit_gain
            st.q_target = self.d_qt if self.n_post else None
            st.lb, st.ub, st.e_off = self.d_lb, self.d_ub, self.Kd
            st.root_box = self.d_lim[0] if self.d_lim else None
            a.step_kernel(self.dmodel, B, st)
            a.solve_raw(self.desc, self.problem, self.result)
            self._pending = bool(integrate)
        else:  # one launch for FK, one per FrameTask, limits + posture, solve, integrate
            a.fk(self.dmodel, B, self.d_q, self.d_T, self.d_Jb)
            for t in range(nf):
                a.frame_task_strided(B, nv, self.d_T + 8 * 12 * t, 12 * nf, self.d_Tt + 8 * 12 * t, 12 * nf,
                                     self.d_Jb + 8 * 6 * nv * t, 6 * nv * nf, self.d_e + 8 * 6 * t, self.K,
                                     self.d_J + 8 * 6 * nv * t, self.Kd * nv)
            a.limits_posture(self.dmodel, B, self.dt, self.config_limit_gain, self.d_q, self.d_qt, self.qt_batched, self.d_lb, self.d_ub,
                             self.d_e if self.n_post else None, self.K, self.Kd)
            a.solve_raw(self.desc, self.problem, self.result)
            if integrate:
                a.integrate_checked(self.dmodel, B, self.d_q, self.d_dq, self.d_status, self.d_fail, self.steps_done)
        self.steps_done += 1

    def solve_pipelined(self, q0: np.ndarray, targets: Sequence[np.ndarray], q_posture: Optional[np.ndarray] = None,
                        safety_break: bool = True, n_chunks: Optional[int] = None, out: Optional[np.ndarray] = None) -> bool:
        """One differential-IK solve of new configurations ``q0`` (no integration), the batch cut into ranges (``n_chunks``
        equal ones, or in the proportions of :data:`PIPELINE_SPLIT`): the upload of one range (``q`` and one ``[B, 12]`` target array per frame task, copy stream), the
        whole-step kernel of the previous one (compute stream) and the results of the one before going home (result
        stream) are in flight together; nothing blocks the host until the closing synchronisation when the arrays are
        page-locked (``pink_amd.pinned_empty``; pageable arrays are staged by the runtime: correct, less overlap).
        ``out [B, nv]`` receives ``dq``.  Results through :meth:`last_step`.  ``False`` -- nothing enqueued -- when the
        whole-step kernel does not serve this model or the solver has no copy stream."""
        a, B, nq, nv, nf = self.api, self.B, self.nq, self.nv, len(self.frames)
        if n_chunks is None:
            cuts = np.rint(np.cumsum((0.0,) + PIPELINE_SPLIT) / sum(PIPELINE_SPLIT) * B).astype(int)
        else:
            cuts = np.array([(B * c) // n_chunks for c in range(n_chunks + 1)])
        cuts = np.unique(cuts)
        if self.fused != "kernel" or not hasattr(a, "put_overlapped") or len(targets) != nf or B < 64:
            return False
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        if q0.shape != (B, nq):
            raise ValueError(f"q0 must have shape {(B, nq)}, got {q0.shape}")
        toks = [getattr(t, "frozen_token", None) for t in targets]  # (FrameTask.freeze_targets)
        if not self.targets_per_frame:
            self._resident = {}
        skip = [tok is not None and self._resident.get(f) == tok for f, tok in enumerate(toks)]
        # ([B, 7] translation + quaternion arrays go up as they are -- 56 B per robot and frame task instead of 96 -- and a
        # device kernel writes the poses: FrameTask.set_target_poses_quat)
        tg = [None if skip[f] else np.ascontiguousarray(t if (np.ndim(t) == 2 and t.shape == (B, 7)) else np.broadcast_to(t, (B, 12)), dtype=np.float64)
              for f, t in enumerate(targets)]
        if out is not None and (out.shape != (B, nv) or out.dtype != np.float64 or not out.flags.c_contiguous):
            raise ValueError(f"out must be a C-contiguous float64 array of shape {(B, nv)}")
        if self.n_post:  # (the first kernel waits for the copy stream: wait_copies below)
            self._put_posture(q0, q_posture, asyn=hasattr(a, "put_async") and hasattr(a, "is_pinned") and a.is_pinned(q0))
        if self._fail_dirty:  # (a solve without integration records no failures: only a rollout leaves some behind)
            a.put(self.d_fail, np.zeros(B, dtype=np.int32))
            self._fail_dirty = False
        self.steps_done, self._pending, self.targets_per_frame = 0, False, True
        # Fully asynchronous ranges need page-locked memory on both ends: a pageable source is staged by the runtime
        # before the call returns (no harm), a pageable destination makes the download wait for its kernel on the host
        # thread -- the uploads of the next range would queue behind it.  Without a page-locked `out` the results come
        # home in one piece at the end, as before.
        asyn = hasattr(a, "put_async") and hasattr(a, "is_pinned") and a.is_pinned(q0)
        back = asyn and out is not None and a.is_pinned(out)
        put = a.put_async if asyn else a.put_overlapped
        res = None
        if back:
            if getattr(self, "_h_status", None) is None:  # page-locked landing buffers of the small result arrays
                self._h_status, self._h_iters = a.pinned_empty((B,), np.int32), a.pinned_empty((B,), np.int32)
            res = (out, self._h_status, self._h_iters)
        # ranges go to the handle's two compute streams in turn: the drain of one range's kernel overlaps the start of the
        # next (on one stream four range kernels cost twice a single launch over the batch: profiles/prof_pipeline_r04.txt)
        two = asyn and hasattr(a, "select_stream") and _os.environ.get("PINKHIP_ONE_COMPUTE_STREAM") != "1"
        # Page-locked result arrays are written by the kernel itself (mapped host memory: the device address of a
        # hipHostMalloc block is its host address): no result stream, no copy per range, nothing left to drain when the
        # last kernel ends -- the device-to-host direction of the link is idle otherwise.  PINKHIP_RESULT_COPIES=1 goes
        # back to the copies (A/B: profiles/ab_api_arrays_r05.txt).
        self._zc = None
        if back and _os.environ.get("PINKHIP_RESULT_COPIES") != "1":
            self._zc = tuple(int(r.ctypes.data) for r in res)
        try:
            return self._pipelined_ranges(a, cuts, q0, tg, toks, put, asyn, back, res, out, safety_break, two)
        finally:
            self._zc = None
            if two:
                a.select_stream(0)

    def _pipelined_ranges(self, a, cuts, q0, tg, toks, put, asyn, back, res, out, safety_break, two) -> bool:
        B, nq, nv = self.B, self.nq, self.nv
        for c in range(len(cuts) - 1):
            lo, hi = int(cuts[c]), int(cuts[c + 1])
            if two:
                a.select_stream(c & 1)
            put(self.d_q + 8 * nq * lo, q0[lo:hi])
            quat = []
            for f, t in enumerate(tg):
                if t is None:
                    continue
                if t.shape[1] == 7:
                    put(self._pq_stage(f, lo), t[lo:hi])
                    quat.append(f)
                else:
                    put(self.d_Tt + 8 * 12 * (B * f + lo), t[lo:hi])
            if asyn:
                a.wait_copies()
            for f in quat:  # (on the range's compute stream, in front of its kernel)
                a.pose_targets(hi - lo, self._pq_stage(f, lo), self.d_Tt + 8 * 12 * (B * f + lo))
            if not self._one_kernel_step(False, lo, hi):
                if c:
                    raise RuntimeError("whole-step kernel refused a later range of the same batch")
                if asyn:
                    a.sync()
                return False
            if back and self._zc is None:
                a.get_async(res[0][lo:hi], self.d_dq + 8 * nv * lo)
                a.get_async(res[1][lo:hi], self.d_status + 4 * lo)
                a.get_async(res[2][lo:hi], self.d_iters + 4 * lo)
        self.steps_done = 1
        self._pipelined = res
        self._resident = dict(enumerate(toks))
        self.bytes_in_last_call = int(q0.nbytes + sum(t.nbytes for t in tg if t is not None))
        # Configuration.check_limits on the whole batch (pink/solve_ik.py:260), after the fact: the velocities of a
        # batch that violates its limits are never handed out -- the copies into a page-locked `out` are already in
        # flight at this point, so a refusal waits for them and blanks the array before it propagates
        try:
            self._check_limits_device(q0, safety_break)
        except BaseException:
            a.sync()
            if back:
                out.fill(np.nan)
            self._pipelined = None
            raise
        return True

    def _one_kernel_step(self, integrate: bool = True, lo: int = 0, hi: Optional[int] = None) -> bool:
        """FK + FrameTask rows + limits + posture + stack + solve + integrate in one launch (robots ``lo .. hi``)."""
        if hi is not None and (lo, hi) != (0, self.B):
            return self._one_kernel_step_range(integrate, lo, hi)
        st = RolloutStep()
        st.q, st.cost, st.T_target, st.T_frames = self.d_q, self.d_cost, self.d_Tt, self.d_T
        st.q_target = self.d_qt if self.n_post else None
        st.dq, st.status, st.iters = getattr(self, "_zc", None) or (self.d_dq, self.d_status, self.d_iters)
        st.first_failure = self.d_fail
        st.config_limit_gain = self.config_limit_gain
        st.target_batched, st.step, st.integrate = self.qt_batched, self.steps_done, int(integrate)
        self._scale_out(st, integrate)
        self._extra(st)
        if self.targets_per_frame:
            st.sT_b, st.sT_f = 12, 12 * self.B
        self._dense_tables(st)
        return self.api.rollout_step(self.desc, self.dmodel, st)

    def _dense_tables(self, st) -> None:
        """The batch-constant tables behind the dense rows the kernel forms on chip: barriers, constraint slots, the
        floating-base limit (box on the root coordinates + constant rows)."""
        if self.d_bar:
            st.barrier_frame, st.barrier_axis, st.barrier_sign, st.barrier_bound, st.barrier_gain, st.barrier_frame2 = self.d_bar
        if self.d_cons:
            st.n_constraint_frames = len(self.cons)
            st.constraint_frame, st.constraint_gain = self.d_cons
        if self.d_lim:
            st.root_box, st.limit_rows, st.limit_h = self.d_lim
            st.n_limit_rows = len(self.lim_h)

    def set_diag_errors(self, errors: Sequence[np.ndarray]) -> None:
        """New batch-constant errors of the extra diagonal tasks (``diag_tasks`` of the constructor, in that order): a
        LowAccelerationTask / JointVelocityTask changes them every control step."""
        off = self.n_post
        for e in errors:
            e = np.atleast_1d(np.asarray(e, dtype=np.float64))
            self._diag_e[off:off + e.shape[0]] = e
            off += e.shape[0]
        if off != max(self.K - self.Kd, 0):
            raise ValueError("errors do not match the diagonal tasks of this rollout")
        if self.d_extra:
            self.api.put(self.d_extra[0], self._diag_e)

    def set_acceleration_limit(self, tables: np.ndarray) -> None:
        """New ``[3, nv]`` tables of the acceleration limit (``Delta_q_prev`` moves with every control step:
        ``AccelerationLimit.set_last_integration``)."""
        if self.d_acc is None:
            raise ValueError("this rollout was built without an acceleration limit")
        self._acc[...] = tables
        self.api.put(self.d_acc, self._acc)

    def _extra(self, st) -> None:
        """The fields of ``pinkhip_rollout_step`` that describe tasks beyond frames + posture."""
        st.posture_task = self.posture_task
        if self.d_acc is not None:
            st.acc_limit = self.d_acc
        if self._extra_tasks:
            st.diag_error = self.d_extra[0]
            if self._const:
                st.n_const_rows = self.n_crow
                st.const_rows, st.const_q0, st.const_b = self.d_extra[1:4]

    def _scale_out(self, st, integrate: bool) -> None:
        """``velocity_out``: a solve without integration writes ``dq / dt`` -- the velocity ``solve_ik`` returns
        (``pink/solve_ik.py:274``) -- instead of ``dq`` (``pinkhip_rollout_step.dq_scale``): no pass over the array on
        the host afterwards.  ``self.scaled`` says whether the last launch did."""
        self.scaled = bool(getattr(self, "velocity_out", False)) and not integrate
        if self.scaled:
            st.dq_scale = 1.0 / self.dt

    def _one_kernel_step_range(self, integrate: bool, lo: int, hi: int) -> bool:
        """The whole-step kernel on the robots ``lo .. hi`` of the resident batch (per-frame target arrays)."""
        nf, nv, nq = len(self.frames), self.nv, self.nq
        if not self.targets_per_frame:
            raise ValueError("ranges of the batch are launched over per-frame target arrays")
        st = RolloutStep()
        st.q, st.cost = self.d_q + 8 * nq * lo, self.d_cost
        st.T_target, st.T_frames = self.d_Tt + 8 * 12 * lo, self.d_T + 8 * 12 * max(nf, 1) * lo
        st.q_target = (self.d_qt + (8 * nq * lo if self.qt_batched else 0)) if self.n_post else None
        dq, status, iters = getattr(self, "_zc", None) or (self.d_dq, self.d_status, self.d_iters)
        st.dq, st.status, st.iters = dq + 8 * nv * lo, status + 4 * lo, iters + 4 * lo
        st.first_failure = self.d_fail + 4 * lo
        st.config_limit_gain = self.config_limit_gain
        st.target_batched, st.step, st.integrate = self.qt_batched, self.steps_done, int(integrate)
        self._scale_out(st, integrate)
        self._extra(st)
        st.sT_b, st.sT_f = 12, 12 * self.B
        self._dense_tables(st)  # (round 5: ranges of a batch with dense rows too -- the tables are the same for every robot)
        self.desc.B = hi - lo
        try:
            return self.api.rollout_step(self.desc, self.dmodel, st)
        finally:
            self.desc.B = self.B

    def flush(self) -> None:
        """Apply the displacement of the last enqueued step (the whole-step kernel integrates lazily, at the
        start of the next step); called by everything that reads the configurations."""
        if self._pending:
            self.api.integrate_checked(self.dmodel, self.B, self.d_q, self.d_dq, self.d_status, self.d_fail, self.steps_done - 1)
            self._pending = False

    def run(self, steps: int, raise_on_failure: bool = True) -> None:
        """``steps`` IK steps for every robot, then one synchronisation.  Raises :class:`NoSolutionFound`
        listing the robots whose QP failed at some step (they stopped moving at that step)."""
        for _ in range(steps):
            self.step()
        self.flush()
        self.api.sync()
        if raise_on_failure:
            idx, status, step = self.failures()
            if idx.size:
                exc = NoSolutionFound(None, None, idx, status)
                exc.steps = step
                raise exc

    def failures(self):
        """``(indices, status, step)`` of the robots whose solve has failed so far (first failure of each)."""
        self.flush()
        f = np.zeros(self.B, np.int32)
        self.api.get(f, self.d_fail)
        idx = np.nonzero(f)[0]
        return idx, f[idx] & 0xFF, f[idx] >> 8

    @staticmethod
    def _check_limits(model: Model, q0: np.ndarray, safety_break: bool, tol: float = 1e-6) -> None:
        """``Configuration.check_limits`` (``pink/configuration.py:166-201``) over the batch of initial
        configurations: first violated joint limit raises (or warns)."""
        lo, up = model.lowerPositionLimit, model.upperPositionLimit
        start = model.root_joint.nq if model.root_joint is not None else 0
        bad = (up > lo + tol) & ((q0 < lo - tol) | (q0 > up + tol))
        bad[:, :start] = False
        if bad.any():
            b, i = (int(v[0]) for v in np.nonzero(bad))
            if safety_break:
                raise NotWithinConfigurationLimits(i, q0[b, i], lo[i], up[i], instance=b)
            import logging

            logging.warning("Value %f at index %d of instance %d is out of limits: [%f, %f]", q0[b, i], i, b, lo[i], up[i])

    def configurations(self) -> np.ndarray:
        self.flush()
        q = np.zeros((self.B, self.nq))
        self.api.get(q, self.d_q)
        return q

    def last_step(self):
        """``(dq, status, iters)`` of the most recent step; the code that solved each robot's QP (index into
        ``pink_amd.batch_solver.PATH_NAMES``) is kept as ``self.last_path``."""
        from .batch_solver import split_iters

        res = getattr(self, "_pipelined", None)
        if res is not None:  # the results of a pipelined solve are on their way (or here): wait, hand them out once
            self._pipelined = None
            self.api.sync()
            it = res[2].copy()
            self.last_path = split_iters(it)
            return res[0], res[1].copy(), it
        dq = np.empty((self.B, self.nv))
        st = np.empty(self.B, np.int32)
        it = np.empty(self.B, np.int32)
        self.api.get(dq, self.d_dq)
        self.api.get(st, self.d_status)
        self.api.get(it, self.d_iters)
        self.last_path = split_iters(it)
        return dq, st, it

    def frame_poses(self) -> np.ndarray:
        """Frame poses computed by the last forward-kinematics launch, ``[B, nf, 12]``."""
        T = np.zeros((self.B, max(len(self.frames), 1), 12))
        self.api.get(T, self.d_T)
        return T

    def free(self) -> None:
        for name in self._BUFFERS:
            self.api.release(getattr(self, name, None))
        for ptr in getattr(self, "d_bar", []) + getattr(self, "d_lim", []) + getattr(self, "d_extra", []) + getattr(self, "d_cons", []):
            self.api.release(ptr)
        self.d_bar, self.d_lim, self.d_extra, self.d_cons = [], [], [], []
        if getattr(self, "d_acc", None) is not None:
            self.api.release(self.d_acc)
            self.d_acc = None
        self.api.model_destroy(self.dmodel)
