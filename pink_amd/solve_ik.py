"""``build_ik`` / ``solve_ik`` with Pink's signatures, plus ``solve_ik_batch``.

Mirrors ``pink/solve_ik.py:152-275``.  The host evaluates the user's tasks /
limits / barriers (Python objects, exactly as in Pink), packs the raw terms
(``pink_amd.batch``) and hands them to the HIP library, which stacks the QP and
solves it.  The only accepted ``solver`` is the MI355X one: there is no
qpsolvers dispatch and no CPU path.
"""

from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .batch import BarrierTerm, DenseTaskTerm, DiagonalTaskTerm, IKBatch, pack_terms, split_box_rows
from .exceptions import NoSolutionFound, PinkError

SOLVER_NAMES = ("mi355x", "pinkhip")
# ``solver=`` strings of the reference (``pink/solve_ik.py:210,270`` forwards them to ``qpsolvers.solve_problem``; its
# README and examples say "quadprog", its tests "daqp" / "proxqp" / "osqp"): accepted as aliases so that existing Pink
# calls run unchanged -- every one of them is served by the MI355X solver, which returns the same minimiser (the QP is
# strictly convex).
REFERENCE_SOLVER_NAMES = ("quadprog", "daqp", "proxqp", "osqp", "clarabel", "cvxopt", "ecos", "gurobi", "highs", "hpipm",
                          "jaxopt_osqp", "kvxopt", "mosek", "nppro", "piqp", "qpalm", "qpax", "qpoases", "qpswift", "qtqp",
                          "scs", "sip")
_aliased = set()


def _check_solver(solver: str) -> None:
    """Accept the MI355X solver and, as aliases, the reference's qpsolvers names (logged once per name)."""
    if solver in SOLVER_NAMES:
        return
    if solver in REFERENCE_SOLVER_NAMES:
        if solver not in _aliased:
            _aliased.add(solver)
            import logging

            logging.getLogger("pink_amd").info("solver=%r is served by the MI355X solver (pinkhip)", solver)
        return
    raise PinkError(f"solver={solver!r}: this build provides the MI355X solver {SOLVER_NAMES} (qpsolvers names "
                    f"{REFERENCE_SOLVER_NAMES} are accepted as aliases of it)")


class IKProblem:
    """What ``pink.build_ik`` returns (a ``qpsolvers.Problem``): ``P, q, G, h, A, b``.

    ``G, h`` are stacked on the host in Pink's order (limits, then barriers;
    ``pink/solve_ik.py:107-122``); ``P, q`` are evaluated lazily by the HIP stack
    kernel.  ``batch`` is the packed single-instance batch handed to the solver.
    """

    def __init__(self, batch: IKBatch, G: Optional[np.ndarray], h: Optional[np.ndarray], A=None, b=None):
        self.batch = batch
        self.G, self.h = G, h
        self.A, self.b = A, b
        self._Pq: Optional[Tuple[np.ndarray, np.ndarray]] = None

    def _stack(self):
        if self._Pq is None:
            from .runtime import default_solver

            H, c = default_solver().stack(self.batch)
            self._Pq = (H[0], c[0])
        return self._Pq

    @property
    def P(self) -> np.ndarray:
        return self._stack()[0]

    @property
    def q(self) -> np.ndarray:
        return self._stack()[1]


def _collect_terms(configuration, tasks, dt, limits, barriers):
    """Evaluate tasks / limits / barriers at one configuration (host side)."""
    nv = configuration.model.nv
    task_terms = [t.as_term(configuration) for t in tasks]
    if limits is None:  # model defaults, pink/solve_ik.py:94-105
        limits = [configuration.model.configuration_limit, configuration.model.velocity_limit]
        fb = getattr(configuration.model, "floating_base_velocity_limit", None)
        if fb is not None:
            limits.append(fb)
    lb = np.full(nv, -np.inf)
    ub = np.full(nv, np.inf)
    dense_rows = []
    G_list, h_list = [], []
    for limit in limits:
        box = limit.compute_box(configuration, dt) if hasattr(limit, "compute_box") else None
        if box is not None:
            idx, lo, up = box
            lb[idx] = np.maximum(lb[idx], lo)
            ub[idx] = np.minimum(ub[idx], up)
            continue
        mv = limit.compute_qp_inequalities(configuration, dt)
        if mv is None:  # pink/solve_ik.py:111
            continue
        blo, bup, Gd, hd = split_box_rows(mv[0], mv[1], nv)
        lb, ub = np.maximum(lb, blo), np.minimum(ub, bup)
        if len(hd):
            dense_rows.append((Gd[None], hd[None]))
    barrier_terms = [b.as_term(configuration) for b in (barriers or [])]
    return nv, task_terms, lb, ub, dense_rows, barrier_terms, limits


def _pink_rows(configuration, dt, limits, barriers):
    """``(G, h)`` exactly as Pink stacks them, for ``IKProblem.G/h``."""
    G_list, h_list = [], []
    for limit in limits:
        mv = limit.compute_qp_inequalities(configuration, dt)
        if mv is not None:
            G_list.append(mv[0])
            h_list.append(mv[1])
    for barrier in barriers or []:
        Gb, hb = barrier.compute_qp_inequalities(configuration, dt)
        G_list.append(Gb)
        h_list.append(hb)
    if not G_list:
        return None, None  # pink/solve_ik.py:120-121
    return np.vstack(G_list), np.hstack(h_list)


def _equalities(configuration, constraints):
    """``A = J``, ``b = -gain e`` of the tasks to enforce strictly (``pink/solve_ik.py:125-149``)."""
    if not constraints:
        return None, None
    A_list = [np.asarray(t.compute_jacobian(configuration), dtype=float) for t in constraints]
    b_list = [-t.gain * np.asarray(t.compute_error(configuration), dtype=float) for t in constraints]
    return np.vstack(A_list), np.hstack(b_list)


def build_ik(configuration, tasks: Iterable, dt: float, damping: float = 1e-12, limits=None, barriers=None,
             constraints=None) -> IKProblem:
    """Build the QP of one IK step (``pink/solve_ik.py:152-203``)."""
    tasks = list(tasks)
    nv, task_terms, lb, ub, dense_rows, barrier_terms, limits = _collect_terms(configuration, tasks, dt, limits, barriers)
    A, b = _equalities(configuration, constraints)
    batch = pack_terms(nv, task_terms, dt, damping, boxes=[(lb[None], ub[None])], dense_rows=dense_rows,
                       barriers=barrier_terms, batch_size=1,
                       equality_rows=[(A[None], b[None])] if A is not None else ())
    G, h = _pink_rows(configuration, dt, limits, barriers)
    return IKProblem(batch, G, h, A, b)


def solve_ik(configuration, tasks: Iterable, dt: float, solver: str = "mi355x", damping: float = 1e-12, limits=None,
             barriers=None, constraints=None, safety_break: bool = True, **kwargs) -> np.ndarray:
    """Velocity tangent to ``configuration`` that best fulfils ``tasks``
    (``pink/solve_ik.py:206-275``).  ``kwargs`` accepts ``max_iter``."""
    _check_solver(solver)
    from .runtime import default_solver

    configuration.check_limits(safety_break=safety_break)  # solve_ik.py:260
    problem = build_ik(configuration, tasks, dt, damping, limits, barriers, constraints)
    result = default_solver().solve(problem.batch, max_iter=int(kwargs.get("max_iter", 0)))
    if not result.all_found:  # solve_ik.py:271-273
        raise NoSolutionFound(problem, result, result.failed_indices(), result.status[result.status != 0])
    return result.dq[0] / dt


def _check_same_length(tasks) -> None:
    """Per-instance task lists must all have the length of the first one: a slot is one task of the stacked QP."""
    n = len(tasks[0])
    for b, tb in enumerate(tasks):
        if not isinstance(tb, (list, tuple)) or len(tb) != n:
            raise PinkError(f"per-instance task lists must have the same length: instance {b} has "
                            f"{len(tb) if isinstance(tb, (list, tuple)) else type(tb).__name__}, instance 0 has {n}")


def _pose12(T) -> np.ndarray:
    return np.hstack([np.asarray(T.rotation, dtype=float).ravel(), np.asarray(T.translation, dtype=float)])


_PAD_ROW_H = 1e30  # right-hand side of the padding rows  0 dq <= PAD


class _SharedSlot:
    """One task object shared by the ``B`` instances of a batch (a sequence of ``B`` references to it)."""

    shared = True

    def __init__(self, task, B: int):
        self.task, self.B = task, B

    def __len__(self) -> int:
        return self.B

    def __getitem__(self, b):
        return self.task

    def __iter__(self):
        return (self.task for _ in range(self.B))


def _task_slots(tasks, B: int):
    """``tasks`` (one list for all instances, or one list per instance) as a list of slots; slot ``k`` holds the
    ``k``-th task of every instance."""
    per_instance = B > 0 and len(tasks) == B and isinstance(tasks[0], (list, tuple))
    if per_instance:
        _check_same_length(tasks)
        return [[tasks[b][k] for b in range(B)] for k in range(len(tasks[0]))]
    return [_SharedSlot(t, B) for t in tasks]


def _batch_kinematics(configurations):
    """:class:`~pink_amd.kinematics_batch.BatchKinematics` of ``configurations`` (a ``ConfigurationBatch`` or a list
    of ``Configuration`` objects of one model), or ``None`` when the list mixes models."""
    from .configuration import ConfigurationBatch
    from .kinematics_batch import BatchKinematics

    if isinstance(configurations, ConfigurationBatch):
        return configurations.kinematics()
    model = configurations[0].model
    if not hasattr(model, "joints") or any(c.model is not model for c in configurations):
        return None
    kin = BatchKinematics(model, np.stack([np.asarray(c.q, dtype=np.float64) for c in configurations]))
    kin._cfgs = dict(enumerate(configurations))  # (evaluators that fall back to per-instance methods use the caller's objects)
    return kin


def pack_configurations(configurations: Sequence, tasks: Sequence, dt: float, damping: float = 1e-12, limits=None,
                        barriers=None, solver_handle=None, gpu_frame_tasks: bool = True, constraints=None) -> IKBatch:
    """Evaluate the same task / limit / barrier objects at every configuration and
    pack the batch.  The task list may also be a list of per-instance lists (one
    target per instance): ``tasks[b]`` is then used for ``configurations[b]``.

    Everything is evaluated for the whole batch at once (:mod:`pink_amd.batch_eval` over one vectorised forward
    kinematics, :mod:`pink_amd.kinematics_batch`): no ``Configuration`` object and no Python loop per instance for the
    task, limit and barrier classes the package ships; user subclasses fall back to their own per-configuration
    methods.  FrameTasks are finished by the HIP frame-task kernel (``log6`` / ``Jlog6`` / ``-Jlog6 J_body``,
    ``pink/tasks/frame_task.py:176-227``) when ``gpu_frame_tasks`` is set.

    ``constraints`` (tasks enforced as equalities, ``pink/solve_ik.py:125-149``) follows the same convention
    as ``tasks``: one list for all instances or one list per instance.
    """
    from . import batch_eval as be

    B = len(configurations)
    if B == 0:
        return pack_terms(0, [], dt, damping, batch_size=0)
    kin = _batch_kinematics(configurations)
    if kin is None:
        return _pack_configurations_per_instance(configurations, tasks, dt, damping, limits, barriers, solver_handle,
                                                 gpu_frame_tasks, constraints)
    model, nv = kin.model, kin.model.nv
    ft_solver = None
    if gpu_frame_tasks:
        from .runtime import default_solver

        ft_solver = solver_handle or default_solver()
    merged = [be.task_term(kin, col, ft_solver) for col in _task_slots(tasks, B)]
    if limits is None:  # model defaults, pink/solve_ik.py:94-105
        model.ensure_limits()
        limits = [model.configuration_limit, model.velocity_limit]
        if getattr(model, "floating_base_velocity_limit", None) is not None:
            limits.append(model.floating_base_velocity_limit)
    lb = np.full((B, nv), -np.inf)
    ub = np.full((B, nv), np.inf)
    dense_rows = []
    for limit in limits:
        rows = be.limit_rows(kin, limit, dt, lb, ub)
        if rows is not None:
            dense_rows.append(rows)
    barrier_terms = [be.barrier_term(kin, bar) for bar in (barriers or [])]
    equality_rows = []
    if constraints:  # pink/solve_ik.py:125-149: A = J, b = -gain e of each task to enforce strictly
        for col in _task_slots(constraints, B):
            t = be.task_term(kin, col, None)
            if isinstance(t, DiagonalTaskTerm):
                k = t.e.shape[1]
                A = np.broadcast_to(np.eye(nv)[t.col0:t.col0 + k], (B, k, nv))
            else:
                A = t.J
            equality_rows.append((A, -t.gain * t.e))
    return pack_terms(nv, merged, dt, damping, boxes=[(lb, ub)], dense_rows=dense_rows, barriers=barrier_terms,
                      batch_size=B, equality_rows=equality_rows)


def _pack_configurations_per_instance(configurations, tasks, dt, damping, limits, barriers, solver_handle, gpu_frame_tasks,
                                      constraints) -> IKBatch:
    """:func:`pack_configurations` one configuration at a time (configurations of different model objects)."""
    from .tasks.frame_task import FrameTask

    B = len(configurations)
    nv = configurations[0].model.nv if B else 0
    per_instance = B > 0 and len(tasks) == B and isinstance(tasks[0], (list, tuple))
    if per_instance:
        _check_same_length(tasks)
    n_tasks = len(tasks[0]) if per_instance else len(tasks)
    merged = []
    for k in range(n_tasks):
        tk = [tasks[b][k] for b in range(B)] if per_instance else [tasks[k]] * B
        t0 = tk[0]
        costs = [np.asarray(t.cost if t.cost is not None else 1.0, dtype=float) for t in tk]
        same_cost = all(c.shape == costs[0].shape and np.array_equal(c, costs[0]) for c in costs)
        if any(t.gain != t0.gain or t.lm_damping != t0.lm_damping for t in tk):
            raise PinkError("gain / lm_damping of one task slot must be the same for every instance of a batch")
        if gpu_frame_tasks and all(isinstance(t, FrameTask) for t in tk):
            from .exceptions import TargetNotSet
            from .runtime import default_solver

            for t in tk:
                if t.transform_target_to_world is None:
                    raise TargetNotSet(f"no target set for frame '{t.frame}'")
            Tf = np.array([_pose12(cfg.get_transform_frame_to_world(t.frame)) for cfg, t in zip(configurations, tk)])
            Tt = np.array([_pose12(t.transform_target_to_world) for t in tk])
            Jb = np.array([cfg.get_frame_jacobian(t.frame) for cfg, t in zip(configurations, tk)])
            e, J = (solver_handle or default_solver()).frame_task_terms(Tf, Tt, Jb)
            cost = costs[0] if same_cost else np.array([np.broadcast_to(c, (6,)) for c in costs])
            merged.append(DenseTaskTerm(J=J, e=e, cost=cost, gain=t0.gain, lm_damping=t0.lm_damping))
            continue
        terms = [t.as_term(cfg) for cfg, t in zip(configurations, tk)]
        e = np.concatenate([t.e for t in terms], axis=0)
        kk = e.shape[1]
        cost = t0.cost if same_cost else np.array([np.broadcast_to(c, (kk,)) for c in costs])
        if isinstance(terms[0], DiagonalTaskTerm):
            merged.append(DiagonalTaskTerm(col0=terms[0].col0, e=e, cost=cost, gain=t0.gain, lm_damping=t0.lm_damping))
        else:
            merged.append(DenseTaskTerm(J=np.concatenate([t.J for t in terms], axis=0), e=e, cost=cost, gain=t0.gain,
                                        lm_damping=t0.lm_damping))
    lbs, ubs, dense, bterms = [], [], [], []
    for cfg in configurations:
        _, _, lb, ub, dr, bt, _ = _collect_terms(cfg, [], dt, limits, barriers)
        lbs.append(lb), ubs.append(ub), dense.append(dr), bterms.append(bt)
    # Dense limit rows: the number a limit returns may differ from instance to instance (a row that is
    # axis-aligned at one configuration went into the box there; a limit may return None).  Every instance's
    # rows are concatenated and padded to the batch maximum with rows  0 dq <= PAD  that can never be active.
    dense_rows = []
    n_rows = [sum(h.shape[1] for _, h in dr) for dr in dense]
    r_max = max(n_rows, default=0)
    if r_max:
        Gp = np.zeros((B, r_max, nv))
        hp = np.full((B, r_max), _PAD_ROW_H)
        for b, dr in enumerate(dense):
            if dr:
                Gp[b, :n_rows[b]] = np.concatenate([G[0] for G, _ in dr], axis=0)
                hp[b, :n_rows[b]] = np.concatenate([h[0] for _, h in dr], axis=0)
        dense_rows.append((Gp, hp))
    barrier_terms = []
    n_bar = {len(bt) for bt in bterms}
    if len(n_bar) > 1:
        raise PinkError("every instance of a batch must evaluate the same list of barriers")
    for k in range(n_bar.pop() if n_bar else 0):
        col = [bt[k] for bt in bterms]
        b0 = col[0]
        if any(t.J_h.shape != b0.J_h.shape or t.safe_displacement_gain != b0.safe_displacement_gain
               or not np.array_equal(np.asarray(t.gain), np.asarray(b0.gain)) for t in col):
            raise PinkError(f"barrier slot {k}: dimension / gains must be the same for every instance of a batch")
        # a barrier whose safe displacement is zero at some configurations (as_term reports None there)
        # contributes zero to the linear term of those instances only
        sd = None
        if any(t.safe_displacement is not None for t in col):
            sd = np.concatenate([np.zeros((1, nv)) if t.safe_displacement is None else t.safe_displacement for t in col], axis=0)
        barrier_terms.append(BarrierTerm(J_h=np.concatenate([t.J_h for t in col], axis=0),
                                         h=np.concatenate([t.h for t in col], axis=0), gain=b0.gain,
                                         safe_displacement_gain=b0.safe_displacement_gain, safe_displacement=sd))
    equality_rows = []
    if constraints:  # pink/solve_ik.py:125-149, per instance
        per_c = B > 0 and len(constraints) == B and isinstance(constraints[0], (list, tuple))
        Ab = [_equalities(cfg, constraints[b] if per_c else constraints) for b, cfg in enumerate(configurations)]
        if len({A.shape for A, _ in Ab}) > 1:
            raise PinkError("constraints= must produce the same number of equality rows for every instance")
        equality_rows.append((np.stack([A for A, _ in Ab]), np.stack([bb for _, bb in Ab])))
    return pack_terms(nv, merged, dt, damping, boxes=[(np.array(lbs), np.array(ubs))] if B else (),
                      dense_rows=dense_rows, barriers=barrier_terms, batch_size=B, equality_rows=equality_rows)


_DECLINED = {"reason": None}
_ROUTE_WARNED = set()


def _declined(reason: str):
    """``return _declined("...")``: the device route is not taken, and why (the first term that declined it; read by the
    one-shot warning of :func:`solve_ik_batch`).  Returns ``None`` like every plan function does when it declines."""
    if _DECLINED["reason"] is None:
        _DECLINED["reason"] = reason
    return None


def _stack_signature(tasks, limits, barriers, constraints) -> tuple:
    def names(objs):
        if objs is None:
            return None
        objs = list(objs)
        if objs and isinstance(objs[0], (list, tuple)):  # per-instance lists: the first instance stands for the batch
            objs = list(objs[0])
        return tuple(type(o).__name__ for o in objs)

    return names(tasks), names(limits), names(barriers), names(constraints)


def _report_route(route: str, B: int, tasks, limits, barriers, constraints, strict_route, warn: bool = True) -> None:
    """A batch of 64 and more that leaves the device route says so ONCE per (stack signature, route) -- the reference's
    own one-shot warning style (``pink/configuration.py:188-201``) -- and raises when the caller asked for
    ``strict_route="device"``: the routes differ by 70 .. 2000 x in cost."""
    reason = _DECLINED["reason"] or "a term the device tables do not hold"
    if strict_route is not None and strict_route != route:
        raise PinkError(f"strict_route={strict_route!r}: this call would take the {route!r} route ({reason})")
    if B < 64 or route == "device" or not warn:
        return
    key = (_stack_signature(tasks, limits, barriers, constraints), route)
    if key not in _ROUTE_WARNED:
        _ROUTE_WARNED.add(key)
        import logging

        logging.getLogger("pink_amd").warning(
            "solve_ik_batch: B = %d leaves the device route for the %r route (%s); pass device_kinematics=True or "
            "strict_route='device' to make this an error", B, route, reason)


def _spec_of(task):
    """``(frame, position cost, orientation cost, gain, lm_damping)`` of a FrameTask, hashable; a RelativeFrameTask has
    ``(frame, root)`` in the first place (a relative slot of the device model, ``pink_amd/rollout.py``)."""
    name = (task.frame, task.root) if hasattr(task, "root") else task.frame
    return (name, tuple(float(v) for v in task.cost[0:3]), tuple(float(v) for v in task.cost[3:6]), float(task.gain), float(task.lm_damping))


def _device_kinematics_plan(configurations, tasks, limits, barriers, constraints):
    """``(model, q [B, nq], frame task specs, target poses, posture, extras, barriers, limit gain, acceleration tables,
    velocity vector, constraints, floating-base limit)`` when the whole batch can be evaluated on the device from the configurations alone --
    FrameTasks / RelativeFrameTasks (one target per instance allowed), one PostureTask, the table-formed tasks of
    :func:`_extra_task`, the model's default limits (:func:`_default_limits_gain`), PositionBarriers (default class-K
    function) and BodySphericalBarriers, equality constraints made of FrameTasks / RelativeFrameTasks (``pink/solve_ik.py:125-149``;
    at most two), one model
    -- else ``None``.  Frames that only a barrier or a constraint needs become slots of the device model with ZERO cost
    (``pink/tasks/task.py:148-166``: nothing enters the objective for them); ``constraints`` is a tuple of
    ``(slot, gain)``."""
    from .barriers.barrier import Barrier
    from .barriers.body_spherical_barrier import BodySphericalBarrier
    from .barriers.position_barrier import PositionBarrier
    from .tasks.frame_task import FrameTask
    from .tasks.relative_frame_task import RelativeFrameTask

    B = len(configurations)
    if B == 0:
        return None
    model = configurations.model if hasattr(configurations, "model") else configurations[0].model
    lim = _default_limits_gain(model, limits)
    if lim is None:
        return _declined("limits other than one ConfigurationLimit + at most one VelocityLimit / AccelerationLimit / FloatingBaseVelocityLimit of this model")
    gain, acc, vmax = lim
    for bar in barriers or ():
        # position barriers with the default class-K function and spherical barriers, neither with a safe displacement of
        # its own, are formed on chip
        if type(bar).compute_safe_displacement is not Barrier.compute_safe_displacement:
            return _declined(f"{type(bar).__name__} with a safe displacement of its own")
        if type(bar) is PositionBarrier:
            if not bar.identity_gain_function:
                return _declined("PositionBarrier with a class-K function of its own")
        elif type(bar) is not BodySphericalBarrier or np.ndim(bar.gain) > 1 or np.size(bar.gain) != 1:
            return _declined(f"barrier {type(bar).__name__}: only PositionBarrier and BodySphericalBarrier rows are formed on chip")
    plan = _device_kinematics_plan_tasks(configurations, tasks)
    if plan is None:
        return _declined("a task the whole-step kernel does not form (FrameTask / RelativeFrameTask, one PostureTask, table-formed tasks shared by the batch)")
    model, q, specs, T, posture, extras = plan
    specs = list(specs)
    as_list = isinstance(T, (list, tuple))
    T = list(T) if as_list else T

    def add_slot(spec, target):  # target [B, 12] (or broadcastable)
        nonlocal T
        specs.append(spec)
        tg = np.broadcast_to(target, (B, 12))
        if as_list:
            T.append(tg)
        else:
            T = np.concatenate([T, np.ascontiguousarray(tg)[:, None, :]], axis=1)
        return len(specs) - 1

    cons = []
    if constraints:
        # the constraint tasks go through the same reading as the objective's tasks: FrameTasks with targets, shared or
        # per instance; their slots carry zero cost
        per_instance = len(constraints) == B and isinstance(constraints[0], (list, tuple))
        flat = [t for c in constraints for t in c] if per_instance else list(constraints)
        if not flat or any(type(t) not in (FrameTask, RelativeFrameTask) for t in flat):
            return _declined("constraints= holds a task that is not a FrameTask / RelativeFrameTask")  # (a RelativeFrameTask constraint is a relative slot: the same rows with a signed ancestor table)
        cfg_like = configurations
        if per_instance and hasattr(configurations, "q") and not isinstance(configurations, (list, tuple)):
            # (a ConfigurationBatch next to per-instance constraint objects: the reader of per-instance task lists only wants
            # `.model` and `.q` of each configuration -- no Configuration object, no forward kinematics)
            import types

            cfg_like = [types.SimpleNamespace(model=model, q=row) for row in configurations.q]
        cplan = _device_kinematics_plan_tasks_raw(cfg_like, constraints)
        if cplan is None or cplan[4] is not None or cplan[5]:
            return _declined("constraints= the device tables cannot hold")
        _, _, cspecs, cT, _, _ = cplan
        if len(cspecs) > 2:
            return _declined("more than two constraint frame tasks")  # (csrc/dispatch.h: kRolloutMaxEqFrames)
        for k, sp in enumerate(cspecs):
            tg = cT[k] if isinstance(cT, (list, tuple)) else cT[:, k]
            slot = add_slot((sp[0], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 1.0, 0.0), tg)
            cons.append((slot, float(sp[3])))
    if barriers:
        ident = np.array([1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, 0, 0, 0])
        have = {sp[0] for sp in specs if not isinstance(sp[0], tuple)}  # (ordinary slots: a barrier needs the world pose)
        for bar in barriers:
            for f in ((bar.frame,) if type(bar) is PositionBarrier else tuple(bar.frames)):
                if f not in have:
                    if not any(fr.name == f for fr in getattr(model, "frames", ())):
                        return _declined(f"barrier frame {f!r} is not a frame of the model")
                    add_slot((f, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 1.0, 0.0), ident)
                    have.add(f)
    if len(specs) > 32 or any(isinstance(sp[0], tuple) and i >= 16 for i, sp in enumerate(specs)):
        return _declined("more frame slots than the device model holds (32; relative slots among the first 16)")
    fb = _explicit_floating_base_limit(model, limits)
    if fb is not None and limits is not None and not any(j.kind == "free_flyer" for j in getattr(model, "joints", ())):
        # (written into the call's limits for a model without a free-flyer root: the device tables have no root box to
        # put it in -- declined here, not as a constructor error of the device model)
        return _declined("a FloatingBaseVelocityLimit on a model without a free-flyer root joint")
    return model, q, specs, T, posture, extras, tuple(barriers or ()), gain, acc, vmax, tuple(cons), fb


def _default_limits_gain(model, limits):
    """``(config_limit_gain, acceleration tables or None)`` when ``limits`` amounts to what the device kernels form from
    tables: the model's default limits (``pink/solve_ik.py:94-105``) -- ``None`` itself, or an explicit list holding
    exactly one ConfigurationLimit and one VelocityLimit of this model (its own velocity vector allowed), at most one
    FloatingBaseVelocityLimit of this model -- optionally with ONE AccelerationLimit of this model on joints behind
    the root (``pink/limits/acceleration_limit.py:158-199``: a box from ``q``, the previous displacement and three
    per-coordinate tables ``a_max / Delta_q_prev / has_configuration_limit``) -- else ``None``."""
    from .limits import AccelerationLimit, ConfigurationLimit, VelocityLimit

    if not hasattr(model, "ensure_limits"):
        return None
    model.ensure_limits()
    if limits is None:
        return float(model.configuration_limit.config_limit_gain), None, None
    from .limits import FloatingBaseVelocityLimit

    cl = [l for l in limits if type(l) is ConfigurationLimit]
    vl = [l for l in limits if type(l) is VelocityLimit]
    al = [l for l in limits if type(l) is AccelerationLimit]
    rest = [l for l in limits if type(l) not in (ConfigurationLimit, VelocityLimit, AccelerationLimit)]
    if len(cl) != 1 or len(vl) > 1 or cl[0].model is not model or (vl and vl[0].model is not model) or len(al) > 1:
        return None
    # (an explicit list holds no FloatingBaseVelocityLimit, or ONE of this model -- the model's own or any other: the plan
    # carries it to the device model, _explicit_floating_base_limit)
    if rest and (len(rest) != 1 or type(rest[0]) is not FloatingBaseVelocityLimit or getattr(rest[0], "model", model) is not model):
        return None
    # (a VelocityLimit built with its own vector -- how joints without a model limit get one, velocity_limit.py:46-58 --
    # is the same table with other numbers: the device model of the call carries that vector instead of the model's)
    vmax = None
    if not vl:  # (a list without a VelocityLimit: no velocity rows -- the table of the call bounds no coordinate)
        vmax = np.full(model.nv, np.inf)
    elif not np.array_equal(vl[0].velocity_limit, np.asarray(model.velocityLimit, dtype=float)):
        # (entries on a free-flyer's tangent coordinates become the box of the root coordinates: DeviceRollout)
        vmax = np.ascontiguousarray(vl[0].velocity_limit, dtype=np.float64)
    acc = None
    if al:
        a = al[0]
        if a.model is not model or any(j.kind == "free_flyer" and j.idx_v in a.indices for j in model.joints):
            return None
        if a.projection_matrix is not None:
            acc = np.zeros((3, model.nv))  # rows: a_max (0 = no bound on the coordinate), Delta_q_prev, has_configuration_limit
            acc[0, a.indices], acc[1, a.indices], acc[2, a.indices] = a.a_max, a.Delta_q_prev, a.has_configuration_limit
    return float(cl[0].config_limit_gain), acc, vmax


def _explicit_floating_base_limit(model, limits):
    """The FloatingBaseVelocityLimit the call's limits hold (``pink/solve_ik.py:94-105``: the model's own by default, or the
    one written into an explicit list), or ``None``."""
    from .limits import FloatingBaseVelocityLimit

    if limits is None:
        return getattr(model.ensure_limits(), "floating_base_velocity_limit", None)
    for l in limits:
        if type(l) is FloatingBaseVelocityLimit:
            return l
    return None


def _barrier_key(bar):
    if hasattr(bar, "frames"):  # BodySphericalBarrier
        return ("spherical", tuple(bar.frames), float(bar.d_min), float(np.asarray(bar.gain, float).ravel()[0]), float(bar.safe_displacement_gain))
    return (bar.frame, tuple(bar.indices), None if bar.p_min is None else tuple(np.asarray(bar.p_min, float)),
            None if bar.p_max is None else tuple(np.asarray(bar.p_max, float)), tuple(np.asarray(bar.gain, float)),
            float(bar.safe_displacement_gain))


def _extra_task(model, t):
    """What the whole-step kernel forms on chip beyond FrameTasks and the PostureTask, for a task object shared by the
    whole batch: ``("const", A, b, q_0, cost, gain, lm)`` -- a dense task with a constant Jacobian
    (LinearHolonomicTask / JointCouplingTask whose ``A`` has no entry on free-flyer coordinates:
    ``pink/tasks/linear_holonomic_task.py:103-148``) -- or ``("diag", col0, e, cost, gain, lm)`` -- an identity-Jacobian
    task with a batch-constant error (``damping_task.py``, ``low_acceleration_task.py:46-84``,
    ``joint_velocity_task.py:59-110``) -- or ``None``."""
    from .tasks.linear_holonomic_task import JointCouplingTask, JointVelocityTask, LinearHolonomicTask
    from .tasks.posture_task import DampingTask, LowAccelerationTask
    from .utils import get_root_joint_dim

    nv, root_nv = model.nv, get_root_joint_dim(model)[1]
    ty = type(t)
    if ty in (LinearHolonomicTask, JointCouplingTask):
        if t.A.shape[1] != nv or any(j.kind == "free_flyer" and np.any(t.A[:, j.idx_v:j.idx_v + 6]) for j in model.joints):
            return None
        q_0 = model.neutral() if t.q_0 is None else np.asarray(t.q_0, dtype=np.float64)
        return ("const", np.ascontiguousarray(t.A, dtype=np.float64), np.asarray(t.b, dtype=np.float64), q_0, t.cost, float(t.gain), float(t.lm_damping))
    if ty is DampingTask:
        return ("diag", root_nv, np.zeros(nv - root_nv), t.cost, float(t.gain), float(t.lm_damping))
    if ty is LowAccelerationTask:
        return ("diag", 0, np.zeros(nv) if t.Delta_q_prev is None else -np.asarray(t.Delta_q_prev, dtype=np.float64), t.cost, float(t.gain), float(t.lm_damping))
    if ty is JointVelocityTask:
        r = 0 if model.root_joint is None else model.root_joint.nv
        if t.target_v is None or t.target_dt is None or t.target_v.shape[0] != nv - r:
            return None  # (the host path raises the reference's errors)
        return ("diag", r, t.target_dt * t.target_v, t.cost, float(t.gain), float(t.lm_damping))
    return None


def _extras_key(extras):
    """Hashable identity of the extra tasks of a plan: everything but the errors of the diagonal ones (those change from
    one control step to the next and are uploaded per call)."""
    out = []
    for x in extras:
        cost = None if x[-3] is None else tuple(np.atleast_1d(np.asarray(x[-3], dtype=float)).tolist())
        if x[0] == "const":
            out.append(("const", x[1].shape, x[1].tobytes(), x[2].tobytes(), x[3].tobytes(), cost, x[-2], x[-1]))
        else:
            out.append(("diag", x[1], x[2].shape[0], cost, x[-2], x[-1]))
    return tuple(out)


def _admissible_plan(plan):
    """Last look at a task plan before the device model is built from it: what the device route's tables cannot hold is
    declined here (``None``: the hybrid / host routes serve the call) instead of surfacing as a constructor error.

    * relative slots live among the first 16 frames of a device model (``csrc/model_tables.h``);
    * constant-row tasks share one reference configuration on chip (``pink_amd/rollout.py``): several
      LinearHolonomicTask / JointCouplingTask objects with different ``q_0`` are brought to the first one's,
      ``A (q - q0_i) - b_i = A (q - q0_0) - (b_i + A (q0_i - q0_0))`` (``pink/tasks/linear_holonomic_task.py:117-148``;
      ``A`` has no entry on free-flyer coordinates, where the difference is not a subtraction)."""
    if plan is None:
        return None
    model, q, specs, targets, posture, extras = plan
    if any(isinstance(sp[0], tuple) and i >= 16 for i, sp in enumerate(specs)):
        return None
    const = [x for x in extras if x[0] == "const"]
    if len(const) > 1 and any(not np.array_equal(x[3], const[0][3]) for x in const):
        q00, out = const[0][3], []
        for x in extras:
            if x[0] == "const" and not np.array_equal(x[3], q00):
                # (A is zero on free-flyer coordinates, where a reference configuration may hold anything -- non-finite
                # entries included: 0 * inf must not reach b)
                shift = np.nan_to_num(model.difference(q00, x[3]), nan=0.0, posinf=0.0, neginf=0.0)
                x = ("const", x[1], x[2] + x[1] @ shift, q00) + tuple(x[4:])
            out.append(x)
        extras = tuple(out)
    return model, q, specs, targets, posture, extras


def _device_kinematics_plan_tasks(configurations, tasks):
    """The task half of :func:`_device_kinematics_plan`: ``(model, q, specs, targets, posture, extras)`` or ``None``."""
    return _admissible_plan(_device_kinematics_plan_tasks_raw(configurations, tasks))


def _device_kinematics_plan_tasks_raw(configurations, tasks):
    from .configuration import ConfigurationBatch
    from .exceptions import TargetNotSet
    from .tasks.frame_task import FrameTask
    from .tasks.posture_task import PostureTask
    from .tasks.relative_frame_task import RelativeFrameTask

    B = len(configurations)
    as_arrays = isinstance(configurations, ConfigurationBatch)
    if not as_arrays and not (len(tasks) == B and isinstance(tasks[0], (list, tuple))):
        # a list of Configuration objects with ONE task list for all of them: the tasks are shared objects, their
        # targets one for all or per instance as arrays (set_target_poses / set_target_batch) -- same rules as below
        model = configurations[0].model
        if any(c.model is not model for c in configurations):
            return None
        configurations = ConfigurationBatch(model, np.stack([np.asarray(c.q, dtype=np.float64) for c in configurations]))
        as_arrays = True
    if as_arrays:
        # arrays in, arrays out: the tasks are shared objects carrying per-instance targets as arrays
        model, q = configurations.model, configurations.q
        specs, targets, posture, extras = [], [], None, []
        for t in tasks:
            if type(t) is FrameTask:
                if t.target_array() is not None:
                    # ([B, 12] poses or [B, 7] translation + quaternion: uploaded as they are, the latter expanded on the device)
                    if t.target_array().shape[0] != B:
                        raise PinkError(f"FrameTask {t.frame!r}: {t.target_array().shape[0]} target poses for {B} configurations")
                    targets.append(t.target_array())
                elif t.transform_target_to_world is not None:
                    targets.append(np.broadcast_to(_pose12(t.transform_target_to_world), (B, 12)))
                else:
                    raise TargetNotSet(f"no target set for frame '{t.frame}'")
                specs.append(_spec_of(t))
            elif type(t) is RelativeFrameTask:
                # the pose of one frame in another (relative_frame_task.py:142-231): one target, in the root frame
                if t.transform_target_to_root is None:
                    raise TargetNotSet(f"target pose of frame '{t.frame}' in frame '{t.root}' is undefined")
                targets.append(np.broadcast_to(_pose12(t.transform_target_to_root), (B, 12)))
                specs.append(_spec_of(t))
            elif type(t) is PostureTask:
                if posture is not None or np.ndim(t.cost) != 0:
                    return None
                if t.target_q_batch is not None:
                    if t.target_q_batch.shape != q.shape:
                        raise PinkError(f"PostureTask: targets {t.target_q_batch.shape} for configurations {q.shape}")
                    qp = t.target_q_batch
                elif t.target_q is not None:
                    qp = np.asarray(t.target_q, dtype=np.float64)  # [nq]: one target for all
                else:
                    return None
                posture = (float(t.cost), float(t.gain), float(t.lm_damping), qp)
            else:
                x = _extra_task(model, t)
                if x is None:
                    return None
                extras.append(x)
        if not specs:
            return None
        return model, q, specs, targets, posture, tuple(extras)  # (targets: one [B, 12] array per frame task, uploaded as they are)
    model = configurations[0].model
    if any(c.model is not model for c in configurations):
        return None
    _check_same_length(tasks)
    slots = [[tasks[b][k] for b in range(B)] for k in range(len(tasks[0]))]
    specs, targets, posture, extras = [], [], None, []
    for col in slots:
        t0 = col[0]
        if type(t0) is RelativeFrameTask:
            if any(type(t) is not RelativeFrameTask or t.frame != t0.frame or t.root != t0.root or t.gain != t0.gain
                   or t.lm_damping != t0.lm_damping or not np.array_equal(t.cost, t0.cost) for t in col):
                return None
            if any(t.transform_target_to_root is None for t in col):
                raise TargetNotSet(f"target pose of frame '{t0.frame}' in frame '{t0.root}' is undefined")
            specs.append(_spec_of(t0))
            targets.append(np.stack([_pose12(t.transform_target_to_root) for t in col]))
            continue
        if type(t0) not in (FrameTask, PostureTask):
            # a task the kernel forms from tables: one object shared by every instance
            x = _extra_task(model, t0) if all(t is t0 for t in col) else None
            if x is None:
                return None
            extras.append(x)
            continue
        if col[-1] is not t0:  # (one task object for the whole batch: nothing to compare)
            ty, g0, l0 = type(t0), t0.gain, t0.lm_damping
            if not all(type(t) is ty and t.gain == g0 and t.lm_damping == l0 for t in col):
                return None
            try:  # the costs of the whole column in one comparison
                costs = np.array([t.cost for t in col], dtype=float)
            except (TypeError, ValueError):
                return None
            if costs.shape[0] != B or (costs != costs[0]).any():
                return None
        if type(t0) is FrameTask:
            if any(t.frame != t0.frame or t.transform_target_to_world is None for t in col):
                return None
            specs.append(_spec_of(t0))
            tg = np.empty((B, 12))
            for b, t in enumerate(col):
                T_b = t.transform_target_to_world
                tg[b, :9] = np.asarray(T_b.rotation, dtype=float).reshape(9)
                tg[b, 9:] = T_b.translation
            targets.append(tg)
        elif type(t0) is PostureTask:
            if posture is not None or any(t.target_q is None for t in col) or np.ndim(t0.cost) != 0:
                return None
            posture = (float(t0.cost), float(t0.gain), float(t0.lm_damping), np.stack([t.target_q for t in col]))
        else:
            return None
    if not specs:
        return None
    q = np.stack([np.asarray(c.q, dtype=np.float64) for c in configurations])
    return model, q, specs, np.stack(targets, axis=1), posture, tuple(extras)


# Device-resident state of the last few (model, batch size, task stack) combinations solve_ik_batch was called with,
# kept per solver (= per device): model tables, buffers and descriptors are built once, a call only moves q and the
# targets in and dq out.
_ROLLOUT_CACHE_MAX = 4
_PIPELINE_MIN_B = 16384  # batches from here on are uploaded and solved in four overlapping ranges
_CACHED_APIS: list = []


def _rollout_cache(api) -> dict:
    cache = getattr(api, "_pinkhip_rollouts", None)
    if cache is None:
        cache = {}
        api._pinkhip_rollouts = cache
        _CACHED_APIS.append(api)
    return cache


def clear_device_cache() -> None:
    """Release the device buffers :func:`solve_ik_batch` keeps between calls."""
    for api in _CACHED_APIS:
        cache = getattr(api, "_pinkhip_rollouts", {})
        for ro in cache.values():
            try:
                ro.free()
            except Exception:  # noqa: BLE001  (a solver that was closed in the meantime)
                pass
        cache.clear()


def _model_fingerprint(model, frames) -> int:
    """Content hash of what a cached device state bakes in from the model: joint limits, velocity limits and the
    placements of the task frames (edits to them must not be served from the cache)."""
    parts = [model.lowerPositionLimit.tobytes(), model.upperPositionLimit.tobytes(), model.velocityLimit.tobytes()]
    names = [n for entry in frames for n in (entry if isinstance(entry, tuple) else (entry,))]  # (frame, root) of a relative slot
    for name in names:
        f = model.frames[model.getFrameId(name)]
        parts.append(np.asarray(f.placement.rotation, dtype=float).tobytes() + np.asarray(f.placement.translation, dtype=float).tobytes())
        parts.append(str(f.joint).encode())
    parts.append(str(len(model.joints)).encode())
    return hash(b"".join(parts))


def _solve_on_device(plan, dt, damping, safety_break, api, max_iter, out=None):
    """FK, task rows, limits and the QP for the whole batch in device kernels (one launch where the whole-step
    kernel covers the model): the host only hands over ``q`` and the targets."""
    from .rollout import DeviceRollout

    model, q, specs, T, posture, extras, bars, limit_gain, acc, vmax, cons, fb = plan
    B = q.shape[0]
    pkey = None if posture is None else posture[:3]
    fkey = None if fb is None else (fb.base_frame, tuple(float(v) for v in fb.twist_max))
    key = (id(model), _model_fingerprint(model, [sp[0] for sp in specs]), B, tuple(specs), float(dt), float(damping), pkey,
           int(max_iter), float(limit_gain), tuple(_barrier_key(b) for b in bars), fkey, _extras_key(extras),
           None if acc is None else (acc[0].tobytes(), acc[2].tobytes()),  # (the previous displacement moves per call)
           None if vmax is None else vmax.tobytes(), cons)
    cache = _rollout_cache(api)
    ro = cache.pop(key, None)
    fresh = False
    if ro is None:
        kw = {}
        if posture is not None:
            kw = dict(posture_cost=posture[0], posture_gain=posture[1], posture_lm_damping=posture[2], q_posture=posture[3])
        ro = DeviceRollout(api, model, q, specs, dt, damping=damping, config_limit_gain=limit_gain,
                           max_iter=max_iter, fused="kernel", safety_break=safety_break, position_barriers=bars, floating_base_limit=fb,
                           constraint_slots=cons,
                           const_tasks=[x[1:] for x in extras if x[0] == "const"], diag_tasks=[x[1:] for x in extras if x[0] == "diag"],
                           acceleration_limit=acc, velocity_limit=vmax, **kw)
        ro._cache_owner = model  # keeps id(model) of the key alive and unique
        ro.velocity_out = True  # (the whole-step kernel hands out dq / dt: no division over the array afterwards)
        fresh = True
    try:
        if not fresh and any(x[0] == "diag" for x in extras):  # (a LowAccelerationTask / JointVelocityTask moves on every step)
            ro.set_diag_errors([x[2] for x in extras if x[0] == "diag"])
        if not fresh and acc is not None:  # (AccelerationLimit.set_last_integration between two control steps)
            ro.set_acceleration_limit(acc)
        # large batches with one target array per frame task: uploads of one range overlap the kernel of the previous
        qp = None if posture is None else posture[3]
        if not (B >= _PIPELINE_MIN_B and isinstance(T, (list, tuple)) and ro.solve_pipelined(q, T, qp, safety_break, out=out)):
            if not fresh:
                ro.reset(q, qp, safety_break)
            ro.set_targets(T)
            ro.step(integrate=False)
        api.sync()
        out = ro.last_step() + (ro.last_path, bool(getattr(ro, "scaled", False)))
    except BaseException:
        try:
            api.sync()  # (asynchronous copies of the failed call may still read / write the buffers released below)
        except Exception:  # noqa: BLE001
            pass
        ro.free()
        raise
    cache[key] = ro  # most recently used last
    while len(cache) > _ROLLOUT_CACHE_MAX:
        cache.pop(next(iter(cache))).free()
    return out


def _slice_plan(plan, lo, hi):
    model, q, specs, T, posture, extras, bars, limit_gain, acc, vmax, cons, fb = plan
    T = [t[lo:hi] for t in T] if isinstance(T, list) else T[lo:hi]
    if posture is not None and np.ndim(posture[3]) == 2:
        posture = posture[:3] + (posture[3][lo:hi],)
    return model, q[lo:hi], specs, T, posture, extras, bars, limit_gain, acc, vmax, cons, fb


def solve_ik_batch(configurations: Sequence, tasks: Sequence, dt: float, solver: str = "mi355x", damping: float = 1e-12,
                   limits=None, barriers=None, constraints=None, safety_break: bool = True, solver_handle=None,
                   device_kinematics: Optional[bool] = None, device_ids: Optional[Sequence[int]] = None,
                   out: Optional[np.ndarray] = None, **kwargs) -> np.ndarray:
    """Batched ``solve_ik``: velocities ``[B, nv]`` for ``B`` configurations.

    ``configurations`` is a list of :class:`Configuration` objects or a :class:`ConfigurationBatch` (one array
    ``q [B, nq]``; tasks then carry per-instance targets as arrays, ``FrameTask.set_target_poses`` /
    ``PostureTask.set_target_batch``): the array form has no per-instance Python work at all.

    Raises :class:`NoSolutionFound` listing the failing instances (the batched
    analogue of ``pink/solve_ik.py:271-273``).

    ``device_kinematics``: when the task stack is FrameTasks / RelativeFrameTasks next to tasks the whole-step kernel forms
    from tables -- one PostureTask, LinearHolonomicTasks / JointCouplingTasks on the joints behind the root, DampingTask,
    LowAccelerationTask, JointVelocityTask (one object each for the whole batch) -- under the model's default limits
    (``limits=None``, or that list written out, optionally with one AccelerationLimit on the joints behind the root),
    forward kinematics, task errors / Jacobians and limits are evaluated by the device kernel from ``q`` alone
    instead of on the host (``None``: do so for batches of 64 and more; ``True``: require it; ``"frame_rows"``: only the
    FrameTask rows on the device, everything else evaluated on the host for the whole batch -- the route any other
    stack of FrameTasks + identity-Jacobian tasks takes).
    Device buffers of the last few call shapes are kept (:func:`clear_device_cache`).

    ``out``: a ``[B, nv]`` float64 array that receives the velocities (like NumPy's ``out=``).  Allocated through
    :func:`pink_amd.pinned_empty` (page-locked), together with ``q`` and the per-instance target arrays, the device
    route moves every byte of the call by DMA while the kernels of the neighbouring ranges run.

    ``strict_route`` (keyword): ``"device"``, ``"hybrid"`` or ``"host-evaluated"`` -- raise :class:`PinkError` naming the
    first term that declined it when the call would take another route (:func:`last_solve_stats` tells which one a call
    took).  Without it a batch of 64 and more that leaves the device route logs ONE warning per stack signature and
    route (logger ``pink_amd``): the routes differ by orders of magnitude in cost.

    ``device_ids``: shard the batch contiguously over these GPUs from this one process (one handle, stream and staging
    area per device, a thread each): no collective, results concatenated on the host.
    """
    _check_solver(solver)
    from .runtime import default_solver

    max_iter = int(kwargs.get("max_iter", 0))
    strict_route = kwargs.get("strict_route")
    if strict_route not in (None, "device", "hybrid", "host-evaluated"):
        raise PinkError(f"strict_route={strict_route!r}: None, 'device', 'hybrid' or 'host-evaluated'")
    _DECLINED["reason"] = None
    if device_ids is not None:
        if solver_handle is not None:
            raise PinkError("device_ids= and solver_handle= are mutually exclusive")
        from .sharding import device_pool

        solver_handle = device_pool(device_ids)
    pool = getattr(solver_handle, "solvers", None)  # a MultiDeviceSolver
    plan = None
    rows_only = isinstance(device_kinematics, str)
    if rows_only:
        if device_kinematics != "frame_rows":
            raise PinkError(f"device_kinematics={device_kinematics!r}: True, False, None or 'frame_rows'")
        device_kinematics = None
    if not rows_only and (device_kinematics or (device_kinematics is None and len(configurations) >= 64)):
        plan = _device_kinematics_plan(configurations, tasks, limits, barriers, constraints)
        if plan is None and device_kinematics:
            raise PinkError("device_kinematics=True needs FrameTasks / RelativeFrameTasks (+ one PostureTask, constant-row and identity "
                            "tasks shared by the batch), the model's default limits (+ one AccelerationLimit), constraints made of at most "
                            "two frame tasks, and barriers that are PositionBarriers or BodySphericalBarriers with the default class-K functions")
    if plan is not None:
        from .batch_solver import BatchResult
        from .rollout import NoWholeStepKernel

        if strict_route not in (None, "device"):
            raise PinkError(f"strict_route={strict_route!r}: this call would take the 'device' route")

        try:
            if pool is not None:
                from .sharding import shard_bounds

                bounds = [shard_bounds(len(configurations), r, len(pool)) for r in range(len(pool))]
                parts = solver_handle.map(lambda r, api: _solve_on_device(_slice_plan(plan, *bounds[r]), dt, damping, safety_break, api, max_iter)
                                          if bounds[r][1] > bounds[r][0] else None)
                parts = [p for p in parts if p is not None]
                dq, status, iters, path = (np.concatenate([p[k] for p in parts]) for k in range(4))
                scaled = all(p[4] for p in parts)
                if not scaled and any(p[4] for p in parts):  # (never: the shards share the model and the kernel)
                    raise PinkError("shards disagree about the scale of dq")
            else:
                dq, status, iters, path, scaled = _solve_on_device(plan, dt, damping, safety_break, solver_handle or default_solver(), max_iter,
                                                           out=out if out is not None and out.shape == (len(configurations), plan[0].nv) else None)
        except NoWholeStepKernel:
            # no instantiation of the whole-step kernel holds this model's rows (more barrier rows / joints than the
            # tables of dispatch.h carry): the host-evaluated path below serves it, unless the caller insisted
            if device_kinematics:
                raise
            plan = _declined("no instantiation of the whole-step kernel holds this model's rows")
        else:
            result = BatchResult(dq, status, iters, path)
            _record_stats(result, "device")
            _report_route("device", len(configurations), tasks, limits, barriers, constraints, strict_route)
            if status.any():
                raise NoSolutionFound(None, result, result.failed_indices(), status[status != 0])
            if scaled:  # the kernel wrote v = dq / dt (pink/solve_ik.py:274)
                if out is not None and dq is not out:
                    np.copyto(out, dq)
                    return out
                return dq
            if out is not None and dq is not out:
                return np.divide(dq, dt, out=out)
            return np.divide(dq, dt, out=dq)  # v = dq / dt, in place: dq is this call's own array
    B = len(configurations)
    if strict_route == "device":  # (decided before any work is done)
        _report_route("hybrid or host-evaluated", B, tasks, limits, barriers, constraints, strict_route)
    if B and hasattr(configurations, "check_limits"):
        configurations.check_limits(safety_break=safety_break)
    elif B:
        from .configuration import ConfigurationBatch

        model = configurations[0].model
        if hasattr(model, "joints") and all(c.model is model for c in configurations):  # (one vectorised check, solve_ik.py:260)
            ConfigurationBatch(model, np.stack([np.asarray(c.q, dtype=np.float64) for c in configurations])).check_limits(safety_break=safety_break)
        else:
            for cfg in configurations:
                cfg.check_limits(safety_break=safety_break)
    api = solver_handle or default_solver()
    if (device_kinematics is None and (B >= 64 or rows_only) and pool is None and kwargs.get("gpu_frame_tasks", True)
            and hasattr(api, "fk_frame_tasks") and hasattr(api, "solve_raw")):
        result = _solve_hybrid(configurations, tasks, dt, damping, limits, barriers, constraints, api, max_iter) if strict_route in (None, "hybrid") else None
        if result is not None:
            _record_stats(result, "hybrid")
            _report_route("hybrid", B, tasks, limits, barriers, constraints, strict_route, warn=not rows_only)
            if not result.all_found:
                raise NoSolutionFound(None, result, result.failed_indices(), result.status[result.status != 0])
            return np.divide(result.dq, dt, out=out if out is not None else result.dq)
    _report_route("host-evaluated", B, tasks, limits, barriers, constraints, strict_route, warn=device_kinematics is None and not rows_only)
    batch = pack_configurations(configurations, tasks, dt, damping, limits, barriers, pool[0] if pool else solver_handle,
                                gpu_frame_tasks=bool(kwargs.get("gpu_frame_tasks", True)), constraints=constraints)
    result = api.solve(batch, max_iter=max_iter)
    _record_stats(result, "host-evaluated")
    if not result.all_found:
        raise NoSolutionFound(batch, result, result.failed_indices(), result.status[result.status != 0])
    return np.divide(result.dq, dt, out=out) if out is not None else result.dq / dt


def _solve_hybrid(configurations, tasks, dt, damping, limits, barriers, constraints, api, max_iter):
    """The host-evaluated stack with its FrameTask rows formed on the device (:mod:`pink_amd.hybrid`): a
    :class:`BatchResult`, or ``None`` when the stack is not of that shape (a dense task other than FrameTasks,
    equality constraints, configurations of several models)."""
    from . import hybrid
    from .batch_solver import BatchResult
    from .configuration import ConfigurationBatch
    from .kinematics_batch import BatchKinematics

    B = len(configurations)
    if isinstance(configurations, ConfigurationBatch):
        model, q, make = configurations.model, configurations.q, configurations.kinematics
    else:
        model = configurations[0].model
        if not hasattr(model, "joints") or any(c.model is not model for c in configurations):
            return None
        q = np.stack([np.asarray(c.q, dtype=np.float64) for c in configurations])
        make = lambda: BatchKinematics(model, q)  # noqa: E731
    slots = _task_slots(tasks, B)
    cslots = _task_slots(constraints, B) if constraints else []
    frame_slots = hybrid.plan(model, slots, cslots)
    if frame_slots is None:
        return None
    if limits is None:  # model defaults, pink/solve_ik.py:94-105
        model.ensure_limits()
        limits = [model.configuration_limit, model.velocity_limit]
        if getattr(model, "floating_base_velocity_limit", None) is not None:
            limits.append(model.floating_base_velocity_limit)
    frames = tuple((slots[k][0].frame, slots[k][0].root) if hasattr(slots[k][0], "root") else slots[k][0].frame for k in frame_slots)
    key = ("hybrid", id(model), _model_fingerprint(model, frames), B, frames)
    cache = _rollout_cache(api)
    state = cache.pop(key, None)
    if state is None:
        state = hybrid.HybridState(api, model, frames, B)
        state._cache_owner = model
    try:
        dq, status, iters, path = hybrid.solve(state, q, make, slots, frame_slots, limits, barriers, dt, damping, max_iter, cslots)
    except BaseException:
        state.free()
        raise
    cache[key] = state
    while len(cache) > _ROLLOUT_CACHE_MAX:
        cache.pop(next(iter(cache))).free()
    return BatchResult(dq, status, iters, path)


def pinned_empty(shape, dtype=np.float64, solver_handle=None) -> np.ndarray:
    """Uninitialised array in page-locked host memory of the (default) solver's device (``pinkhip_host_alloc``): arrays
    a caller fills in place and hands to :func:`solve_ik_batch` -- ``ConfigurationBatch(model, q)``,
    ``FrameTask.set_target_poses(..., out=)``, ``out=`` -- then move by DMA at the PCIe rate, overlapped with the
    kernels, instead of being staged by the runtime."""
    from .runtime import default_solver

    return (solver_handle or default_solver()).pinned_empty(shape, dtype)


_LAST_STATS: dict = {}


def _record_stats(result, route: str) -> None:
    """Keeps the result of the call; the figures are worked out when :func:`last_solve_stats` asks for them (three passes
    over the batch: 0.2 ms of a 1.5 ms call at B = 65 536)."""
    _LAST_STATS.clear()
    _LAST_STATS.update(route=route, _result=result)


def last_solve_stats() -> dict:
    """What the last :func:`solve_ik_batch` call of this process did: ``route`` (``"device"``: kinematics, rows and QP
    formed on the device from ``q``; ``"hybrid"``: FrameTask rows formed on the device from ``q``, the other tasks /
    limits / barriers evaluated on the host, QP on the device; ``"host-evaluated"``: everything evaluated on the host,
    QP on the device), ``instances``, ``failed``, ``iters_mean`` and ``paths`` -- the share of the batch per solver path
    (:meth:`pink_amd.batch_solver.BatchResult.path_fractions`; ``handover`` is the share that paid for both solvers)."""
    result = _LAST_STATS.pop("_result", None)
    if result is not None:
        _LAST_STATS.update(instances=int(result.status.shape[0]), failed=int((result.status != 0).sum()),
                           iters_mean=float(result.iters.mean()) if result.iters.size else 0.0, paths=result.path_fractions())
    return dict(_LAST_STATS)
