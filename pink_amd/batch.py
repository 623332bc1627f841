"""Packed batch of differential-IK problems: the data contract of the HIP path.

A batch holds ``B`` independent instances of the QP that ``pink.build_ik``
assembles one at a time (reference ``pink/solve_ik.py:198-203``), kept as *raw
terms* so that the stacking ``H = damping I + sum J^T W^2 J + mu I``,
``c = sum gain J^T W^2 e`` (``pink/tasks/task.py:145-167``) is fused into the
solve kernel instead of being materialised on the host.

Layout (all float64, C-contiguous; this is what ``include/pinkhip.h`` takes):

* ``J  [B, Kd, nv]`` rows of the tasks whose Jacobian is dense (FrameTask, ...),
  task after task;
* ``e  [B, K]`` task errors: the ``Kd`` dense rows first, then the rows of the
  *diagonal* tasks (tasks whose Jacobian is ``eye(nv)[col0:col0+k]``:
  PostureTask ``pink/tasks/posture_task.py:128-129``, DampingTask, ...), for
  which no Jacobian is stored at all;
* ``cost [K]`` (or ``[B, K]``) per-row weights ``w`` (the diagonal of ``W``);
* ``lb, ub [B, nv]`` per-coordinate box: every limit row of the form
  ``+-e_i dq <= h`` (ConfigurationLimit ``pink/limits/configuration_limit.py
  :117-120``, VelocityLimit ``pink/limits/velocity_limit.py:118-120``) is merged
  into it -- same feasible set, hence the same unique minimiser; unbounded
  coordinates hold ``-+inf``;
* ``Gd [B, md, nv]``, ``hd [B, md]`` the remaining (dense) inequality rows:
  barriers ``pink/barriers/barrier.py:246-254`` and anything else.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

TASK_DENSE = 0
TASK_DIAGONAL = 1


@dataclass
class DenseTaskTerm:
    """A task given by its batched Jacobian and error (``Task.compute_jacobian``
    / ``compute_error`` evaluated per instance)."""

    J: np.ndarray  # [B, k, nv]
    e: np.ndarray  # [B, k]
    cost: object = None  # None | scalar | [k]  (pink/tasks/task.py:148-156)
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class DiagonalTaskTerm:
    """A task whose Jacobian is ``eye(nv)[col0:col0+k]`` (never stored)."""

    col0: int
    e: np.ndarray  # [B, k]
    cost: object = None
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class BarrierTerm:
    """A control-barrier function: rows ``-J_h/dt dq <= gain*h`` plus the optional
    safe-displacement regulariser (``pink/barriers/barrier.py:190-203,246-254``)."""

    J_h: np.ndarray  # [B, d, nv]
    h: np.ndarray  # [B, d]  barrier values (after the class-K function)
    gain: object = 1.0  # scalar | [d]
    safe_displacement_gain: float = 0.0
    safe_displacement: Optional[np.ndarray] = None  # [B, nv]


def _expand_cost(cost, k: int) -> np.ndarray:
    if cost is None:
        return np.ones(k)
    w = np.asarray(cost, dtype=np.float64)
    if w.ndim == 0:
        return np.full(k, float(w))
    if w.shape[-1] != k:
        raise ValueError(f"cost has {w.shape[-1]} entries, task has {k} rows")
    return w


@dataclass
class IKBatch:
    """Packed batch; see the module docstring for the layout."""

    nv: int
    J: np.ndarray
    e: np.ndarray
    cost: np.ndarray
    task_rows: np.ndarray  # int32 [T+1] offsets into the K rows of ``e``
    task_kind: np.ndarray  # int32 [T]
    task_col0: np.ndarray  # int32 [T]   first column of a diagonal task
    gain: np.ndarray  # [T]
    lm_damping: np.ndarray  # [T]
    lb: np.ndarray
    ub: np.ndarray
    Gd: np.ndarray
    hd: np.ndarray
    barrier_rows: np.ndarray  # int32 [nb+1] offsets into the md dense rows
    barrier_safe_gain: np.ndarray  # [nb]
    c_extra: Optional[np.ndarray] = None  # [B, nv]
    damping: float = 1e-12
    dt: float = 1e-3
    n_eq: int = 0  # the first n_eq rows of Gd/hd are equalities (constraints=, solve_ik.py:125-149)
    meta: dict = field(default_factory=dict)

    @property
    def B(self) -> int:
        return int(self.e.shape[0])

    @property
    def K(self) -> int:
        return int(self.e.shape[1])

    @property
    def Kd(self) -> int:
        return int(self.J.shape[1])

    @property
    def md(self) -> int:
        return int(self.Gd.shape[1])

    @property
    def T(self) -> int:
        return int(self.task_kind.size)

    def bytes_per_qp(self) -> int:
        """Algorithmic HBM bytes of one stack+solve:
        ``8 (Kd nv + K + 2 nv + md (nv+1) + nv) + 4``."""
        nv = self.nv
        return 8 * (self.Kd * nv + self.K + 2 * nv + self.md * (nv + 1) + nv) + 4

    def bytes_per_stack(self) -> int:
        """Algorithmic HBM bytes of one stack-only evaluation (J, e -> H, c)."""
        nv = self.nv
        return 8 * (self.Kd * nv + self.K + nv * nv + nv)

    def slice(self, lo: int, hi: int) -> "IKBatch":
        """Instances ``[lo, hi)`` (contiguous shard)."""
        cost = self.cost[lo:hi] if self.cost.ndim == 2 else self.cost
        return IKBatch(
            self.nv, self.J[lo:hi], self.e[lo:hi], cost, self.task_rows, self.task_kind,
            self.task_col0, self.gain, self.lm_damping, self.lb[lo:hi], self.ub[lo:hi],
            self.Gd[lo:hi], self.hd[lo:hi], self.barrier_rows, self.barrier_safe_gain,
            None if self.c_extra is None else self.c_extra[lo:hi], self.damping, self.dt, self.n_eq,
            dict(self.meta),
        )


def pack_terms(
    nv: int,
    tasks: Sequence[object],
    dt: float,
    damping: float = 1e-12,
    boxes: Sequence[tuple] = (),
    dense_rows: Sequence[tuple] = (),
    barriers: Sequence[BarrierTerm] = (),
    batch_size: Optional[int] = None,
    equality_rows: Sequence[tuple] = (),
) -> IKBatch:
    """Pack per-task / per-limit terms into an :class:`IKBatch`.

    ``boxes`` holds ``(lb, ub)`` pairs (``[B, nv]`` or ``[nv]``) that are
    intersected; ``dense_rows`` holds ``(G [B, r, nv], h [B, r])`` pairs;
    ``equality_rows`` holds ``(A [B, r, nv], b [B, r])`` pairs enforced as ``A dq = b``
    (they become the leading dense rows).
    Dense tasks are moved ahead of diagonal ones (the objective is a sum, so
    the order is immaterial: ``pink/solve_ik.py:57-60``).
    """
    B = batch_size
    for t in tasks:
        B = t.e.shape[0] if B is None else B
    for b in barriers:
        B = b.h.shape[0] if B is None else B
    for lb, ub in boxes:
        if B is None and np.ndim(lb) == 2:
            B = np.shape(lb)[0]
    if B is None:
        B = 1

    dense = [t for t in tasks if isinstance(t, DenseTaskTerm)]
    diag = [t for t in tasks if isinstance(t, DiagonalTaskTerm)]
    if len(dense) + len(diag) != len(tasks):
        raise TypeError("tasks must be DenseTaskTerm or DiagonalTaskTerm")

    rows = [0]
    kinds: List[int] = []
    col0: List[int] = []
    gains: List[float] = []
    lms: List[float] = []
    costs: List[np.ndarray] = []
    errs: List[np.ndarray] = []
    Js: List[np.ndarray] = []
    cost_batched = False
    for t in dense:
        Jt = np.asarray(t.J, dtype=np.float64)
        if Jt.ndim != 3 or Jt.shape[0] != B or Jt.shape[2] != nv:
            raise ValueError(f"dense task Jacobian must be [B={B}, k, nv={nv}], got {Jt.shape}")
        k = Jt.shape[1]
        Js.append(Jt)
        errs.append(np.asarray(t.e, dtype=np.float64).reshape(B, k))
        w = _expand_cost(t.cost, k)
        cost_batched |= w.ndim == 2
        costs.append(w)
        rows.append(rows[-1] + k)
        kinds.append(TASK_DENSE)
        col0.append(0)
        gains.append(float(t.gain))
        lms.append(float(t.lm_damping))
    for t in diag:
        et = np.asarray(t.e, dtype=np.float64)
        k = et.shape[1]
        if t.col0 < 0 or t.col0 + k > nv:
            raise ValueError("diagonal task exceeds the tangent space")
        errs.append(et.reshape(B, k))
        w = _expand_cost(t.cost, k)
        cost_batched |= w.ndim == 2
        costs.append(w)
        rows.append(rows[-1] + k)
        kinds.append(TASK_DIAGONAL)
        col0.append(int(t.col0))
        gains.append(float(t.gain))
        lms.append(float(t.lm_damping))

    Kd = sum(j.shape[1] for j in Js)
    J = np.concatenate(Js, axis=1) if Js else np.zeros((B, 0, nv))
    e = np.concatenate(errs, axis=1) if errs else np.zeros((B, 0))
    if cost_batched:
        costs = [np.broadcast_to(w, (B, w.shape[-1])) for w in costs]
        cost = np.concatenate(costs, axis=1) if costs else np.zeros((B, 0))
    else:
        cost = np.concatenate(costs) if costs else np.zeros(0)
    assert J.shape[1] == Kd

    lb = ub = None  # (the first box is copied: no passes over [B, nv] to fill with +-inf and intersect)
    for blo, bhi in boxes:
        blo = np.broadcast_to(np.asarray(blo, dtype=np.float64), (B, nv))
        bhi = np.broadcast_to(np.asarray(bhi, dtype=np.float64), (B, nv))
        lb = np.array(blo) if lb is None else np.maximum(lb, blo)  # (a copy: the batch owns its arrays)
        ub = np.array(bhi) if ub is None else np.minimum(ub, bhi)
    if lb is None:
        lb, ub = np.full((B, nv), -np.inf), np.full((B, nv), np.inf)

    G_list = [np.asarray(A, dtype=np.float64) for A, _ in equality_rows]
    h_list = [np.asarray(bb, dtype=np.float64) for _, bb in equality_rows]
    n_eq = sum(g.shape[1] for g in G_list)
    G_list += [np.asarray(G, dtype=np.float64) for G, _ in dense_rows]
    h_list += [np.asarray(h, dtype=np.float64) for _, h in dense_rows]
    brow = [sum(g.shape[1] for g in G_list)]
    bsafe: List[float] = []
    c_extra = None
    for b in barriers:
        Jh = np.asarray(b.J_h, dtype=np.float64)
        d = Jh.shape[1]
        g = np.asarray(b.gain, dtype=np.float64)
        g = np.full(d, float(g)) if g.ndim == 0 else g
        G_list.append(-Jh / dt)  # barrier.py:246
        h_list.append(g * np.asarray(b.h, dtype=np.float64))  # barrier.py:247-252
        brow.append(brow[-1] + d)
        bsafe.append(float(b.safe_displacement_gain))
        if b.safe_displacement is not None and b.safe_displacement_gain > 1e-6:
            rho = b.safe_displacement_gain / np.sum(Jh * Jh, axis=(1, 2))  # barrier.py:196-198
            term = -rho[:, None] * np.asarray(b.safe_displacement, dtype=np.float64)
            c_extra = term if c_extra is None else c_extra + term
    Gd = np.concatenate(G_list, axis=1) if G_list else np.zeros((B, 0, nv))
    hd = np.concatenate(h_list, axis=1) if h_list else np.zeros((B, 0))

    return IKBatch(
        nv=nv,
        J=np.ascontiguousarray(J),
        e=np.ascontiguousarray(e),
        cost=np.ascontiguousarray(cost),
        task_rows=np.asarray(rows, dtype=np.int32),
        task_kind=np.asarray(kinds, dtype=np.int32),
        task_col0=np.asarray(col0, dtype=np.int32),
        gain=np.asarray(gains, dtype=np.float64),
        lm_damping=np.asarray(lms, dtype=np.float64),
        lb=np.ascontiguousarray(lb),
        ub=np.ascontiguousarray(ub),
        Gd=np.ascontiguousarray(Gd),
        hd=np.ascontiguousarray(hd),
        barrier_rows=np.asarray(brow, dtype=np.int32),
        barrier_safe_gain=np.asarray(bsafe, dtype=np.float64),
        c_extra=None if c_extra is None else np.ascontiguousarray(c_extra),
        damping=float(damping),
        dt=float(dt),
        n_eq=int(n_eq),
    )


def split_box_rows(G: Optional[np.ndarray], h: Optional[np.ndarray], nv: int):
    """Split one instance's stacked ``G dq <= h`` (``pink/solve_ik.py:107-122``)
    into a per-coordinate box and the rows that are not axis-aligned.

    A row with a single non-zero ``a`` in column ``i`` reads ``a dq_i <= h`` and
    tightens ``ub_i = h/a`` (``a > 0``) or ``lb_i = h/a`` (``a < 0``).  Returns
    ``(lb, ub, G_dense, h_dense)``.
    """
    lb = np.full(nv, -np.inf)
    ub = np.full(nv, np.inf)
    if G is None or len(G) == 0:
        return lb, ub, np.zeros((0, nv)), np.zeros(0)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    nnz = np.count_nonzero(G, axis=1)
    axis = nnz == 1
    cols = np.argmax(G != 0.0, axis=1)
    for r in np.nonzero(axis)[0]:
        i = cols[r]
        a = G[r, i]
        if a > 0:
            ub[i] = min(ub[i], h[r] / a)
        else:
            lb[i] = max(lb[i], h[r] / a)
    keep = ~axis & (nnz > 0)
    # all-zero rows 0 <= h carry no constraint unless h < 0 (infeasible): keep those
    zero_bad = (nnz == 0) & (h < 0)
    keep |= zero_bad
    return lb, ub, G[keep], h[keep]
