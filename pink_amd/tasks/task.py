"""Task base class: same surface as ``pink/tasks/task.py:23-171``.

``compute_error`` / ``compute_jacobian`` stay abstract host-side methods; the
arithmetic of ``compute_qp_objective`` (``task.py:145-167``) runs on the GPU
(``ik_stack_kernel`` through the C ABI), also when called for a single task.
"""

from __future__ import annotations

import abc
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from ..batch import DenseTaskTerm, DiagonalTaskTerm, pack_terms


class Task(abc.ABC):
    """Kinematic task with ``cost``, ``gain`` and ``lm_damping`` (``task.py:38-64``)."""

    cost: Optional[Union[float, Sequence[float], np.ndarray]]
    gain: float
    lm_damping: float

    def __init__(self, cost=None, gain: float = 1.0, lm_damping: float = 0.0):
        self.cost = cost
        self.gain = gain
        self.lm_damping = lm_damping

    @abc.abstractmethod
    def compute_error(self, configuration) -> np.ndarray:
        """Task error ``e(q)`` of ``J(q) dq = -gain e(q)`` (``task.py:66-98``)."""

    @abc.abstractmethod
    def compute_jacobian(self, configuration) -> np.ndarray:
        """Task Jacobian ``J(q)``, shape ``(k, nv)`` (``task.py:100-113``)."""

    def diagonal_col0(self, configuration) -> Optional[int]:
        """First tangent column when the Jacobian is ``eye(nv)[col0:col0+k]`` (then the
        Jacobian is never materialised for the GPU), else ``None``."""
        return None

    def as_term(self, configuration):
        """This task evaluated at ``configuration`` as a term of a batch of one."""
        e = np.asarray(self.compute_error(configuration), dtype=np.float64)
        col0 = self.diagonal_col0(configuration)
        if col0 is not None:
            return DiagonalTaskTerm(col0=col0, e=e[None], cost=self.cost, gain=self.gain, lm_damping=self.lm_damping)
        J = np.asarray(self.compute_jacobian(configuration), dtype=np.float64)
        return DenseTaskTerm(J=J[None], e=e[None], cost=self.cost, gain=self.gain, lm_damping=self.lm_damping)

    def compute_qp_objective(self, configuration) -> Tuple[np.ndarray, np.ndarray]:
        """``(H, c)`` with ``H = (W J)^T (W J) + mu I``, ``c = -(W(-gain e))^T W J``
        (``task.py:115-167``), evaluated by the HIP stack kernel."""
        from ..runtime import default_solver

        nv = configuration.model.nv
        batch = pack_terms(nv, [self.as_term(configuration)], dt=1.0, damping=0.0, batch_size=1)
        H, c = default_solver().stack(batch)
        return H[0], c[0]

    @abc.abstractmethod
    def __repr__(self):
        """Human-readable representation of the task."""
