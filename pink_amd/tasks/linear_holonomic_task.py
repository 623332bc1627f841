"""Linear holonomic tasks ``A (q (-) q_0) = b`` and the tasks derived from them
(``pink/tasks/linear_holonomic_task.py``, ``joint_coupling_task.py``,
``joint_velocity_task.py``).  Host-side row producers for the stack + solve kernel.
"""

from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np

from ..exceptions import TaskDefinitionError, TaskJacobianNotSet
from .task import Task


class LinearHolonomicTask(Task):
    """``e(q) = A (q (-) q_0) - b`` with Jacobian ``A d(q (-) q_0)/dq``."""

    def __init__(self, A: np.ndarray, b: np.ndarray, q_0: Optional[np.ndarray],
                 cost: Optional[Union[float, Sequence[float], np.ndarray]] = None, lm_damping: float = 0.0,
                 gain: float = 1.0) -> None:
        super().__init__(cost=cost, gain=gain, lm_damping=lm_damping)
        A, b = np.asarray(A, dtype=float), np.asarray(b, dtype=float)
        if b.shape[0] != A.shape[0]:
            raise TaskDefinitionError(f"Shape mismatch between A.shape={A.shape} and b.shape={b.shape}")
        self.A, self.b, self.q_0 = A, b, q_0

    def _reference(self, configuration) -> np.ndarray:
        if self.A.shape[1] != configuration.model.nv:
            raise TaskJacobianNotSet
        return configuration.model.neutral() if self.q_0 is None else self.q_0

    def compute_error(self, configuration) -> np.ndarray:
        """``linear_holonomic_task.py:103-127``."""
        q_ref = self._reference(configuration)
        return self.A @ configuration.model.difference(q_ref, configuration.q) - self.b

    def compute_jacobian(self, configuration) -> np.ndarray:
        """``A dDifference(q_0, q, ARG1)`` (``linear_holonomic_task.py:129-148``)."""
        q_ref = self._reference(configuration)
        return self.A @ configuration.model.d_difference(q_ref, configuration.q)

    def __repr__(self):
        return (f"LinearHolonomicTask(A={self.A}, b={self.b}, q_0={self.q_0}, cost={self.cost}, "
                f"gain={self.gain}, lm_damping={self.lm_damping})")


class JointCouplingTask(LinearHolonomicTask):
    """``sum_i ratio_i q_i = 0`` over the named joints (``joint_coupling_task.py:20-90``)."""

    def __init__(self, joint_names: Sequence[str], ratios: Sequence[float], cost: float, configuration,
                 lm_damping: float = 0.0, gain: float = 1.0) -> None:
        if len(joint_names) != len(ratios):
            raise TaskDefinitionError("one ratio per joint name")
        model = configuration.model
        A = np.zeros((1, model.nv))
        for name, ratio in zip(joint_names, ratios):
            j = model.joints[model.getJointId(name)]
            A[:, j.idx_v:j.idx_v + j.nv] = ratio
        super().__init__(A, np.zeros(1), model.neutral(), cost=cost, gain=gain, lm_damping=lm_damping)
        self.joint_names, self.ratios = joint_names, ratios

    def __repr__(self):
        return (f"JointCouplingTask(joint_names={self.joint_names}, ratios={self.ratios}, cost={self.cost}, "
                f"gain={self.gain}, lm_damping={self.lm_damping})")


class JointVelocityTask(Task):
    """Track a target tangent velocity of the actuated joints (``joint_velocity_task.py``):
    ``e = dt v_target`` -- the reference's ``compute_error`` returns the target displacement itself
    (``joint_velocity_task.py:59-80``; pinned by ``tests/golden/pink_round4.npz``, produced by the reference's class) --
    ``J = I`` on the columns after the root joint."""

    def __init__(self, cost: float) -> None:
        super().__init__(cost=cost, gain=1.0, lm_damping=0.0)
        self.target_v: Optional[np.ndarray] = None
        self.target_dt: Optional[float] = None

    def set_target(self, target_v: np.ndarray, dt: float) -> None:
        self.target_v = np.asarray(target_v, dtype=float).copy()
        self.target_dt = float(dt)

    def _root_nv(self, configuration) -> int:
        root = configuration.model.root_joint
        return 0 if root is None else root.nv

    def diagonal_col0(self, configuration) -> Optional[int]:
        return self._root_nv(configuration)

    def compute_error(self, configuration) -> np.ndarray:
        from ..exceptions import TargetNotSet

        if self.target_v is None or self.target_dt is None:
            raise TargetNotSet("no target set for joint velocity task")
        k = configuration.model.nv - self._root_nv(configuration)
        if self.target_v.shape[0] != k:
            raise TaskDefinitionError(f"target velocity has dimension {self.target_v.shape[0]}, expected {k}")
        return self.target_dt * self.target_v

    def compute_jacobian(self, configuration) -> np.ndarray:
        nv, r = configuration.model.nv, self._root_nv(configuration)
        return np.eye(nv)[r:]

    def __repr__(self):
        return f"JointVelocityTask(cost={self.cost})"
