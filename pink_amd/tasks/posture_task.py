"""Posture task and the other identity-Jacobian tasks (``pink/tasks/posture_task.py``,
``damping_task.py``, ``low_acceleration_task.py``)."""

from __future__ import annotations

from typing import Optional

import numpy as np

from ..exceptions import TargetNotSet
from ..utils import get_root_joint_dim
from .task import Task


class _ActuatedIdentityTask(Task):
    """Jacobian ``eye(nv)[root_nv:]``: passed to the GPU as a diagonal task."""

    def diagonal_col0(self, configuration) -> Optional[int]:
        return get_root_joint_dim(configuration.model)[1]

    def compute_jacobian(self, configuration) -> np.ndarray:
        _, root_nv = get_root_joint_dim(configuration.model)
        return configuration.tangent.eye[root_nv:, :]  # posture_task.py:128-129


class PostureTask(_ActuatedIdentityTask):
    """Regulate joint angles to a reference posture (``posture_task.py:38-107``)."""

    def __init__(self, cost: float, lm_damping: float = 0.0, gain: float = 1.0):
        super().__init__(cost=cost, gain=gain, lm_damping=lm_damping)
        self.target_q: Optional[np.ndarray] = None
        self.target_q_batch: Optional[np.ndarray] = None  # [B, nq] (set_target_batch)

    def set_target(self, target_q: np.ndarray) -> None:
        self.target_q = np.array(target_q, dtype=float)
        self.target_q_batch = None  # (one target source is live at a time)

    def set_target_from_configuration(self, configuration) -> None:
        self.set_target(configuration.q)

    def set_target_batch(self, target_q: np.ndarray) -> None:
        """One reference posture per instance of a batch, ``[B, nq]`` (for ``solve_ik_batch`` on a
        :class:`pink_amd.ConfigurationBatch`)."""
        self.target_q_batch = np.ascontiguousarray(target_q, dtype=np.float64)
        self.target_q = None

    def compute_error(self, configuration) -> np.ndarray:
        """``q (-) q*`` on the actuated coordinates (``posture_task.py:100-107``; the code,
        not the docstring, is authoritative there)."""
        if self.target_q is None:
            raise TargetNotSet("no posture target")
        _, root_nv = get_root_joint_dim(configuration.model)
        return configuration.model.difference(self.target_q, configuration.q)[root_nv:]

    def __repr__(self):
        return f"PostureTask(cost={self.cost}, gain={self.gain}, lm_damping={self.lm_damping})"


class DampingTask(_ActuatedIdentityTask):
    """Minimise joint velocities: zero error (``damping_task.py:24-43``)."""

    def __init__(self, cost: float):
        super().__init__(cost=cost, gain=1.0, lm_damping=0.0)

    def compute_error(self, configuration) -> np.ndarray:
        _, root_nv = get_root_joint_dim(configuration.model)
        return np.zeros(configuration.model.nv - root_nv)

    def __repr__(self):
        return f"DampingTask(cost={self.cost})"


class LowAccelerationTask(Task):
    """Minimise the change of velocity: ``e = -dt v_prev``, ``J = I``
    (``low_acceleration_task.py:34-84``)."""

    def __init__(self, cost: float):
        super().__init__(cost=cost, gain=1.0, lm_damping=0.0)
        self.Delta_q_prev: Optional[np.ndarray] = None

    def set_last_integration(self, v_prev: np.ndarray, dt: float) -> None:
        self.Delta_q_prev = np.asarray(v_prev, dtype=float) * dt

    def diagonal_col0(self, configuration) -> Optional[int]:
        return 0

    def compute_error(self, configuration) -> np.ndarray:
        if self.Delta_q_prev is None:
            return np.zeros(configuration.model.nv)
        return -self.Delta_q_prev

    def compute_jacobian(self, configuration) -> np.ndarray:
        return configuration.tangent.eye

    def __repr__(self):
        return f"LowAccelerationTask(cost={self.cost})"
