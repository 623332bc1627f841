"""Pose of one frame relative to another (``pink/tasks/relative_frame_task.py``).

Host-side row producer: six rows ``(J, e)`` per instance that go through the same
stack + solve kernel as every other task.
"""

from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np

from ..exceptions import TargetNotSet, TaskDefinitionError
from ..lie import SE3, Jlog6, log6
from .task import Task


class RelativeFrameTask(Task):
    """Regulate the pose of ``frame`` in ``root`` to ``transform_target_to_root``."""

    def __init__(self, frame: str, root: str, position_cost: Union[float, Sequence[float], np.ndarray],
                 orientation_cost: Union[float, Sequence[float], np.ndarray], lm_damping: float = 0.0,
                 gain: float = 1.0) -> None:
        super().__init__(cost=np.ones(6), gain=gain, lm_damping=lm_damping)
        self.frame = frame
        self.root = root
        self.transform_target_to_root: Optional[SE3] = None
        self.set_position_cost(position_cost)
        self.set_orientation_cost(orientation_cost)

    def _set_cost(self, sl: slice, value, what: str) -> None:
        if isinstance(value, (int, float)):
            if value < 0.0:
                raise TaskDefinitionError(f"{what} cost should be a non-negative float or vector")
        else:
            value = np.asarray(value, dtype=float)
            if (value < 0.0).any():
                raise TaskDefinitionError(f"{what} cost should be a non-negative float or vector")
        self.cost[sl] = value

    def set_position_cost(self, position_cost) -> None:
        """Weights of the three linear rows (``relative_frame_task.py:84-101``)."""
        self._set_cost(slice(0, 3), position_cost, "position")

    def set_orientation_cost(self, orientation_cost) -> None:
        """Weights of the three angular rows (``relative_frame_task.py:103-121``)."""
        self._set_cost(slice(3, 6), orientation_cost, "orientation")

    def set_target(self, transform_target_to_root: SE3) -> None:
        self.transform_target_to_root = transform_target_to_root.copy()

    def set_target_from_configuration(self, configuration) -> None:
        self.set_target(configuration.get_transform(self.frame, self.root))

    def _frame_to_target(self, configuration):
        if self.transform_target_to_root is None:
            raise TargetNotSet(f"target pose of frame '{self.frame}' in frame '{self.root}' is undefined")
        T_rf = configuration.get_transform(self.frame, self.root)
        return T_rf, self.transform_target_to_root.actInv(T_rf)

    def compute_error(self, configuration) -> np.ndarray:
        """``log6(T_rt^-1 T_rf)``, a body twist of the frame (``relative_frame_task.py:142-176``)."""
        return log6(self._frame_to_target(configuration)[1])

    def compute_jacobian(self, configuration) -> np.ndarray:
        """``Jlog6(T_tf) (fJ_0f - Ad(T_fr) rJ_0r)`` (``relative_frame_task.py:178-231``)."""
        T_rf, T_tf = self._frame_to_target(configuration)
        J_f = configuration.get_frame_jacobian(self.frame)
        J_r = configuration.get_frame_jacobian(self.root)
        return Jlog6(T_tf) @ (J_f - T_rf.actionInverse @ J_r)

    def __repr__(self):
        return (f"RelativeFrameTask(frame={self.frame}, root={self.root}, gain={self.gain}, "
                f"orientation_cost={self.cost[3:6]}, position_cost={self.cost[0:3]}, "
                f"transform_target_to_root={self.transform_target_to_root})")
