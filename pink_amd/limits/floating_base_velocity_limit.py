"""Velocity limit of the floating base (``pink/limits/floating_base_velocity_limit.py``)."""

from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np

from .limit import Limit


def _as_velocity_vector(values, label: str) -> np.ndarray:
    array = np.asarray(values, dtype=float)
    if array.ndim == 0:
        array = np.repeat(array, 3)
    array = array.flatten()
    if array.shape != (3,):
        raise ValueError(f"{label} must be a scalar or an iterable of length 3, got shape {array.shape}")
    return array


class FloatingBaseVelocityLimit(Limit):
    """Bounds on the twist of a frame attached to ``root_joint``, expressed in that frame
    (``floating_base_velocity_limit.py:57-148``).  Its rows are dense on the root columns, so
    they travel as dense inequality rows of the packed batch."""

    def __init__(self, model, base_frame: Optional[str], max_linear_velocity: Union[Sequence[float], float],
                 max_angular_velocity: Union[Sequence[float], float]):
        self.model = model
        self.linear_max = _as_velocity_vector(max_linear_velocity, "max_linear_velocity")
        self.angular_max = _as_velocity_vector(max_angular_velocity, "max_angular_velocity")
        self.twist_max = np.hstack([self.linear_max, self.angular_max])
        root = model.root_joint
        if root is None:
            raise ValueError("FloatingBaseVelocityLimit requires a floating-base root joint.")
        self.root_idx_v, self.root_nv = root.idx_v, root.nv
        root_id = model.joints.index(root)
        if base_frame is None:
            candidates = [f.name for f in model.frames if f.joint == root_id]
            if not candidates:
                raise ValueError("Model does not expose a frame attached to 'root_joint'.")
            base_frame = candidates[0]
        frame = model.frames[model.getFrameId(base_frame)]
        if frame.joint != root_id:
            raise ValueError(f"Frame '{base_frame}' is not attached to the root joint.")
        self.base_frame = base_frame

    def compute_qp_inequalities(self, configuration, dt: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        finite = np.isfinite(self.twist_max)
        if not finite.any():
            return None
        J = configuration.get_frame_jacobian(self.base_frame).copy()
        J[:, :self.root_idx_v] = 0.0  # only the root twist columns (:128-141)
        J[:, self.root_idx_v + self.root_nv:] = 0.0
        rows = J[finite, :]
        bounds = dt * self.twist_max[finite]
        return np.vstack([rows, -rows]), np.hstack([bounds, bounds])
