"""Frame position barrier (``pink/barriers/position_barrier.py``)."""

from __future__ import annotations

from typing import List, Optional, Union

import numpy as np

from ..exceptions import NoPositionLimitProvided
from .barrier import Barrier


class PositionBarrier(Barrier):
    """Keep the world position of a frame inside ``[p_min, p_max]`` along ``indices``."""

    def __init__(self, frame: str, indices: Optional[List[int]] = None, p_min: Optional[np.ndarray] = None,
                 p_max: Optional[np.ndarray] = None, gain: Union[float, np.ndarray] = 1.0,
                 safe_displacement_gain: float = 0.0):
        indices = [0, 1, 2] if indices is None else indices
        if p_min is None and p_max is None:
            raise NoPositionLimitProvided(f"Position barrier for frame {frame} requires either p_min or p_max")
        dim = len(indices) * ((p_min is not None) + (p_max is not None))
        if isinstance(gain, np.ndarray) and len(gain) != dim:
            gain = np.tile(gain, 2)  # position_barrier.py:81-82
        super().__init__(dim, gain=gain, safe_displacement_gain=safe_displacement_gain)
        self.indices, self.frame, self.p_min, self.p_max = indices, frame, p_min, p_max

    def compute_barrier(self, configuration) -> np.ndarray:
        pos = configuration.get_transform_frame_to_world(self.frame).translation
        parts = []
        if self.p_min is not None:
            parts.append(pos[self.indices] - self.p_min)
        if self.p_max is not None:
            parts.append(self.p_max - pos[self.indices])
        return np.concatenate(parts)  # position_barrier.py:109-121

    def compute_jacobian(self, configuration) -> np.ndarray:
        J_lin = configuration.get_frame_jacobian(self.frame)[:3]
        R = configuration.get_transform_frame_to_world(self.frame).rotation
        J_world = (R @ J_lin)[self.indices]  # position_barrier.py:136-145
        parts = []
        if self.p_min is not None:
            parts.append(J_world.copy())
        if self.p_max is not None:
            parts.append(-J_world.copy())
        return np.vstack(parts)
