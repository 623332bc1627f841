"""Minimum distance between two frames (``pink/barriers/body_spherical_barrier.py``).

Host-side producer of one dense inequality row per instance; the row goes through the
barrier path of the stack + solve kernel (``-J_h/dt``, ``gain alpha(h)``).
"""

from __future__ import annotations

from typing import Tuple, Union

import numpy as np

from ..exceptions import NegativeMinimumDistance
from .barrier import Barrier


class BodySphericalBarrier(Barrier):
    """``h(q) = |p_1(q) - p_2(q)|^2 - d_min^2 >= 0`` with the saturating class-K function
    ``h / (1 + |h|)`` (``body_spherical_barrier.py:45-72``)."""

    def __init__(self, frames: Tuple[str, str], d_min: float, gain: Union[float, np.ndarray] = 1.0,
                 safe_displacement_gain: float = 3.0):
        if d_min < 0.0:
            raise NegativeMinimumDistance("The minimum distance threshold must be non-negative.")
        super().__init__(1, gain=gain, gain_function=lambda h: h / (1.0 + np.abs(h)),
                         safe_displacement_gain=safe_displacement_gain)
        self.frames, self.d_min = frames, d_min

    def _offset(self, configuration) -> np.ndarray:
        T1 = configuration.get_transform_frame_to_world(self.frames[0])
        T2 = configuration.get_transform_frame_to_world(self.frames[1])
        return T1.translation - T2.translation

    def compute_barrier(self, configuration) -> np.ndarray:
        d = self._offset(configuration)
        return np.array([d @ d - self.d_min ** 2])

    def compute_jacobian(self, configuration) -> np.ndarray:
        """``2 (p_1 - p_2)^T (R_1 J_1,lin - R_2 J_2,lin)`` (``body_spherical_barrier.py:105-140``)."""
        d = self._offset(configuration)
        lin = []
        for f in self.frames:
            R = configuration.get_transform_frame_to_world(f).rotation
            lin.append(R @ configuration.get_frame_jacobian(f)[:3])
        return (2.0 * d @ (lin[0] - lin[1]))[None]
