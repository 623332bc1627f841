"""Joint-velocity limit (``pink/limits/velocity_limit.py``)."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ..exceptions import PinkError
from .limit import Limit


class VelocityLimit(Limit):
    """Velocity-limited joints; an explicit ``velocity_limit`` vector overrides the
    model's (``velocity_limit.py:38-86``)."""

    def __init__(self, model, velocity_limit: Optional[np.ndarray] = None):
        if velocity_limit is None:
            velocity_limit = np.asarray(model.velocityLimit, dtype=float)
        else:
            velocity_limit = np.asarray(velocity_limit, dtype=float).flatten()
            if model.nv > 0 and velocity_limit.shape[0] != model.nv:
                raise PinkError(f"{velocity_limit.shape=} but {model.nv=}")
        has_limit = np.logical_and(velocity_limit < 1e20, velocity_limit > 1e-10)  # :61-64
        index_list = []
        for joint in model.joints:
            if joint.idx_v >= 0 and has_limit[joint.idx_v:joint.idx_v + joint.nv].all():
                index_list.extend(range(joint.idx_v, joint.idx_v + joint.nv))
        self.indices = np.array(index_list, dtype=int)
        self.indices.setflags(write=False)
        self.projection_matrix = np.eye(model.nv)[self.indices] if len(index_list) else None
        self.joints = [j for j in model.joints if j.idx_v >= 0 and has_limit[j.idx_v:j.idx_v + j.nv].all()]
        self.model = model
        self.velocity_limit = velocity_limit

    def compute_box(self, configuration, dt: float):
        if self.projection_matrix is None:
            return None
        v = dt * self.velocity_limit[self.indices]
        return self.indices, -v, v

    def compute_qp_inequalities(self, configuration, dt: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        if self.projection_matrix is None:
            return None
        v_max = self.velocity_limit[self.indices]
        P = self.projection_matrix
        return np.vstack([P, -P]), np.hstack([dt * v_max, dt * v_max])  # :118-121
