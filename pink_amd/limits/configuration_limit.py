"""Joint-position limit (``pink/limits/configuration_limit.py``)."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .limit import Limit


class ConfigurationLimit(Limit):
    """Bounded joints of a model; ``config_limit_gain`` steers away from the limits
    (``configuration_limit.py:40-80``)."""

    def __init__(self, model, config_limit_gain: float = 0.5):
        assert 0.0 < config_limit_gain <= 1.0
        lower, upper = np.asarray(model.lowerPositionLimit), np.asarray(model.upperPositionLimit)
        has_limit = np.logical_and(upper < 1e20, upper > lower + 1e-10)  # :50-56
        index_list = []
        for joint in model.joints:  # a joint is bounded when all its coordinates are (:58-71)
            if joint.idx_q >= 0 and has_limit[joint.idx_q:joint.idx_q + joint.nq].all():
                index_list.extend(range(joint.idx_v, joint.idx_v + joint.nv))
        self.indices = np.array(index_list, dtype=int)
        self.indices.setflags(write=False)
        self.projection_matrix = np.eye(model.nv)[self.indices] if len(index_list) else None
        self.config_limit_gain = config_limit_gain
        self.joints = [j for j in model.joints if j.idx_q >= 0 and has_limit[j.idx_q:j.idx_q + j.nq].all()]
        self.model = model

    def compute_box(self, configuration, dt: float):
        if self.projection_matrix is None:
            return None
        dq_max = self.model.difference(configuration.q, self.model.upperPositionLimit)  # :111-116
        dq_min = self.model.difference(configuration.q, self.model.lowerPositionLimit)
        g = self.config_limit_gain
        return self.indices, g * dq_min[self.indices], g * dq_max[self.indices]

    def compute_qp_inequalities(self, configuration, dt: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        box = self.compute_box(configuration, dt)
        if box is None:
            return None
        _, p_min, p_max = box
        P = self.projection_matrix
        return np.vstack([P, -P]), np.hstack([p_max, -p_min])  # :117-121
