"""Joint-acceleration limit (``pink/limits/acceleration_limit.py``)."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .limit import Limit


class AccelerationLimit(Limit):
    """Finite-difference acceleration bound plus the braking-distance bound toward the
    configuration limits (``acceleration_limit.py:44-117``).  Every row is ``+-e_i``: the limit
    feeds the merged box of the packed batch."""

    def __init__(self, model, acceleration_limit: np.ndarray):
        a = np.asarray(acceleration_limit, dtype=float)
        has_acc = np.logical_and(a < 1e20, a > 1e-10)  # :60-63
        lower, upper = np.asarray(model.lowerPositionLimit), np.asarray(model.upperPositionLimit)
        has_cfg = np.logical_and(upper < 1e20, upper > lower + 1e-10)  # :72-78
        index_list, cfg_list = [], []
        for joint in model.joints:
            if joint.idx_v >= 0 and has_acc[joint.idx_v:joint.idx_v + joint.nv].all():
                index_list.extend(range(joint.idx_v, joint.idx_v + joint.nv))
                cfg_list.extend([bool(has_cfg[joint.idx_q:joint.idx_q + joint.nq].all())] * joint.nv)
        self.indices = np.array(index_list, dtype=int)
        self.indices.setflags(write=False)
        dim = len(index_list)
        self.projection_matrix = np.eye(model.nv)[self.indices] if dim else None
        self.a_max = a[self.indices] if dim else np.empty(0)
        self.has_configuration_limit = np.array(cfg_list, dtype=bool)
        self.Delta_q_prev = np.zeros(dim)
        self.model = model

    def set_last_integration(self, v_prev: np.ndarray, dt: float) -> None:
        """Latest integrated velocity and its duration (``acceleration_limit.py:106-117``)."""
        self.Delta_q_prev = (np.asarray(v_prev, dtype=float) * dt)[self.indices]

    def compute_box(self, configuration, dt: float):
        if self.projection_matrix is None:
            return None
        m = self.model
        dq_max = m.difference(configuration.q, m.upperPositionLimit)[self.indices]  # :158-163
        dq_max = np.where(self.has_configuration_limit, dq_max, np.inf)
        dq_min = m.difference(m.lowerPositionLimit, configuration.q)[self.indices]  # :167-174
        dq_min = np.where(self.has_configuration_limit, dq_min, np.inf)
        dt_sq = dt * dt
        with np.errstate(invalid="ignore"):
            upper = np.minimum(self.a_max * dt_sq + self.Delta_q_prev, dt * np.sqrt(2 * self.a_max * dq_max))  # :186-199
            lower = np.minimum(self.a_max * dt_sq - self.Delta_q_prev, dt * np.sqrt(2 * self.a_max * dq_min))
        return self.indices, -lower, upper

    def compute_qp_inequalities(self, configuration, dt: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        box = self.compute_box(configuration, dt)
        if box is None:
            return None
        _, neg_lower, upper = box
        P = self.projection_matrix
        return np.vstack([P, -P]), np.hstack([upper, -neg_lower])
