"""Limit base class (``pink/limits/limit.py:17-45``)."""

from __future__ import annotations

import abc
from typing import Optional, Tuple

import numpy as np


class Limit(abc.ABC):
    """A limit contributes rows ``G(q) dq <= h(q)`` or ``None`` when it has none."""

    @abc.abstractmethod
    def compute_qp_inequalities(self, configuration, dt: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """Pair ``(G, h)`` or ``None``."""

    def compute_box(self, configuration, dt: float):
        """Optional fast path: ``(indices, lower, upper)`` when every row is ``+-e_i``
        (then the rows are merged into the per-coordinate box without materialising G)."""
        return None
