"""Process-wide default :class:`BatchSolver` (one handle on ``PINKHIP_DEVICE``, default 0)."""

from __future__ import annotations

import os
from typing import Optional

from .batch_solver import BatchSolver

_default: Optional[BatchSolver] = None


def default_solver() -> BatchSolver:
    """The lazily created solver the Pink-style API uses.  Raises ``PinkHipError``
    when the HIP library or an MI355X is missing: there is no CPU path."""
    global _default
    if _default is None:
        _default = BatchSolver(device_id=int(os.environ.get("PINKHIP_DEVICE", "0")))
    return _default


def set_default_solver(solver) -> None:
    """Install a solver object (anything with ``solve(batch)`` / ``stack(batch)``)."""
    global _default
    _default = solver
