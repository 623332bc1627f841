"""Exceptions of the IK front end; same names and payloads as ``pink/exceptions.py``."""

from __future__ import annotations

from typing import Optional

import numpy as np


class PinkError(Exception):
    """Base class (``pink/exceptions.py:11``)."""


class ConfigurationError(PinkError):
    """Invalid configuration vector."""


class FrameNotFound(PinkError):
    """A frame name is not in the model (``pink/exceptions.py:19-33``)."""

    def __init__(self, name: str, frames) -> None:
        self.name = name
        names = [getattr(f, "name", f) for f in frames]
        self.message = f'Name "{name}" is not a robot frame name in {names}'
        super().__init__(self.message)


class NoPositionLimitProvided(PinkError):
    """A position barrier got neither ``p_min`` nor ``p_max``."""


class NoSolutionFound(PinkError):
    """The QP solver did not find a solution (``pink/exceptions.py:49-67``).

    ``problem`` is the QP that failed and ``results`` what the backend reported,
    as in the reference; a batched solve additionally lists the failing
    instances in ``indices`` with their ``status`` codes (``include/pinkhip.h``).
    """

    def __init__(self, problem, results, indices: Optional[np.ndarray] = None, status: Optional[np.ndarray] = None) -> None:
        super().__init__("QP solver did not find a solution to the differential IK problem")
        self.problem = problem
        self.results = results
        self.indices = indices
        self.status = status


class NotWithinConfigurationLimits(PinkError):
    """A configuration violates its limits (``pink/exceptions.py:70-108``)."""

    def __init__(self, joint: int, value: float, lower: float, upper: float, instance: Optional[int] = None) -> None:
        self.joint, self.value, self.lower, self.upper = joint, value, lower, upper
        self.instance = instance
        self.message = f"Joint {joint} violates configuration limits {lower} <= {value} <= {upper}"
        super().__init__(self.message)


class TargetNotSet(PinkError):
    """A task target is read before being set."""


class TaskDefinitionError(PinkError):
    """Ill-formed task definition."""


class TaskJacobianNotSet(PinkError):
    """A task Jacobian is read before being set."""


class InvalidCollisionPairs(PinkError):
    """Ill-formed set of collision pairs (``pink/exceptions.py``)."""


class NegativeMinimumDistance(PinkError):
    """A barrier is given a negative minimum distance (``pink/exceptions.py``)."""
