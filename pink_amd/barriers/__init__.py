"""Control barrier functions (``pink/barriers``)."""
from .barrier import Barrier
from .position_barrier import PositionBarrier

__all__ = ["Barrier", "PositionBarrier"]
