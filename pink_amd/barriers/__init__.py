"""Control barrier functions (``pink/barriers``)."""
from .barrier import Barrier
from .body_spherical_barrier import BodySphericalBarrier
from .position_barrier import PositionBarrier

__all__ = ["Barrier", "PositionBarrier", "BodySphericalBarrier"]
