"""Control barrier functions (``pink/barriers``)."""
from .barrier import Barrier
from .body_spherical_barrier import BodySphericalBarrier
from .position_barrier import PositionBarrier
from .self_collision_barrier import PairDistance, SelfCollisionBarrier, SpherePairs

__all__ = ["Barrier", "PositionBarrier", "BodySphericalBarrier", "SelfCollisionBarrier", "SpherePairs", "PairDistance"]
