"""Self-collision avoidance barrier (``pink/barriers/self_collision_barrier.py``).

The reference reads nearest points from hpp-fcl / coal through Pinocchio's collision data
(``configuration.collision_data.distanceResults``), which is not available offline.  What the barrier
itself computes from them is restated here unchanged: barrier values ``d_k - d_min`` of the ``dim``
closest pairs (``self_collision_barrier.py:114-128``) and one dense Jacobian row per pair,
``n^T J_p^1 + (r_1 x n)^T J_w^1 - n^T J_p^2 - (r_2 x n)^T J_w^2`` (``:165-224``).  The rows go through
the dense-row path of the stack + solve kernel like every other barrier.

The distance query is a plug: any callable ``configuration -> sequence of PairDistance`` (what a collision
library reports per pair: parent joints, nearest points in the world frame, distance).  ``SpherePairs`` is
a built-in one for spheres rigidly attached to joints -- smooth convex geometry, the case the reference's
docstring declares well defined -- so that the barrier can be used and tested without a collision library.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np

from ..exceptions import InvalidCollisionPairs, NegativeMinimumDistance
from .barrier import Barrier


@dataclass
class PairDistance:
    """One collision pair as a distance query reports it (``hppfcl::DistanceResult`` + the parent joints)."""

    joint1: int  # parent joint of the first body   (geometryObjects[cp.first].parentJoint)
    joint2: int
    point1: np.ndarray  # nearest point on the first body, world frame  (getNearestPoint1)
    point2: np.ndarray
    min_distance: float


class SpherePairs:
    """Distance query for pairs of spheres attached to joints: ``(joint1, centre1, radius1, joint2, centre2,
    radius2)`` with the centres given in the joint frames."""

    def __init__(self, pairs: Sequence[Tuple[int, Sequence[float], float, int, Sequence[float], float]]):
        self.pairs = [(int(j1), np.asarray(c1, float), float(r1), int(j2), np.asarray(c2, float), float(r2))
                      for j1, c1, r1, j2, c2, r2 in pairs]

    def __call__(self, configuration):
        out = []
        for j1, c1, r1, j2, c2, r2 in self.pairs:
            p1 = configuration.oMi[j1].act(c1)
            p2 = configuration.oMi[j2].act(c2)
            d = p2 - p1
            dist = float(np.linalg.norm(d))
            u = d / dist if dist > 0 else np.zeros(3)
            out.append(PairDistance(j1, j2, p1 + r1 * u, p2 - r2 * u, dist - r1 - r2))
        return out


class SelfCollisionBarrier(Barrier):
    """``h_k(q) = d(p_k^1, p_k^2) - d_min`` over the ``n_collision_pairs`` closest pairs."""

    def __init__(self, n_collision_pairs: int, gain: Union[float, np.ndarray] = 1.0, safe_displacement_gain: float = 1.0,
                 d_min: float = 0.02, distance_query: Optional[Callable] = None):
        if d_min < 0.0:
            raise NegativeMinimumDistance("The minimum distance threshold must be non-negative.")
        if n_collision_pairs < 0:
            raise InvalidCollisionPairs("The number of collision pairs must be non-negative.")
        super().__init__(dim=n_collision_pairs, gain=gain, safe_displacement_gain=safe_displacement_gain)
        self.d_min = d_min
        self.distance_query = distance_query

    def _pairs(self, configuration):
        query = self.distance_query or getattr(configuration, "collision_pairs", None)
        if query is None:
            raise InvalidCollisionPairs("no distance query: pass distance_query= (e.g. SpherePairs) or give the "
                                        "configuration a collision_pairs(configuration) callable")
        pairs = list(query(configuration))
        if len(pairs) < self.dim:  # self_collision_barrier.py:107-113
            raise InvalidCollisionPairs(f"The number of collision pairs ({len(pairs)}) is less than the barrier dimension ({self.dim}).")
        return pairs

    @staticmethod
    def _closest(distances: np.ndarray, dim: int) -> np.ndarray:
        return np.argpartition(-distances, -dim)[-dim:] if dim else np.zeros(0, dtype=int)  # :125-127, :181-183

    def compute_barrier(self, configuration) -> np.ndarray:
        pairs = self._pairs(configuration)
        distances = np.array([p.min_distance - self.d_min for p in pairs])
        return distances[self._closest(distances, self.dim)]

    def compute_jacobian(self, configuration) -> np.ndarray:
        pairs = self._pairs(configuration)
        nv = configuration.model.nv
        J = np.zeros((self.dim, nv))
        distances = np.array([p.min_distance for p in pairs])
        for i, k in enumerate(self._closest(distances, self.dim)):
            p = pairs[int(k)]
            w1, w2 = np.asarray(p.point1, float), np.asarray(p.point2, float)
            if np.allclose(w1, w2):  # touching: the normal is undefined, the row stays zero (:197-200)
                continue
            n = (w1 - w2) / np.linalg.norm(w1 - w2)
            r1 = w1 - configuration.oMi[p.joint1].translation
            r2 = w2 - configuration.oMi[p.joint2].translation
            J1 = configuration.get_joint_jacobian_world_aligned(p.joint1)
            J2 = configuration.get_joint_jacobian_world_aligned(p.joint2)
            # n^T J_p + (r x n)^T J_w for the first body, minus the same for the second (n_2 = -n_1)
            J[i] = n @ J1[:3] + np.cross(r1, n) @ J1[3:] - (n @ J2[:3] + np.cross(r2, n) @ J2[3:])
        return np.nan_to_num(J)
