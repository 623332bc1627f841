"""Barrier base class (``pink/barriers/barrier.py:20-278``)."""

from __future__ import annotations

import abc
from typing import Callable, Optional, Tuple, Union

import numpy as np

from ..batch import BarrierTerm


class Barrier(abc.ABC):
    """``h(q) >= 0`` kept through ``dh/dq dq/dt + alpha(h) >= 0`` plus an optional
    safe-displacement regulariser (``barrier.py:62-99``)."""

    def __init__(self, dim: int, gain: Union[float, np.ndarray] = 1.0,
                 gain_function: Optional[Callable[[float], float]] = None, safe_displacement_gain: float = 0.0):
        self.dim = dim
        self.gain = gain if isinstance(gain, np.ndarray) else np.ones(dim) * gain
        self.identity_gain_function = gain_function is None  # (the device-resident path forms h on chip: identity only)
        self.gain_function = gain_function if gain_function is not None else (lambda x: x)
        self.safe_displacement = np.zeros(dim)
        self.safe_displacement_gain = safe_displacement_gain

    @abc.abstractmethod
    def compute_barrier(self, configuration) -> np.ndarray:
        """Barrier values ``h(q)``."""

    @abc.abstractmethod
    def compute_jacobian(self, configuration) -> np.ndarray:
        """``dh/dq``, shape ``(dim, nv)``."""

    def compute_safe_displacement(self, configuration) -> np.ndarray:
        return np.zeros(configuration.model.nv)  # barrier.py:134-149

    def as_term(self, configuration) -> BarrierTerm:
        """This barrier at ``configuration`` as a term of a batch of one: the class-K
        function is applied here, the rest (``-J/dt``, ``gain*h``, ``r/|J|^2``) is
        assembled by the packer / the kernel."""
        J = np.asarray(self.compute_jacobian(configuration), dtype=float)
        h = np.asarray(self.compute_barrier(configuration), dtype=float)
        hk = np.array([self.gain_function(h[i]) for i in range(self.dim)])
        dq_safe = None
        if self.safe_displacement_gain > 1e-6:
            self.safe_displacement = self.compute_safe_displacement(configuration)
            if np.any(self.safe_displacement):
                dq_safe = np.asarray(self.safe_displacement, dtype=float)[None]
        return BarrierTerm(J_h=J[None], h=hk[None], gain=self.gain, safe_displacement_gain=self.safe_displacement_gain,
                           safe_displacement=dq_safe)

    def compute_qp_objective(self, configuration) -> Tuple[np.ndarray, np.ndarray]:
        """``(H, c)`` of the regulariser (``barrier.py:151-203``), from the HIP stack kernel."""
        from ..batch import pack_terms
        from ..runtime import default_solver

        nv = configuration.model.nv
        batch = pack_terms(nv, [], dt=1.0, damping=0.0, barriers=[self.as_term(configuration)], batch_size=1)
        H, c = default_solver().stack(batch)
        return H[0], c[0]

    def compute_qp_inequalities(self, configuration, dt: float = 1e-3) -> Tuple[np.ndarray, np.ndarray]:
        """``G = -J_h/dt``, ``h_i = gain_i alpha(h_i(q))`` (``barrier.py:205-254``)."""
        J = np.asarray(self.compute_jacobian(configuration), dtype=float)
        hv = np.asarray(self.compute_barrier(configuration), dtype=float)
        return -J / dt, np.array([self.gain[i] * self.gain_function(hv[i]) for i in range(self.dim)])

    def __repr__(self) -> str:
        return (f"Barrier(gain={self.gain}, safe_displacement_gain={self.safe_displacement_gain}, dim={self.dim})")
