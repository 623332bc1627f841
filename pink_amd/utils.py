"""Small helpers mirroring ``pink/utils.py``."""

from __future__ import annotations

from typing import Tuple

import numpy as np


class VectorSpace:
    """Read-only ``eye`` / ``ones`` / ``zeros`` of a tangent space (``pink/utils.py:77-113``)."""

    def __init__(self, dim: int):
        self.dim = dim
        self.__eye = np.eye(dim)
        self.__ones = np.ones(dim)
        self.__zeros = np.zeros(dim)
        for a in (self.__eye, self.__ones, self.__zeros):
            a.setflags(write=False)

    @property
    def eye(self) -> np.ndarray:
        return self.__eye

    @property
    def ones(self) -> np.ndarray:
        return self.__ones

    @property
    def zeros(self) -> np.ndarray:
        return self.__zeros


def get_root_joint_dim(model) -> Tuple[int, int]:
    """``(nq, nv)`` of the joint named ``"root_joint"`` or ``(0, 0)`` (``pink/utils.py:40-54``)."""
    root = getattr(model, "root_joint", None)
    if root is not None:
        return root.nq, root.nv
    return 0, 0
