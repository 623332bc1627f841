"""Frame task: regulate the pose of a robot frame (``pink/tasks/frame_task.py``)."""

from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np

from ..exceptions import TargetNotSet, TaskDefinitionError
from ..lie import SE3, Jlog6, log6
from .task import Task


import itertools

_FREEZE_TOKENS = itertools.count(1)


class FrozenTargets(np.ndarray):
    """A ``[B, 12]`` target array its owner promised not to modify (:meth:`FrameTask.freeze_targets`): read-only, and
    carrying a token by which a device state recognises what it already holds."""

    frozen_token: Optional[int] = None


class FrameTask(Task):
    """6-D pose task; cost is ``[position x3, orientation x3]`` (``frame_task.py:44-127``)."""

    def __init__(self, frame: str, position_cost, orientation_cost, lm_damping: float = 0.0, gain: float = 1.0):
        super().__init__(cost=np.ones(6), gain=gain, lm_damping=lm_damping)
        self.frame = frame
        self.transform_target_to_world: Optional[SE3] = None
        self.target_poses: Optional[np.ndarray] = None  # [B, 12] per-instance targets (set_target_poses)
        self.set_position_cost(position_cost)
        self.set_orientation_cost(orientation_cost)

    def _set_cost(self, values, sl: slice, what: str) -> None:
        v = np.asarray(values, dtype=float)
        if v.ndim > 0 and v.shape != (3,):
            raise TaskDefinitionError(f"{what} cost should be a float or a vector of 3, got shape {v.shape}")
        if (v < 0.0).any():
            raise TaskDefinitionError(f"{what} cost should be >= 0")
        self.cost[sl] = v

    def set_position_cost(self, position_cost: Union[float, Sequence[float], np.ndarray]) -> None:
        self._set_cost(position_cost, slice(0, 3), "position")

    def set_orientation_cost(self, orientation_cost: Union[float, Sequence[float], np.ndarray]) -> None:
        self._set_cost(orientation_cost, slice(3, 6), "orientation")

    @property
    def position_cost(self):
        return self.cost[0:3]

    @property
    def orientation_cost(self):
        return self.cost[3:6]

    def set_target(self, transform_target_to_world: SE3) -> None:
        """One target for every configuration the task is evaluated at (``frame_task.py:129-137``); replaces per-instance
        targets set earlier (one target source is live at a time)."""
        self.transform_target_to_world = transform_target_to_world.copy()
        self.target_poses = None

    def set_target_from_configuration(self, configuration) -> None:
        self.set_target(configuration.get_transform_frame_to_world(self.frame))

    def set_target_poses(self, rotations: np.ndarray, translations: np.ndarray, out: Optional[np.ndarray] = None) -> None:
        """One target per instance of a batch, as arrays: ``rotations [B, 3, 3]``, ``translations [B, 3]``
        (target frame to world).  Used by :func:`pink_amd.solve_ik_batch` with a
        :class:`pink_amd.ConfigurationBatch`; instance ``b`` then plays ``set_target(SE3(R[b], p[b]))``.
        ``out [B, 12]`` (e.g. from :func:`pink_amd.pinned_empty`) receives the packed poses and is kept as the task's
        target array: a control loop refills it in place."""
        R = np.asarray(rotations, dtype=np.float64)
        t = np.asarray(translations, dtype=np.float64)
        if R.ndim != 3 or R.shape[1:] != (3, 3) or t.shape != (R.shape[0], 3):
            raise TaskDefinitionError(f"rotations [B, 3, 3] and translations [B, 3] expected, got {R.shape} and {t.shape}")
        if out is None:
            out = np.empty((R.shape[0], 12))
        elif out.shape != (R.shape[0], 12) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise TaskDefinitionError(f"out must be a C-contiguous float64 array of shape {(R.shape[0], 12)}")
        out[:, :9] = R.reshape(-1, 9)
        out[:, 9:] = t
        self.target_poses = out
        self.transform_target_to_world = None  # (replaces a single target set earlier)

    def freeze_targets(self) -> None:
        """Declare the per-instance targets set by :meth:`set_target_poses` unchanged until the next ``set_target*`` call:
        the array becomes read-only and :func:`pink_amd.solve_ik_batch` uploads it to a device ONCE per cached device state
        instead of with every call (6.3 MB per frame task at B = 65 536: three fifths of what a call at the headline
        shape sends across PCIe are targets).  Without it every call uploads the array as it is then -- a control loop
        may refill it in place between two calls.

        The task keeps a READ-ONLY VIEW of the array; the caller's own array object is left as it is (an ``out=`` buffer
        can be handed to the next :meth:`set_target_poses`, which ends the freeze).  The promise is the caller's: a
        write into the buffer while the freeze lasts is NOT seen by the device."""
        if self.target_poses is None:
            raise TargetNotSet(f"no per-instance targets set for frame '{self.frame}'")
        arr = self.target_poses
        if getattr(arr, "frozen_token", None) is None:
            view = arr.view(FrozenTargets)
            view.flags.writeable = False  # (the view's flag: the base array stays writeable)
            view.frozen_token = next(_FREEZE_TOKENS)
            self.target_poses = view

    def compute_error(self, configuration) -> np.ndarray:
        """Body twist from the frame to its target, ``log6(T_frame^-1 T_target)``
        (``frame_task.py:176-193``)."""
        if self.transform_target_to_world is None:
            raise TargetNotSet(f"no target set for frame '{self.frame}'")
        T_fw = configuration.get_transform_frame_to_world(self.frame)
        return log6(T_fw.actInv(self.transform_target_to_world))

    def compute_jacobian(self, configuration) -> np.ndarray:
        """``-Jlog6(T_target^-1 T_frame) @ J_frame`` with the body Jacobian of the frame
        (``frame_task.py:217-227``)."""
        if self.transform_target_to_world is None:
            raise TargetNotSet(f"no target set for frame '{self.frame}'")
        T_fw = configuration.get_transform_frame_to_world(self.frame)
        T_ft = self.transform_target_to_world.actInv(T_fw)
        return -Jlog6(T_ft) @ configuration.get_frame_jacobian(self.frame)

    def __repr__(self):
        return (f"FrameTask(frame={self.frame!r}, gain={self.gain}, orientation_cost={self.orientation_cost}, "
                f"position_cost={self.position_cost}, lm_damping={self.lm_damping})")
