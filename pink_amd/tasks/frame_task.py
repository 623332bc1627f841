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


def poses_from_pq(pq: np.ndarray) -> np.ndarray:
    """``[B, 12]`` poses (rotation row-major, translation) of ``[B, 7]`` = translation + quaternion ``(x, y, z, w)``: the
    host-side statement of ``ik_pose_targets_thread`` (``repo:pink_amd/csrc/ik_kinematics.h``), same operations."""
    pq = np.asarray(pq, dtype=np.float64)
    n = 1.0 / np.sqrt(np.einsum("bi,bi->b", pq[:, 3:], pq[:, 3:]))
    x, y, z, w = (pq[:, 3 + i] * n for i in range(4))
    out = np.empty((pq.shape[0], 12))
    out[:, 0] = 1.0 - 2.0 * (y * y + z * z)
    out[:, 1] = 2.0 * (x * y - z * w)
    out[:, 2] = 2.0 * (x * z + y * w)
    out[:, 3] = 2.0 * (x * y + z * w)
    out[:, 4] = 1.0 - 2.0 * (x * x + z * z)
    out[:, 5] = 2.0 * (y * z - x * w)
    out[:, 6] = 2.0 * (x * z - y * w)
    out[:, 7] = 2.0 * (y * z + x * w)
    out[:, 8] = 1.0 - 2.0 * (x * x + y * y)
    out[:, 9:] = pq[:, :3]
    return out


class FrameTask(Task):
    """6-D pose task; cost is ``[position x3, orientation x3]`` (``frame_task.py:44-127``)."""

    def __init__(self, frame: str, position_cost, orientation_cost, lm_damping: float = 0.0, gain: float = 1.0):
        super().__init__(cost=np.ones(6), gain=gain, lm_damping=lm_damping)
        self.frame = frame
        self.transform_target_to_world: Optional[SE3] = None
        self.target_poses: Optional[np.ndarray] = None  # [B, 12] per-instance targets (set_target_poses)
        self.target_pq: Optional[np.ndarray] = None  # [B, 7] per-instance targets as translation + quaternion (set_target_poses_quat)
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
        self.target_poses = self.target_pq = None

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
        self.target_pq = None
        self.transform_target_to_world = None  # (replaces a single target set earlier)

    def set_target_poses_quat(self, translations: np.ndarray, quaternions: np.ndarray, out: Optional[np.ndarray] = None) -> None:
        """One target per instance as ``translations [B, 3]`` and unit ``quaternions [B, 4]`` in Pinocchio's order
        ``(x, y, z, w)`` -- what ``pin.SE3ToXYZQUAT`` returns, 7 numbers per pose instead of 12: a moving-target call of
        :func:`pink_amd.solve_ik_batch` then sends 56 B per frame task and robot across PCIe instead of 96 B, and a device
        kernel writes the rotation matrices next to the other targets (``pinkhip_pose_targets_device``; the quaternion is
        normalised there).  Instance ``b`` plays ``set_target(pin.XYZQUATToSE3([*t[b], *quat[b]]))``.  ``out [B, 7]`` (e.g.
        from :func:`pink_amd.pinned_empty`) receives ``[t, quat]`` and is kept as the task's target array."""
        t = np.asarray(translations, dtype=np.float64)
        qt = np.asarray(quaternions, dtype=np.float64)
        if t.ndim != 2 or t.shape[1] != 3 or qt.shape != (t.shape[0], 4):
            raise TaskDefinitionError(f"translations [B, 3] and quaternions [B, 4] expected, got {t.shape} and {qt.shape}")
        if not (np.abs(np.einsum("bi,bi->b", qt, qt) - 1.0) < 1e-6).all():
            raise TaskDefinitionError("quaternions must have unit norm (x, y, z, w)")
        if out is None:
            out = np.empty((t.shape[0], 7))
        elif out.shape != (t.shape[0], 7) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise TaskDefinitionError(f"out must be a C-contiguous float64 array of shape {(t.shape[0], 7)}")
        out[:, :3] = t
        out[:, 3:] = qt
        self.target_pq = out
        self.target_poses = None
        self.transform_target_to_world = None

    def target_array(self) -> Optional[np.ndarray]:
        """The per-instance target array that is live: ``[B, 12]`` poses, ``[B, 7]`` translation + quaternion, or ``None``."""
        return self.target_pq if self.target_pq is not None else self.target_poses

    def poses12(self) -> Optional[np.ndarray]:
        """Per-instance targets as ``[B, 12]`` poses whatever they were given as (the host-evaluated routes read this)."""
        if self.target_pq is None:
            return self.target_poses
        return poses_from_pq(self.target_pq)

    def freeze_targets(self) -> None:
        """Declare the per-instance targets set by :meth:`set_target_poses` unchanged until the next ``set_target*`` call:
        the array becomes read-only and :func:`pink_amd.solve_ik_batch` uploads it to a device ONCE per cached device state
        instead of with every call (6.3 MB per frame task at B = 65 536: three fifths of what a call at the headline
        shape sends across PCIe are targets).  Without it every call uploads the array as it is then -- a control loop
        may refill it in place between two calls.

        The task keeps a READ-ONLY VIEW of the array; the caller's own array object is left as it is (an ``out=`` buffer
        can be handed to the next :meth:`set_target_poses`, which ends the freeze).  The promise is the caller's: a
        write into the buffer while the freeze lasts is NOT seen by the device."""
        arr = self.target_array()
        if arr is None:
            raise TargetNotSet(f"no per-instance targets set for frame '{self.frame}'")
        if getattr(arr, "frozen_token", None) is None:
            view = arr.view(FrozenTargets)
            view.flags.writeable = False  # (the view's flag: the base array stays writeable)
            view.frozen_token = next(_FREEZE_TOKENS)
            if self.target_pq is not None:
                self.target_pq = view
            else:
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
