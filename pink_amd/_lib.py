"""ctypes binding of ``libpinkhip.so`` (the C ABI declared in ``include/pinkhip.h``).

No PyTorch, no pybind: the host side stays plain Python + NumPy and talks to the
HIP library through ``extern "C"`` entry points.  There is no CPU fallback: if
the library cannot be loaded, or no gfx950 device is visible, the calls raise.
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

from .batch import IKBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpinkhip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)


class PinkHipError(RuntimeError):
    """API-level failure of the HIP library (negative return code)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"pinkhip error {code}: {message}")
        self.code = code


class Desc(ctypes.Structure):
    """``pinkhip_desc``."""

    _fields_ = [
        ("B", ctypes.c_int64),
        ("nv", ctypes.c_int32),
        ("T", ctypes.c_int32),
        ("Kd", ctypes.c_int32),
        ("K", ctypes.c_int32),
        ("md", ctypes.c_int32),
        ("n_eq", ctypes.c_int32),
        ("task_rows", c_int32_p),
        ("task_kind", c_int32_p),
        ("task_col0", c_int32_p),
        ("gain", c_double_p),
        ("lm_damping", c_double_p),
        ("n_barriers", ctypes.c_int32),
        ("barrier_rows", c_int32_p),
        ("barrier_safe_gain", c_double_p),
        ("damping", ctypes.c_double),
        ("dt", ctypes.c_double),
        ("cost_is_batched", ctypes.c_int32),
        ("max_iter", ctypes.c_int32),
        ("n_free_lead", ctypes.c_int32),
    ]


class Problem(ctypes.Structure):
    """``pinkhip_problem`` (host or device addresses)."""

    _fields_ = [(n, ctypes.c_void_p) for n in ("J", "e", "cost", "lb", "ub", "Gd", "hd", "c_extra")]


class Result(ctypes.Structure):
    """``pinkhip_result``."""

    _fields_ = [(n, ctypes.c_void_p) for n in ("dq", "status", "iters")]


class DeviceInfo(ctypes.Structure):
    """``pinkhip_device_info``."""

    _fields_ = [
        ("device_id", ctypes.c_int32),
        ("compute_units", ctypes.c_int32),
        ("wavefront_size", ctypes.c_int32),
        ("clock_mhz", ctypes.c_int32),
        ("total_mem_bytes", ctypes.c_int64),
        ("lds_per_cu_bytes", ctypes.c_int64),
        ("name", ctypes.c_char * 128),
        ("gcn_arch", ctypes.c_char * 64),
    ]


class Step(ctypes.Structure):
    """``pinkhip_step``."""

    _fields_ = [
        ("q", ctypes.c_void_p), ("dq_prev", ctypes.c_void_p), ("status", ctypes.c_void_p), ("first_failure", ctypes.c_void_p),
        ("step", ctypes.c_int32), ("target_batched", ctypes.c_int32),
        ("T_target", ctypes.c_void_p), ("T_frames", ctypes.c_void_p),
        ("e", ctypes.c_void_p), ("sE", ctypes.c_int64), ("J", ctypes.c_void_p), ("sJ", ctypes.c_int64),
        ("dt", ctypes.c_double), ("config_limit_gain", ctypes.c_double),
        ("q_target", ctypes.c_void_p), ("lb", ctypes.c_void_p), ("ub", ctypes.c_void_p), ("e_off", ctypes.c_int32),
        ("root_box", ctypes.c_void_p),
    ]


class RolloutStep(ctypes.Structure):
    """``pinkhip_rollout_step``."""

    _fields_ = [
        ("q", ctypes.c_void_p), ("cost", ctypes.c_void_p), ("T_target", ctypes.c_void_p), ("T_frames", ctypes.c_void_p),
        ("q_target", ctypes.c_void_p), ("dq", ctypes.c_void_p), ("status", ctypes.c_void_p), ("iters", ctypes.c_void_p),
        ("first_failure", ctypes.c_void_p), ("config_limit_gain", ctypes.c_double),
        ("target_batched", ctypes.c_int32), ("step", ctypes.c_int32), ("integrate", ctypes.c_int32),
        ("barrier_frame", ctypes.c_void_p), ("barrier_axis", ctypes.c_void_p), ("barrier_sign", ctypes.c_void_p),
        ("barrier_bound", ctypes.c_void_p), ("barrier_gain", ctypes.c_void_p),
        ("sT_b", ctypes.c_int64), ("sT_f", ctypes.c_int64),
        ("root_box", ctypes.c_void_p), ("n_limit_rows", ctypes.c_int32), ("limit_rows", ctypes.c_void_p),
        ("limit_h", ctypes.c_void_p), ("dq_scale", ctypes.c_double),
        ("n_const_rows", ctypes.c_int32), ("const_rows", ctypes.c_void_p), ("const_q0", ctypes.c_void_p), ("const_b", ctypes.c_void_p),
        ("posture_task", ctypes.c_int32), ("diag_error", ctypes.c_void_p), ("acc_limit", ctypes.c_void_p),
        ("n_constraint_frames", ctypes.c_int32), ("constraint_frame", ctypes.c_void_p), ("constraint_gain", ctypes.c_void_p),
        ("barrier_frame2", ctypes.c_void_p),
    ]


# every symbol include/pinkhip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "pinkhip_version", "pinkhip_device_count", "pinkhip_create", "pinkhip_destroy",
    "pinkhip_last_error", "pinkhip_get_device_info", "pinkhip_solve_host", "pinkhip_last_kernel_ms", "pinkhip_solve_device",
    "pinkhip_stack_host", "pinkhip_stack_device", "pinkhip_frame_task_host", "pinkhip_frame_task_device",
    "pinkhip_frame_task_strided_device", "pinkhip_model_create", "pinkhip_model_destroy", "pinkhip_fk_device",
    "pinkhip_fk_frame_tasks_device", "pinkhip_step_device", "pinkhip_rollout_step_device",
    "pinkhip_limits_posture_device", "pinkhip_check_limits_device", "pinkhip_integrate_device", "pinkhip_integrate_checked_device",
    "pinkhip_pose_targets_device",
    "pinkhip_comm_get_unique_id", "pinkhip_comm_init", "pinkhip_comm_gather", "pinkhip_comm_gather_bytes",
    "pinkhip_comm_allgather_bytes", "pinkhip_comm_destroy",
    "pinkhip_host_alloc", "pinkhip_host_free", "pinkhip_malloc", "pinkhip_free",
    "pinkhip_memcpy_h2d", "pinkhip_memcpy_h2d_overlapped", "pinkhip_memcpy_h2d_async", "pinkhip_stream_wait_copies", "pinkhip_select_compute_stream",
    "pinkhip_memcpy_d2h_async", "pinkhip_memcpy_d2h", "pinkhip_memcpy_d2d", "pinkhip_sync", "pinkhip_timer_start",
    "pinkhip_timer_stop",
)

_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load ``libpinkhip.so`` (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PINKHIP_LIBRARY", LIB_PATH)
    if not os.path.exists(p):
        raise PinkHipError(
            -4, f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
        )
    lib = ctypes.CDLL(p)
    vp = ctypes.c_void_p
    lib.pinkhip_version.restype = ctypes.c_int
    lib.pinkhip_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    lib.pinkhip_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.pinkhip_destroy.argtypes = [vp]
    lib.pinkhip_last_error.argtypes = [vp]
    lib.pinkhip_last_error.restype = ctypes.c_char_p
    lib.pinkhip_get_device_info.argtypes = [vp, ctypes.POINTER(DeviceInfo)]
    for name in ("pinkhip_solve_host", "pinkhip_solve_device"):
        getattr(lib, name).argtypes = [vp, ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    for name in ("pinkhip_stack_host", "pinkhip_stack_device"):
        getattr(lib, name).argtypes = [vp, ctypes.POINTER(Desc), ctypes.POINTER(Problem), vp, vp]
    for name in ("pinkhip_frame_task_host", "pinkhip_frame_task_device"):
        getattr(lib, name).argtypes = [vp, ctypes.c_int64, ctypes.c_int32, vp, vp, vp, vp, vp]
    i64, i32, f64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_double
    lib.pinkhip_frame_task_strided_device.argtypes = [vp, i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64]
    lib.pinkhip_model_create.argtypes = [vp, vp, ctypes.POINTER(vp)]
    lib.pinkhip_model_destroy.argtypes = [vp, vp]
    lib.pinkhip_fk_device.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.pinkhip_fk_frame_tasks_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, i64]
    lib.pinkhip_step_device.argtypes = [vp, vp, i64, ctypes.POINTER(Step)]
    lib.pinkhip_rollout_step_device.argtypes = [vp, ctypes.POINTER(Desc), vp, ctypes.POINTER(RolloutStep)]
    lib.pinkhip_limits_posture_device.argtypes = [vp, vp, i64, f64, f64, vp, vp, i32, vp, vp, vp, i32, i32]
    lib.pinkhip_check_limits_device.argtypes = [vp, vp, i64, vp, ctypes.c_double, ctypes.POINTER(ctypes.c_int64)]
    lib.pinkhip_integrate_device.argtypes = [vp, vp, i64, vp, vp]
    lib.pinkhip_pose_targets_device.argtypes = [vp, i64, vp, vp]
    lib.pinkhip_integrate_checked_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, i32]
    lib.pinkhip_comm_get_unique_id.argtypes = [ctypes.c_char_p]
    lib.pinkhip_comm_init.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lib.pinkhip_comm_gather.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int]
    lib.pinkhip_comm_gather_bytes.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int]
    lib.pinkhip_comm_allgather_bytes.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_comm_destroy.argtypes = [vp]
    lib.pinkhip_host_alloc.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_int64]
    lib.pinkhip_host_free.argtypes = [vp, vp]
    lib.pinkhip_malloc.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_int64]
    lib.pinkhip_free.argtypes = [vp, vp]
    lib.pinkhip_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_memcpy_h2d_overlapped.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_memcpy_h2d_async.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_memcpy_d2h_async.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_stream_wait_copies.argtypes = [vp]
    lib.pinkhip_select_compute_stream.argtypes = [vp, ctypes.c_int32]
    lib.pinkhip_memcpy_d2d.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.pinkhip_sync.argtypes = [vp]
    lib.pinkhip_timer_start.argtypes = [vp]
    lib.pinkhip_timer_stop.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    if path is None:
        _lib = lib
    return lib


class PackedArgs:
    """Keeps the NumPy buffers behind a (Desc, Problem) pair alive."""

    def __init__(self, batch: IKBatch, max_iter: int = 0):
        self.batch = batch
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        self.task_rows = i32(batch.task_rows)
        self.task_kind = i32(batch.task_kind)
        self.task_col0 = i32(batch.task_col0)
        self.gain = f64(batch.gain)
        self.lm = f64(batch.lm_damping)
        self.barrier_rows = i32(batch.barrier_rows)
        self.barrier_safe_gain = f64(batch.barrier_safe_gain)
        self.J, self.e, self.cost = f64(batch.J), f64(batch.e), f64(batch.cost)
        self.lb, self.ub = f64(batch.lb), f64(batch.ub)
        self.Gd, self.hd = f64(batch.Gd), f64(batch.hd)
        self.c_extra = None if batch.c_extra is None else f64(batch.c_extra)
        B, nv = batch.B, batch.nv
        if self.J.shape != (B, batch.Kd, nv) or self.lb.shape != (B, nv) or self.ub.shape != (B, nv):
            raise ValueError("inconsistent batch shapes")
        if self.Gd.shape != (B, batch.md, nv) or self.hd.shape != (B, batch.md):
            raise ValueError("inconsistent dense-row shapes")
        d = Desc()
        d.B, d.nv, d.T, d.Kd, d.K, d.md, d.n_eq = B, nv, batch.T, batch.Kd, batch.K, batch.md, int(batch.n_eq)
        d.task_rows = self.task_rows.ctypes.data_as(c_int32_p)
        d.task_kind = self.task_kind.ctypes.data_as(c_int32_p)
        d.task_col0 = self.task_col0.ctypes.data_as(c_int32_p)
        d.gain = self.gain.ctypes.data_as(c_double_p)
        d.lm_damping = self.lm.ctypes.data_as(c_double_p)
        d.n_barriers = int(self.barrier_safe_gain.size)
        d.barrier_rows = self.barrier_rows.ctypes.data_as(c_int32_p)
        d.barrier_safe_gain = self.barrier_safe_gain.ctypes.data_as(c_double_p)
        d.damping, d.dt = float(batch.damping), float(batch.dt)
        d.cost_is_batched = int(self.cost.ndim == 2)
        d.max_iter = int(max_iter)
        # how many leading coordinates carry no bound in any instance (the root of a free-flyer): lets the library solve
        # nv = 33 / 34 box-only batches two per wavefront (include/pinkhip.h, n_free_lead).  Only looked for where it matters.
        d.n_free_lead = 0
        if 32 < nv <= 34 and batch.md == 0 and B > 0:
            free = np.isneginf(self.lb[:, :8]).all(axis=0) & np.isposinf(self.ub[:, :8]).all(axis=0)
            d.n_free_lead = int(free.argmin()) if not free.all() else int(free.size)
        self.desc = d

    def host_problem(self) -> Problem:
        p = Problem()
        p.J, p.e, p.cost = self.J.ctypes.data, self.e.ctypes.data, self.cost.ctypes.data
        p.lb, p.ub = self.lb.ctypes.data, self.ub.ctypes.data
        p.Gd, p.hd = self.Gd.ctypes.data, self.hd.ctypes.data
        p.c_extra = None if self.c_extra is None else self.c_extra.ctypes.data
        return p

    def streams(self):
        """(name, array) pairs of the per-instance inputs, for device uploads."""
        out = [("J", self.J), ("e", self.e), ("cost", self.cost), ("lb", self.lb), ("ub", self.ub),
               ("Gd", self.Gd), ("hd", self.hd)]
        if self.c_extra is not None:
            out.append(("c_extra", self.c_extra))
        return out
