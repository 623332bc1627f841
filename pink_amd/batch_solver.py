"""Batched IK solves on one MI355X through the C ABI (``include/pinkhip.h``).

``BatchSolver`` owns one ``pinkhip_handle`` (one device, one stream).  Batches
can be solved straight from host memory (``solve``), or uploaded once and
re-solved from HBM (``upload`` / ``solve_device`` / ``download``), which is what
the benchmark times.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import PackedArgs, PinkHipError, Problem, Result
from .batch import IKBatch


PATH_NAMES = ("tableau", "handover", "routed", "goldfarb_idnani")  # PINKHIP_PATH_* of include/pinkhip.h
_PATH_SHIFT = 24


def split_iters(raw: np.ndarray) -> np.ndarray:
    """``iters[b]`` as the library writes it carries the code that solved the instance in its high bits
    (``PINKHIP_ITERS_PATH``): returns that code per instance and leaves the iteration count in ``raw`` (in place)."""
    path = (raw >> _PATH_SHIFT).astype(np.int8)
    np.bitwise_and(raw, (1 << _PATH_SHIFT) - 1, out=raw)
    return path


@dataclass
class BatchResult:
    """Output of one batched solve."""

    dq: np.ndarray  # [B, nv] displacement (divide by dt for the velocity, solve_ik.py:274)
    status: np.ndarray  # [B] int32, 0 = optimal (see include/pinkhip.h)
    iters: np.ndarray  # [B] int32 active-set iterations
    path: Optional[np.ndarray] = None  # [B] int8: which code solved the instance (index into PATH_NAMES)

    @property
    def all_found(self) -> bool:
        return bool((self.status == 0).all())

    def failed_indices(self) -> np.ndarray:
        return np.nonzero(self.status != 0)[0]

    def path_fractions(self) -> Dict[str, float]:
        """Share of the batch per solver path: ``tableau`` (sweep-tableau kernel, KKT-certified), ``handover`` (its
        result failed the certificate: solved again by the Goldfarb-Idnani code in the same launch -- the instance paid
        both), ``routed`` (sent there before the tableau iteration by the conditioning estimate), ``goldfarb_idnani``
        (that kernel by dispatch)."""
        if self.path is None or self.path.size == 0:
            return {}
        n = np.bincount(self.path.astype(np.int64), minlength=len(PATH_NAMES))
        return {name: float(n[k]) / self.path.size for k, name in enumerate(PATH_NAMES)}


class DeviceBatch:
    """A batch resident in HBM together with its output buffers."""

    def __init__(self, solver: "BatchSolver", args: PackedArgs, out_ptrs=None):
        self.solver = solver
        self.args = args
        self.ptrs: Dict[str, int] = {}
        self.nbytes = 0
        b = args.batch
        for name, arr in args.streams():
            self.ptrs[name] = solver._malloc(max(arr.nbytes, 8))
            solver._h2d(self.ptrs[name], arr)
            self.nbytes += arr.nbytes
        # outputs: library-allocated, or caller-owned device buffers (e.g. tensors of
        # a framework that will run a collective on them)
        self.owns_outputs = out_ptrs is None
        if out_ptrs is None:
            self.d_dq = solver._malloc(max(8 * b.B * b.nv, 8))
            self.d_status = solver._malloc(max(4 * b.B, 8))
            self.d_iters = solver._malloc(max(4 * b.B, 8))
        else:
            self.d_dq, self.d_status, self.d_iters = (int(p) for p in out_ptrs)
        self.d_H: Optional[int] = None
        self.d_c: Optional[int] = None
        p = Problem()
        for name in ("J", "e", "cost", "lb", "ub", "Gd", "hd", "c_extra"):
            setattr(p, name, self.ptrs.get(name))
        self.problem = p
        r = Result()
        r.dq, r.status, r.iters = self.d_dq, self.d_status, self.d_iters
        self.result = r

    def free(self) -> None:
        s = self.solver
        outs = [self.d_dq, self.d_status, self.d_iters] if self.owns_outputs else []
        for ptr in list(self.ptrs.values()) + outs + [self.d_H, self.d_c]:
            if ptr:
                s._free(ptr)
        self.ptrs = {}
        self.d_dq = self.d_status = self.d_iters = self.d_H = self.d_c = None


class BatchSolver:
    """One handle of the HIP library bound to ``device_id``."""

    def __init__(self, device_id: int = 0, library: Optional[ctypes.CDLL] = None):
        self._lib = library or _lib.load_library()
        h = ctypes.c_void_p()
        rc = self._lib.pinkhip_create(ctypes.byref(h), int(device_id))
        if rc != 0:
            raise PinkHipError(rc, (self._lib.pinkhip_last_error(None) or b"").decode())
        self._h = h
        self.device_id = device_id

    @staticmethod
    def sharded(device_ids, solver_factory=None):
        """A :class:`pink_amd.sharding.MultiDeviceSolver` over these GPUs: same ``solve(batch)``, the batch cut into
        contiguous shards, one handle + stream + host thread per device, all in this process."""
        from .sharding import MultiDeviceSolver

        return MultiDeviceSolver(device_ids, solver_factory)

    # -- plumbing --------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise PinkHipError(rc, (self._lib.pinkhip_last_error(self._h) or b"").decode())

    def _malloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._check(self._lib.pinkhip_malloc(self._h, ctypes.byref(p), int(nbytes)))
        return p.value

    def _free(self, ptr: int) -> None:
        self._check(self._lib.pinkhip_free(self._h, ctypes.c_void_p(ptr)))

    def _h2d(self, dptr: int, arr: np.ndarray) -> None:
        self._check(self._lib.pinkhip_memcpy_h2d(self._h, ctypes.c_void_p(dptr), arr.ctypes.data, arr.nbytes))

    def _d2h(self, arr: np.ndarray, dptr: int) -> None:
        self._check(self._lib.pinkhip_memcpy_d2h(self._h, arr.ctypes.data, ctypes.c_void_p(dptr), arr.nbytes))

    def close(self) -> None:
        if self._h is not None and self._h.value is not None:
            self._lib.pinkhip_destroy(self._h)
            self._h.value = None  # finalizers of page-locked arrays see a closed handle (the driver frees them)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self) -> dict:
        info = _lib.DeviceInfo()
        self._check(self._lib.pinkhip_get_device_info(self._h, ctypes.byref(info)))
        return dict(
            device_id=info.device_id, compute_units=info.compute_units,
            wavefront_size=info.wavefront_size, clock_mhz=info.clock_mhz,
            total_mem_bytes=info.total_mem_bytes, lds_per_cu_bytes=info.lds_per_cu_bytes,
            name=info.name.decode(), gcn_arch=info.gcn_arch.decode(),
        )

    # -- host-memory path ------------------------------------------------------
    def solve(self, batch: IKBatch, max_iter: int = 0, out: Optional[BatchResult] = None) -> BatchResult:
        """Stack and solve every instance of ``batch`` (host buffers in, host out).  ``out`` re-uses the result
        arrays of an earlier call (e.g. page-locked ones from :meth:`pinned_result`)."""
        a = PackedArgs(batch, max_iter)
        B, nv = batch.B, batch.nv
        if out is not None:
            dq, status, iters = out.dq, out.status, out.iters
            if dq.shape != (B, nv) or status.shape != (B,) or iters.shape != (B,):
                raise ValueError("out has the wrong shape")
            # the library writes through the raw pointers: anything but contiguous, writable float64 / int32 / int32
            # arrays would be corrupted or misread silently
            for name, arr, dt in (("dq", dq, np.float64), ("status", status, np.int32), ("iters", iters, np.int32)):
                if not isinstance(arr, np.ndarray) or arr.dtype != dt or not arr.flags.c_contiguous or not arr.flags.writeable:
                    raise ValueError(f"out.{name} must be a C-contiguous, writable {np.dtype(dt).name} array")
        else:
            dq = np.zeros((B, nv))
            status = np.zeros(B, dtype=np.int32)
            iters = np.zeros(B, dtype=np.int32)
        r = Result()
        r.dq, r.status, r.iters = dq.ctypes.data, status.ctypes.data, iters.ctypes.data
        p = a.host_problem()
        self._check(self._lib.pinkhip_solve_host(self._h, ctypes.byref(a.desc), ctypes.byref(p), ctypes.byref(r)))
        ms = ctypes.c_float(-1.0)
        self._check(self._lib.pinkhip_last_kernel_ms(self._h, ctypes.byref(ms)))
        self.last_kernel_ms = float(ms.value) if ms.value >= 0.0 else None  # (device time of this call's solve kernels)
        return BatchResult(dq, status, iters, split_iters(iters))

    def pinned_result(self, B: int, nv: int) -> BatchResult:
        """Result arrays in page-locked memory, for ``solve(..., out=)``."""
        return BatchResult(self.pinned_empty((B, nv)), self.pinned_empty((B,), np.int32), self.pinned_empty((B,), np.int32))

    def stack(self, batch: IKBatch) -> Tuple[np.ndarray, np.ndarray]:
        """QP objective only: ``H [B, nv, nv]``, ``c [B, nv]`` (``build_ik``'s P, q)."""
        a = PackedArgs(batch)
        B, nv = batch.B, batch.nv
        H = np.zeros((B, nv, nv))
        c = np.zeros((B, nv))
        p = a.host_problem()
        self._check(self._lib.pinkhip_stack_host(self._h, ctypes.byref(a.desc), ctypes.byref(p), H.ctypes.data, c.ctypes.data))
        return H, c

    def frame_task_terms(self, T_frame: np.ndarray, T_target: np.ndarray, J_body: np.ndarray):
        """Batched FrameTask error and Jacobian on the GPU (``frame_task.py:148-227``).

        ``T_frame``, ``T_target``: ``[B, 12]`` poses (rotation row-major, then translation);
        ``J_body``: ``[B, 6, nv]`` body Jacobians.  Returns ``(e [B, 6], J [B, 6, nv])``.
        """
        Tf = np.ascontiguousarray(T_frame, dtype=np.float64).reshape(-1, 12)
        Tt = np.ascontiguousarray(T_target, dtype=np.float64).reshape(-1, 12)
        Jb = np.ascontiguousarray(J_body, dtype=np.float64)
        B, _, nv = Jb.shape
        if Tf.shape[0] != B or Tt.shape[0] != B or Jb.shape[1] != 6:
            raise ValueError("inconsistent frame-task shapes")
        e = np.zeros((B, 6))
        J = np.zeros((B, 6, nv))
        self._check(self._lib.pinkhip_frame_task_host(self._h, B, nv, Tf.ctypes.data, Tt.ctypes.data, Jb.ctypes.data,
                                                      e.ctypes.data, J.ctypes.data))
        return e, J

    # -- page-locked host buffers ------------------------------------------------
    def pinned_empty(self, shape, dtype=np.float64) -> np.ndarray:
        """Uninitialised NumPy array in page-locked host memory (``pinkhip_host_alloc``): ``solve`` copies such
        arrays by DMA at the PCIe rate.  The memory is released when the array (and every view of it) is gone."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes == 0:
            return np.empty(shape, dtype=dtype)
        p = ctypes.c_void_p()
        self._check(self._lib.pinkhip_host_alloc(self._h, ctypes.byref(p), nbytes))
        buf = (ctypes.c_char * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        lib, h, addr = self._lib, self._h, p.value
        import weakref

        ranges = self.__dict__.setdefault("_pinned_ranges", {})
        ranges[addr] = nbytes

        def release():
            ranges.pop(addr, None)
            if h.value is not None:
                lib.pinkhip_host_free(h, ctypes.c_void_p(addr))

        weakref.finalize(buf, release)
        return arr

    def is_pinned(self, arr: np.ndarray) -> bool:
        """``arr`` lies inside a block handed out by :meth:`pinned_empty` (copies to / from it are asynchronous DMA)."""
        a = arr.ctypes.data
        return any(base <= a and a + arr.nbytes <= base + n for base, n in self.__dict__.get("_pinned_ranges", {}).items())

    def pin(self, batch: IKBatch) -> IKBatch:
        """Copy of ``batch`` whose per-instance streams live in page-locked memory (for repeated ``solve`` calls
        on buffers that are refilled in place: ``np.copyto(pinned.J, J)``)."""
        import dataclasses

        def mv(a):
            if a is None:
                return None
            out = self.pinned_empty(a.shape, a.dtype)
            np.copyto(out, a)
            return out

        return dataclasses.replace(batch, J=mv(batch.J), e=mv(batch.e), cost=mv(batch.cost), lb=mv(batch.lb), ub=mv(batch.ub),
                                   Gd=mv(batch.Gd), hd=mv(batch.hd), c_extra=mv(batch.c_extra))

    # -- HBM-resident path -----------------------------------------------------
    def upload(self, batch: IKBatch, max_iter: int = 0, out_ptrs=None) -> DeviceBatch:
        """Copy a batch to HBM.  ``out_ptrs = (dq, status, iters)`` device addresses
        makes the solve write into caller-owned buffers instead of library ones."""
        return DeviceBatch(self, PackedArgs(batch, max_iter), out_ptrs)

    def solve_device(self, dev: DeviceBatch) -> None:
        """Enqueue one stack+solve pass over a resident batch (asynchronous)."""
        self._check(self._lib.pinkhip_solve_device(self._h, ctypes.byref(dev.args.desc), ctypes.byref(dev.problem), ctypes.byref(dev.result)))

    def stack_device(self, dev: DeviceBatch) -> None:
        b = dev.args.batch
        if dev.d_H is None:
            dev.d_H = self._malloc(max(8 * b.B * b.nv * b.nv, 8))
            dev.d_c = self._malloc(max(8 * b.B * b.nv, 8))
        self._check(self._lib.pinkhip_stack_device(self._h, ctypes.byref(dev.args.desc), ctypes.byref(dev.problem), ctypes.c_void_p(dev.d_H), ctypes.c_void_p(dev.d_c)))

    def download(self, dev: DeviceBatch) -> BatchResult:
        b = dev.args.batch
        dq = np.zeros((b.B, b.nv))
        status = np.zeros(b.B, dtype=np.int32)
        iters = np.zeros(b.B, dtype=np.int32)
        if b.B:
            self._d2h(dq, dev.d_dq)
            self._d2h(status, dev.d_status)
            self._d2h(iters, dev.d_iters)
        return BatchResult(dq, status, iters, split_iters(iters))

    def download_stack(self, dev: DeviceBatch) -> Tuple[np.ndarray, np.ndarray]:
        b = dev.args.batch
        H = np.zeros((b.B, b.nv, b.nv))
        c = np.zeros((b.B, b.nv))
        if b.B:
            self._d2h(H, dev.d_H)
            self._d2h(c, dev.d_c)
        return H, c

    # -- raw device-pointer interface used by pink_amd.rollout.DeviceRollout ------------
    def alloc(self, nbytes: int) -> int:
        return self._malloc(max(int(nbytes), 8))

    def release(self, ptr: int) -> None:
        if ptr:
            self._free(ptr)

    def put(self, ptr: int, arr: np.ndarray) -> None:
        self._h2d(ptr, np.ascontiguousarray(arr))

    def put_overlapped(self, ptr: int, arr: np.ndarray) -> None:
        """Upload on the copy stream (``pinkhip_memcpy_h2d_overlapped``): does not wait for the kernels already
        enqueued, which keep running while the bytes move."""
        arr = np.ascontiguousarray(arr)
        self._check(self._lib.pinkhip_memcpy_h2d_overlapped(self._h, ctypes.c_void_p(ptr), arr.ctypes.data, arr.nbytes))

    def put_async(self, ptr: int, arr: np.ndarray) -> None:
        """Enqueue an upload on the copy stream and return (``pinkhip_memcpy_h2d_async``).  ``arr`` must be C-contiguous
        and stay untouched until :meth:`sync`; from page-locked memory (:meth:`pinned_empty`) it is a DMA the host does
        not wait for."""
        if not arr.flags.c_contiguous:
            raise ValueError("put_async needs a C-contiguous array")
        self._check(self._lib.pinkhip_memcpy_h2d_async(self._h, ctypes.c_void_p(ptr), arr.ctypes.data, arr.nbytes))

    def wait_copies(self) -> None:
        """Kernels enqueued from now on start after the uploads enqueued so far (``pinkhip_stream_wait_copies``)."""
        self._check(self._lib.pinkhip_stream_wait_copies(self._h))

    def select_stream(self, index: int) -> None:
        """Enqueue on compute stream ``index`` (0 or 1) from now on (``pinkhip_select_compute_stream``)."""
        self._check(self._lib.pinkhip_select_compute_stream(self._h, int(index)))

    def get_async(self, arr: np.ndarray, ptr: int) -> None:
        """Enqueue a download ordered after the kernels enqueued so far and return (``pinkhip_memcpy_d2h_async``);
        ``arr`` is valid after :meth:`sync`."""
        if not arr.flags.c_contiguous or not arr.flags.writeable:
            raise ValueError("get_async needs a C-contiguous, writable array")
        self._check(self._lib.pinkhip_memcpy_d2h_async(self._h, arr.ctypes.data, ctypes.c_void_p(ptr), arr.nbytes))

    def get(self, arr: np.ndarray, ptr: int) -> None:
        self._d2h(arr, ptr)

    def copy_d2d(self, dst: int, src: int, nbytes: int) -> None:
        """Stream-ordered device-to-device copy (asynchronous)."""
        self._check(self._lib.pinkhip_memcpy_d2d(self._h, ctypes.c_void_p(dst), ctypes.c_void_p(src), int(nbytes)))

    def model_create(self, desc) -> int:
        m = ctypes.c_void_p()
        self._check(self._lib.pinkhip_model_create(self._h, ctypes.byref(desc), ctypes.byref(m)))
        return m.value

    def model_destroy(self, model: int) -> None:
        self._check(self._lib.pinkhip_model_destroy(self._h, ctypes.c_void_p(model)))

    def fk(self, model: int, B: int, q: int, T_frames: int, J_body: int) -> None:
        self._check(self._lib.pinkhip_fk_device(self._h, ctypes.c_void_p(model), B, q, T_frames, J_body))

    def fk_frame_tasks(self, model: int, B: int, q: int, T_target: int, T_frames, e: int, sE: int, J: int, sJ: int) -> None:
        """Forward kinematics + the FrameTask rows of every model frame in one launch."""
        self._check(self._lib.pinkhip_fk_frame_tasks_device(self._h, ctypes.c_void_p(model), B, q, T_target, T_frames,
                                                            e, sE, J, sJ))

    def step_kernel(self, model: int, B: int, args) -> None:
        """Whole control step around the solve in one launch (``pinkhip_step_device``)."""
        self._check(self._lib.pinkhip_step_device(self._h, ctypes.c_void_p(model), B, ctypes.byref(args)))

    def rollout_step(self, desc, model: int, args) -> bool:
        """The whole control step in one kernel (``pinkhip_rollout_step_device``).  Returns ``False`` when no
        instantiation fits the model (the caller then uses ``step_kernel`` + ``solve_raw``)."""
        rc = self._lib.pinkhip_rollout_step_device(self._h, ctypes.byref(desc), ctypes.c_void_p(model), ctypes.byref(args))
        if rc == -5:  # PINKHIP_E_UNSUPPORTED
            return False
        self._check(rc)
        return True

    def frame_task_strided(self, B, nv, Tf, sTf, Tt, sTt, Jb, sJb, e, sE, J, sJ) -> None:
        self._check(self._lib.pinkhip_frame_task_strided_device(self._h, B, nv, Tf, sTf, Tt, sTt, Jb, sJb, e, sE, J, sJ))

    def limits_posture(self, model, B, dt, gain, q, q_target, batched, lb, ub, e, K, e_off) -> None:
        self._check(self._lib.pinkhip_limits_posture_device(self._h, ctypes.c_void_p(model), B, dt, gain, q, q_target,
                                                            batched, lb, ub, e, K, e_off))

    def check_limits(self, model, B, q, tol: float = 1e-6) -> int:
        """``b * nq + i`` of the first configuration entry of the device batch ``q`` outside its joint limits, or -1
        (``pinkhip_check_limits_device``; synchronises)."""
        bad = ctypes.c_int64(-1)
        self._check(self._lib.pinkhip_check_limits_device(self._h, ctypes.c_void_p(model), B, q, float(tol), ctypes.byref(bad)))
        return int(bad.value)

    def integrate(self, model, B, q, dq) -> None:
        self._check(self._lib.pinkhip_integrate_device(self._h, ctypes.c_void_p(model), B, q, dq))

    def pose_targets(self, B: int, pq: int, T: int) -> None:
        """``[B, 7]`` translation + quaternion targets at device address ``pq`` -> ``[B, 12]`` poses at ``T`` (asynchronous,
        on the current compute stream)."""
        self._check(self._lib.pinkhip_pose_targets_device(self._h, B, ctypes.c_void_p(pq), ctypes.c_void_p(T)))

    def integrate_checked(self, model, B, q, dq, status, first_failure, step) -> None:
        """``integrate`` that leaves instances with ``status != 0`` untouched and records the first failure."""
        self._check(self._lib.pinkhip_integrate_checked_device(self._h, ctypes.c_void_p(model), B, q, dq, status,
                                                               first_failure, int(step)))

    def solve_raw(self, desc, problem, result) -> None:
        self._check(self._lib.pinkhip_solve_device(self._h, ctypes.byref(desc), ctypes.byref(problem), ctypes.byref(result)))

    # -- RCCL gather of dq (one handle per GPU / process) ----------------------------
    def comm_unique_id(self) -> bytes:
        """128-byte RCCL id, created on one rank and shipped to the others by the caller."""
        buf = ctypes.create_string_buffer(128)
        self._check(self._lib.pinkhip_comm_get_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, nranks: int) -> None:
        self._check(self._lib.pinkhip_comm_init(self._h, unique_id, int(rank), int(nranks)))

    def comm_gather(self, d_send: int, d_recv: Optional[int], count: int, root: int = 0) -> None:
        """``count`` doubles from every rank's device buffer to ``root``'s ``[nranks * count]``."""
        self._check(self._lib.pinkhip_comm_gather(self._h, ctypes.c_void_p(d_send), ctypes.c_void_p(d_recv or 0),
                                                  int(count), int(root)))

    def comm_gather_bytes(self, d_send: int, d_recv: Optional[int], nbytes: int, root: int = 0) -> None:
        self._check(self._lib.pinkhip_comm_gather_bytes(self._h, ctypes.c_void_p(d_send), ctypes.c_void_p(d_recv or 0),
                                                        int(nbytes), int(root)))

    def comm_allgather_bytes(self, d_send: int, d_recv: int, nbytes: int) -> None:
        self._check(self._lib.pinkhip_comm_allgather_bytes(self._h, ctypes.c_void_p(d_send), ctypes.c_void_p(d_recv), int(nbytes)))

    def comm_destroy(self) -> None:
        self._check(self._lib.pinkhip_comm_destroy(self._h))

    def sync(self) -> None:
        self._check(self._lib.pinkhip_sync(self._h))

    def timer_start(self) -> None:
        self._check(self._lib.pinkhip_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = ctypes.c_float(0.0)
        self._check(self._lib.pinkhip_timer_stop(self._h, ctypes.byref(ms)))
        return float(ms.value)
