"""Shared fixtures.  ``-m "not gpu"`` runs everywhere; ``-m gpu`` needs an MI355X."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


@pytest.fixture(scope="session")
def built():
    """Everything native is compiled (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    g.build()
    return g


@pytest.fixture(scope="session")
def emu(built):
    """CPU wave emulator running the real kernel source (tests/emu)."""
    from pink_amd._lib import Desc, Problem, Result

    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libpinkemu.so"))
    lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_emu_stack_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.c_void_p, ctypes.c_void_p]
    lib.pinkhip_emu_last_error.restype = ctypes.c_char_p
    lib.pinkhip_emu_frame_task_host.argtypes = [ctypes.c_longlong, ctypes.c_int] + [ctypes.c_void_p] * 5
    return EmuSolver(lib)


@pytest.fixture
def emu_async(emu):
    """The emulator behind the ASYNCHRONOUS half of BatchSolver's surface (page-locked arrays, copy / result streams, two
    compute streams): everything completes immediately, so the host logic of the pipelined array call -- ranges, frozen
    targets, results written into page-locked arrays by the kernel -- runs under ``-m "not gpu"`` too."""
    return AsyncEmuSolver(emu.lib)


class EmuSolver:
    """Same surface as pink_amd.batch_solver.BatchSolver.solve/stack, on the emulator."""

    def __init__(self, lib):
        self.lib = lib

    def solve(self, batch, max_iter=0):
        from pink_amd._lib import PackedArgs, Result
        from pink_amd.batch_solver import BatchResult, split_iters

        a = PackedArgs(batch, max_iter)
        dq = np.zeros((batch.B, batch.nv))
        st = np.zeros(batch.B, np.int32)
        it = np.zeros(batch.B, np.int32)
        r = Result()
        r.dq, r.status, r.iters = dq.ctypes.data, st.ctypes.data, it.ctypes.data
        p = a.host_problem()
        rc = self.lib.pinkhip_emu_solve_host(ctypes.byref(a.desc), ctypes.byref(p), ctypes.byref(r))
        if rc != 0:
            raise RuntimeError(self.lib.pinkhip_emu_last_error().decode())
        return BatchResult(dq, st, it, split_iters(it))

    def frame_task_terms(self, T_frame, T_target, J_body):
        Tf = np.ascontiguousarray(T_frame, dtype=np.float64).reshape(-1, 12)
        Tt = np.ascontiguousarray(T_target, dtype=np.float64).reshape(-1, 12)
        Jb = np.ascontiguousarray(J_body, dtype=np.float64)
        B, _, nv = Jb.shape
        e = np.zeros((B, 6))
        J = np.zeros((B, 6, nv))
        self.lib.pinkhip_emu_frame_task_host.argtypes = [ctypes.c_longlong, ctypes.c_int] + [ctypes.c_void_p] * 5
        self.lib.pinkhip_emu_frame_task_host(B, nv, Tf.ctypes.data, Tt.ctypes.data, Jb.ctypes.data, e.ctypes.data, J.ctypes.data)
        return e, J

    # -- raw "device" pointer interface of pink_amd.rollout.DeviceRollout, on host memory ----
    def _bufs(self):
        if not hasattr(self, "_mem"):
            self._mem = {}
        return self._mem

    def alloc(self, nbytes):
        buf = np.zeros(max(int(nbytes), 8) // 8 + 1, dtype=np.float64)
        buf[:] = np.nan  # never-written device memory is garbage: make it visible
        self._bufs()[buf.ctypes.data] = buf
        return buf.ctypes.data

    def release(self, ptr):
        self._bufs().pop(ptr, None)

    def put(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        ctypes.memmove(ptr, arr.ctypes.data, arr.nbytes)

    put_overlapped = put  # (no streams here: the emulator runs every launch to completion)

    def get(self, arr, ptr):
        ctypes.memmove(arr.ctypes.data, ptr, arr.nbytes)

    def sync(self):
        pass

    def model_create(self, desc):
        m = ctypes.c_void_p()
        self.lib.pinkhip_emu_model_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        rc = self.lib.pinkhip_emu_model_create(ctypes.byref(desc), ctypes.byref(m))
        if rc != 0:
            raise RuntimeError(self.lib.pinkhip_emu_last_error().decode())
        return m.value

    def model_destroy(self, model):
        self.lib.pinkhip_emu_model_destroy.argtypes = [ctypes.c_void_p]
        self.lib.pinkhip_emu_model_destroy(ctypes.c_void_p(model))

    def fk(self, model, B, q, T_frames, J_body):
        vp = ctypes.c_void_p
        self.lib.pinkhip_emu_fk.argtypes = [vp, ctypes.c_longlong, vp, vp, vp]
        self.lib.pinkhip_emu_fk(model, B, q, T_frames, J_body)

    def fk_frame_tasks(self, model, B, q, T_target, T_frames, e, sE, J, sJ):
        vp, ll = ctypes.c_void_p, ctypes.c_longlong
        self.lib.pinkhip_emu_fk_frame_tasks.argtypes = [vp, ll, vp, vp, vp, vp, ll, vp, ll]
        self.lib.pinkhip_emu_fk_frame_tasks(model, B, q, T_target, T_frames, e, sE, J, sJ)

    def step_kernel(self, model, B, args):
        from pink_amd._lib import Step

        self.lib.pinkhip_emu_step.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.POINTER(Step)]
        self.lib.pinkhip_emu_step(model, B, ctypes.byref(args))

    def rollout_step(self, desc, model, args):
        from pink_amd._lib import Desc, RolloutStep

        self.lib.pinkhip_emu_rollout_step.argtypes = [ctypes.POINTER(Desc), ctypes.c_void_p, ctypes.POINTER(RolloutStep)]
        rc = self.lib.pinkhip_emu_rollout_step(ctypes.byref(desc), model, ctypes.byref(args))
        if rc == -5:
            return False
        if rc != 0:
            raise RuntimeError(self.lib.pinkhip_emu_last_error().decode())
        return True

    def frame_task_strided(self, B, nv, Tf, sTf, Tt, sTt, Jb, sJb, e, sE, J, sJ):
        vp, ll = ctypes.c_void_p, ctypes.c_longlong
        self.lib.pinkhip_emu_frame_task_strided.argtypes = [ll, ctypes.c_int, vp, ll, vp, ll, vp, ll, vp, ll, vp, ll]
        self.lib.pinkhip_emu_frame_task_strided(B, nv, Tf, sTf, Tt, sTt, Jb, sJb, e, sE, J, sJ)

    def limits_posture(self, model, B, dt, gain, q, q_target, batched, lb, ub, e, K, e_off):
        vp = ctypes.c_void_p
        self.lib.pinkhip_emu_limits_posture.argtypes = [vp, ctypes.c_longlong, ctypes.c_double, ctypes.c_double, vp, vp,
                                                        ctypes.c_int, vp, vp, vp, ctypes.c_int, ctypes.c_int]
        self.lib.pinkhip_emu_limits_posture(model, B, dt, gain, q, q_target, batched, lb, ub, e, K, e_off)

    def check_limits(self, model, B, q, tol=1e-6):
        vp = ctypes.c_void_p
        bad = ctypes.c_longlong(-1)
        self.lib.pinkhip_emu_check_limits.argtypes = [vp, ctypes.c_longlong, vp, ctypes.c_double, ctypes.POINTER(ctypes.c_longlong)]
        self.lib.pinkhip_emu_check_limits(model, B, q, tol, ctypes.byref(bad))
        return int(bad.value)

    def integrate(self, model, B, q, dq):
        vp = ctypes.c_void_p
        self.lib.pinkhip_emu_integrate.argtypes = [vp, ctypes.c_longlong, vp, vp]
        self.lib.pinkhip_emu_integrate(model, B, q, dq)

    def pose_targets(self, B, pq, T):
        vp = ctypes.c_void_p
        self.lib.pinkhip_emu_pose_targets.argtypes = [ctypes.c_longlong, vp, vp]
        self.lib.pinkhip_emu_pose_targets(B, pq, T)

    def integrate_checked(self, model, B, q, dq, status, first_failure, step):
        vp = ctypes.c_void_p
        self.lib.pinkhip_emu_integrate_checked.argtypes = [vp, ctypes.c_longlong, vp, vp, vp, vp, ctypes.c_int]
        self.lib.pinkhip_emu_integrate_checked(model, B, q, dq, status, first_failure, step)

    def solve_raw(self, desc, problem, result):
        rc = self.lib.pinkhip_emu_solve_host(ctypes.byref(desc), ctypes.byref(problem), ctypes.byref(result))
        if rc != 0:
            raise RuntimeError(self.lib.pinkhip_emu_last_error().decode())

    def stack(self, batch):
        from pink_amd._lib import PackedArgs

        a = PackedArgs(batch)
        H = np.zeros((batch.B, batch.nv, batch.nv))
        c = np.zeros((batch.B, batch.nv))
        p = a.host_problem()
        rc = self.lib.pinkhip_emu_stack_host(ctypes.byref(a.desc), ctypes.byref(p), H.ctypes.data, c.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.lib.pinkhip_emu_last_error().decode())
        return H, c


@pytest.fixture(scope="session")
def gpu_solver(built):
    """A BatchSolver on device 0; only requested by gpu-marked tests."""
    from pink_amd.batch_solver import BatchSolver

    s = BatchSolver(0)
    yield s
    s.close()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "pink_build_ik.npz"))


class AsyncEmuSolver(EmuSolver):
    """EmuSolver + put_async / get_async / wait_copies / select_stream / pinned_empty / is_pinned (see ``emu_async``)."""

    def __init__(self, lib):
        super().__init__(lib)
        self._pinned, self.streams_selected, self.async_copies, self.async_gets = [], [], 0, 0

    def pinned_empty(self, shape, dtype=np.float64):
        arr = np.empty(shape, dtype=dtype)
        self._pinned.append(arr)  # (kept alive: the registry is by address)
        return arr

    def is_pinned(self, arr):
        a = arr.ctypes.data
        return any(p.ctypes.data <= a and a + arr.nbytes <= p.ctypes.data + p.nbytes for p in self._pinned)

    def put_async(self, ptr, arr):
        self.async_copies += 1  # (a pageable source is legal: the runtime stages it before the call returns)
        self.put(ptr, arr)

    def get_async(self, arr, ptr):
        assert self.is_pinned(arr)
        self.async_gets += 1
        self.get(arr, ptr)

    def wait_copies(self):
        pass

    def select_stream(self, index):
        self.streams_selected.append(int(index))
