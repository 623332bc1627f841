"""ctypes front end of ``oracle/liboracle.so`` (TEST INFRASTRUCTURE ONLY).

See ``oracle/gi_oracle.c`` for what the library restates and its parity status.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "gi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_gi_work_size.restype = ctypes.c_long
        _lib.oracle_gi_work_size.argtypes = [ctypes.c_int, ctypes.c_int]
        _lib.oracle_gi_solve_eq.restype = ctypes.c_int
        _lib.oracle_gi_solve_eq.argtypes = [
            ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _ip, ctypes.c_int, _dp,
        ]
        _lib.oracle_solve_ik_batch.restype = ctypes.c_int
        _lib.oracle_solve_ik_batch.argtypes = [
            ctypes.c_long, ctypes.c_int, ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_int, _dp, _dp,
            ctypes.c_double, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _ip, _ip, _dp, _dp,
            ctypes.c_int, ctypes.c_int,
        ]
    return _lib


def _d(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_ip)


def gi_solve(P, q, G=None, h=None, max_iter: int = 0, meq: int = 0):
    """One QP through the C Goldfarb-Idnani; returns (x, status, iters, lam).  The first
    ``meq`` rows of ``(G, h)`` are equalities."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n = q.size
    m = 0 if G is None else int(np.asarray(h).size)
    Gc = None if m == 0 else np.ascontiguousarray(G, dtype=np.float64)
    hc = None if m == 0 else np.ascontiguousarray(h, dtype=np.float64)
    x = np.zeros(n)
    lam = np.zeros(max(m, 1))
    it = ctypes.c_int(0)
    work = np.zeros(lib().oracle_gi_work_size(n, m))
    st = lib().oracle_gi_solve_eq(n, _d(P), _d(q), m, int(meq), _d(Gc), _d(hc), _d(x), _d(lam), ctypes.byref(it), max_iter, _d(work))
    return x, st, it.value, lam[:m]


def solve_ik_batch(
    J, e, cost, gain, lm, rows, damping, G=None, h=None, diag_extra=None, c_extra=None,
    want_Hc: bool = False, solve: bool = True, nthreads: int = 1, meq: int = 0,
):
    """Pink-form batch -> dq.  Shapes: J [B,K,nv], e [B,K], cost [K] or [B,K],
    G [B,m,nv], h [B,m].  Returns dict(dq, status, iters[, H, c])."""
    J = np.ascontiguousarray(J, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    B, K, nv = J.shape
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    T = rows.size - 1
    assert rows[-1] == K and e.shape == (B, K)
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    cost_batched = int(cost.ndim == 2)
    gain = np.ascontiguousarray(gain, dtype=np.float64)
    lm = np.ascontiguousarray(lm, dtype=np.float64)
    m = 0 if G is None else int(np.asarray(G).shape[1])
    Gc = None if m == 0 else np.ascontiguousarray(G, dtype=np.float64)
    hc = None if m == 0 else np.ascontiguousarray(h, dtype=np.float64)
    de = None if diag_extra is None else np.ascontiguousarray(diag_extra, dtype=np.float64)
    ce = None if c_extra is None else np.ascontiguousarray(c_extra, dtype=np.float64)
    dq = np.zeros((B, nv))
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    H = np.zeros((B, nv, nv)) if want_Hc else None
    c = np.zeros((B, nv)) if want_Hc else None
    rc = lib().oracle_solve_ik_batch(
        B, nv, T, _i(rows), _d(J), _d(e), _d(cost), cost_batched, _d(gain), _d(lm), float(damping),
        _d(de), _d(ce), m, int(meq), _d(Gc), _d(hc), _d(dq), _i(status), _i(iters), _d(H), _d(c),
        int(solve), int(nthreads),
    )
    if rc != 0:
        raise MemoryError("oracle_solve_ik_batch failed to allocate scratch")
    out = {"dq": dq, "status": status, "iters": iters}
    if want_Hc:
        out["H"] = H
        out["c"] = c
    return out
