"""Development check on the CPU wave emulator built for ONE instantiation (seconds to build):

  (cd tests/emu && g++ -O1 -std=c++17 -fPIC -shared -DPINKHIP_DEV_NV=30 -DPINKHIP_DEV_W=32 -DPINKHIP_DEV_MD=0 \
      -o /tmp/libpinkemu_dev.so emu_kernels.cpp emu_part.cpp)
  python scripts/emu_dev_check.py /tmp/libpinkemu_dev.so draco3 [B]

Solves the synthetic batches of one configuration (tight / kinematic bounds, tracking regime) on the emulator and
compares with the C oracle: max |dq - dq_ref|, statuses, mean trips, solver paths.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import c_oracle  # noqa: E402
from pink_amd import synthetic  # noqa: E402
from pink_amd._lib import Desc, Problem, Result  # noqa: E402
from conftest import EmuSolver  # noqa: E402


def main():
    libp = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "draco3"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    lib = ctypes.CDLL(libp)
    lib.pinkhip_emu_solve_host.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Problem), ctypes.POINTER(Result)]
    lib.pinkhip_emu_last_error.restype = ctypes.c_char_p
    emu = EmuSolver(lib)
    worst = 0.0
    for kw in (dict(bounds="tight"), dict(bounds="kinematic"), dict(bounds="kinematic", error_scale=0.02),
               dict(bounds="tight", jacobians="kinematic"), dict(bounds="tight", seed=7)):
        terms = synthetic.make_terms(name, B, **kw)
        ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms), nthreads=8)
        out = emu.solve(synthetic.pack(terms))
        ok = ref["status"] == 0
        err = float(np.abs(out.dq - ref["dq"])[ok].max()) if ok.any() else 0.0
        worst = max(worst, err)
        it = out.iters & 0xFFFFFF
        path = out.iters >> 24
        print(f"{name} {kw}: max|dq-ref| {err:.2e} status equal {np.array_equal(out.status, ref['status'])} trips mean {it.mean():.2f} "
              f"pair-max {np.maximum(it[0::2], it[1::2]).mean():.2f} max {it.max()} (oracle {ref['iters'].mean():.2f}) paths {np.bincount(path).tolist()}")
    print("worst", worst)


if __name__ == "__main__":
    main()
