"""Per-section cycle breakdown of the packed solve kernel (profiling build, not shipped).

    python scripts/section_clock.py build      # here: cross-compile libpinkhip_clock.so
    python scripts/section_clock.py [config]   # on the GPU box: run and print the breakdown

The profiling library is the product source compiled with -DPINKHIP_SECTION_CLOCK: every 64th wave
adds the s_memtime cycles spent in each section to a device array (ik_kernels_packed.h, PINKHIP_TICK).
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

LIB = os.environ.get("PINKHIP_CLOCK_LIBRARY", os.path.join(ge.CSRC, "libpinkhip_clock.so"))
SECTIONS_PACKED = ["stacking", "Cholesky", "J = L^-T, x0", "selection", "d = J^T n (row)", "norms, v, sync", "z, w",
                   "r = P d1", "steps, x/u update", "add (J2, P column)", "drop", "exit"]
SECTIONS_SWEEP = ["stacking + parking", "initial sweeps", "x0", "selection", "column (+ refinement)", "steps, x/u update",
                  "column of the leaving", "pivot", "exit"]
SECTIONS = SECTIONS_PACKED if os.environ.get("PINKHIP_SOLVER") == "packed" else SECTIONS_SWEEP


def build():
    """One-kernel profiling library: make DEV=1 SECTION_CLOCK=1 (one instantiation: CLOCK_NV / CLOCK_MD / CLOCK_W,
    default the headline one; the sweep-tableau kernel exports the counters unless PINKHIP_SOLVER=packed)."""
    nv, md, w = (os.environ.get(k, d) for k, d in (("CLOCK_NV", "30"), ("CLOCK_MD", "0"), ("CLOCK_W", "32")))
    extra = "" if os.environ.get("PINKHIP_SOLVER") == "packed" else "-DPINKHIP_CLOCK_SWEEP"
    if os.environ.get("PINKHIP_SOLVER") != "packed" and int(nv) + int(md) > int(w):
        extra = "-DPINKHIP_CLOCK_SWEEPX"  # (more tableau rows than lanes: the kernel with virtual dense rows, ik_sweepx.h)
    if os.environ.get("PINKHIP_SOLVER") == "packed" and md != "0":
        extra = "-DPINKHIP_CLOCK_DENSE=1"
    out = os.environ.get("CLOCK_OUT", "libpinkhip_clock.so")
    subprocess.run(["make", "-j4", "DEV=1", "SECTION_CLOCK=1", f"DEVNV={nv}", f"DEVMD={md}", f"DEVW={w}", f"EXTRA={extra}",
                    f"OBJDIR=build_clock_{nv}_{md}", f"OUT={out}"], cwd=ge.CSRC, check=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        return
    os.environ["PINKHIP_LIBRARY"] = LIB
    import numpy as np
    from pink_amd import synthetic, _lib
    from pink_amd.batch_solver import BatchSolver
    name = sys.argv[1] if len(sys.argv) > 1 else "draco3"
    bounds = sys.argv[2] if len(sys.argv) > 2 else "tight"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    s = BatchSolver(0)
    lib = _lib.load_library()
    lib.pinkhip_debug_section_clock.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    if bounds == "tracking":  # (bench.py's tracking_small_errors regime: a controller following a slowly moving target)
        t = synthetic.make_terms(name, B, bounds="kinematic", jacobians="kinematic", error_scale=0.02)
    else:
        t = synthetic.make_terms(name, B, bounds=bounds, jacobians="dense" if bounds == "tight" else "kinematic")
    dev = s.upload(synthetic.pack(t))
    out = (ctypes.c_uint64 * 16)()
    s.solve_device(dev)
    s.sync()
    lib.pinkhip_debug_section_clock(s._h, out)  # clear the warm-up launch
    s.timer_start()
    s.solve_device(dev)
    ms = s.timer_stop()
    assert lib.pinkhip_debug_section_clock(s._h, out) == 0
    r = s.download(dev)
    c = np.array(list(out), dtype=np.float64)[:len(SECTIONS)]
    print(f"{name} {bounds} B={B}: {ms:.3f} ms (instrumented), mean iterations {r.iters.mean():.1f}")
    per_wave = 64 // int(os.environ.get("CLOCK_W", "32"))
    sampled = (B // per_wave + 63) // 64  # every 64th wave adds its cycles
    for n, v in zip(SECTIONS, c):
        print(f"  {n:22s} {100 * v / c.sum():5.1f} %   {v / c.sum() * ms * 1e3 :8.1f} us of the launch   {v / sampled:9.0f} cycles per sampled wave")


if __name__ == "__main__":
    main()
