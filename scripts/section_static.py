#!/usr/bin/env python3
"""Static instruction mix per section of the stack+solve kernel: compiles one instantiation with the
PINKHIP_SECTION_CLOCK markers (s_memtime pairs) to assembly and counts the instructions between consecutive
markers.  Sections follow PINKHIP_TICK(k) in ik_kernels_packed.h; conditional blocks are counted in full.

    python scripts/section_static.py [NV W DENSE [extra hipcc flags]]      (default 30 32 0)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pink_amd", "csrc")
NAMES = ["stacking", "cholesky+inverse", "x0", "selection", "d = J^T n", "norms/householder", "z, w", "r = P d1",
         "steps, x/u", "add", "drop", "exit"]


def main():
    nv, w, dense = (sys.argv[1:4] + ["30", "32", "0"][len(sys.argv[1:4]):])[:3]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=200000",
                        "-DPINKHIP_SECTION_CLOCK", f"-DPINKHIP_TU_NV={nv}", f"-DPINKHIP_TU_W={w}", f"-DPINKHIP_TU_DENSE={dense}",
                        "--cuda-device-only", "-S", os.path.join(CSRC, "tu_packed.hip"), "-o", out] + sys.argv[4:],
                       check=True, capture_output=True)
        lines = [l.strip() for l in open(out)]
    segs, cur, n_mark = [], {}, 0
    in_kernel = False
    for l in lines:
        if l.startswith("_ZN7pinkhip22ik_solve_packed_kernel") and ":" in l.split(";")[0]:
            in_kernel = True
            continue
        l = l.split(";")[0].strip()
        if not in_kernel or not l or l.startswith((".", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if op == "s_endpgm":
            break
        if op in ("s_memtime", "s_memrealtime"):
            n_mark += 1
            if n_mark == 1:  # the initial clock_prev read: what precedes it is the prologue
                cur = {}
            elif n_mark % 2 == 0:  # first read of a TICK closes the section
                segs.append(cur)
                cur = {}
            continue
        kind = ("fp64" if re.match(r"v_(fma|fmac|mul|add|min|max|rsq|rcp)_f64", op) else
                "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
                "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu")
        cur[kind] = cur.get(kind, 0) + 1
        cur.setdefault("ops", {})
        cur["ops"][op] = cur["ops"].get(op, 0) + 1
    segs.append(cur)
    print(f"ik_solve_packed_kernel<{nv},{w},{dense}> with clock markers: static instructions per section")
    print(f"{'section':22s} {'fp64':>6s} {'valu':>6s} {'lds':>5s} {'vmem':>5s} {'salu':>6s}   top non-fp64 VALU")
    for i, s in enumerate(segs):
        name = NAMES[i] if i < len(NAMES) else f"#{i}"
        top = sorted(((n, o) for o, n in s.get("ops", {}).items() if o.startswith("v_") and "f64" not in o), reverse=True)[:6]
        print(f"{name:22s} {s.get('fp64', 0):6d} {s.get('valu', 0):6d} {s.get('lds', 0):5d} {s.get('vmem', 0):5d} "
              f"{s.get('salu', 0):6d}   " + ", ".join(f"{o} {n}" for n, o in top))


if __name__ == "__main__":
    main()
