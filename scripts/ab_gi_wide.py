#!/usr/bin/env python3
"""A/B of builds of the Goldfarb-Idnani kernel on the weakly regularised nv = 50 shapes (examples/humanoid_jvrc.py as
shipped: no posture task): ms per 65 536 and the largest difference of dq / statuses against the first library.
    python scripts/ab_gi_wide.py lib_a.so lib_b.so ...      (one subprocess per library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [("jvrc_noposture", "tight"), ("jvrc_noposture", "tracking"), ("jvrc", "tight")]


def child(tag):
    import numpy as np
    from pink_amd import synthetic
    from pink_amd.batch_solver import BatchSolver
    s = BatchSolver(0)
    for name, bounds in CASES:
        if bounds == "tracking":
            t = synthetic.make_terms(name, 65536, bounds="kinematic", jacobians="kinematic", error_scale=0.02)
        else:
            t = synthetic.make_terms(name, 65536, bounds=bounds, jacobians="dense")
        dev = s.upload(synthetic.pack(t))
        for _ in range(2):
            s.solve_device(dev)
        s.sync()
        ms = []
        for _ in range(5):
            s.timer_start()
            s.solve_device(dev)
            ms.append(s.timer_stop())
        r = s.download(dev)
        np.savez(f"/tmp/ab_gi_{tag}_{name}_{bounds}.npz", dq=r.dq, status=r.status, iters=r.iters & 0xFFFF)
        print(f"  {name:16s} {bounds:9s} {min(ms):7.3f} ms (median {sorted(ms)[2]:7.3f})  iterations {float((r.iters & 0xFFFF).mean()):6.2f}  "
              f"statuses {np.bincount(r.status, minlength=4).tolist()}", flush=True)
        dev.free()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        return
    import numpy as np
    libs = sys.argv[1:]
    for i, lib in enumerate(libs):
        print(f"{lib}  (PINKHIP_SOLVER={os.environ.get('PINKHIP_SOLVER', 'auto')})", flush=True)
        subprocess.run([sys.executable, __file__, "--child", str(i)], env=dict(os.environ, PINKHIP_LIBRARY=os.path.abspath(lib)), check=True)
        if i:
            for name, bounds in CASES:
                a, b = (np.load(f"/tmp/ab_gi_{k}_{name}_{bounds}.npz") for k in (0, i))
                print(f"    vs first: {name} {bounds}: |dq - dq_0| max {np.abs(a['dq'] - b['dq']).max():.3e}, statuses equal {bool((a['status'] == b['status']).all())}, "
                      f"iterations equal {float((a['iters'] == b['iters']).mean()):.5f}", flush=True)


if __name__ == "__main__":
    main()
