"""Ad-hoc GPU check: parity vs the C oracle on every config + first timings."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle
from pink_amd import synthetic
from pink_amd.batch_solver import BatchSolver

out = {}
s = BatchSolver(0)
print(s.device_info())
for name, B in (("ur5", 1024), ("draco3", 1024), ("jvrc", 512)):
    for bounds, jac in (("tight", "dense"), ("kinematic", "kinematic")):
        t = synthetic.make_terms(name, B, bounds=bounds, jacobians=jac)
        ref = c_oracle.solve_ik_batch(**synthetic.pink_form(t), want_Hc=True, nthreads=0)
        pk = synthetic.pack(t)
        H, c = s.stack(pk)
        r = s.solve(pk)
        err = np.abs(r.dq - ref["dq"]).max()
        print(f"{name:7s} {bounds:9s} B={B} stack dH={np.abs(H-ref['H']).max():.2e} dc={np.abs(c-ref['c']).max():.2e} "
              f"solve max|ddq|={err:.3e} status!=0: {(r.status!=0).sum()} iters gpu/ref mean {r.iters.mean():.1f}/{ref['iters'].mean():.1f} same_iters={(r.iters==ref['iters']).mean():.3f}")
        out[f"{name}_{bounds}"] = dict(err=float(err), bad=int((r.status != 0).sum()))
# timing
for name, B in (("draco3", 65536), ("ur5", 65536), ("jvrc", 32768)):
    for bounds in ("tight", "kinematic"):
        t = synthetic.make_terms(name, B, bounds=bounds, jacobians="dense" if bounds == "tight" else "kinematic")
        pk = synthetic.pack(t)
        dev = s.upload(pk)
        s.solve_device(dev); s.sync()
        reps = 5
        s.timer_start()
        for _ in range(reps):
            s.solve_device(dev)
        ms = s.timer_stop() / reps
        r = s.download(dev)
        s.timer_start()
        for _ in range(reps):
            s.stack_device(dev)
        ms_stack = s.timer_stop() / reps
        print(f"TIMING {name} {bounds} B={B}: solve {ms:.3f} ms -> {B/ms*1e3/1e6:.2f} M solves/s; iters mean {r.iters.mean():.1f}; "
              f"stack {ms_stack:.3f} ms -> {B*pk.bytes_per_stack()/ms_stack/1e6:.1f} GB/s; bytes/qp {pk.bytes_per_qp()}")
        out[f"time_{name}_{bounds}"] = dict(ms=ms, msolves=B / ms * 1e3 / 1e6, ms_stack=ms_stack)
        if name == "draco3" and bounds == "tight":
            # PCIe-inclusive: host buffers in, host buffers out (pinkhip_solve_host)
            s.solve(pk)
            t0 = time.perf_counter()
            for _ in range(3):
                s.solve(pk)
            host_ms = (time.perf_counter() - t0) / 3 * 1e3
            print(f"HOST-PATH {name} B={B}: {host_ms:.2f} ms per call incl. H2D/D2H -> {B/host_ms*1e3/1e6:.2f} M solves/s "
                  f"({dev.nbytes/1e6:.0f} MB in)")
            out["host_path_ms"] = host_ms
        dev.free()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_check.json"), "w"), indent=1)
