import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from pink_amd import synthetic
from pink_amd.batch_solver import BatchSolver
s=BatchSolver(0)
for name,B in (("draco3",524288),("ur5",2097152)):
    t=synthetic.make_terms(name,B); pk=synthetic.pack(t); dev=s.upload(pk)
    s.solve_device(dev); s.sync()
    s.timer_start()
    for _ in range(3): s.solve_device(dev)
    ms=s.timer_stop()/3
    r=s.download(dev)
    print(f"{name} B={B}: {ms:.2f} ms {B/ms/1e3:.1f} M/s bad={(r.status!=0).sum()} lb_ok={(r.dq>=pk.lb-1e-12).all()} ub_ok={(r.dq<=pk.ub+1e-12).all()}")
    dev.free()
