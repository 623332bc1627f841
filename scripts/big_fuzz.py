import sys, time
sys.path.insert(0, '/root/repo')
from pink_amd.batch_solver import BatchSolver
from tests import parity_suite as ps
s = BatchSolver(0)
t0 = time.time()
n = ps.fuzz(s, range(10000, 14000))
print("fuzz: instances checked", n, "in %.0f s" % (time.time() - t0))
t0 = time.time()
n = ps.kkt_certificate(s, range(20000, 22500))
print("kkt: instances certified", n, "in %.0f s" % (time.time() - t0))
