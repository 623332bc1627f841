import sys, os, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
import conftest, parity_suite as ps
from pink_amd._lib import Desc, Problem, Result
from pink_amd.batch import pack_terms, DenseTaskTerm, DiagonalTaskTerm
from oracle import c_oracle
from pink_amd.batch_solver import BatchSolver
emu = BatchSolver(0)
sd = int(sys.argv[1])
rng = np.random.default_rng(sd)
nv = int(rng.integers(1, 34)); B = int(rng.integers(1, 7))
neq = int(rng.integers(0, min(3, nv) + 1)) if rng.random() < 0.4 else 0
mdi = int(rng.integers(0, 5)) if rng.random() < 0.5 else 0
k = int(rng.integers(1, 7))
J = rng.normal(0, 0.5, size=(B, k, nv)); e = 0.1 * rng.normal(size=(B, k)); ep = rng.uniform(-0.5, 0.5, size=(B, nv))
tight = 10 ** rng.uniform(-3, -1)
lb = -rng.uniform(0.2 * tight, tight, size=(B, nv)); ub = rng.uniform(0.2 * tight, tight, size=(B, nv))
m = rng.random(size=(B, nv)); lb[m < 0.1] = -np.inf; ub[(m > 0.1) & (m < 0.2)] = np.inf
pinned = rng.random(size=(B, nv)) < 0.05; lb[pinned] = ub[pinned] = 0.0
A = rng.normal(size=(B, neq, nv)); bv = 0.01 * rng.normal(size=(B, neq))
Gi = rng.normal(size=(B, mdi, nv)); hi = rng.uniform(-0.01, 0.05, size=(B, mdi))
if mdi >= 2 and rng.random() < 0.3: Gi[:, 1], hi[:, 1] = Gi[:, 0], hi[:, 0] + 0.01
lm = float(rng.choice([0.0, 0.5])); cost = rng.uniform(0.5, 2, size=k)
print("nv", nv, "B", B, "neq", neq, "mdi", mdi, "k", k, "tight", tight, "lm", lm)
batch = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=cost, lm_damping=lm), DiagonalTaskTerm(col0=0, e=ep, cost=0.1)], 0.005, 1e-12, boxes=[(lb, ub)], dense_rows=[(Gi, hi)] if mdi else (), equality_rows=[(A, bv)] if neq else (), batch_size=B)
for solver in ("sweep", "packed"):
    os.environ["PINKHIP_SOLVER"] = solver
    out = emu.solve(batch)
    print(solver, out.status, out.iters)
    b = int(sys.argv[2]) if len(sys.argv) > 2 else B - 1
    x = out.dq[b]
    print(" box viol", max((lb[b] - x).max(), (x - ub[b]).max()), " eq resid", np.abs(A[b] @ x - bv[b]).max() if neq else None, " ineq viol", (Gi[b] @ x - hi[b]).max() if mdi else None)
eye = np.eye(nv); hb = np.concatenate([ub, -lb], axis=1); hb = np.where(np.isfinite(hb), hb, 1e30)
G = np.concatenate([A, np.broadcast_to(eye, (B, nv, nv)), np.broadcast_to(-eye, (B, nv, nv)), Gi], axis=1); h = np.concatenate([bv, hb, hi], axis=1)
ref = c_oracle.solve_ik_batch(np.concatenate([J, np.broadcast_to(eye, (B, nv, nv))], axis=1), np.concatenate([e, ep], axis=1), np.concatenate([cost, np.full(nv, 0.1)]), np.ones(2), np.array([lm, 0.0]), np.array([0, k, k + nv], np.int32), 1e-12, G, h, meq=neq, want_Hc=True)
print("oracle", ref["status"])
print("pinned cols of b:", np.nonzero(pinned[b])[0], " A[b] cols:", None if not neq else A[b].shape)
# feasibility by LP-ish check: least squares on equalities restricted to free coords
import scipy.optimize as so
res = so.linprog(np.zeros(nv), A_ub=Gi[b] if mdi else None, b_ub=hi[b] if mdi else None, A_eq=A[b] if neq else None, b_eq=bv[b] if neq else None, bounds=list(zip(np.where(np.isfinite(lb[b]), lb[b], None), np.where(np.isfinite(ub[b]), ub[b], None))))
print("linprog feasible:", res.status, res.message)
