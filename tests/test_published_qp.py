"""Known-answer vectors published for the reference's QP stack (quadprog behind qpsolvers).

The reference delegates the solve to ``qpsolvers.solve_problem(..., solver="quadprog")``
(``pink/solve_ik.py:270``); neither package is present in /root/reference or installable here, so the
QP half of the oracle is pinned to the answers they publish:

* the worked example of Goldfarb & Idnani (1983) that quadprog documents for ``solve.QP``
  (min 1/2 x'Dx - d'x, A'x >= b): solution, multipliers and the order of activation;
* the example of the qpsolvers README (inequalities + one equality), solved there with quadprog.

The oracle's two independent Goldfarb-Idnani implementations, an independent SciPy solve (SLSQP) and the
HIP kernel (emulator here, MI355X when marked gpu) must all reproduce them.
"""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import c_oracle
from oracle import pink_oracle as po
from pink_amd.batch import DenseTaskTerm, pack_terms

# Goldfarb & Idnani example / quadprog `solve.QP` documentation
GI_D = np.eye(3)
GI_d = np.array([0.0, 5.0, 0.0])
GI_A = np.array([[-4.0, 2.0, 0.0], [-3.0, 1.0, -2.0], [0.0, 0.0, 1.0]])  # columns are the constraints
GI_b = np.array([-8.0, 2.0, 0.0])
GI_X = np.array([0.4761905, 1.0476190, 2.0952381])
GI_LAGRANGIAN = np.array([0.0, 0.2380952, 2.0952381])
GI_VALUE = -2.380952

# qpsolvers README example
M = np.array([[1.0, 2.0, 0.0], [-8.0, 3.0, 2.0], [0.0, 1.0, 1.0]])
QS_P, QS_q = M.T @ M, np.array([3.0, 2.0, 3.0]) @ M
QS_G = np.array([[1.0, 2.0, 1.0], [2.0, 0.0, 1.0], [-1.0, 2.0, -1.0]])
QS_h = np.array([3.0, 2.0, -2.0])
QS_A, QS_b = np.array([[1.0, 1.0, 1.0]]), np.array([1.0])
QS_X = np.array([0.30769231, -0.69230769, 1.38461538])


def _slsqp(P, q, G, h, A=None, b=None):
    cons = [{"type": "ineq", "fun": lambda x: h - G @ x, "jac": lambda x: -G}]
    if A is not None:
        cons.append({"type": "eq", "fun": lambda x: A @ x - b, "jac": lambda x: A})
    r = minimize(lambda x: 0.5 * x @ P @ x + q @ x, np.zeros(len(q)), jac=lambda x: P @ x + q, constraints=cons,
                 method="SLSQP", options={"ftol": 1e-12, "maxiter": 200})
    return r.x  # (SLSQP may stop on its line-search criterion at this tolerance; the iterate is what is compared)


def test_published_vectors_are_consistent_with_an_independent_solver():
    x = _slsqp(GI_D, -GI_d, -GI_A.T, -GI_b)
    assert np.abs(x - GI_X).max() < 5e-7 and abs(0.5 * x @ x - GI_d @ x - GI_VALUE) < 1e-6
    x = _slsqp(QS_P, QS_q, QS_G, QS_h, QS_A, QS_b)
    assert np.abs(x - QS_X).max() < 5e-8


def test_oracle_reproduces_the_goldfarb_idnani_example():
    G, h = -GI_A.T, -GI_b
    x, st, _, lam = c_oracle.gi_solve(GI_D, -GI_d, G, h)
    r = po.goldfarb_idnani(GI_D, -GI_d, G, h)
    for xs, ls in ((x, lam), (r.x, r.multipliers)):
        assert np.abs(xs - GI_X).max() < 5e-8
        assert np.abs(ls - GI_LAGRANGIAN).max() < 5e-8  # constraints 3 and 2 are the active ones
    assert st == 0 and r.found and sorted(r.active) == [1, 2]
    stat, viol, _ = po.kkt_residuals(GI_D, -GI_d, G, h, GI_X, active_tol=1e-6)  # 7 published digits
    assert stat < 1e-6 and viol < 1e-6


def test_oracle_reproduces_the_qpsolvers_readme_example():
    G = np.vstack([QS_A, QS_G])  # equality first (meq = 1), as pink/solve_ik.py:140-149 passes A, b
    h = np.concatenate([QS_b, QS_h])
    x, st, _, _ = c_oracle.gi_solve(QS_P, QS_q, G, h, meq=1)
    assert st == 0 and np.abs(x - QS_X).max() < 5e-9


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def solver(request):
    return request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")


def test_kernel_reproduces_both_examples(solver):
    # H = J^T J, c = J^T e with gain 1: (J, e) = (I, -d) and (M, [3, 2, 3]); damping 0
    t = DenseTaskTerm(J=np.eye(3)[None], e=-GI_d[None], cost=None, gain=1.0, lm_damping=0.0)
    batch = pack_terms(3, [t], dt=1.0, damping=0.0, dense_rows=[((-GI_A.T)[None], (-GI_b)[None])])
    out = solver.solve(batch)
    assert out.status[0] == 0 and np.abs(out.dq[0] - GI_X).max() < 5e-8
    H, c = solver.stack(batch)
    assert np.allclose(H[0], GI_D) and np.allclose(c[0], -GI_d)

    t = DenseTaskTerm(J=M[None], e=np.array([3.0, 2.0, 3.0])[None], cost=None, gain=1.0, lm_damping=0.0)
    batch = pack_terms(3, [t], dt=1.0, damping=0.0, dense_rows=[(QS_G[None], QS_h[None])],
                       equality_rows=[(QS_A[None], QS_b[None])])
    out = solver.solve(batch)
    assert out.status[0] == 0 and np.abs(out.dq[0] - QS_X).max() < 5e-9
    H, c = solver.stack(batch)
    assert np.allclose(H[0], QS_P) and np.allclose(c[0], QS_q)
