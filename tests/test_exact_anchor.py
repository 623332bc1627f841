"""The QP half against EXACT arithmetic (oracle/exact_qp.py: the KKT system of the final active set in 50-digit arithmetic,
P and q formed in the same arithmetic from the task rows as pink/tasks/task.py:145-167 states them).

quadprog -- what pink/solve_ik.py:270 reaches through qpsolvers -- cannot be installed here; but the QP is strictly convex,
so ANY correct solver returns its one minimiser up to round-off, and the minimiser itself can be computed to 35 digits.
The HIP kernels (emulator here, MI355X under -m gpu) and the fp64 oracle are held to it: within 1e-10 on BASELINE's
configurations, and -- where H is weakly regularised (examples/humanoid_jvrc.py:69-81 as shipped: cond(H) ~ 1e13) and no
fp64 solver can promise 1e-8 -- not farther from the exact minimiser than ten times the fp64 oracle is."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle.exact_qp import anchor_report, exact_minimiser
from pink_amd import synthetic
from tests.test_published_qp import GI_A, GI_D, GI_LAGRANGIAN, GI_X, GI_b, GI_d, M, QS_A, QS_G, QS_X, QS_b, QS_h


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def solver(request):
    return request.param, request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")


def test_exact_minimiser_reproduces_the_published_examples():
    """Goldfarb & Idnani's worked example as quadprog documents it (solution AND multipliers) and the qpsolvers README
    example (one equality): the exact solver started from a WRONG active set (the origin) finds them."""
    one = (np.ones(3), [1.0], [0.0], [0, 3])
    x, info = exact_minimiser(np.eye(3), -GI_d, *one, 0.0, -GI_A.T, -GI_b, np.zeros(3))
    assert np.abs(x - GI_X).max() < 5e-8 and info["active"] == [1, 2]
    assert np.abs(np.array(info["multipliers"]) - GI_LAGRANGIAN[1:]).max() < 5e-8
    x, info = exact_minimiser(M, np.array([3.0, 2.0, 3.0]), *one, 0.0, np.vstack([QS_A, QS_G]), np.concatenate([QS_b, QS_h]), np.zeros(3), meq=1)
    assert np.abs(x - QS_X).max() < 5e-9 and info["active"][0] == 0


@pytest.mark.parametrize("name,kw", [("ur5", dict(bounds="tight")), ("draco3", dict(bounds="tight")), ("draco3", dict(bounds="kinematic", jacobians="kinematic")),
                                     ("draco3b", dict(bounds="tight")), ("jvrc", dict(bounds="tight"))],
                         ids=["ur5", "draco3-tight", "draco3-kinematic", "draco3-barriers", "jvrc-barriers"])
def test_baseline_configurations_against_the_exact_minimiser(solver, name, kw):
    where, s = solver
    B, n = (12, 3) if where == "emu" else (4096, 12)
    terms = synthetic.make_terms(name, B, **kw)
    ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms))
    out = s.solve(synthetic.pack(terms))
    assert (out.status == 0).all() and (ref["status"] == 0).all()
    rep = anchor_report(lambda lo, hi: synthetic.pink_form(terms.slice(lo, hi)), terms.damping, out.dq, ref["dq"], n, n)
    assert rep["instances"] >= n and rep["active_set_changes_from_the_guess"] == 0, rep
    assert rep["max_abs_err_vs_exact"] < 1e-10 and rep["oracle_max_abs_err_vs_exact"] < 1e-10, rep  # (north_star: 1e-8)


@pytest.mark.parametrize("kw", [dict(bounds="tight", jacobians="dense"), dict(bounds="kinematic", jacobians="kinematic", error_scale=0.05)],
                         ids=["tight", "tracking"])
def test_weakly_regularised_against_the_exact_minimiser(solver, kw):
    """examples/humanoid_jvrc.py:69-81,112-114 as shipped (no posture task, damping 1e-12): the instances where the
    kernel and the fp64 oracle differ MOST, and the first ones, against the exact minimiser.  The kernel must be within
    the contract's 1e-8 or, where fp64 cannot deliver that, not farther than ten times the oracle's own distance."""
    where, s = solver
    B, n = (16, 4) if where == "emu" else (65536, 32)
    terms = synthetic.make_terms("jvrc_noposture", B, **kw)
    out = s.solve(synthetic.pack(terms))
    dq_ref = np.empty_like(out.dq)
    st_ref = np.empty(B, np.int32)
    for lo in range(0, B, 8192):  # (the dense G of Pink's form is 70 kB per instance)
        r = c_oracle.solve_ik_batch(**synthetic.pink_form(terms.slice(lo, min(lo + 8192, B))), nthreads=16)
        dq_ref[lo:lo + 8192], st_ref[lo:lo + 8192] = r["dq"], r["status"]
    assert (out.status == 0).all() and (st_ref == 0).all()
    rep = anchor_report(lambda lo, hi: synthetic.pink_form(terms.slice(lo, hi)), terms.damping, out.dq, dq_ref, n, n)
    print(where, kw["bounds"], {k: v for k, v in rep.items() if k != "per_instance"})
    for b, eg, eo in rep["per_instance"]:
        assert eg <= max(1e-8, 10.0 * eo), (b, eg, eo)
