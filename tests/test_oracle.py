"""The CPU oracle against the reference's known-answer tests and independent solvers.

Each test names the reference test it restates (paths relative to the reference
checkout).  No GPU, no HIP library involved.
"""
import numpy as np
import pytest
from scipy.optimize import lsq_linear

from oracle import c_oracle
from oracle import pink_oracle as po
from tests.cases import config_case


def _rand_task(rng, k=6, nv=9):
    return rng.normal(size=(k, nv)), rng.normal(size=k)


def test_unit_cost_objective_is_JtJ_and_etJ():
    """tests/test_frame_task.py:123-141: unit cost, lm=0 => H == J^T J, c == e^T J."""
    J, e = _rand_task(np.random.default_rng(0))
    for cost in (None, 1.0, np.ones(6)):
        H, c = po.task_objective(J, e, cost, 1.0, 0.0)
        assert np.allclose(H, J.T @ J, rtol=1e-14, atol=1e-14)
        assert np.allclose(c, e @ J, rtol=1e-14, atol=1e-14)


def test_zero_cost_is_row_deletion():
    """tests/test_frame_task.py:143-181."""
    rng = np.random.default_rng(1)
    J, e = _rand_task(rng)
    for keep in (slice(0, 3), slice(3, 6), slice(1, 2), slice(5, 6)):
        cost = np.zeros(6)
        cost[keep] = 1.0
        H, c = po.task_objective(J, e, cost, 1.0, 0.0)
        assert np.allclose(H, J[keep].T @ J[keep], atol=1e-14)
        assert np.allclose(c, e[keep] @ J[keep], atol=1e-14)


def test_weight_enters_squared_and_gain_scales_c():
    """pink/tasks/task.py:158-166: H = J^T W^2 J, c = gain J^T W^2 e, mu = lm gain^2 |W e|^2."""
    rng = np.random.default_rng(2)
    J, e = _rand_task(rng)
    w = rng.uniform(0.5, 2.0, size=6)
    H, c = po.task_objective(J, e, w, 0.7, 0.3)
    mu = 0.3 * 0.49 * np.sum((w * e) ** 2)
    assert np.allclose(H, J.T @ np.diag(w**2) @ J + mu * np.eye(9), rtol=1e-13)
    assert np.allclose(c, 0.7 * J.T @ (w**2 * e), rtol=1e-13)


def test_lm_damping_inert_at_zero_error_active_otherwise():
    """tests/test_frame_task.py:183-215."""
    rng = np.random.default_rng(3)
    J, e = _rand_task(rng)
    H1, c1 = po.task_objective(J, 0 * e, 1.0, 1.0, 1e-8)
    H2, c2 = po.task_objective(J, 0 * e, 1.0, 1.0, 1e-4)
    assert np.array_equal(H1, H2) and np.array_equal(c1, c2)
    H1, c1 = po.task_objective(J, e, 1.0, 1.0, 1e-8)
    H2, c2 = po.task_objective(J, e, 1.0, 1.0, 1e-1)
    x1 = po.goldfarb_idnani(H1 + 1e-9 * np.eye(9), c1).x
    x2 = po.goldfarb_idnani(H2 + 1e-9 * np.eye(9), c2).x
    assert np.linalg.norm(x2) < np.linalg.norm(x1)  # it is a damping
    assert np.linalg.norm(x2 - x1) > 1e-6


def test_low_acceleration_golden():
    """tests/test_low_acceleration_task.py:34-42: J = I, e = -dt v_prev, cost 1 =>
    H = I, c = -v_prev dt."""
    v_prev = np.array([1.0, 2.0, 3.0, 4.0, -3.0, -2.0])
    dt = 1.234e-2
    H, c = po.task_objective(np.eye(6), -v_prev * dt, 1.0, 1.0, 0.0)
    assert np.linalg.norm(H - np.eye(6)) < 1e-10
    assert np.linalg.norm(c + v_prev * dt) < 1e-10


def test_damping_task_golden():
    """tests/test_damping_task.py:34-39: J = I, e = 0 => H = I, c = 0."""
    H, c = po.task_objective(np.eye(7), np.zeros(7), 1.0, 1.0, 0.0)
    assert np.array_equal(H, np.eye(7)) and not c.any()


def test_no_task_gives_zero_velocity_and_no_rows():
    """tests/test_solve_ik.py:67-87."""
    P, q, G, h = po.build_qp(5, [], 1e-12)
    assert G is None and h is None
    assert np.array_equal(P, 1e-12 * np.eye(5)) and not q.any()
    assert np.array_equal(po.goldfarb_idnani(P, q).x, np.zeros(5))


def test_limit_rows_shapes_and_values():
    """tests/test_limits.py:22-33, tests/test_velocity_limit.py:46-55,
    tests/test_configuration_limit.py:75-99."""
    nv, dt = 6, 1e-3
    q = np.zeros(nv)
    q_min, q_max = -np.ones(nv), 2 * np.ones(nv)
    q_max[2] = 1e30  # unbounded joint drops out (configuration_limit.py:50-56)
    idx = po.configuration_limit_indices(q_min, q_max)
    assert idx.tolist() == [0, 1, 3, 4, 5]
    G, h = po.configuration_limit_rows(q, q_min, q_max, idx, nv)
    assert G.shape == (10, nv) and h.shape == (10,)
    assert np.allclose(h[:5], 0.5 * 2.0) and np.allclose(h[5:], 0.5 * 1.0)
    v_max = np.full(nv, 2.0)
    v_max[0] = 0.0  # below 1e-10 -> not limited (velocity_limit.py:61-64)
    vi = po.velocity_limit_indices(v_max)
    G, h = po.velocity_limit_rows(v_max, vi, nv, dt)
    assert G.shape == (10, nv) and np.allclose(h, dt * 2.0)
    assert po.velocity_limit_rows(v_max, np.zeros(0, int), nv, dt) is None
    assert po.qp_inequalities([None, None]) == (None, None)


def test_barrier_rows_and_objective():
    """pink/barriers/barrier.py:193-201,246-254; tests/test_barrier.py:34-45 (shapes)."""
    rng = np.random.default_rng(4)
    Jh = rng.normal(size=(3, 8))
    hv = rng.uniform(0, 1, size=3)
    G, h = po.barrier_rows(Jh, hv, 100.0, 0.01)
    assert G.shape == (3, 8) and np.allclose(G, -Jh / 0.01) and np.allclose(h, 100 * hv)
    H, c = po.barrier_objective(Jh, 1.0, 8)
    assert np.allclose(H, np.eye(8) / np.linalg.norm(Jh) ** 2) and not c.any()
    H, c = po.barrier_objective(Jh, 1e-7, 8)  # below the 1e-6 threshold: disabled
    assert not H.any()


@pytest.mark.parametrize("name,bounds,jac", [("ur5", "tight", "dense"), ("draco3", "tight", "dense"),
                                             ("draco3", "kinematic", "kinematic"), ("jvrc", "tight", "dense")])
def test_numpy_and_c_goldfarb_idnani_agree_with_kkt(name, bounds, jac):
    _, pf = config_case(name, bounds, jac, 6)
    out = c_oracle.solve_ik_batch(**pf, want_Hc=True)
    assert (out["status"] == 0).all()
    for b in range(6):
        P, q = out["H"][b], out["c"][b]
        res = po.goldfarb_idnani(P, q, pf["G"][b], pf["h"][b])
        assert res.found
        assert np.abs(res.x - out["dq"][b]).max() < 1e-11
        stat, viol, lam = po.kkt_residuals(P, q, pf["G"][b], pf["h"][b], out["dq"][b])
        assert stat < 1e-10 * max(1.0, np.abs(q).max()) and viol < 1e-12 and (lam >= 0).all()


def test_bvls_cross_check_on_box_problems():
    """Independent solver: scipy BVLS on the Cholesky-transformed least squares."""
    batch, pf = config_case("draco3", "tight", "dense", 12)
    out = c_oracle.solve_ik_batch(**pf, want_Hc=True)
    for b in range(12):
        L = np.linalg.cholesky(out["H"][b])
        r = lsq_linear(L.T, -np.linalg.solve(L, out["c"][b]), bounds=(batch.lb[b], batch.ub[b]),
                       method="bvls", tol=1e-15, max_iter=500)
        assert np.abs(r.x - out["dq"][b]).max() < 1e-11


def test_infeasible_and_not_pd_are_reported():
    P = np.eye(2)
    q = np.zeros(2)
    G = np.array([[1.0, 0.0], [-1.0, 0.0]])
    h = np.array([-1.0, -1.0])  # x0 <= -1 and x0 >= 1
    assert po.goldfarb_idnani(P, q, G, h).status == po.STATUS_INFEASIBLE
    assert c_oracle.gi_solve(P, q, G, h)[1] == po.STATUS_INFEASIBLE
    Pbad = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert po.goldfarb_idnani(Pbad, q).status == po.STATUS_NOT_PD
    assert c_oracle.gi_solve(Pbad, q)[1] == po.STATUS_NOT_PD


def test_duplicate_rows_do_not_change_the_minimiser():
    """Pink emits [P;-P] twice (configuration + velocity limit); merging them into
    one box keeps the feasible set, hence the minimiser."""
    batch, pf = config_case("ur5", "tight", "dense", 8)
    out = c_oracle.solve_ik_batch(**pf, want_Hc=True)
    nv = batch.nv
    eye = np.eye(nv)
    for b in range(8):
        G = np.vstack([eye, -eye])
        h = np.hstack([batch.ub[b], -batch.lb[b]])
        x, st, _, _ = c_oracle.gi_solve(out["H"][b], out["c"][b], G, h)
        assert st == 0 and np.abs(x - out["dq"][b]).max() < 1e-13
