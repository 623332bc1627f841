"""Parity of the HIP path on a real MI355X, through the C ABI (libpinkhip.so).

Small/medium batches are compared with the CPU oracle instance by instance;
BASELINE.json's full sizes are checked through size-independent properties
(KKT conditions against the GPU-stacked H, c; feasibility; determinism;
permutation equivariance) plus an oracle comparison on a random sample.
"""
import numpy as np
import pytest

from oracle import c_oracle
from tests import parity_suite as ps
from tests.cases import CONFIG_CASES, config_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["ur5", "draco3", "barrier", "equality", "safe"])
def test_golden_fixture(gpu_solver, golden, name):
    ps.golden(gpu_solver, golden, name)


@pytest.mark.parametrize("name,bounds,jac", CONFIG_CASES)
def test_baseline_configs(gpu_solver, name, bounds, jac):
    ps.config(gpu_solver, name, bounds, jac, B=4096 if name == "ur5" else 2048)


@pytest.mark.parametrize("nv", [1, 2, 5, 6, 7, 8, 9, 12, 13, 16, 17, 24, 25, 30, 31, 32, 33, 40, 41, 48, 56, 57, 64])
def test_every_padding_class(gpu_solver, nv):
    ps.random_dims(gpu_solver, nv, B=256, seed=100 + nv, root=min(2, nv - 1) if nv > 3 else 0)


@pytest.mark.parametrize("nv,md", [(6, 1), (12, 5), (30, 6), (31, 32), (50, 12), (12, 48), (30, 64), (64, 64)])
def test_dense_inequality_rows(gpu_solver, nv, md):
    ps.random_dims(gpu_solver, nv, B=256, seed=500 + nv, md=md)


def test_lm_damping_and_rank_deficient_tasks(gpu_solver):
    ps.random_dims(gpu_solver, 14, B=512, seed=7, lm=0.5, rank_deficient=True)


def test_no_diagonal_task(gpu_solver):
    ps.random_dims(gpu_solver, 6, B=512, seed=8, Kd_tasks=3, diag=False, lm=1.0)


def test_edge_cases(gpu_solver):
    ps.empty_task_list(gpu_solver)
    ps.fulfilled_tasks_give_zero(gpu_solver)
    ps.infeasible(gpu_solver)
    ps.infeasible_dense_rows(gpu_solver)
    ps.not_positive_definite(gpu_solver)
    ps.mixed_status_batch(gpu_solver)
    ps.max_iter_is_reported(gpu_solver)
    ps.batched_cost(gpu_solver)
    ps.many_dense_rows_chunked_staging(gpu_solver)
    ps.empty_batch(gpu_solver)
    ps.unconstrained(gpu_solver)


@pytest.mark.parametrize("nv,n_eq,md", [(6, 2, 0), (12, 3, 2), (30, 6, 3), (50, 4, 2)])
def test_equality_constraints(gpu_solver, nv, n_eq, md):
    ps.equality_constraints(gpu_solver, nv, n_eq, md, B=512, seed=900 + nv)
    ps.equality_edge_cases(gpu_solver)


def test_fuzz(gpu_solver):
    assert ps.fuzz(gpu_solver, range(7000, 7400)) > 1000


@pytest.mark.parametrize("kernel", ["sweep", "packed"])
def test_fuzz_weakly_regularised(gpu_solver, kernel, monkeypatch):
    """The regime where the explicitly updated inverse of the sweep tableau loses its accuracy (DESIGN.md 3.1 "What an
    explicit inverse cannot do"): its results are certified, handed over or routed to the Goldfarb-Idnani code, verdicts
    included.  ZERO uncertified draws: statuses equal the oracle's on every draw, and every instance is either within
    100 eps cond(H) max|x| of the oracle's dq or -- flat directions at cond(H) ~ 1e10 -- passes the KKT + objective
    certificate of parity_suite.certify_point (computed here, on the QP as stated); anything else fails the test.
    The range holds seed 415035 (off by 1.3e-6 on both kernels in round 3's wide fuzz) and seeds 520171 / 537045 of
    round 4's (a dense row active with a multiplier of -1e-8 after the closing corrections: stationary with it, 0.09 from
    the minimiser at cond(H) = 2e8 -- the certificate now holds the multipliers of active rows to their sign)."""
    monkeypatch.setenv("PINKHIP_SOLVER", kernel)
    del ps.CERTIFIED[:]
    n = ps.fuzz(gpu_solver, list(range(200001, 204001, 2)) + [415035, 520171, 537045], ill=True)
    assert n > 5000
    print(f"{kernel}: {n} feasible instances, {len(ps.CERTIFIED)} accepted on their KKT / objective certificate: {ps.CERTIFIED[:8]}")


def test_fuzz_wide(gpu_solver):
    """Eight thousand more draws (three seconds on the GPU), and the draw that showed what a dependence test on a
    subtractively updated curvature misses: seed 12979's fourth instance is infeasible, the last entering bound
    depends on the active set, and the tableau's diagonal entry was 6e-11 of its initial value -- pure round-off, on
    either side of a 1e-10 threshold depending on how the compiler contracts the FMAs (emulator: infeasible, GPU:
    "optimal" with an equality violated by 4e-3).  The kernel now forms small curvatures again as w^T H w."""
    assert ps.fuzz(gpu_solver, [12979]) >= 3
    assert ps.fuzz(gpu_solver, range(20000, 28000)) > 20000


def _kkt_batch(H, c, lb, ub, dq, Gd=None, hd=None):
    """Vectorised KKT check for box (+ dense) QPs; returns (stationarity, violation)."""
    g = np.einsum("bij,bj->bi", H, dq) + c
    scale = 1.0 + np.abs(c).max(axis=1, keepdims=True)
    tol = 1e-9
    with np.errstate(invalid="ignore"):  # (unbounded coordinates: inf - inf, masked by isfinite)
        at_lb = np.isfinite(lb) & (dq <= lb + tol * (1 + np.abs(lb)))
        at_ub = np.isfinite(ub) & (dq >= ub - tol * (1 + np.abs(ub)))
    viol = max(float(np.max(np.where(np.isfinite(lb), lb - dq, -1.0))), float(np.max(np.where(np.isfinite(ub), dq - ub, -1.0))))
    if Gd is not None and Gd.shape[1]:
        slack = hd - np.einsum("bmj,bj->bm", Gd, dq)
        viol = max(viol, float(-slack.min()))
        act = slack <= tol * (1 + np.abs(hd))
        # remove the dense-row part of the gradient by least squares on the active rows
        for b in np.nonzero(act.any(axis=1))[0]:
            A = Gd[b][act[b]].T
            free = ~(at_lb[b] | at_ub[b])
            if free.any():
                lam, *_ = np.linalg.lstsq(A[free], -g[b][free], rcond=None)
                g[b] = g[b] + A @ lam
    free = ~(at_lb | at_ub)
    stat = np.abs(np.where(free, g, 0.0) / scale).max()
    # multipliers: gradient must push outward at active bounds
    sign_ok = (np.where(at_lb & ~at_ub, g, 0.0) >= -1e-8 * scale).all() and (np.where(at_ub & ~at_lb, g, 0.0) <= 1e-8 * scale).all()
    return float(stat), float(viol), bool(sign_ok)


FULL_SIZE = [  # BASELINE.json configs 2-4 at their full batch size, plus the two other regimes the bench line carries
    ("draco3", 65536, dict(bounds="tight", jacobians="dense")),
    ("draco3b", 65536, dict(bounds="tight", jacobians="dense")),  # + two barriers: 36 tableau rows on 32 lanes (ik_sweepx.h)
    ("jvrc", 65536, dict(bounds="tight", jacobians="dense")),
    ("ur5", 4096, dict(bounds="tight", jacobians="dense")),
    ("draco3", 65536, dict(bounds="kinematic", jacobians="kinematic")),
    ("draco3", 65536, dict(bounds="kinematic", jacobians="kinematic", error_scale=0.02)),
]


@pytest.mark.parametrize("name,B,kw", FULL_SIZE, ids=lambda v: v if isinstance(v, str) else (str(v) if isinstance(v, int) else "-".join(f"{k}={x}" for k, x in v.items())))
def test_full_size_properties(gpu_solver, name, B, kw):
    """SURVEY.md 8(d) parity procedure on EVERY instance of the batch: dq against the C oracle (max over the batch,
    absolute and relative), status histogram, active-set agreement, KKT residuals of the GPU solution against the QP
    the pinned stacking builds; then determinism, permutation equivariance and host path == device path."""
    from oracle.parity_report import parity_report
    from pink_amd import synthetic

    s = gpu_solver
    terms = synthetic.make_terms(name, B, **kw)
    batch = synthetic.pack(terms)
    dev = s.upload(batch)
    s.solve_device(dev)
    s.stack_device(dev)
    s.sync()
    out = s.download(dev)
    H, c = s.download_stack(dev)
    assert (out.status == 0).all()
    rep = parity_report(lambda lo, hi: synthetic.pink_form(terms.slice(lo, hi)), batch, out.dq, out.status, H_gpu=H)
    assert rep["instances_compared"] == B and rep["status_mismatch"] == 0
    assert rep["max_abs_err"] <= 1e-10, rep  # north_star: 1e-8
    assert rep["max_H_err_rel"] <= 1e-12, rep
    assert rep["kkt_stationarity_max"] < 1e-9 and rep["kkt_violation_max"] < 1e-11 and rep["kkt_multiplier_sign_max"] < 1e-8, rep
    assert rep["active_set_equal_frac"] >= 0.999, rep  # (a bound met to within the activity tolerance may differ)
    # the same KKT check with the GPU-stacked (H, c)
    stat, viol, sign_ok = _kkt_batch(H, c, batch.lb, batch.ub, out.dq, batch.Gd, batch.hd)
    assert stat < 1e-9 and viol < 1e-11 and sign_ok
    # determinism: a second pass is bit-identical
    s.solve_device(dev)
    s.sync()
    out2 = s.download(dev)
    assert np.array_equal(out.dq, out2.dq) and np.array_equal(out.iters, out2.iters)
    dev.free()
    # permutation equivariance + host path == device path (bitwise)
    perm = np.random.default_rng(1).permutation(B)[:8192]
    sub_batch = batch.slice(0, B)
    for f in ("J", "e", "lb", "ub", "Gd", "hd"):
        setattr(sub_batch, f, np.ascontiguousarray(getattr(batch, f)[perm]))
    out3 = s.solve(sub_batch)
    assert np.array_equal(out3.dq, out.dq[perm])


@pytest.mark.parametrize("kw", [dict(bounds="tight", jacobians="dense"), dict(bounds="kinematic", jacobians="kinematic", error_scale=0.05)],
                         ids=["tight", "tracking"])
def test_full_size_weakly_regularised(gpu_solver, kw):
    """examples/humanoid_jvrc.py:69-81,112-114 as it is -- nv = 50, four FrameTasks, NO posture task, damping = 1e-12
     -- at B = 65 536: cond(H) ~ 1e13-1e14, so dq is determined to cond(H) eps only along the
    21 weighted directions' complement and the north-star 1e-8 on dq is not a property two correct solvers can share.
    Every instance is held to what IS determined: statuses equal to the oracle's, KKT residuals of the QP as stated,
    an objective not above the oracle's, and dq on the weighted rows (W J dq: what the tasks see).  The stack is rank
    deficient by construction (21 weighted rows on 50 coordinates, no LM term): the Goldfarb-Idnani kernel by dispatch."""
    from oracle.parity_report import parity_report
    from pink_amd import synthetic

    s, B = gpu_solver, 65536
    terms = synthetic.make_terms("jvrc_noposture", B, **kw)
    batch = synthetic.pack(terms)
    out = s.solve(batch)
    assert (out.status == 0).all()
    fr = out.path_fractions()
    assert fr["goldfarb_idnani"] == 1.0, fr
    rep = parity_report(lambda lo, hi: synthetic.pink_form(terms.slice(lo, hi)), batch, out.dq, out.status)
    assert rep["instances_compared"] == B and rep["status_mismatch"] == 0, rep
    # (cond(H) = 1e14: a bound is met to cond(H) eps |step| ~ 1e-10, the objective to 1e-11 of its value -- on the
    # emulator: stationarity 5e-10, violation 6e-11, objective within +-2e-11 of the oracle's, W J dq within 5e-10)
    assert rep["kkt_stationarity_max"] < 1e-8 and rep["kkt_violation_max"] < 1e-9 and rep["kkt_multiplier_sign_max"] < 1e-8, rep
    assert rep["objective_gap_rel_max"] <= 1e-9, rep
    # what the tasks see: the weighted task rows of the step, against the oracle's
    n = 4096
    ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms.slice(0, n)), nthreads=16)
    Jw = batch.J[:n] * batch.cost[None, :batch.Kd, None]
    d_task = np.abs(np.einsum("bkj,bj->bk", Jw, out.dq[:n] - ref["dq"])).max()
    assert d_task < 1e-8, d_task


def test_rccl_gather_single_rank(gpu_solver):
    """pinkhip_comm_*: the dq gather over RCCL, exercised with a one-rank communicator (the
    only size a 1-GPU box allows); multi-rank use follows the same calls with the id shipped
    to every process."""
    s = gpu_solver
    batch, _ = config_case("ur5", "tight", "dense", 256)
    dev = s.upload(batch)
    s.solve_device(dev)
    s.sync()
    out = s.download(dev)
    uid = s.comm_unique_id()
    assert len(uid) == 128
    s.comm_init(uid, 0, 1)
    try:
        n = 256 * 6
        recv = s._malloc(8 * n)
        s.comm_gather(dev.d_dq, recv, n, root=0)
        s.sync()
        got = np.zeros((256, 6))
        s._d2h(got, recv)
        assert np.array_equal(got, out.dq)
        s._free(recv)
        # byte gather / all-gather of the int32 status and iteration counts (what solve_sharded moves)
        for fn in (lambda d: s.comm_gather_bytes(dev.d_iters, d, 4 * 256, 0), lambda d: s.comm_allgather_bytes(dev.d_iters, d, 4 * 256)):
            recv = s._malloc(4 * 256)
            fn(recv)
            s.sync()
            it = np.zeros(256, np.int32)
            s._d2h(it, recv)
            assert np.array_equal(it, out.iters)
            s._free(recv)
        # RcclComm-shaped device gather helper used by pink_amd.sharding (one rank: the gather is a copy)
        from pink_amd.comm import HostRendezvous, RcclComm

        class OneRank(RcclComm):
            def __init__(self, solver):  # communicator already initialised above
                self.solver, self.rdzv, self.rank, self.world = solver, HostRendezvous(0, 1), 0, 1

        d = OneRank(s).gather_device(dev.d_dq, 8 * n, None)
        s.sync()
        got2 = np.zeros((256, 6))
        s.get(got2, d)
        assert np.array_equal(got2, out.dq)
        s.release(d)
    finally:
        s.comm_destroy()
        dev.free()


def test_pinned_host_path_equals_pageable(gpu_solver):
    """pinkhip_solve_host from page-locked buffers (chunked, overlapped H2D) returns bit for bit what the pageable
    single-copy path returns, across several chunks and with dense rows / equalities in the batch."""
    s = gpu_solver
    batch, _ = config_case("jvrc", "tight", "dense", 6000)  # ~13.8 kB per instance: three chunks of 32 MB
    ref = s.solve(batch)
    pb, pr = s.pin(batch), s.pinned_result(batch.B, batch.nv)
    out = s.solve(pb, out=pr)
    assert out.dq is pr.dq and np.array_equal(pr.dq, ref.dq) and np.array_equal(pr.status, ref.status) and np.array_equal(pr.iters, ref.iters)
    assert (ref.status == 0).all()


def test_api_errors(gpu_solver):
    from pink_amd._lib import PinkHipError

    batch, _ = config_case("ur5", "tight", "dense", 4)
    batch.dt = 0.0
    with pytest.raises(PinkHipError):
        gpu_solver.solve(batch)


def test_kkt_certificate_independent_of_the_oracle_solver(gpu_solver):
    assert ps.kkt_certificate(gpu_solver, range(9000, 9400)) > 500


def test_small_stack_packing(gpu_solver):
    ps.small_stack_packing(gpu_solver)
