"""Parity checks of the HIP kernels against the CPU oracle.

The same functions are run on the CPU wave emulator (``test_emulator_parity.py``,
small batches, no GPU) and on the MI355X through the C ABI
(``test_gpu_parity.py``).  ``solver`` is anything with ``solve(batch)`` and
``stack(batch)``.

Tolerance: north_star demands |dq - dq_ref| <= 1e-8 (fp64); the suite holds the
kernels to 1e-10 absolute (what is observed is ~1e-14).
"""
import numpy as np

from oracle import c_oracle
from oracle import pink_oracle as po
from pink_amd.batch import DenseTaskTerm, DiagonalTaskTerm, pack_terms
from tests.cases import config_case, golden_case, golden_equalities, random_case

TOL_DQ = 1e-10


def check_against_oracle(solver, batch, pf, tol=TOL_DQ, expect_status=None):
    ref = c_oracle.solve_ik_batch(**pf, want_Hc=True, nthreads=0)
    H, c = solver.stack(batch)
    scale = max(1.0, np.abs(ref["H"]).max())
    assert np.abs(H - ref["H"]).max() <= 1e-13 * scale
    assert np.abs(c - ref["c"]).max() <= 1e-13 * max(1.0, np.abs(ref["c"]).max())
    out = solver.solve(batch)
    if expect_status is None:
        assert (ref["status"] == 0).all()
        assert (out.status == 0).all(), out.status[out.status != 0][:10]
        # fp64 round-off grows with cond(H): two correct solvers agree to ~cond*eps*|x|
        cond = np.linalg.cond(ref["H"])
        xmax = np.abs(ref["dq"]).max(axis=1)
        tol_b = np.maximum(tol, 100 * np.finfo(float).eps * cond * xmax)
        err = np.abs(out.dq - ref["dq"]).max(axis=1)
        assert (err <= tol_b).all(), (err.max(), cond.max())
        assert err.max() <= 1e-8, err.max()  # north_star bound, unconditionally
    else:
        assert (out.status == expect_status).all(), out.status
        assert (ref["status"] == expect_status).all(), ref["status"]
    return out, ref


def golden(solver, golden_npz, name):
    """Fixture produced by the reference's own build_ik: H, c vs its (P, q); dq vs
    the oracle solving the reference's own (P, q, G, h)."""
    batch, P, q, G, h = golden_case(golden_npz, name)
    A, b = golden_equalities(golden_npz, name)
    H, c = solver.stack(batch)
    assert np.allclose(H[0], P, rtol=1e-13, atol=1e-15)
    assert np.allclose(c[0], q, rtol=1e-13, atol=1e-15)
    out = solver.solve(batch)
    if A is None:
        x, st, _, _ = c_oracle.gi_solve(P, q, G, h)
    else:  # the reference's own (A, b) lead the rows as equalities (quadprog's meq)
        assert batch.n_eq == len(b)
        assert np.array_equal(batch.Gd[0, :batch.n_eq], A) and np.array_equal(batch.hd[0, :batch.n_eq], b)
        x, st, _, _ = c_oracle.gi_solve(P, q, np.vstack([A, G]), np.hstack([b, h]), meq=len(b))
        assert np.abs(A @ out.dq[0] - b).max() < 1e-12
    assert st == 0 and out.status[0] == 0
    assert np.abs(out.dq[0] - x).max() <= TOL_DQ
    stat, viol, _ = po.kkt_residuals(P, q, G, h, out.dq[0], A=A, b=b)
    assert stat < 1e-10 and viol < 1e-11


def config(solver, name, bounds, jac, B):
    batch, pf = config_case(name, bounds, jac, B)
    return check_against_oracle(solver, batch, pf)


def random_dims(solver, nv, B, seed, **kw):
    batch, pf = random_case(nv, B, seed, **kw)
    return check_against_oracle(solver, batch, pf)


def empty_task_list(solver, nv=7, B=3):
    """tests/test_solve_ik.py:79-87: no task => dq = 0 (H = damping I, c = 0)."""
    lb = -np.ones((B, nv))
    ub = np.ones((B, nv))
    batch = pack_terms(nv, [], 0.01, 1e-12, boxes=[(lb, ub)], batch_size=B)
    H, c = solver.stack(batch)
    assert np.array_equal(H, np.broadcast_to(1e-12 * np.eye(nv), (B, nv, nv))) and not c.any()
    out = solver.solve(batch)
    assert (out.status == 0).all() and not out.dq.any()


def fulfilled_tasks_give_zero(solver, nv=10, B=4):
    """tests/test_solve_ik.py:89-102: zero error => dq = 0 whatever the bounds (0 inside)."""
    rng = np.random.default_rng(5)
    J = rng.normal(size=(B, 6, nv))
    t = [DenseTaskTerm(J=J, e=np.zeros((B, 6)), cost=1.0), DiagonalTaskTerm(col0=0, e=np.zeros((B, nv)), cost=0.1)]
    batch = pack_terms(nv, t, 0.01, 1e-12, boxes=[(-np.ones((B, nv)), np.ones((B, nv)))], batch_size=B)
    out = solver.solve(batch)
    assert (out.status == 0).all() and np.abs(out.dq).max() < 1e-15


def infeasible(solver):
    """Crossed bounds: quadprog's "constraints are inconsistent" -> status 2 (NoSolutionFound)."""
    nv, B = 5, 2
    t = [DiagonalTaskTerm(col0=0, e=np.ones((B, nv)), cost=1.0)]
    lb = np.full((B, nv), -1.0)
    ub = np.full((B, nv), 1.0)
    lb[:, 2], ub[:, 2] = 0.5, 0.25
    batch = pack_terms(nv, t, 0.01, 1e-12, boxes=[(lb, ub)], batch_size=B)
    out = solver.solve(batch)
    assert (out.status == 2).all()


def infeasible_dense_rows(solver):
    nv, B = 6, 2
    t = [DiagonalTaskTerm(col0=0, e=np.ones((B, nv)), cost=1.0)]
    g = np.ones((B, 1, nv))
    G = np.concatenate([g, -g], axis=1)  # sum x <= -1 and -sum x <= -1
    h = -np.ones((B, 2))
    batch = pack_terms(nv, t, 0.01, 1e-12, dense_rows=[(G, h)], batch_size=B)
    out = solver.solve(batch)
    assert (out.status == 2).all()


def not_positive_definite(solver):
    """No task, zero damping: H = 0 -> quadprog's "matrix G is not positive definite" -> status 3."""
    nv, B = 4, 2
    batch = pack_terms(nv, [], 0.01, 0.0, boxes=[(-np.ones((B, nv)), np.ones((B, nv)))], batch_size=B)
    out = solver.solve(batch)
    assert (out.status == 3).all()


def mixed_status_batch(solver):
    """One bad instance must not disturb its neighbours (per-instance status)."""
    batch, pf = random_case(12, 6, 77, md=0)
    batch.lb[3, 4], batch.ub[3, 4] = 1.0, -1.0
    out = solver.solve(batch)
    ref = c_oracle.solve_ik_batch(**pf)
    good = np.array([0, 1, 2, 4, 5])
    assert out.status[3] == 2 and (out.status[good] == 0).all()
    assert np.abs(out.dq[good] - ref["dq"][good]).max() <= TOL_DQ


def max_iter_is_reported(solver):
    batch, _ = config_case("draco3", "tight", "dense", 2)
    out = solver.solve(batch, max_iter=3)
    assert (out.status == 1).all() and (out.iters == 4).all()


def batched_cost(solver):
    """cost given per instance ([B, K]) equals solving each instance with its own cost."""
    rng = np.random.default_rng(9)
    nv, B = 9, 4
    J = rng.normal(size=(B, 6, nv))
    e = 0.1 * rng.normal(size=(B, 6))
    ep = rng.uniform(-0.3, 0.3, size=(B, nv))
    cost = rng.uniform(0.5, 2.0, size=(B, 6))
    lb, ub = -0.02 * np.ones((B, nv)), 0.02 * np.ones((B, nv))
    tb = [DenseTaskTerm(J=J, e=e, cost=cost, lm_damping=0.1), DiagonalTaskTerm(col0=0, e=ep, cost=0.3)]
    out = solver.solve(pack_terms(nv, tb, 0.01, 1e-12, boxes=[(lb, ub)], batch_size=B))
    for b in range(B):
        t1 = [DenseTaskTerm(J=J[b:b + 1], e=e[b:b + 1], cost=cost[b], lm_damping=0.1),
              DiagonalTaskTerm(col0=0, e=ep[b:b + 1], cost=0.3)]
        o1 = solver.solve(pack_terms(nv, t1, 0.01, 1e-12, boxes=[(lb[b:b + 1], ub[b:b + 1])], batch_size=1))
        assert np.array_equal(o1.dq[0], out.dq[b])


def many_dense_rows_chunked_staging(solver):
    """Kd larger than one LDS staging chunk (rows are streamed in pieces)."""
    nv, B = 6, 3
    rng = np.random.default_rng(11)
    tasks, Js, es, rows = [], [], [], [0]
    for _ in range(20):  # 120 rows of nv=6: staging chunk is 13 rows at NV=8
        J = rng.normal(size=(B, 6, nv))
        e = 0.05 * rng.normal(size=(B, 6))
        tasks.append(DenseTaskTerm(J=J, e=e, cost=1.0))
        Js.append(J), es.append(e), rows.append(rows[-1] + 6)
    lb, ub = -0.01 * np.ones((B, nv)), 0.01 * np.ones((B, nv))
    batch = pack_terms(nv, tasks, 0.01, 1e-12, boxes=[(lb, ub)], batch_size=B)
    eye = np.eye(nv)
    G = np.broadcast_to(np.vstack([eye, -eye]), (B, 2 * nv, nv))
    h = np.concatenate([ub, -lb], axis=1)
    pf = dict(J=np.concatenate(Js, axis=1), e=np.concatenate(es, axis=1), cost=np.ones(120), gain=np.ones(20),
              lm=np.zeros(20), rows=np.array(rows, np.int32), damping=1e-12, G=np.ascontiguousarray(G), h=h)
    check_against_oracle(solver, batch, pf)


def empty_batch(solver):
    batch, _ = config_case("ur5", "tight", "dense", 4)
    out = solver.solve(batch.slice(0, 0))
    assert out.dq.shape == (0, 6) and out.status.shape == (0,)


def unconstrained(solver):
    """limits=[] (solve_ik.py:187-189): no rows at all -> dq = -H^-1 c."""
    batch, pf = random_case(15, 5, 3)
    batch.lb[:] = -np.inf
    batch.ub[:] = np.inf
    ref = c_oracle.solve_ik_batch(**{**pf, "G": None, "h": None}, want_Hc=True)
    out = solver.solve(batch)
    assert (out.status == 0).all() and (out.iters == 0).all()
    assert np.abs(out.dq - ref["dq"]).max() <= TOL_DQ
    x = -np.linalg.solve(ref["H"], ref["c"][..., None])[..., 0]
    assert np.abs(out.dq - x).max() <= 1e-9


def equality_constraints(solver, nv, n_eq, md_ineq, B, seed):
    """constraints= (pink/solve_ik.py:125-149): A dq = b rows are the leading dense rows."""
    rng = np.random.default_rng(seed)
    J = rng.normal(0, 0.5, size=(B, 6, nv))
    e = 0.1 * rng.normal(size=(B, 6))
    ep = rng.uniform(-0.5, 0.5, size=(B, nv))
    A = rng.normal(size=(B, n_eq, nv))
    bvec = 0.01 * rng.normal(size=(B, n_eq))
    lb = -rng.uniform(0.01, 0.05, size=(B, nv))
    ub = rng.uniform(0.01, 0.05, size=(B, nv))
    Gi = rng.normal(size=(B, md_ineq, nv))
    hi = rng.uniform(0.0, 0.05, size=(B, md_ineq))
    batch = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=1.0), DiagonalTaskTerm(col0=0, e=ep, cost=0.1)], 0.005, 1e-12,
                       boxes=[(lb, ub)], dense_rows=[(Gi, hi)] if md_ineq else (), equality_rows=[(A, bvec)], batch_size=B)
    assert batch.n_eq == n_eq and batch.md == n_eq + md_ineq
    out = solver.solve(batch)
    eye = np.eye(nv)
    G = np.concatenate([A, np.broadcast_to(eye, (B, nv, nv)), np.broadcast_to(-eye, (B, nv, nv)), Gi], axis=1)
    h = np.concatenate([bvec, ub, -lb, hi], axis=1)
    ref = c_oracle.solve_ik_batch(np.concatenate([J, np.broadcast_to(eye, (B, nv, nv))], axis=1),
                                  np.concatenate([e, ep], axis=1), np.concatenate([np.ones(6), np.full(nv, 0.1)]),
                                  np.ones(2), np.zeros(2), np.array([0, 6, 6 + nv], np.int32), 1e-12, G, h, meq=n_eq,
                                  nthreads=0)
    ok = ref["status"] == 0
    assert (out.status == ref["status"]).all()
    assert np.abs(out.dq[ok] - ref["dq"][ok]).max() <= TOL_DQ
    assert np.abs(np.einsum("bmj,bj->bm", A, out.dq)[ok] - bvec[ok]).max() < 1e-12


def equality_edge_cases(solver):
    """A duplicated (consistent) equality is skipped; an inconsistent one reports status 2."""
    nv, B = 5, 2
    t = [DiagonalTaskTerm(col0=0, e=np.ones((B, nv)), cost=1.0)]
    A = np.zeros((B, 2, nv))
    A[:, 0, :2] = 1.0
    A[:, 1, :2] = 2.0
    box = [(-np.ones((B, nv)), np.ones((B, nv)))]
    ok = solver.solve(pack_terms(nv, t, 0.01, 1e-12, boxes=box, equality_rows=[(A, np.tile([0.2, 0.4], (B, 1)))], batch_size=B))
    assert (ok.status == 0).all() and np.abs(ok.dq[:, 0] + ok.dq[:, 1] - 0.2).max() < 1e-13
    bad = solver.solve(pack_terms(nv, t, 0.01, 1e-12, boxes=box, equality_rows=[(A, np.tile([0.2, 0.5], (B, 1)))], batch_size=B))
    assert (bad.status == 2).all()


def fuzz(solver, seeds, nv_lo=1, nv_hi=34, md_hi=5, ill=False, free_lead=0):
    """Random mixes of box bounds (some missing, some with lb == ub), dense inequality rows (some
    duplicated), equalities, LM damping and dimensions; infeasible draws must be reported as such
    by both sides.  ``free_lead``: the first so many coordinates carry no bound and the draw has no rows (with 33 / 34
    coordinates: the instantiation that eliminates the leading coordinates, round 6) -- applied after every draw of the
    seed, so that the seeds of the other shapes keep their problems."""
    n_checked = 0
    n_refuted0 = len(REFUTED)
    for sd in seeds:
        rng = np.random.default_rng(sd)
        nv = int(rng.integers(nv_lo, nv_hi))
        B = int(rng.integers(1, 7))
        neq = int(rng.integers(0, min(3, nv) + 1)) if rng.random() < 0.4 else 0
        mdi = int(rng.integers(0, md_hi)) if rng.random() < 0.5 else 0
        k = int(rng.integers(1, 7))
        J = rng.normal(0, 0.5, size=(B, k, nv))
        e = 0.1 * rng.normal(size=(B, k))
        ep = rng.uniform(-0.5, 0.5, size=(B, nv))
        tight = 10 ** rng.uniform(-3, -1)
        lb = -rng.uniform(0.2 * tight, tight, size=(B, nv))
        ub = rng.uniform(0.2 * tight, tight, size=(B, nv))
        m = rng.random(size=(B, nv))
        lb[m < 0.1] = -np.inf
        ub[(m > 0.1) & (m < 0.2)] = np.inf
        pinned = rng.random(size=(B, nv)) < 0.05
        lb[pinned] = ub[pinned] = 0.0
        A = rng.normal(size=(B, neq, nv))
        bv = 0.01 * rng.normal(size=(B, neq))
        Gi = rng.normal(size=(B, mdi, nv))
        hi = rng.uniform(-0.01, 0.05, size=(B, mdi))
        if mdi >= 2 and rng.random() < 0.3:
            Gi[:, 1], hi[:, 1] = Gi[:, 0], hi[:, 0] + 0.01
        lm = float(rng.choice([0.0, 0.5]))
        cost = rng.uniform(0.5, 2, size=k)
        if free_lead:
            lb[:, :free_lead], ub[:, :free_lead] = -np.inf, np.inf
            neq, mdi, A, bv, Gi, hi = 0, 0, A[:, :0], bv[:, :0], Gi[:, :0], hi[:, :0]
        # ill: a weak regulariser (posture cost down to 1e-5: cond(H) up to ~1e11) -- the tolerance below follows cond(H)
        dcost = float(10 ** rng.uniform(-5, -1)) if ill else 0.1
        if ill:
            lm = float(rng.choice([0.0, 1e-6]))
        batch = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=cost, lm_damping=lm), DiagonalTaskTerm(col0=0, e=ep, cost=dcost)],
                           0.005, 1e-12, boxes=[(lb, ub)], dense_rows=[(Gi, hi)] if mdi else (),
                           equality_rows=[(A, bv)] if neq else (), batch_size=B)
        out = solver.solve(batch)
        eye = np.eye(nv)
        hb = np.concatenate([ub, -lb], axis=1)
        hb = np.where(np.isfinite(hb), hb, 1e30)
        G = np.concatenate([A, np.broadcast_to(eye, (B, nv, nv)), np.broadcast_to(-eye, (B, nv, nv)), Gi], axis=1)
        h = np.concatenate([bv, hb, hi], axis=1)
        ref = c_oracle.solve_ik_batch(np.concatenate([J, np.broadcast_to(eye, (B, nv, nv))], axis=1),
                                      np.concatenate([e, ep], axis=1), np.concatenate([cost, np.full(nv, dcost)]),
                                      np.ones(2), np.array([lm, 0.0]), np.array([0, k, k + nv], np.int32), 1e-12, G, h,
                                      meq=neq, want_Hc=True)
        # An "inconsistent constraints" verdict of the oracle (quadprog's rule: the entering row depends on the active ones
        # and none can be dropped) is REFUTED by a point that is feasible and KKT-stationary -- in the weakly regularised
        # draws round-off alone can produce that verdict (seed 965963: the C and the NumPy restatement disagree with each
        # other, an LP confirms the rows are consistent).  Only that direction is accepted, and only on the certificate.
        for b in np.nonzero(out.status != ref["status"])[0]:
            assert ill and out.status[b] == 0 and ref["status"][b] == 2, (sd, out.status, ref["status"])
            certify_point(ref["H"][b], ref["c"][b], G[b, neq:], h[b, neq:], out.dq[b], None, A[b] if neq else None, bv[b] if neq else None,
                          tag=(sd, int(b), "oracle said inconsistent"))
            # ... and the point is THE minimiser: the exact-arithmetic solve (oracle/exact_qp.py) started from its active set
            from oracle.exact_qp import exact_minimiser

            xe, info = exact_minimiser(np.vstack([J[b], eye]), np.concatenate([e[b], ep[b]]), np.concatenate([cost, np.full(nv, dcost)]), [1.0, 1.0],
                                       [lm, 0.0], [0, k, k + nv], 1e-12, G[b], h[b], out.dq[b], meq=neq)
            assert np.abs(out.dq[b] - xe).max() <= 1e-8 * max(1.0, float(np.abs(xe).max())), (sd, int(b), "refuted verdict, but not the exact minimiser")
            REFUTED.append((sd, int(b)))
        ok = (ref["status"] == 0) & (out.status == 0)
        if ok.any():
            cond = np.linalg.cond(ref["H"][ok])
            xmax = np.abs(ref["dq"][ok]).max(axis=1)
            err = np.abs(out.dq[ok] - ref["dq"][ok]).max(axis=1)
            within = err <= np.maximum(TOL_DQ, 100 * np.finfo(float).eps * cond * xmax)
            # An instance beyond that tolerance is not waved through: along the flat directions of a weakly regularised
            # H two correct solvers may differ by more than cond(H) eps |x|, but then the point must CERTIFY itself --
            # KKT residuals of the QP as stated and an objective no worse than the oracle's -- or the draw fails.
            for k in np.nonzero(~within)[0]:
                b = int(np.nonzero(ok)[0][k])
                certify_point(ref["H"][b], ref["c"][b], G[b, neq:], h[b, neq:], out.dq[b], ref["dq"][b],
                              A[b] if neq else None, bv[b] if neq else None, tag=(sd, b, float(err[k])))
                # ... and, beyond the certificate's own tolerances, against the EXACT minimiser: the kernel's point may not
                # be farther from it than ten times the fp64 oracle's is (or 1e-8) -- the rule of tests/test_exact_anchor.py
                exact_anchor_check(J[b], e[b], ep[b], cost, dcost, lm, G[b], h[b], neq, out.dq[b], ref["dq"][b], tag=(sd, b))
                CERTIFIED.append((sd, b, float(err[k]), float(cond[k])))
            n_checked += int(ok.sum())
    # The refuted verdicts of this call: listed (the GPU log shows them under -s / -rP) and BOUNDED -- the harness accepts
    # "the kernel returned the certified exact minimiser where the oracle's rule said inconsistent" as the rare round-off
    # event it is (one seed in 284 000 in round 5), not as a way for a kernel to pass by disagreeing with the oracle.
    new = REFUTED[n_refuted0:]
    if new:
        print(f"parity_suite.fuzz: oracle verdict 'inconsistent' refuted by a certified exact minimiser on (seed, instance) {new}")
    n_seeds = len(seeds) if hasattr(seeds, "__len__") else n_checked
    assert len(new) <= 1 + n_seeds // 20000, ("too many refuted oracle verdicts", new)
    return n_checked


def exact_anchor_check(J, e, ep, cost, dcost, lm, G, h, neq, x, x_ref, tag=None):
    """``|x - x_exact| <= max(1e-8 scale, 10 |x_ref - x_exact|)`` with the minimiser from oracle/exact_qp.py (50-digit
    KKT solve).  A draw the exact solver cannot settle (degenerate active set) is recorded in EXACT_UNSETTLED, not failed:
    the certificate has already passed."""
    from oracle.exact_qp import exact_minimiser

    nv = J.shape[1]
    try:
        xe, _ = exact_minimiser(np.vstack([J, np.eye(nv)]), np.concatenate([e, ep]), np.concatenate([cost, np.full(nv, dcost)]), [1.0, 1.0],
                                [lm, 0.0], [0, J.shape[0], J.shape[0] + nv], 1e-12, G, h, x, meq=neq)
    except (ZeroDivisionError, RuntimeError, AssertionError):
        EXACT_UNSETTLED.append(tag)
        return
    eg, eo = float(np.abs(x - xe).max()), float(np.abs(x_ref - xe).max())
    assert eg <= max(1e-8 * max(1.0, float(np.abs(xe).max())), 10.0 * eo), ("farther from the exact minimiser than 10 x the oracle", tag, eg, eo)


CERTIFIED = []  # (seed, instance, |dq - dq_ref|, cond(H)) of the draws accepted on their certificate instead of on dq
EXACT_UNSETTLED = []  # certified draws whose exact-arithmetic anchor could not be computed (degenerate active sets)
REFUTED = []  # (seed, instance) where the oracle's "inconsistent" verdict was refuted by the kernel's certified point


def certify_point(P, q, G, h, x, x_ref, A=None, b=None, tag=None):
    """``x`` is the minimiser of  min 1/2 x'Px + q'x, Gx <= h, Ax = b  as far as fp64 can tell: primal feasible to
    1e-9 (1 + |h|), stationary with non-negative multipliers to 1e-9 (|q| + |P| |x|), and an objective not above the
    oracle's by more than 1e-12 (1 + |f|).  Independent of how far x is from x_ref."""
    stat, viol, lam, nu = po.kkt_residuals(P, q, G, h, x, A=A, b=b, with_nu=True)
    scale = max(1.0, float(np.abs(q).max()), float(np.abs(P).max() * np.abs(x).max()))
    hf = np.abs(h)[np.abs(h) < 1e29]  # (rows  x_i <= 1e30  stand for missing bounds)
    hmax = float(hf.max(initial=0.0))
    assert viol <= 1e-9 * (1.0 + hmax), ("violation", tag, viol)
    assert stat <= 1e-9 * scale, ("stationarity", tag, stat, scale)
    if x_ref is None:  # (no reference point: feasibility + stationarity are the whole certificate)
        return
    f = lambda v: 0.5 * v @ P @ v + q @ v  # noqa: E731
    gap = (f(x) - f(x_ref)) / (1.0 + abs(f(x_ref)))
    # Two nearly feasible KKT points differ in objective by what the multipliers make of their feasibility round-off
    # (f moves by lam_i per unit of slack in row i): with multipliers of 6e5 (seed 961094: an equality next to pinned
    # coordinates) one ulp of slack is 1e-11 of objective, and the comparison allows for exactly that, no more.
    viol_ref = max(0.0, float((G @ x_ref - h).max(initial=0.0)) if len(G) else 0.0, float(np.abs(A @ x_ref - b).max()) if A is not None else 0.0)
    slack_noise = viol + viol_ref + 4.0 * np.finfo(float).eps * (1.0 + hmax)
    allow = (float(np.abs(lam).sum()) + float(np.abs(nu).sum())) * slack_noise / (1.0 + abs(f(x_ref)))
    assert gap <= 1e-12 + allow, ("objective above the oracle's", tag, gap, allow)


def kkt_certificate(solver, seeds):
    """Independent of any Goldfarb-Idnani restatement: the solution the kernel returns must satisfy the KKT
    conditions of the QP the *reference* would build (stationarity with non-negative multipliers on the
    active rows = dual feasibility + complementarity, primal feasibility, equalities) -- a strictly convex
    QP has one minimiser, so the certificate proves dq is it.  Random boxes, dense rows, barriers with safe
    displacements, equalities, LM damping."""
    from pink_amd.batch import BarrierTerm

    n_checked = 0
    for sd in seeds:
        rng = np.random.default_rng(sd)
        nv = int(rng.integers(2, 34))
        B = int(rng.integers(1, 5))
        k = int(rng.integers(1, 7))
        neq = int(rng.integers(0, min(3, nv - 1) + 1)) if rng.random() < 0.5 else 0
        mdi = int(rng.integers(0, 4))
        nbar = int(rng.integers(0, 3))
        dt = 0.005
        J = rng.normal(0, 0.5, size=(B, k, nv))
        e = 0.1 * rng.normal(size=(B, k))
        ep = rng.uniform(-0.5, 0.5, size=(B, nv))
        cost = rng.uniform(0.5, 2, size=k)
        lm = float(rng.choice([0.0, 0.3]))
        gain = float(rng.uniform(0.4, 1.0))
        lb = -rng.uniform(0.002, 0.05, size=(B, nv))
        ub = rng.uniform(0.002, 0.05, size=(B, nv))
        m = rng.random(size=(B, nv))
        lb[m < 0.15] = -np.inf
        ub[(m > 0.15) & (m < 0.3)] = np.inf
        A = rng.normal(size=(B, neq, nv))
        bv = 0.005 * rng.normal(size=(B, neq))
        Gi = rng.normal(size=(B, mdi, nv))
        hi = rng.uniform(0.0, 0.05, size=(B, mdi))
        bars = []
        for _ in range(nbar):
            Jh = rng.normal(0, 0.3, size=(B, 3, nv))
            bars.append(BarrierTerm(J_h=Jh, h=rng.uniform(0.0, 0.05, size=(B, 3)), gain=100.0,
                                    safe_displacement_gain=float(rng.choice([0.0, 1.0, 2.5])),
                                    safe_displacement=0.01 * rng.normal(size=(B, nv)) if rng.random() < 0.6 else None))
        batch = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=cost, gain=gain, lm_damping=lm), DiagonalTaskTerm(col0=0, e=ep, cost=0.1)],
                           dt, 1e-12, boxes=[(lb, ub)], dense_rows=[(Gi, hi)] if mdi else (), barriers=bars,
                           equality_rows=[(A, bv)] if neq else (), batch_size=B)
        out = solver.solve(batch)
        eye = np.eye(nv)
        for b in range(B):
            # the QP as pink.build_ik states it (oracle stacking is pinned by the reference fixtures)
            tasks = [(J[b], e[b], cost, gain, lm), (eye, ep[b], 0.1, 1.0, 0.0)]
            bt = [(t.J_h[b], t.h[b], 100.0, t.safe_displacement_gain,
                   None if t.safe_displacement is None else t.safe_displacement[b]) for t in bars]
            fin_u, fin_l = np.isfinite(ub[b]), np.isfinite(lb[b])
            blocks = [(eye[fin_u], ub[b][fin_u]), (-eye[fin_l], -lb[b][fin_l])]
            if mdi:
                blocks.append((Gi[b], hi[b]))
            P, q, G, h = po.build_qp(nv, tasks, 1e-12, blocks, bt, dt)
            if out.status[b] != 0:
                continue  # infeasible draws are covered by fuzz() against the oracle's verdict
            x = out.dq[b]
            stat, viol, lam = po.kkt_residuals(P, q, G, h, x, A=A[b] if neq else None, b=bv[b] if neq else None)
            scale = max(1.0, np.abs(q).max())
            assert stat <= 1e-9 * scale and viol <= 1e-10, (sd, b, stat, viol)
            n_checked += 1
    return n_checked


def small_stack_packing(solver):
    """nv <= 8 without barriers takes the stack kernel that packs eight instances per wavefront (two per MFMA
    tile): every batch size around the packing boundaries, LM damping, per-instance costs, a diagonal task that
    does not start at column 0."""
    rng = np.random.default_rng(21)
    for nv in (1, 3, 6, 8):
        for B in (1, 7, 8, 9, 17):
            k = int(rng.integers(1, 9))
            J = rng.normal(0, 0.5, size=(B, k, nv))
            e = 0.1 * rng.normal(size=(B, k))
            cost = rng.uniform(0.5, 2.0, size=(B, k)) if B % 2 else rng.uniform(0.5, 2.0, size=k)
            root = int(rng.integers(0, nv))
            ep = rng.uniform(-0.5, 0.5, size=(B, nv - root))
            tasks = [DenseTaskTerm(J=J, e=e, cost=cost, gain=0.8, lm_damping=0.3),
                     DiagonalTaskTerm(col0=root, e=ep, cost=0.2, gain=0.9, lm_damping=0.1)]
            batch = pack_terms(nv, tasks, 0.01, 1e-9, batch_size=B)
            H, c = solver.stack(batch)
            for b in range(B):
                cb = cost[b] if cost.ndim == 2 else cost
                Hr, cr = po.qp_objective(nv, [(J[b], e[b], cb, 0.8, 0.3), (np.eye(nv)[root:], ep[b], 0.2, 0.9, 0.1)], 1e-9)
                assert np.abs(H[b] - Hr).max() <= 1e-13 * max(1.0, np.abs(Hr).max()), (nv, B, b)
                assert np.abs(c[b] - cr).max() <= 1e-13 * max(1.0, np.abs(cr).max()), (nv, B, b)
