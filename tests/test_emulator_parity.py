"""The HIP kernel source, executed on the CPU wave emulator, against the oracle.

This is the SIMT logic check that runs without a GPU (``-m "not gpu"``); the
same suite runs on the MI355X in ``test_gpu_parity.py``.
"""
import pytest

from tests import parity_suite as ps
from tests.cases import CONFIG_CASES


@pytest.mark.parametrize("name", ["ur5", "draco3", "barrier", "equality", "safe"])
def test_golden_fixture(emu, golden, name):
    ps.golden(emu, golden, name)


@pytest.mark.parametrize("name,bounds,jac", CONFIG_CASES)
def test_baseline_configs(emu, name, bounds, jac):
    ps.config(emu, name, bounds, jac, B=3 if name == "jvrc" else 5)


@pytest.mark.parametrize("nv", [1, 2, 5, 6, 7, 8, 9, 12, 13, 16, 17, 24, 25, 30, 31, 32, 33, 40, 41, 48, 56, 57, 64])
def test_every_padding_class(emu, nv):
    ps.random_dims(emu, nv, B=2, seed=100 + nv, root=min(2, nv - 1) if nv > 3 else 0)


@pytest.mark.parametrize("nv,md", [(6, 1), (12, 5), (30, 6), (31, 32), (12, 48), (30, 64)])
def test_dense_inequality_rows(emu, nv, md):
    ps.random_dims(emu, nv, B=3, seed=500 + nv, md=md)


def test_lm_damping_and_rank_deficient_tasks(emu):
    ps.random_dims(emu, 14, B=3, seed=7, lm=0.5, rank_deficient=True)


def test_no_diagonal_task(emu):
    ps.random_dims(emu, 6, B=3, seed=8, Kd_tasks=3, diag=False, lm=1.0)


def test_empty_task_list(emu):
    ps.empty_task_list(emu)


def test_fulfilled_tasks_give_zero(emu):
    ps.fulfilled_tasks_give_zero(emu)


def test_infeasible(emu):
    ps.infeasible(emu)
    ps.infeasible_dense_rows(emu)


def test_not_positive_definite(emu):
    ps.not_positive_definite(emu)


def test_mixed_status_batch(emu):
    ps.mixed_status_batch(emu)


def test_max_iter(emu):
    ps.max_iter_is_reported(emu)


def test_batched_cost(emu):
    ps.batched_cost(emu)


def test_chunked_staging(emu):
    ps.many_dense_rows_chunked_staging(emu)


def test_empty_batch(emu):
    ps.empty_batch(emu)


def test_unconstrained(emu):
    ps.unconstrained(emu)


@pytest.mark.parametrize("nv,n_eq,md", [(6, 2, 0), (12, 3, 2), (30, 6, 3), (50, 4, 2)])
def test_equality_constraints(emu, nv, n_eq, md):
    ps.equality_constraints(emu, nv, n_eq, md, B=3, seed=900 + nv)


def test_equality_edge_cases(emu):
    ps.equality_edge_cases(emu)


def test_fuzz(emu):
    assert ps.fuzz(emu, range(5000, 5060)) > 100
    assert ps.fuzz(emu, [12979]) >= 3  # (see tests/test_gpu_parity.py::test_fuzz_wide)


def test_fuzz_weakly_regularised(emu):
    """tests/test_gpu_parity.py::test_fuzz_weakly_regularised on the emulator: draws the tableau cannot certify take
    the hand-over to the Goldfarb-Idnani code (80013: coordinates without bounds at 1e2 before; 80291: NaN behind
    status 0 before; 80135, 102469: false "inconsistent")."""
    assert ps.fuzz(emu, [80011, 80013, 80015, 80037, 80135, 80261, 80291, 80389, 102469, 520171, 537045], ill=True) > 20
    # 415035 (cond(H) 1.9e9): |dq - dq_ref| = 1.3e-6 is beyond cond eps |x|, so the draw is accepted on its certificate --
    # and on the exact-arithmetic anchor: the kernel's point is not farther from the exact minimiser than 10 x the oracle's
    del ps.CERTIFIED[:], ps.EXACT_UNSETTLED[:]
    assert ps.fuzz(emu, [415035], ill=True) >= 1
    assert [c[:2] for c in ps.CERTIFIED] == [(415035, 0)] and not ps.EXACT_UNSETTLED


def test_seeds_the_wide_gpu_fuzz_of_round_6_stopped_on(emu):
    """Three weakly regularised draws of scripts/gpu_fuzz.py 3000000 100000 (28 .. 30 coordinates with three or four dense
    rows, conditioning estimates 3e8 .. 2e9): the tableau's point passed its certificate 2e-5 .. 8e-5 from the exact minimiser,
    the Goldfarb-Idnani oracle is within 4e-9 of it -- on round 5's kernels as well.  The certificate cannot see an error
    along a direction in which the residual is below its own round-off; such instances go to the Goldfarb-Idnani code by
    their conditioning estimate (PINKHIP_SWEEP_ROUTE_COND = 1e8 since)."""
    assert ps.fuzz(emu, [3063897, 3074349, 3085897], ill=True) >= 12


def test_kkt_certificate_independent_of_the_oracle_solver(emu):
    assert ps.kkt_certificate(emu, range(9000, 9060)) > 80
    # seed 704011 (wide GPU fuzz, round 5): two equalities with negative multipliers at a vertex -- the certificate's own
    # multiplier fit has to leave them sign-free
    assert ps.kkt_certificate(emu, [704011]) == 3


def test_small_stack_packing(emu):
    ps.small_stack_packing(emu)


def test_seeds_the_wide_gpu_fuzz_of_round_5_stopped_on(emu, monkeypatch):
    """Two draws of scripts/gpu_fuzz.py 900000 120000 where the CHECKER, not the kernel, was at its limit:
    965963 (weakly regularised, duplicated dense rows): the C oracle reports "inconsistent constraints" by quadprog's rule
    although the rows are consistent (its NumPy twin and an LP say so) -- the kernel's point is feasible and stationary, which
    refutes the verdict; 961094 (an equality next to pinned coordinates, multipliers of 6e5): the two points agree to
    5e-10 and their objectives differ by what those multipliers make of 1e-13 of equality residual."""
    for solver in ("sweep", "packed"):
        monkeypatch.setenv("PINKHIP_SOLVER", solver)
        del ps.REFUTED[:]
        assert ps.fuzz(emu, [965963], ill=True) == 4
        assert ps.REFUTED == [(965963, 3)]
        assert ps.fuzz(emu, [961094]) == 3
