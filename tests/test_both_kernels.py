"""Both stack + solve kernels on the shapes where the dispatch rule (``prefer_sweep``, pink_amd/csrc/dispatch.h) would
pick only one of them: ``PINKHIP_SOLVER=packed`` forces the Goldfarb-Idnani kernel (ik_kernels_packed.h),
``PINKHIP_SOLVER=sweep`` the sweep-tableau kernel (ik_sweep.h) wherever it is instantiated, ``PINKHIP_SOLVER=sweepx``
the one with virtual dense rows (ik_sweepx.h) wherever THAT is instantiated (dense rows on up to 32 coordinates; the
sweep-tableau kernel elsewhere).  Emulator under
``-m "not gpu"``, MI355X under ``-m gpu``; same checks against the oracle as the main parity suites."""
import pytest

from tests import parity_suite as ps


@pytest.fixture(params=["packed", "sweep", "sweepx"])
def forced(request, monkeypatch):
    monkeypatch.setenv("PINKHIP_SOLVER", request.param)
    return request.param


def _suite(solver, golden, seeds, B):
    for name in ("ur5", "draco3", "barrier", "equality", "safe"):
        ps.golden(solver, golden, name)
    for name, bounds, jac in (("ur5", "tight", "dense"), ("draco3", "tight", "dense"), ("draco3", "kinematic", "kinematic"),
                              ("jvrc", "tight", "dense")):
        ps.config(solver, name, bounds, jac, B=B)
    for nv in (3, 6, 8, 12, 16, 30, 33, 50, 64):
        ps.random_dims(solver, nv, B=2, seed=100 + nv, root=min(2, nv - 1) if nv > 3 else 0)
    for nv, md in ((6, 1), (12, 4), (14, 7), (24, 8), (30, 2), (30, 6), (30, 8), (32, 5), (50, 6), (31, 32)):
        ps.random_dims(solver, nv, B=3, seed=500 + nv, md=md)
    for nv, n_eq, md in ((6, 2, 0), (12, 3, 2), (30, 6, 3), (30, 3, 4), (50, 4, 2)):
        ps.equality_constraints(solver, nv, n_eq, md, B=3, seed=900 + nv)
    ps.equality_edge_cases(solver)
    ps.infeasible(solver)
    ps.infeasible_dense_rows(solver)
    ps.not_positive_definite(solver)
    ps.mixed_status_batch(solver)
    ps.max_iter_is_reported(solver)
    ps.empty_task_list(solver)
    ps.fulfilled_tasks_give_zero(solver)
    ps.unconstrained(solver)
    assert ps.fuzz(solver, seeds) > 0
    assert ps.kkt_certificate(solver, [s + 4000 for s in seeds]) > 0


def test_forced_kernel_emulator(emu, golden, forced):
    _suite(emu, golden, range(5100, 5125), B=3)


@pytest.mark.gpu
def test_forced_kernel_gpu(gpu_solver, golden, forced):
    _suite(gpu_solver, golden, range(7500, 7700), B=256)
