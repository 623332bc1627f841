"""Round 6: nv = 33 / 34 box-only batches whose leading coordinates carry no bound (Draco3 with its free-flyer root,
examples/humanoid_draco3.py:55-56: nv = 33) are solved two per wavefront -- ik_solve_sweep_kernel<34, 0, 32> eliminates the
first two coordinates from the stated problem (H' = H_rr - H_re H_ee^-1 H_er), solves the rest on a 32-lane group and
recovers them afterwards.  Same minimiser as the oracle (pink/solve_ik.py:206-275 through Goldfarb-Idnani); an instance
that bounds a leading coordinate after all, or a caller that does not declare them free, is served all the same.
Emulator here, MI355X under -m gpu."""
import numpy as np
import pytest

from oracle import c_oracle
from pink_amd import synthetic
from pink_amd._lib import PackedArgs


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def solver(request):
    return request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")


def _config(nv):
    name = f"freeflyer_nv{nv}"
    synthetic.CONFIGS[name] = dict(synthetic.CONFIGS["draco3_freeflyer"], nv=nv, config_id=40 + nv)
    return name


@pytest.mark.parametrize("nv", [33, 34])
@pytest.mark.parametrize("regime", [dict(bounds="tight"), dict(bounds="kinematic", jacobians="kinematic", error_scale=0.02)])
def test_front_coordinates_are_eliminated_and_recovered(solver, nv, regime):
    terms = synthetic.make_terms(_config(nv), 37, **regime)  # (odd: the last wave holds one instance)
    batch = synthetic.pack(terms)
    assert PackedArgs(batch).desc.n_free_lead == 6  # the six root coordinates of the stand-in carry no bound
    ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms))
    out = solver.solve(batch)
    assert (ref["status"] == 0).all() and np.array_equal(out.status, ref["status"])
    assert (out.path == 0).all()  # the tableau kernel, no hand-over
    assert np.abs(out.dq - ref["dq"]).max() < 1e-10
    # the eliminated coordinates move (the test would pass trivially on zeros)
    assert np.abs(out.dq[:, :2]).max() > 1e-6


def test_a_bound_on_a_front_coordinate_sends_the_instance_to_the_other_kernel(solver, monkeypatch):
    """The descriptor's n_free_lead is a hint: instances 3 and 8 bound coordinate 1 although the caller declared the leading
    coordinates free -- they are solved (by the Goldfarb-Idnani code, in the same launch), the others stay on the tableau."""
    import pink_amd._lib as lib

    terms = synthetic.make_terms(_config(33), 12, bounds="tight")
    batch = synthetic.pack(terms)
    batch.lb[[3, 8], 1], batch.ub[[3, 8], 1] = -1e-4, 2e-4
    real = lib.PackedArgs.__init__

    def declare_free(self, b, max_iter=0):
        real(self, b, max_iter)
        self.desc.n_free_lead = 6

    monkeypatch.setattr(lib.PackedArgs, "__init__", declare_free)
    out = solver.solve(batch)
    monkeypatch.undo()
    pf = synthetic.pink_form(terms)
    honest = solver.solve(batch)  # (n_free_lead = 1 now: the 64-lane instantiation)
    assert PackedArgs(batch).desc.n_free_lead == 1
    assert (out.status == 0).all() and (honest.status == 0).all()
    assert set(np.nonzero(out.path != 0)[0]) == {3, 8}
    assert np.abs(out.dq - honest.dq).max() < 1e-10
    assert (out.dq[[3, 8], 1] >= -1e-4 - 1e-12).all() and (out.dq[[3, 8], 1] <= 2e-4 + 1e-12).all()
    del pf


def test_undeclared_batches_take_the_wide_group(solver):
    """A fixed-base robot with 33 joints (every coordinate bounded): n_free_lead = 0, the 64-lane instantiation, same
    minimiser as the oracle."""
    name = "fixed_nv33"
    synthetic.CONFIGS[name] = dict(synthetic.CONFIGS["draco3_freeflyer"], root_nv=0, config_id=77)
    terms = synthetic.make_terms(name, 9, bounds="tight")
    batch = synthetic.pack(terms)
    assert PackedArgs(batch).desc.n_free_lead == 0
    ref = c_oracle.solve_ik_batch(**synthetic.pink_form(terms))
    out = solver.solve(batch)
    assert np.array_equal(out.status, ref["status"]) and np.abs(out.dq - ref["dq"])[ref["status"] == 0].max() < 1e-10


def test_random_problems_behind_unbounded_leading_coordinates(solver):
    """parity_suite.fuzz at 33 / 34 coordinates with 2 .. 6 unbounded leading ones and no rows: random mixes of bounds (some
    missing, some pinned), LM damping and batch sizes through the instantiation that eliminates two coordinates --
    statuses and velocities of the oracle (scripts/gpu_fuzz.py runs thousands of these on the MI355X)."""
    import parity_suite as ps

    for sd in range(5000, 5012):
        assert ps.fuzz(solver, [sd], nv_lo=33, nv_hi=35, free_lead=2 + sd % 5) > 0
    # (the dispatch took the instantiation under test)
    rng_terms = synthetic.make_terms(_config(34), 3, bounds="tight")
    assert solver.solve(synthetic.pack(rng_terms)).path.max() == 0


# ---------------------------------------------------------------------------------------------------------------
# Round-5 review, item 6(a): the DEVICE route with barrier rows pinned to the reference's rows directly -- not through the
# host classes.  The whole-step kernel (emulator / MI355X) forms PositionBarrier and BodySphericalBarrier rows on chip; its
# velocity must be the minimiser of the QP whose barrier rows and regulariser are the ones the REFERENCE's classes produced
# (tests/golden/pink_round4.npz, `pb_*` cases: `*/G, h, H, c` and `*/sph_*/G, h, H, c`), solved by the oracle from those
# arrays -- pink_amd's barrier classes do not enter the expected value.
@pytest.fixture(scope="module")
def golden4():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pink_round4.npz"))


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def route_backend(request):
    import pink_amd
    from pink_amd.runtime import set_default_solver

    s = request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")
    set_default_solver(s)
    yield request.param
    pink_amd.clear_device_cache()
    set_default_solver(None)


@pytest.mark.parametrize("case", ["pb_arm", "pb_humanoid"])
def test_device_route_barrier_rows_are_the_references(route_backend, golden4, case):
    import pink_amd
    from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik_batch
    from pink_amd.barriers import BodySphericalBarrier, PositionBarrier

    g = golden4
    n, ff, dt = int(g[f"{case}/n"]), bool(g[f"{case}/ff"]), float(g[f"{case}/dt"])
    m = build_chain(n, free_flyer=ff, seed=4)
    cfg = Configuration(m, g[f"{case}/q"].copy())
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    T = cfg.get_transform_frame_to_world("tool0").copy()
    T.translation = T.translation + np.array([0.02, -0.01, 0.05])
    ft.set_target(T)
    po = PostureTask(cost=1e-2)
    po.set_target(m.neutral())
    base = pink_amd.build_ik(cfg, [ft, po], dt)  # the stack without barriers: P, q, limit rows (pinned to the reference elsewhere)
    cb = ConfigurationBatch(m, np.tile(cfg.q, (66, 1)))
    variants = []
    for name in ("max_z", "box_xy", "min_all"):
        kw = {}
        for k in ("indices", "p_min", "p_max", "gain", "safe_displacement_gain"):
            if f"{case}/{name}/{k}" in g:
                v = g[f"{case}/{name}/{k}"]
                kw[k] = [int(i) for i in v] if k == "indices" else (float(v) if k == "safe_displacement_gain" else v.copy())
        variants.append((name, PositionBarrier("tool0", **kw)))
    for name in ("far", "near"):
        variants.append((f"sph_{name}", BodySphericalBarrier(("tool0", "joint_2"), float(g[f"{case}/sph_{name}/d_min"]), gain=g[f"{case}/sph_{name}/gain"].copy(),
                                                              safe_displacement_gain=float(g[f"{case}/sph_{name}/safe_displacement_gain"]))))
    for name, bar in variants:
        V = solve_ik_batch(cb, [ft, po], dt, barriers=[bar], device_kinematics=True)
        assert pink_amd.last_solve_stats()["route"] == "device", name
        # the reference's QP: objective + the reference barrier's regulariser, limit rows + the reference barrier's rows
        P = base.P + g[f"{case}/{name}/H"]
        q = base.q + g[f"{case}/{name}/c"]
        G = np.vstack([base.G, g[f"{case}/{name}/G"]])
        h = np.concatenate([base.h, g[f"{case}/{name}/h"]])
        x_ref, status, _, _ = c_oracle.gi_solve(P, q, G, h)
        assert status == 0, name
        dq = V * dt
        assert np.abs(dq - x_ref[None, :]).max() < 1e-9 * max(1.0, np.abs(x_ref).max()), (case, name)
        assert np.array_equal(V, np.tile(V[0], (V.shape[0], 1)))  # (the same robot 66 times)
