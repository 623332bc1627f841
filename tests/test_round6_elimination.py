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
