"""Generate golden fixtures by running the REFERENCE's own code (run here, once).

The reference (``/root/reference``, stephane-caron/pink) cannot be imported as
is: it needs ``pinocchio`` and ``qpsolvers``, neither installable offline.  Its
*stacking* half is pure NumPy though, so this script installs two stub modules
(just the names ``pink`` touches at import time plus ``pin.difference`` for
vector-space models), imports the real ``pink`` from the read-only checkout and
records what ``pink.build_ik`` / ``Task.compute_qp_objective`` /
``ConfigurationLimit`` / ``VelocityLimit`` / ``Barrier`` produce on seeded
synthetic terms.  Nothing of the reference is copied: only its *outputs* are
stored (``tests/golden/pink_build_ik.npz``).

    python tests/golden/make_golden.py          # needs /root/reference

The QP *solve* cannot be produced this way (quadprog is absent); see
``oracle/pink_oracle.py`` for how that half is anchored.
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE = os.environ.get("PINK_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    pin = types.ModuleType("pinocchio")
    for name in ("GeometryData", "GeometryModel", "Data", "Model", "SE3", "RobotWrapper"):
        setattr(pin, name, type(name, (), {}))
    pin.ReferenceFrame = types.SimpleNamespace(LOCAL=0, WORLD=1, LOCAL_WORLD_ALIGNED=2)
    pin.__version__ = "3.0.0-stub"
    pin.difference = lambda model, q0, q1: np.asarray(q1) - np.asarray(q0)  # vector space
    viz = types.ModuleType("pinocchio.visualize")
    viz.MeshcatVisualizer = type("MeshcatVisualizer", (), {})
    viz.ViserVisualizer = type("ViserVisualizer", (), {})
    pin.visualize = viz
    sys.modules["pinocchio"] = pin
    sys.modules["pinocchio.visualize"] = viz

    qps = types.ModuleType("qpsolvers")

    class Problem:
        def __init__(self, P, q, G=None, h=None, A=None, b=None, lb=None, ub=None):
            self.P, self.q, self.G, self.h, self.A, self.b = P, q, G, h, A, b

    qps.Problem = Problem
    qps.Solution = type("Solution", (), {})
    qps.solve_problem = None
    qps.available_solvers = []
    sys.modules["qpsolvers"] = qps


class FakeJoint:
    def __init__(self, i):
        self.idx_q, self.nq, self.idx_v, self.nv = i, 1, i, 1


class FakeModel:
    """A vector-space model: nq == nv, one single-dof joint per coordinate."""

    def __init__(self, nv, q_min, q_max, v_max):
        self.nv = self.nq = nv
        self.lowerPositionLimit = np.asarray(q_min, float)
        self.upperPositionLimit = np.asarray(q_max, float)
        self.velocityLimit = np.asarray(v_max, float)
        self.joints = [FakeJoint(i) for i in range(nv)]

    def hasConfigurationLimit(self):
        return np.ones(self.nq, dtype=bool)


def time_reference_build_ik(terms, repeats=10, sample=16):
    """Wall time of the reference's own ``pink.build_ik`` (``pink/solve_ik.py:152-203``: the stacking half of
    ``solve_ik``; its QP solve needs quadprog) called once per instance on a synthetic batch of
    ``pink_amd.synthetic`` terms -- bench.py's CPU baseline B0'.  Needs /root/reference."""
    import statistics
    import time

    install_stubs()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import pink
    from pink.limits import ConfigurationLimit, VelocityLimit
    from pink.tasks import Task
    from pink.utils import VectorSpace

    class SyntheticTask(Task):
        def __init__(self, J, e, **kw):
            super().__init__(**kw)
            self.J, self.e = J, e

        def compute_error(self, configuration):
            return self.e

        def compute_jacobian(self, configuration):
            return self.J

        def __repr__(self):
            return "SyntheticTask()"

    nv, n = terms.nv, min(sample, terms.B)
    q_min = np.full(nv, -np.inf)
    q_max = np.full(nv, np.inf)
    v_max = np.full(nv, np.inf)
    problems = []
    for b in range(n):
        q = np.zeros(nv)
        q_min[terms.limit_idx] = 2.0 * terms.cfg_lo[b]  # gamma = 0.5: gamma (q_min - q) = cfg_lo
        q_max[terms.limit_idx] = 2.0 * terms.cfg_hi[b]
        v_max[terms.limit_idx] = terms.vel[b] / terms.dt
        model = FakeModel(nv, q_min.copy(), q_max.copy(), v_max.copy())
        cfg = type("Cfg", (), {})()
        cfg.model, cfg.q, cfg.tangent = model, q, VectorSpace(nv)
        tasks = [SyntheticTask(t.J[b], t.e[b], cost=np.asarray(t.cost, float), gain=t.gain, lm_damping=t.lm_damping)
                 for t in terms.dense_tasks]
        for t in terms.diag_tasks:
            k = t.e.shape[1]
            tasks.append(SyntheticTask(np.eye(nv)[t.col0:t.col0 + k], t.e[b], cost=float(t.cost), gain=t.gain,
                                       lm_damping=t.lm_damping))
        problems.append((cfg, tasks, [ConfigurationLimit(model), VelocityLimit(model)]))

    def run():
        for cfg, tasks, limits in problems:
            pink.build_ik(cfg, tasks, terms.dt, damping=terms.damping, limits=limits)

    run()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {"available": True, "calls_per_s": n / med, "us_per_call_median": med / n * 1e6, "us_per_call_best": min(ts) / n * 1e6,
            "repeats": repeats, "sample": n, "threads": 1, "pink_version": pink.__version__,
            "what": "pink.build_ik only (objective + inequality stacking in NumPy); Pinocchio and the quadprog solve excluded"}


def main():
    install_stubs()
    sys.path.insert(0, REFERENCE)
    import pink
    from pink.barriers import Barrier
    from pink.limits import ConfigurationLimit, VelocityLimit
    from pink.tasks import Task
    from pink.utils import VectorSpace

    class FakeConfiguration:
        def __init__(self, model, q):
            self.model = model
            self.q = q
            self.tangent = VectorSpace(model.nv)

    class SyntheticTask(Task):
        def __init__(self, J, e, **kw):
            super().__init__(**kw)
            self.J, self.e = J, e

        def compute_error(self, configuration):
            return self.e

        def compute_jacobian(self, configuration):
            return self.J

        def __repr__(self):
            return "SyntheticTask()"

    class SyntheticBarrier(Barrier):
        def __init__(self, J_h, h_val, dq_safe=None, **kw):
            super().__init__(len(h_val), **kw)
            self.J_h, self.h_val, self.dq_safe = J_h, h_val, dq_safe

        def compute_barrier(self, configuration):
            return self.h_val

        def compute_jacobian(self, configuration):
            return self.J_h

        def compute_safe_displacement(self, configuration):
            # the hook a subclass overrides (pink/barriers/barrier.py:134-149); default is zero
            if self.dq_safe is None:
                return super().compute_safe_displacement(configuration)
            return self.dq_safe

    rng = np.random.default_rng(20260924)
    out = {}
    cases = [
        dict(name="ur5", nv=6, root=0, frames=[(1.0, 1.0)], lm=1.0, posture=1e-3, dt=0.005, nbar=0),
        dict(name="draco3", nv=30, root=6, frames=[(1.0, 1.0), (1.0, 0.0), (1.0, 1.0), (4.0, 4.0)], lm=0.0,
             posture=1e-1, dt=0.005, nbar=0),
        dict(name="barrier", nv=12, root=0, frames=[(1.0, 3.0), (2.0, 0.5)], lm=1e-3, posture=1e-2, dt=0.01, nbar=2),
        # constraints=[task]: A = J, b = -gain e (pink/solve_ik.py:125-149)
        dict(name="equality", nv=9, root=0, frames=[(1.0, 1.0), (2.0, 1.0)], lm=0.0, posture=1e-1, dt=0.005, nbar=0,
             n_constraints=2),
        # barriers whose compute_safe_displacement is overridden: c gains -rho dq_safe (pink/barriers/barrier.py:193-201)
        dict(name="safe", nv=14, root=6, frames=[(1.0, 1.0), (1.0, 0.5)], lm=1e-2, posture=1e-2, dt=0.01, nbar=2,
             safe=True, safe_gains=(1.0, 3.0)),
    ]
    for cs in cases:
        nv, root, dt = cs["nv"], cs["root"], cs["dt"]
        n_act = nv - root
        q = rng.uniform(-1.0, 1.0, size=nv)
        q_min = np.full(nv, -np.inf)
        q_max = np.full(nv, np.inf)
        v_max = np.full(nv, np.inf)
        q_min[root:] = q[root:] - rng.uniform(0.01, 0.2, size=n_act)
        q_max[root:] = q[root:] + rng.uniform(0.01, 0.2, size=n_act)
        v_max[root:] = rng.uniform(1.0, 10.0, size=n_act)
        model = FakeModel(nv, q_min, q_max, v_max)
        cfg = FakeConfiguration(model, q)
        tasks, Js, es, costs, gains, lms = [], [], [], [], [], []
        for i, (pc, oc) in enumerate(cs["frames"]):
            J = rng.normal(0, 0.5, size=(6, nv))
            e = 0.1 * rng.normal(size=6)
            cost = np.array([pc] * 3 + [oc] * 3)
            gain = [1.0, 0.85, 0.5, 1.0][i]
            tasks.append(SyntheticTask(J, e, cost=cost, gain=gain, lm_damping=cs["lm"]))
            Js.append(J), es.append(e), costs.append(cost), gains.append(gain), lms.append(cs["lm"])
        # posture-like task: identity Jacobian on the actuated coordinates, scalar float cost
        Jp = np.eye(nv)[root:]
        ep = rng.uniform(-0.5, 0.5, size=n_act)
        tasks.append(SyntheticTask(Jp, ep, cost=float(cs["posture"]), gain=1.0, lm_damping=0.0))
        barriers, bJ, bh, bgain, bsafe = [], [], [], [], []
        bdq = []
        for ib in range(cs["nbar"]):
            Jh = rng.normal(0, 0.3, size=(3, nv))
            hv = rng.uniform(0.0, 0.05, size=3)
            sgain = cs.get("safe_gains", (1.0,) * cs["nbar"])[ib]
            dq_safe = 0.02 * rng.normal(size=nv) if cs.get("safe") else None
            barriers.append(SyntheticBarrier(Jh, hv, dq_safe, gain=100.0, safe_displacement_gain=sgain))
            bJ.append(Jh), bh.append(hv), bgain.append(100.0), bsafe.append(sgain)
            if dq_safe is not None:
                bdq.append(dq_safe)
        constraints, cJ, ce, cgain = [], [], [], []
        for ic in range(cs.get("n_constraints", 0)):
            k = 2 + ic
            Jc = rng.normal(0, 0.5, size=(k, nv))
            ec = 0.01 * rng.normal(size=k)
            gc = [0.7, 1.0][ic]
            constraints.append(SyntheticTask(Jc, ec, cost=1.0, gain=gc))
            cJ.append(Jc), ce.append(ec), cgain.append(gc)
        limits = [ConfigurationLimit(model), VelocityLimit(model)]
        problem = pink.build_ik(cfg, tasks, dt, damping=1e-12, limits=limits, barriers=barriers or None,
                                constraints=constraints or None)
        H_tasks = [t.compute_qp_objective(cfg) for t in tasks]
        n = cs["name"]
        out[f"{n}/nv"] = nv
        out[f"{n}/root"] = root
        out[f"{n}/dt"] = dt
        out[f"{n}/q"] = q
        out[f"{n}/q_min"], out[f"{n}/q_max"], out[f"{n}/v_max"] = q_min, q_max, v_max
        out[f"{n}/J"] = np.stack(Js)
        out[f"{n}/e"] = np.stack(es)
        out[f"{n}/cost"] = np.stack(costs)
        out[f"{n}/gain"] = np.array(gains)
        out[f"{n}/lm"] = np.array(lms)
        out[f"{n}/e_posture"] = ep
        out[f"{n}/posture_cost"] = cs["posture"]
        if bJ:
            out[f"{n}/barrier_J"] = np.stack(bJ)
            out[f"{n}/barrier_h"] = np.stack(bh)
            out[f"{n}/barrier_gain"] = np.array(bgain)
            out[f"{n}/barrier_safe_gain"] = np.array(bsafe)
        if bdq:
            out[f"{n}/barrier_dq_safe"] = np.stack(bdq)
        for ic in range(len(cJ)):
            out[f"{n}/constraint{ic}_J"], out[f"{n}/constraint{ic}_e"] = cJ[ic], ce[ic]
            out[f"{n}/constraint{ic}_gain"] = cgain[ic]
        if cJ:
            out[f"{n}/n_constraints"] = len(cJ)
            out[f"{n}/A"], out[f"{n}/b"] = problem.A, problem.b
        out[f"{n}/P"], out[f"{n}/qvec"] = problem.P, problem.q
        out[f"{n}/G"], out[f"{n}/h"] = problem.G, problem.h
        out[f"{n}/H_task0"], out[f"{n}/c_task0"] = H_tasks[0]
        out[f"{n}/config_limit_indices"] = limits[0].indices
        out[f"{n}/velocity_limit_indices"] = limits[1].indices
    np.savez_compressed(os.path.join(HERE, "pink_build_ik.npz"), **out)
    print("wrote", os.path.join(HERE, "pink_build_ik.npz"), "with", len(out), "arrays; pink", pink.__version__)


if __name__ == "__main__":
    main()
