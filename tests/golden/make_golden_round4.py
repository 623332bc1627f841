"""Golden vectors for the task / limit classes the whole-step kernel learnt to form in round 4, produced by running the
REFERENCE's own classes (run here, once; needs /root/reference):

  pink.limits.AccelerationLimit.compute_qp_inequalities      (pink/limits/acceleration_limit.py:119-199)
  pink.tasks.LinearHolonomicTask / JointCouplingTask          (pink/tasks/linear_holonomic_task.py, joint_coupling_task.py)
  pink.tasks.DampingTask, LowAccelerationTask, JointVelocityTask, PostureTask (compute_error / compute_jacobian)

on vector-space models (one single-dof joint per coordinate: the stub `pinocchio` of make_golden.py supplies
`pin.difference` / `pin.dDifference` for those), seeded inputs.  Nothing of the reference is copied: only its OUTPUTS
are stored (tests/golden/pink_round4.npz); tests/test_oracle_golden.py holds pink_amd's restatements -- and through them
the tables the device kernels read -- to these numbers.

    python tests/golden/make_golden_round4.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


class Joint:
    def __init__(self, i):
        self.idx_q, self.nq, self.idx_v, self.nv = i, 1, i, 1


class Model(mg.FakeModel):
    """make_golden's vector-space model with joint names (joint_1 .. joint_nv; index 0 is the universe)."""

    def __init__(self, nv, q_min, q_max, v_max):
        super().__init__(nv, q_min, q_max, v_max)
        self.names = ["universe"] + [f"joint_{k}" for k in range(1, nv + 1)]
        self.joints = [types.SimpleNamespace(idx_q=-1, nq=0, idx_v=-1, nv=0)] + [Joint(i) for i in range(nv)]

    def existJointName(self, name):
        return name in self.names

    def getJointId(self, name):
        return self.names.index(name)


def main():
    mg.install_stubs()
    import pinocchio as pin

    pin.ARG1 = 1
    pin.dDifference = lambda model, q0, q1, arg: np.eye(len(np.asarray(q1)))  # vector space
    pin.neutral = lambda model: np.zeros(model.nq)
    sys.path.insert(0, mg.REFERENCE)
    from pink.limits import AccelerationLimit
    from pink.tasks import DampingTask, JointCouplingTask, JointVelocityTask, LinearHolonomicTask, LowAccelerationTask, PostureTask
    from pink.utils import VectorSpace

    rng = np.random.default_rng(20260925)
    out = {}
    for case, nv in (("arm7", 7), ("arm12", 12)):
        q = rng.uniform(-1.0, 1.0, size=nv)
        q_min = q - rng.uniform(0.005, 0.6, size=nv)
        q_max = q + rng.uniform(0.005, 0.6, size=nv)
        q_min[2], q_max[2] = -np.inf, np.inf  # a joint without configuration limits
        v_max = rng.uniform(1.0, 10.0, size=nv)
        model = Model(nv, q_min, q_max, v_max)
        model.hasConfigurationLimit = lambda q_min=q_min: np.isfinite(q_min)
        cfg = types.SimpleNamespace(model=model, q=q, tangent=VectorSpace(nv))
        dt = float(rng.choice([1e-3, 5e-3, 2e-2]))
        a = 10 ** rng.uniform(0.5, 3.0, size=nv)
        a[4] = np.inf  # a joint without an acceleration limit
        v_prev = rng.normal(size=nv) * 0.5
        acc = AccelerationLimit(model, a.copy())
        acc.set_last_integration(v_prev, dt)
        G, h = acc.compute_qp_inequalities(cfg, dt)
        out[f"{case}/nv"], out[f"{case}/dt"], out[f"{case}/q"] = nv, dt, q
        out[f"{case}/q_min"], out[f"{case}/q_max"], out[f"{case}/v_max"] = q_min, q_max, v_max
        out[f"{case}/a_max"], out[f"{case}/v_prev"] = a, v_prev
        out[f"{case}/acc_G"], out[f"{case}/acc_h"] = G, h
        # linear holonomic / joint coupling
        A = rng.normal(size=(3, nv))
        b = 0.1 * rng.normal(size=3)
        q_0 = rng.uniform(-0.5, 0.5, size=nv)
        lh = LinearHolonomicTask(A, b, q_0, cost=[1.0, 2.0, 0.5], lm_damping=1e-3, gain=0.8)
        out[f"{case}/lh_A"], out[f"{case}/lh_b"], out[f"{case}/lh_q0"] = A, b, q_0
        out[f"{case}/lh_e"], out[f"{case}/lh_J"] = lh.compute_error(cfg), lh.compute_jacobian(cfg)
        Hc = lh.compute_qp_objective(cfg)
        out[f"{case}/lh_H"], out[f"{case}/lh_c"] = Hc[0], Hc[1]
        names, ratios = ["joint_2", "joint_3", "joint_6"], [1.0, -0.5, 2.0]
        jc = JointCouplingTask(names, ratios, 100.0, cfg, lm_damping=5e-7)
        out[f"{case}/jc_ratios"] = np.array(ratios)
        out[f"{case}/jc_e"], out[f"{case}/jc_J"] = jc.compute_error(cfg), jc.compute_jacobian(cfg)
        Hc = jc.compute_qp_objective(cfg)
        out[f"{case}/jc_H"], out[f"{case}/jc_c"] = Hc[0], Hc[1]
        # identity-Jacobian tasks
        dm = DampingTask(cost=0.3)
        out[f"{case}/damp_e"], out[f"{case}/damp_J"] = dm.compute_error(cfg), dm.compute_jacobian(cfg)
        la = LowAccelerationTask(cost=0.2)
        la.set_last_integration(v_prev, dt)
        out[f"{case}/la_e"], out[f"{case}/la_J"] = la.compute_error(cfg), la.compute_jacobian(cfg)
        jv = JointVelocityTask(cost=0.1)
        v_t = rng.normal(size=nv) * 0.3
        jv.set_target(v_t, dt)
        out[f"{case}/jv_target"] = v_t
        out[f"{case}/jv_e"], out[f"{case}/jv_J"] = jv.compute_error(cfg), jv.compute_jacobian(cfg)
        po = PostureTask(cost=0.4, lm_damping=1e-2, gain=0.6)
        q_star = rng.uniform(-0.5, 0.5, size=nv)
        po.set_target(q_star)
        out[f"{case}/posture_target"] = q_star
        out[f"{case}/posture_e"], out[f"{case}/posture_J"] = po.compute_error(cfg), po.compute_jacobian(cfg)
        Hc = po.compute_qp_objective(cfg)
        out[f"{case}/posture_H"], out[f"{case}/posture_c"] = Hc[0], Hc[1]
    path = os.path.join(HERE, "pink_round4.npz")
    np.savez(path, **out)
    print("wrote", path, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
