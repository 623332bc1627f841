"""Golden vectors for the task / limit classes the whole-step kernel learnt to form in round 4, produced by running the
REFERENCE's own classes (run here, once; needs /root/reference):

  pink.limits.AccelerationLimit.compute_qp_inequalities      (pink/limits/acceleration_limit.py:119-199)
  pink.tasks.LinearHolonomicTask / JointCouplingTask          (pink/tasks/linear_holonomic_task.py, joint_coupling_task.py)
  pink.tasks.DampingTask, LowAccelerationTask, JointVelocityTask, PostureTask (compute_error / compute_jacobian)
  pink.barriers.PositionBarrier.compute_qp_inequalities / compute_qp_objective  (position_barrier.py:95-153, barrier.py:151-254)
  pink.barriers.SelfCollisionBarrier (self_collision_barrier.py:85-224; round 6: over a stub of Pinocchio's collision data)

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
        from pink.limits import ConfigurationLimit, VelocityLimit

        cl = ConfigurationLimit(model, config_limit_gain=0.7)  # (an explicit gain: honoured by the device's coordinate_box)
        Gc, hc = cl.compute_qp_inequalities(cfg, dt)
        Gv, hv = VelocityLimit(model).compute_qp_inequalities(cfg, dt)
        out[f"{case}/cl_G"], out[f"{case}/cl_h"], out[f"{case}/vl_G"], out[f"{case}/vl_h"] = Gc, hc, Gv, hv
        po = PostureTask(cost=0.4, lm_damping=1e-2, gain=0.6)
        q_star = rng.uniform(-0.5, 0.5, size=nv)
        po.set_target(q_star)
        out[f"{case}/posture_target"] = q_star
        out[f"{case}/posture_e"], out[f"{case}/posture_J"] = po.compute_error(cfg), po.compute_jacobian(cfg)
        Hc = po.compute_qp_objective(cfg)
        out[f"{case}/posture_H"], out[f"{case}/posture_c"] = Hc[0], Hc[1]
    # PositionBarrier (pink/barriers/position_barrier.py:95-153 on top of barrier.py:151-254): the reference's class,
    # driven by a pink_amd Configuration -- it only asks the configuration for a frame's pose and body Jacobian, which
    # the NumPy kinematics stand-in of this repo supplies (so the kinematics are ours, the barrier arithmetic -- bounds,
    # signs, tiled gains, safe-displacement regulariser -- the reference's)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from pink.barriers import PositionBarrier

    from pink_amd import Configuration, build_chain

    for case, n, ff in (("pb_arm", 7, False), ("pb_humanoid", 9, True)):
        m = build_chain(n, free_flyer=ff, seed=4)
        q = m.neutral()
        for j in m.joints:
            if j.kind != "free_flyer":
                q[j.idx_q] = rng.uniform(-0.8, 0.8)
        cfg = Configuration(m, q)
        p = cfg.get_transform_frame_to_world("tool0").translation
        dt = 5e-3
        out[f"{case}/q"], out[f"{case}/dt"], out[f"{case}/n"], out[f"{case}/ff"] = q, dt, n, int(ff)
        variants = {
            "max_z": dict(indices=[2], p_max=np.array([p[2] + 0.01]), gain=np.array([50.0]), safe_displacement_gain=0.0),
            "box_xy": dict(indices=[0, 1], p_min=p[:2] - 0.2, p_max=p[:2] + 0.3, gain=np.array([100.0, 80.0]), safe_displacement_gain=1.0),
            "min_all": dict(p_min=p - np.array([0.05, 0.1, 0.02]), gain=np.array([10.0, 20.0, 30.0]), safe_displacement_gain=3.0),
        }
        for name, kw in variants.items():
            bar = PositionBarrier("tool0", **kw)
            G, h = bar.compute_qp_inequalities(cfg, dt)
            H, c = bar.compute_qp_objective(cfg)
            out[f"{case}/{name}/G"], out[f"{case}/{name}/h"], out[f"{case}/{name}/H"], out[f"{case}/{name}/c"] = G, h, H, c
            for k, v in kw.items():
                out[f"{case}/{name}/{k}"] = np.asarray(v, dtype=float)
        # BodySphericalBarrier (pink/barriers/body_spherical_barrier.py): two frames kept d_min apart, its own class-K function
        from pink.barriers import BodySphericalBarrier

        d = float(np.linalg.norm(cfg.get_transform_frame_to_world("tool0").translation - cfg.get_transform_frame_to_world("joint_2").translation))
        for name, kw in (("far", dict(d_min=0.5 * d, gain=np.array([40.0]), safe_displacement_gain=2.0)),
                         ("near", dict(d_min=0.98 * d, gain=np.array([5.0]), safe_displacement_gain=0.0))):
            sb = BodySphericalBarrier(("tool0", "joint_2"), **kw)
            G, h = sb.compute_qp_inequalities(cfg, dt)
            H, c = sb.compute_qp_objective(cfg)
            out[f"{case}/sph_{name}/G"], out[f"{case}/sph_{name}/h"], out[f"{case}/sph_{name}/H"], out[f"{case}/sph_{name}/c"] = G, h, H, c
            out[f"{case}/sph_{name}/d_min"], out[f"{case}/sph_{name}/gain"] = kw["d_min"], kw["gain"]
            out[f"{case}/sph_{name}/safe_displacement_gain"] = kw["safe_displacement_gain"]
    # FloatingBaseVelocityLimit (pink/limits/floating_base_velocity_limit.py:60-148): the reference's class on an adapter
    # that shows it this repo's model through the pin.Model names it uses; pin.getFrameJacobian hands out the stand-in's
    # body Jacobian.  Identity, offset and rotated placements of the base frame; one unbounded component.
    from pink.limits import FloatingBaseVelocityLimit

    from pink_amd.lie import SE3, exp6

    class ModelView:
        def __init__(self, m):
            self.m, self.nv, self.nq = m, m.nv, m.nq
            self.joints = m.joints
            self.frames = [types.SimpleNamespace(name=f.name, parentJoint=f.joint) for f in m.frames]

        def existJointName(self, name):
            return any(j.name == name for j in self.m.joints)

        def getJointId(self, name):
            return self.m.getJointId(name)

        def existFrame(self, name):
            return any(f.name == name for f in self.m.frames)

        def getFrameId(self, name):
            return self.m.getFrameId(name)

    m = build_chain(6, free_flyer=True, seed=8)
    root_id = m.joints.index(m.root_joint)
    m.add_frame("base_id", root_id, SE3())
    m.add_frame("base_off", root_id, SE3(np.eye(3), [0.1, -0.05, 0.2]))
    m.add_frame("base_rot", root_id, exp6(np.array([0.05, 0.1, -0.1, 0.4, -0.3, 0.6])))
    q = m.neutral()
    for j in m.joints:
        if j.kind != "free_flyer":
            q[j.idx_q] = rng.uniform(-0.8, 0.8)
    M0 = exp6(rng.normal(size=6) * 0.5)
    from pink_amd.configuration import _rot_to_quat

    q[0:3], q[3:7] = M0.translation, _rot_to_quat(M0.rotation)
    cfg = Configuration(m, q)
    view = ModelView(m)
    pin.getFrameJacobian = lambda model, data, fid, rf: np.array(data.get_frame_jacobian(model.frames[fid].name))
    out["fb/q"], out["fb/dt"] = q, 5e-3
    for name in ("base_id", "base_off", "base_rot"):
        lim = FloatingBaseVelocityLimit(view, name, [0.3, 0.2, np.inf], 0.5)
        G, h = lim.compute_qp_inequalities(types.SimpleNamespace(data=cfg), 5e-3)
        out[f"fb/{name}/G"], out[f"fb/{name}/h"] = G, h
    # FrameTask / RelativeFrameTask (pink/tasks/frame_task.py:148-227, relative_frame_task.py:142-231): the reference's
    # classes decide WHICH transforms are composed and with which sign; pin.log / pin.Jlog6 are stubbed with the
    # independent SE(3) maps of oracle/se3_oracle.py (matrix logarithm through SciPy; Jacobian by central differences of
    # mpmath's 50-digit matrix logarithm, good to 1e-15: round 5, the second-order double-precision difference of round 4
    # was good to 1e-8) -- not with the closed forms the product uses.
    from oracle import se3_oracle

    from pink.tasks import FrameTask, RelativeFrameTask

    def mat(T):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = T.rotation, T.translation
        return M

    pin.log = lambda T: types.SimpleNamespace(vector=se3_oracle.log6(mat(T)))
    pin.Jlog6 = lambda T: se3_oracle.jlog6_mp(mat(T))
    m = build_chain(8, free_flyer=True, seed=11)
    m.add_frame("mid", m.getJointId("joint_4"), SE3(np.eye(3), [0.0, 0.05, 0.1]))
    q = m.neutral()
    for j in m.joints:
        if j.kind != "free_flyer":
            q[j.idx_q] = rng.uniform(-0.8, 0.8)
    M0 = exp6(rng.normal(size=6) * 0.5)
    q[0:3], q[3:7] = M0.translation, _rot_to_quat(M0.rotation)
    cfg = Configuration(m, q)
    out["ft/q"] = q
    for name, scale in (("small", 1e-2), ("large", 0.6)):
        ft = FrameTask("tool0", position_cost=1.0, orientation_cost=0.5)
        tgt = cfg.get_transform_frame_to_world("tool0") * exp6(scale * rng.normal(size=6))
        ft.set_target(tgt)
        out[f"ft/{name}/target"] = np.r_[np.asarray(tgt.rotation).ravel(), tgt.translation]
        out[f"ft/{name}/e"], out[f"ft/{name}/J"] = ft.compute_error(cfg), ft.compute_jacobian(cfg)
        rt = RelativeFrameTask("tool0", "mid", position_cost=1.0, orientation_cost=0.5)
        rtg = cfg.get_transform("tool0", "mid") * exp6(scale * rng.normal(size=6))
        rt.set_target(rtg)
        out[f"ft/{name}/rel_target"] = np.r_[np.asarray(rtg.rotation).ravel(), rtg.translation]
        out[f"ft/{name}/rel_e"], out[f"ft/{name}/rel_J"] = rt.compute_error(cfg), rt.compute_jacobian(cfg)
    # Configuration.check_limits (pink/configuration.py:166-201): the reference's method, called unbound on a stand-in
    # that has .q and .model -- tolerance, "no limit" joints, the skipped root coordinates, which entry is reported
    import pink
    from pink.exceptions import NotWithinConfigurationLimits

    for case, ff in (("cl_arm", False), ("cl_humanoid", True)):
        mm = build_chain(6, free_flyer=ff, seed=3, limit=1.0)
        lo, up = mm.lowerPositionLimit, mm.upperPositionLimit
        view = ModelView(mm)
        view.lowerPositionLimit, view.upperPositionLimit = lo, up
        r = 7 if ff else 0
        base = mm.neutral()
        qs, verdicts = [], []
        trials = [(), ((r + 2, up[r + 2] + 5e-7),), ((r + 2, up[r + 2] + 2e-6),), ((r + 4, lo[r + 4] - 3e-6), (r + 1, up[r + 1] + 1.0)),
                  ((r + 0, lo[r + 0] - 0.5),)]
        if ff:
            trials.append(((1, 50.0),))  # a root coordinate far away: not checked
        for edits in trials:
            q = base.copy()
            for i, v in edits:
                q[i] = v
            try:
                pink.Configuration.check_limits(types.SimpleNamespace(q=q, model=view), 1e-6, True)
                verdicts.append([-1, 0.0, 0.0, 0.0])
            except NotWithinConfigurationLimits as exc:
                verdicts.append([exc.joint, exc.value, exc.lower, exc.upper])
            qs.append(q)
        out[f"{case}/q"], out[f"{case}/verdict"] = np.array(qs), np.array(verdicts, dtype=float)
    # ------------------------------------------------------------------------------------------------------------------
    # The whole stacking path at once: the reference's OWN pink.build_ik (pink/solve_ik.py:152-203) over the reference's OWN
    # FrameTask x 3, RelativeFrameTask, PostureTask, JointCouplingTask x 2, DampingTask, ConfigurationLimit, VelocityLimit,
    # AccelerationLimit, FloatingBaseVelocityLimit and PositionBarrier objects -- the stack of examples/humanoid_draco3.py and
    # then some -- on a floating-base robot whose kinematics come from this repo's stand-in.  pin.Jlog6 stays the 50-digit
    # difference of the matrix logarithm set above (1e-15; round 4 used a fourth-order double-precision difference, 1e-11,
    # which cond(P) amplified to 1e-8 on the minimiser), so that the QP the reference builds can be held to 1e-10.
    pin.difference = lambda model, q0, q1: model.m.difference(np.asarray(q0, dtype=float), np.asarray(q1, dtype=float))
    pin.neutral = lambda model: model.m.neutral()
    pin.dDifference = lambda model, q0, q1, arg: model.m.d_difference(np.asarray(q0, dtype=float), np.asarray(q1, dtype=float))
    from pink.limits import ConfigurationLimit, VelocityLimit

    m = build_chain(12, free_flyer=True, seed=21, limit=2.5, velocity=6.0)
    root_id = m.joints.index(m.root_joint)
    m.add_frame("base", root_id, exp6(np.array([0.02, 0.0, 0.05, 0.1, -0.2, 0.3])))
    m.add_frame("mid", m.getJointId("joint_5"), SE3(np.eye(3), [0.0, 0.05, 0.1]))
    q = m.neutral()
    for j in m.joints:
        if j.kind != "free_flyer":
            q[j.idx_q] = rng.uniform(-0.8, 0.8)
    M0 = exp6(rng.normal(size=6) * 0.4)
    q[0:3], q[3:7] = M0.translation, _rot_to_quat(M0.rotation)
    cfg = Configuration(m, q)
    view = ModelView(m)
    view.lowerPositionLimit, view.upperPositionLimit, view.velocityLimit = m.lowerPositionLimit, m.upperPositionLimit, m.velocityLimit
    view.hasConfigurationLimit = lambda: np.isfinite(m.upperPositionLimit)
    ref_cfg = types.SimpleNamespace(q=cfg.q, model=view, tangent=cfg.tangent, data=cfg, get_transform_frame_to_world=cfg.get_transform_frame_to_world,
                                    get_transform=cfg.get_transform, get_frame_jacobian=cfg.get_frame_jacobian)
    dt, damping = 5e-3, 1e-12
    tasks, targets = [], {}
    for k, (frame, pc, oc, lm, gain) in enumerate((("tool0", 1.0, 1.0, 1e-3, 1.0), ("joint_4", [1.0, 2.0, 0.5], 0.0, 0.0, 0.85), ("joint_9", 4.0, 4.0, 1e-2, 0.5))):
        t = FrameTask(frame, position_cost=pc, orientation_cost=oc, lm_damping=lm, gain=gain)
        tgt = cfg.get_transform_frame_to_world(frame) * exp6(0.05 * rng.normal(size=6))
        t.set_target(tgt)
        targets[f"frame{k}"] = np.r_[np.asarray(tgt.rotation).ravel(), tgt.translation]
        tasks.append(t)
    rt = RelativeFrameTask("tool0", "mid", position_cost=0.8, orientation_cost=0.3, lm_damping=1e-3, gain=0.9)
    rtg = cfg.get_transform("tool0", "mid") * exp6(0.03 * rng.normal(size=6))
    rt.set_target(rtg)
    targets["rel"] = np.r_[np.asarray(rtg.rotation).ravel(), rtg.translation]
    po = PostureTask(cost=1e-1)
    q_star = m.neutral()
    po.set_target(q_star)
    jc1 = JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 100.0, ref_cfg, lm_damping=5e-7)
    jc2 = JointCouplingTask(["joint_7", "joint_8"], [1.0, -1.0], 100.0, ref_cfg, lm_damping=5e-7)
    dm = DampingTask(cost=1e-2)
    tasks += [rt, po, jc1, jc2, dm]
    a_max = np.r_[np.full(6, np.inf), np.full(12, 300.0)]
    acc = AccelerationLimit(view, a_max.copy())
    v_prev = np.r_[np.zeros(6), rng.normal(size=12) * 0.3]
    acc.set_last_integration(v_prev, dt)
    fb = FloatingBaseVelocityLimit(view, "base", [0.4, 0.3, 0.5], 0.8)
    limits = [ConfigurationLimit(view, config_limit_gain=0.6), VelocityLimit(view), acc, fb]
    p = cfg.get_transform_frame_to_world("tool0").translation
    bar = PositionBarrier("tool0", indices=[2], p_max=np.array([p[2] + 0.01]), gain=np.array([50.0]), safe_displacement_gain=1.0)
    problem = pink.build_ik(ref_cfg, tasks, dt, damping=damping, limits=limits, barriers=[bar])
    out["full/q"], out["full/dt"], out["full/a_max"], out["full/v_prev"], out["full/bar_pmax"] = q, dt, a_max, v_prev, p[2] + 0.01
    for k, v in targets.items():
        out[f"full/target_{k}"] = v
    out["full/P"], out["full/c"], out["full/G"], out["full/h"] = problem.P, problem.q, problem.G, problem.h
    # ... and a fixed-base arm with the identity-Jacobian tasks that carry state of the previous step (LowAccelerationTask,
    # JointVelocityTask -- whose error sign the first run of this file corrected) under ConfigurationLimit + VelocityLimit +
    # AccelerationLimit: the reference's build_ik again
    m = build_chain(7, free_flyer=False, seed=31, limit=2.2, velocity=5.0)
    q = m.neutral()
    for j in m.joints:
        q[j.idx_q] = rng.uniform(-0.9, 0.9)
    cfg = Configuration(m, q)
    view = ModelView(m)
    view.lowerPositionLimit, view.upperPositionLimit, view.velocityLimit = m.lowerPositionLimit, m.upperPositionLimit, m.velocityLimit
    view.hasConfigurationLimit = lambda: np.isfinite(m.upperPositionLimit)
    ref_cfg = types.SimpleNamespace(q=cfg.q, model=view, tangent=cfg.tangent, data=cfg, get_transform_frame_to_world=cfg.get_transform_frame_to_world,
                                    get_transform=cfg.get_transform, get_frame_jacobian=cfg.get_frame_jacobian)
    dt = 1e-2
    ft = FrameTask("tool0", position_cost=[1.0, 1.0, 2.0], orientation_cost=0.2, lm_damping=1e-2, gain=0.7)
    tgt = cfg.get_transform_frame_to_world("tool0") * exp6(0.08 * rng.normal(size=6))
    ft.set_target(tgt)
    ft2 = FrameTask("joint_4", position_cost=0.5, orientation_cost=0.0)
    tgt2 = cfg.get_transform_frame_to_world("joint_4") * exp6(0.04 * rng.normal(size=6))
    ft2.set_target(tgt2)
    po = PostureTask(cost=5e-2, gain=0.8)
    q_star = rng.uniform(-0.3, 0.3, size=m.nq)
    po.set_target(q_star)
    v_prev = rng.normal(size=m.nv) * 0.4
    la = LowAccelerationTask(cost=0.05)
    la.set_last_integration(v_prev, dt)
    jv = JointVelocityTask(cost=0.08)
    v_t = rng.normal(size=m.nv) * 0.3
    jv.set_target(v_t, dt)
    a_max = rng.uniform(30.0, 200.0, size=m.nv)
    acc = AccelerationLimit(view, a_max.copy())
    acc.set_last_integration(v_prev, dt)
    problem = pink.build_ik(ref_cfg, [ft, ft2, po, la, jv], dt, damping=1e-12, limits=[ConfigurationLimit(view), VelocityLimit(view), acc])
    out["arm/q"], out["arm/dt"], out["arm/q_star"], out["arm/v_prev"], out["arm/v_t"], out["arm/a_max"] = q, dt, q_star, v_prev, v_t, a_max
    out["arm/target0"] = np.r_[np.asarray(tgt.rotation).ravel(), tgt.translation]
    out["arm/target1"] = np.r_[np.asarray(tgt2.rotation).ravel(), tgt2.translation]
    out["arm/P"], out["arm/c"], out["arm/G"], out["arm/h"] = problem.P, problem.q, problem.G, problem.h
    # ... and equality constraints (pink/solve_ik.py:125-149): the same arm, constraints=[FrameTask] next to a FrameTask and a posture
    hold = FrameTask("joint_6", position_cost=1.0, orientation_cost=1.0, gain=0.6)
    hold_t = cfg.get_transform_frame_to_world("joint_6") * exp6(2e-4 * rng.normal(size=6))
    hold.set_target(hold_t)
    problem = pink.build_ik(ref_cfg, [ft, po], dt, damping=1e-12, limits=[ConfigurationLimit(view), VelocityLimit(view)], constraints=[hold])
    out["eq/hold_target"] = np.r_[np.asarray(hold_t.rotation).ravel(), hold_t.translation]
    out["eq/P"], out["eq/c"], out["eq/G"], out["eq/h"], out["eq/A"], out["eq/b"] = problem.P, problem.q, problem.G, problem.h, problem.A, problem.b
    # ------------------------------------------------------------------------------------------------------------------
    # SelfCollisionBarrier (pink/barriers/self_collision_barrier.py:85-224), round 6: the reference's class on a stand-in
    # that shows it exactly what it reads from Pinocchio's collision data -- collision_model.collisionPairs / geometryObjects
    # (parent joints), collision_data.distanceResults (min_distance, getNearestPoint1 / 2), data.oMi, and
    # pin.getJointJacobian(..., LOCAL_WORLD_ALIGNED) / pin.skew -- with the distance results of sphere pairs attached to the
    # joints of this repo's stand-in robot (the smooth convex case the reference's docstring declares well defined).  Which
    # pairs are the closest, the sign of each term of a row, d_min and the base class's rows and objective: the reference's.
    # (its own generator: nothing above this line changes when this section does)
    from pink.barriers import SelfCollisionBarrier

    from pink_amd.barriers.self_collision_barrier import SpherePairs

    rng6 = np.random.default_rng(20260930)
    pin.skew = lambda v: np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    pin.getJointJacobian = lambda model, data, jid, rf: np.array(data.get_joint_jacobian_world_aligned(jid))
    for case, n, ff in (("sc_arm", 7, False), ("sc_humanoid", 9, True)):
        m = build_chain(n, free_flyer=ff, seed=6)
        q = m.neutral()
        for j in m.joints:
            if j.kind != "free_flyer":
                q[j.idx_q] = rng6.uniform(-0.9, 0.9)
        if ff:
            M0 = exp6(rng6.normal(size=6) * 0.4)
            q[0:3], q[3:7] = M0.translation, _rot_to_quat(M0.rotation)
        cfg = Configuration(m, q)
        nj = len(m.joints)
        pairs = []
        for _ in range(5):
            j1, j2 = sorted(int(v) for v in rng6.choice(np.arange(nj), size=2, replace=False))
            pairs.append((j1, 0.05 * rng6.normal(size=3), float(rng6.uniform(0.01, 0.04)), j2, 0.05 * rng6.normal(size=3), float(rng6.uniform(0.01, 0.04))))
        res = SpherePairs(pairs)(cfg)
        geoms = []
        for pr in res:
            geoms += [types.SimpleNamespace(parentJoint=pr.joint1), types.SimpleNamespace(parentJoint=pr.joint2)]
        coll_model = types.SimpleNamespace(collisionPairs=[types.SimpleNamespace(first=2 * k, second=2 * k + 1) for k in range(len(res))], geometryObjects=geoms)
        coll_data = types.SimpleNamespace(distanceResults=[
            types.SimpleNamespace(min_distance=pr.min_distance, getNearestPoint1=(lambda pr=pr: pr.point1), getNearestPoint2=(lambda pr=pr: pr.point2)) for pr in res])
        ref_cfg = types.SimpleNamespace(q=cfg.q, model=types.SimpleNamespace(nv=m.nv), data=cfg, collision_model=coll_model, collision_data=coll_data)
        dt = 5e-3
        out[f"{case}/q"], out[f"{case}/dt"], out[f"{case}/n"], out[f"{case}/ff"] = q, dt, n, int(ff)
        out[f"{case}/pairs"] = np.array([[j1, *c1, r1, j2, *c2, r2] for j1, c1, r1, j2, c2, r2 in pairs], dtype=float)
        for name, kw in (("all", dict(n_collision_pairs=len(pairs), gain=20.0, safe_displacement_gain=2.0, d_min=0.02)),
                         ("closest2", dict(n_collision_pairs=2, gain=5.0, safe_displacement_gain=0.0, d_min=0.05))):
            sb = SelfCollisionBarrier(**kw)
            hb, Jb = sb.compute_barrier(ref_cfg), sb.compute_jacobian(ref_cfg)
            G, h = sb.compute_qp_inequalities(ref_cfg, dt)
            H, c = sb.compute_qp_objective(ref_cfg)
            out[f"{case}/{name}/barrier"], out[f"{case}/{name}/J"] = hb, Jb
            out[f"{case}/{name}/G"], out[f"{case}/{name}/h"], out[f"{case}/{name}/H"], out[f"{case}/{name}/c"] = G, h, H, c
            for k, v in kw.items():
                out[f"{case}/{name}/{k}"] = np.asarray(v, dtype=float)
    path = os.path.join(HERE, "pink_round4.npz")
    np.savez(path, **out)
    print("wrote", path, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
