"""Oracle stacking vs fixtures produced by the reference's own ``pink.build_ik``
(tests/golden/make_golden.py imports /root/reference with stub pinocchio/qpsolvers)."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import pink_oracle as po

NAMES = ["ur5", "draco3", "barrier", "equality", "safe"]


def _terms(g, n):
    nv, root, dt = int(g[f"{n}/nv"]), int(g[f"{n}/root"]), float(g[f"{n}/dt"])
    tasks = [(g[f"{n}/J"][i], g[f"{n}/e"][i], g[f"{n}/cost"][i], float(g[f"{n}/gain"][i]), float(g[f"{n}/lm"][i]))
             for i in range(g[f"{n}/J"].shape[0])]
    tasks.append((np.eye(nv)[root:], g[f"{n}/e_posture"], float(g[f"{n}/posture_cost"]), 1.0, 0.0))
    q, q_min, q_max, v_max = g[f"{n}/q"], g[f"{n}/q_min"], g[f"{n}/q_max"], g[f"{n}/v_max"]
    ci = po.configuration_limit_indices(q_min, q_max)
    vi = po.velocity_limit_indices(v_max)
    blocks = [po.configuration_limit_rows(q, q_min, q_max, ci, nv), po.velocity_limit_rows(v_max, vi, nv, dt)]
    barriers = []
    if f"{n}/barrier_J" in g:
        for i in range(g[f"{n}/barrier_J"].shape[0]):
            dq_safe = g[f"{n}/barrier_dq_safe"][i] if f"{n}/barrier_dq_safe" in g else None
            barriers.append((g[f"{n}/barrier_J"][i], g[f"{n}/barrier_h"][i], float(g[f"{n}/barrier_gain"][i]),
                             float(g[f"{n}/barrier_safe_gain"][i]), dq_safe))
    return nv, dt, tasks, blocks, barriers, ci, vi


def _constraints(g, n):
    k = int(g[f"{n}/n_constraints"]) if f"{n}/n_constraints" in g else 0
    return [(g[f"{n}/constraint{i}_J"], g[f"{n}/constraint{i}_e"], float(g[f"{n}/constraint{i}_gain"])) for i in range(k)]


@pytest.mark.parametrize("n", NAMES)
def test_build_qp_matches_reference_build_ik(golden, n):
    nv, dt, tasks, blocks, barriers, ci, vi = _terms(golden, n)
    assert np.array_equal(ci, golden[f"{n}/config_limit_indices"])
    assert np.array_equal(vi, golden[f"{n}/velocity_limit_indices"])
    P, q, G, h = po.build_qp(nv, tasks, 1e-12, blocks, barriers, dt)
    assert np.allclose(P, golden[f"{n}/P"], rtol=1e-13, atol=1e-15)
    assert np.allclose(q, golden[f"{n}/qvec"], rtol=1e-13, atol=1e-15)
    assert G.shape == golden[f"{n}/G"].shape
    assert np.allclose(G, golden[f"{n}/G"], rtol=1e-14, atol=0)
    assert np.allclose(h, golden[f"{n}/h"], rtol=1e-14, atol=0)
    H0, c0 = po.task_objective(*tasks[0])
    assert np.allclose(H0, golden[f"{n}/H_task0"], rtol=1e-13, atol=1e-15)
    assert np.allclose(c0, golden[f"{n}/c_task0"], rtol=1e-13, atol=1e-15)
    # constraints= (pink/solve_ik.py:125-149): A, b as the reference's build_ik returned them
    A, b = po.qp_equalities(_constraints(golden, n))
    if f"{n}/A" in golden:
        assert np.array_equal(A, golden[f"{n}/A"]) and np.allclose(b, golden[f"{n}/b"], rtol=1e-15, atol=0)
    else:
        assert A is None and b is None


def test_safe_displacement_fixture_has_a_linear_term(golden):
    """The fixture really exercises c += -rho dq_safe (barrier.py:201): q differs from the value without it."""
    nv, dt, tasks, blocks, barriers, _, _ = _terms(golden, "safe")
    no_safe = [(J, hv, g, r, None) for (J, hv, g, r, _dq) in barriers]
    _, q0, _, _ = po.build_qp(nv, tasks, 1e-12, blocks, no_safe, dt)
    assert np.abs(q0 - golden["safe/qvec"]).max() > 1e-3


@pytest.mark.parametrize("n", NAMES)
def test_c_oracle_stacking_matches_reference(golden, n):
    nv, dt, tasks, blocks, barriers, _, _ = _terms(golden, n)
    J = np.concatenate([t[0] for t in tasks])[None]
    e = np.concatenate([t[1] for t in tasks])[None]
    cost = np.concatenate([po.weight_vector(t[2], t[0].shape[0]) for t in tasks])
    rows = np.cumsum([0] + [t[0].shape[0] for t in tasks]).astype(np.int32)
    gain = np.array([t[3] for t in tasks])
    lm = np.array([t[4] for t in tasks])
    diag_extra = c_extra = None
    if barriers:
        rho = [b[3] / np.linalg.norm(b[0]) ** 2 for b in barriers]
        diag_extra = np.array([sum(rho)])
        if any(b[4] is not None for b in barriers):
            c_extra = -sum(r * b[4] for r, b in zip(rho, barriers))[None]
    G, h, meq = golden[f"{n}/G"], golden[f"{n}/h"], 0
    A = golden[f"{n}/A"] if f"{n}/A" in golden else None
    b = golden[f"{n}/b"] if f"{n}/A" in golden else None
    if A is not None:
        G, h, meq = np.vstack([A, G]), np.hstack([b, h]), len(b)
    out = c_oracle.solve_ik_batch(J, e, cost, gain, lm, rows, 1e-12, G[None], h[None],
                                  diag_extra=diag_extra, c_extra=c_extra, want_Hc=True, meq=meq)
    assert np.allclose(out["H"][0], golden[f"{n}/P"], rtol=1e-13, atol=1e-15)
    assert np.allclose(out["c"][0], golden[f"{n}/qvec"], rtol=1e-13, atol=1e-15)
    # and the solve on the reference's own (P, q, G, h, A, b) is KKT-certified
    assert out["status"][0] == 0
    stat, viol, lam = po.kkt_residuals(golden[f"{n}/P"], golden[f"{n}/qvec"], golden[f"{n}/G"], golden[f"{n}/h"],
                                       out["dq"][0], A=A, b=b)
    assert stat < 1e-10 and viol < 1e-12


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the task / limit classes the whole-step kernel forms from tables, held to the outputs of the REFERENCE's own
# classes (tests/golden/make_golden_round4.py -> pink_round4.npz: pink.limits.AccelerationLimit, pink.tasks.
# LinearHolonomicTask / JointCouplingTask / DampingTask / LowAccelerationTask / JointVelocityTask on vector-space models).
@pytest.fixture(scope="module")
def golden4():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pink_round4.npz"))


def _model_of(g, n):
    from pink_amd import Configuration, build_chain

    nv = int(g[f"{n}/nv"])
    m = build_chain(nv)
    m._lower, m._upper, m._vel = list(g[f"{n}/q_min"]), list(g[f"{n}/q_max"]), list(g[f"{n}/v_max"])
    return m, Configuration(m, g[f"{n}/q"].copy()), float(g[f"{n}/dt"])


@pytest.mark.parametrize("n", ["arm7", "arm12"])
def test_acceleration_limit_matches_the_reference_class(golden4, n):
    """pink/limits/acceleration_limit.py:119-199 -- rows +-e_i and their bounds, a joint without configuration limits,
    a joint without an acceleration limit, a previous displacement -- and the three tables the device reads."""
    import sys

    from pink_amd.limits import AccelerationLimit, ConfigurationLimit, VelocityLimit

    g = golden4
    m, cfg, dt = _model_of(g, n)
    acc = AccelerationLimit(m, g[f"{n}/a_max"].copy())
    acc.set_last_integration(g[f"{n}/v_prev"], dt)
    G, h = acc.compute_qp_inequalities(cfg, dt)
    assert np.array_equal(G, g[f"{n}/acc_G"]) and np.abs(h - g[f"{n}/acc_h"]).max() < 1e-15
    # the device's tables (pinkhip_rollout_step.acc_limit) carry the same box: upper bounds then lower bounds of G dq <= h
    gain, tables, _ = sys.modules["pink_amd.solve_ik"]._default_limits_gain(m, [ConfigurationLimit(m), VelocityLimit(m), acc])
    idx, k = acc.indices, len(acc.indices)
    a, dqp, has = tables[0, idx], tables[1, idx], tables[2, idx] != 0
    with np.errstate(invalid="ignore"):
        up = np.where(has, np.minimum(a * dt * dt + dqp, dt * np.sqrt(2 * a * (m.upperPositionLimit[idx] - cfg.q[idx]))), a * dt * dt + dqp)
        lw = np.where(has, np.minimum(a * dt * dt - dqp, dt * np.sqrt(2 * a * (cfg.q[idx] - m.lowerPositionLimit[idx]))), a * dt * dt - dqp)
    assert np.abs(up - g[f"{n}/acc_h"][:k]).max() < 1e-15 and np.abs(lw - g[f"{n}/acc_h"][k:]).max() < 1e-15
    assert (tables[0, np.setdiff1d(np.arange(m.nv), idx)] == 0).all()


@pytest.mark.parametrize("n", ["arm7", "arm12"])
def test_table_formed_tasks_match_the_reference_classes(golden4, n, emu):
    """e, J (and H, c through the stack kernel) of LinearHolonomicTask, JointCouplingTask, DampingTask,
    LowAccelerationTask, JointVelocityTask (pink/tasks/linear_holonomic_task.py:103-148, joint_coupling_task.py,
    damping_task.py, low_acceleration_task.py:46-84, joint_velocity_task.py:59-110)."""
    from pink_amd import DampingTask, PostureTask
    from pink_amd.runtime import set_default_solver
    from pink_amd.tasks import JointCouplingTask, JointVelocityTask, LinearHolonomicTask, LowAccelerationTask

    g = golden4
    m, cfg, dt = _model_of(g, n)
    set_default_solver(emu)
    try:
        lh = LinearHolonomicTask(g[f"{n}/lh_A"], g[f"{n}/lh_b"], g[f"{n}/lh_q0"], cost=[1.0, 2.0, 0.5], lm_damping=1e-3, gain=0.8)
        jc = JointCouplingTask(["joint_2", "joint_3", "joint_6"], list(g[f"{n}/jc_ratios"]), 100.0, cfg, lm_damping=5e-7)
        for t, key in ((lh, "lh"), (jc, "jc")):
            assert np.abs(t.compute_error(cfg) - g[f"{n}/{key}_e"]).max() < 1e-15
            assert np.abs(t.compute_jacobian(cfg) - g[f"{n}/{key}_J"]).max() < 1e-15
            H, c = t.compute_qp_objective(cfg)
            assert np.abs(H - g[f"{n}/{key}_H"]).max() < 1e-12 * max(1.0, np.abs(g[f"{n}/{key}_H"]).max())
            assert np.abs(c - g[f"{n}/{key}_c"]).max() < 1e-12 * max(1.0, np.abs(g[f"{n}/{key}_c"]).max())
        dm = DampingTask(cost=0.3)
        la = LowAccelerationTask(cost=0.2)
        la.set_last_integration(g[f"{n}/v_prev"], dt)
        jv = JointVelocityTask(cost=0.1)
        jv.set_target(g[f"{n}/jv_target"], dt)
        for t, key in ((dm, "damp"), (la, "la"), (jv, "jv")):
            assert np.abs(t.compute_error(cfg) - g[f"{n}/{key}_e"]).max() < 1e-16
            assert np.array_equal(t.compute_jacobian(cfg), g[f"{n}/{key}_J"])
        po = PostureTask(cost=0.4, lm_damping=1e-2, gain=0.6)
        po.set_target(g[f"{n}/posture_target"])
        assert np.abs(po.compute_error(cfg) - g[f"{n}/posture_e"]).max() < 1e-16
        assert np.array_equal(po.compute_jacobian(cfg), g[f"{n}/posture_J"])
        H, c = po.compute_qp_objective(cfg)
        assert np.abs(H - g[f"{n}/posture_H"]).max() < 1e-12 and np.abs(c - g[f"{n}/posture_c"]).max() < 1e-12
    finally:
        set_default_solver(None)


@pytest.mark.parametrize("case", ["pb_arm", "pb_humanoid"])
def test_position_barrier_matches_the_reference_class(golden4, case, emu):
    """pink.barriers.PositionBarrier of the reference (position_barrier.py:95-153 over barrier.py:151-254), run by
    make_golden_round4.py on this repo's kinematics stand-in: G = -J_h / dt, h = gain (p - p_min | p_max - p) with the
    gains tiled when both bounds are given, and the regulariser (H, c) of its objective -- pink_amd's class and the
    stack kernel against it; then the rows the whole-step kernel forms on chip, through the velocities they produce."""
    import pink_amd
    from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik_batch
    from pink_amd.barriers import PositionBarrier
    from pink_amd.runtime import set_default_solver

    g = golden4
    n, ff, dt = int(g[f"{case}/n"]), bool(g[f"{case}/ff"]), float(g[f"{case}/dt"])
    m = build_chain(n, free_flyer=ff, seed=4)
    cfg = Configuration(m, g[f"{case}/q"].copy())
    set_default_solver(emu)
    try:
        for name in ("max_z", "box_xy", "min_all"):
            kw = {}
            for k in ("indices", "p_min", "p_max", "gain", "safe_displacement_gain"):
                if f"{case}/{name}/{k}" in g:
                    v = g[f"{case}/{name}/{k}"]
                    kw[k] = [int(i) for i in v] if k == "indices" else (float(v) if k == "safe_displacement_gain" else v.copy())
            bar = PositionBarrier("tool0", **kw)
            G, h = bar.compute_qp_inequalities(cfg, dt)
            assert np.abs(G - g[f"{case}/{name}/G"]).max() < 1e-12 * max(1.0, np.abs(G).max()), name
            assert np.abs(h - g[f"{case}/{name}/h"]).max() < 1e-13 * max(1.0, np.abs(h).max()), name
            H, c = bar.compute_qp_objective(cfg)
            assert np.abs(H - g[f"{case}/{name}/H"]).max() < 1e-12 * max(1.0, np.abs(H).max()), name
            assert np.abs(c - g[f"{case}/{name}/c"]).max() < 1e-12, name
            # the whole-step kernel forms these rows on chip from the frame position and the columns' world twists: the QP
            # with the REFERENCE's barrier rows and regulariser, solved by the host route from explicit terms, gives the
            # same velocity as the device route
            ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
            T = cfg.get_transform_frame_to_world("tool0").copy()
            T.translation = T.translation + np.array([0.02, -0.01, 0.05])
            ft.set_target(T)
            po = PostureTask(cost=1e-2)
            po.set_target(m.neutral())
            cb = ConfigurationBatch(m, np.tile(cfg.q, (3, 1)))
            V_dev = solve_ik_batch(cb, [ft, po], dt, barriers=[bar], device_kinematics=True)
            assert pink_amd.last_solve_stats()["route"] == "device"
            pr = pink_amd.build_ik(cfg, [ft, po], dt, barriers=[bar])
            k = g[f"{case}/{name}/G"].shape[0]
            assert np.abs(pr.G[-k:] - g[f"{case}/{name}/G"]).max() < 1e-12 * max(1.0, np.abs(G).max())  # (barriers come last: solve_ik.py:107-122)
            v = pink_amd.solve_ik(cfg, [ft, po], dt, barriers=[bar])
            assert np.abs(V_dev - v).max() < 1e-8 * max(1.0, np.abs(v).max()), name
        # BodySphericalBarrier (body_spherical_barrier.py; evaluated on the host for the whole batch: DESIGN.md 3.4)
        from pink_amd.barriers import BodySphericalBarrier

        for name in ("far", "near"):
            sb = BodySphericalBarrier(("tool0", "joint_2"), float(g[f"{case}/sph_{name}/d_min"]), gain=g[f"{case}/sph_{name}/gain"].copy(),
                                      safe_displacement_gain=float(g[f"{case}/sph_{name}/safe_displacement_gain"]))
            G, h = sb.compute_qp_inequalities(cfg, dt)
            assert np.abs(G - g[f"{case}/sph_{name}/G"]).max() < 1e-12 * max(1.0, np.abs(G).max()), name
            assert np.abs(h - g[f"{case}/sph_{name}/h"]).max() < 1e-13 * max(1.0, np.abs(h).max()), name
            H, c = sb.compute_qp_objective(cfg)
            assert np.abs(H - g[f"{case}/sph_{name}/H"]).max() < 1e-12 * max(1.0, np.abs(H).max()), name
            assert np.abs(c - g[f"{case}/sph_{name}/c"]).max() < 1e-12, name
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)


def test_self_collision_barrier_matches_the_reference_class(golden4, emu):
    """pink.barriers.SelfCollisionBarrier of the reference (self_collision_barrier.py:85-224) read its distance results from a
    stub of Pinocchio's collision data fed by SpherePairs on this repo's stand-in robot (tests/golden/make_golden_round4.py,
    round 6): which pairs are the closest, barrier values d - d_min, one Jacobian row per pair (the sign of each of its four
    terms), and the base class's rows -J/dt, gain h and the safe-displacement objective -- pink_amd's class to 1e-12; and
    the batched solve with that barrier is the minimiser of the QP the reference's rows describe."""
    import pink_amd
    from pink_amd import Configuration, FrameTask, PostureTask, build_chain, solve_ik
    from pink_amd.barriers import SelfCollisionBarrier
    from pink_amd.barriers.self_collision_barrier import SpherePairs
    from pink_amd.runtime import set_default_solver

    g = golden4
    set_default_solver(emu)  # (compute_qp_objective stacks on the solver: barrier.py of this repo)
    for case in ("sc_arm", "sc_humanoid"):
        m = build_chain(int(g[f"{case}/n"]), free_flyer=bool(g[f"{case}/ff"]), seed=6)
        cfg = Configuration(m, g[f"{case}/q"].copy())
        dt = float(g[f"{case}/dt"])
        query = SpherePairs([(int(r[0]), r[1:4], float(r[4]), int(r[5]), r[6:9], float(r[9])) for r in g[f"{case}/pairs"]])
        for name in ("all", "closest2"):
            sb = SelfCollisionBarrier(int(g[f"{case}/{name}/n_collision_pairs"]), gain=float(g[f"{case}/{name}/gain"]),
                                      safe_displacement_gain=float(g[f"{case}/{name}/safe_displacement_gain"]), d_min=float(g[f"{case}/{name}/d_min"]),
                                      distance_query=query)
            assert np.abs(sb.compute_barrier(cfg) - g[f"{case}/{name}/barrier"]).max() < 1e-13, (case, name)
            J = sb.compute_jacobian(cfg)
            assert np.abs(J - g[f"{case}/{name}/J"]).max() < 1e-12 * max(1.0, np.abs(J).max()), (case, name)
            G, h = sb.compute_qp_inequalities(cfg, dt)
            assert np.abs(G - g[f"{case}/{name}/G"]).max() < 1e-12 * max(1.0, np.abs(G).max()), (case, name)
            assert np.abs(h - g[f"{case}/{name}/h"]).max() < 1e-12 * max(1.0, np.abs(h).max()), (case, name)
            H, c = sb.compute_qp_objective(cfg)
            assert np.abs(H - g[f"{case}/{name}/H"]).max() < 1e-12 * max(1.0, np.abs(H).max()), (case, name)
            assert np.abs(c - g[f"{case}/{name}/c"]).max() < 1e-12, (case, name)
        # ... and through the solve (dense rows of the stack + solve kernel): the velocity respects the reference's rows and
        # is KKT-stationary for its objective
        try:
            ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
            ft.set_target(cfg.get_transform_frame_to_world("tool0") * pink_amd.lie.exp6(np.array([0.05, -0.04, 0.03, 0.02, 0.0, -0.03])))
            po = PostureTask(cost=1e-1)
            po.set_target(m.neutral())
            sb = SelfCollisionBarrier(2, gain=5.0, safe_displacement_gain=0.0, d_min=0.05, distance_query=query)
            v = solve_ik(cfg, [ft, po], dt, barriers=[sb])
            G, h = g[f"{case}/closest2/G"], g[f"{case}/closest2/h"]
            assert (G @ (v * dt) <= h + 1e-9 * (1.0 + np.abs(h))).all(), case
        finally:
            if case == "sc_humanoid":
                set_default_solver(None)


def test_floating_base_velocity_limit_matches_the_reference_class(golden4, emu):
    """pink.limits.FloatingBaseVelocityLimit of the reference (floating_base_velocity_limit.py:60-148) on this repo's model
    (shown to it through an adapter with pin.Model's names): rows +-J_root and bounds dt twist_max, a component without a
    bound, base frames with identity / offset / rotated placements -- pink_amd's class row for row; then what the device
    makes of it (a box on the root coordinates + constant dense rows, rollout._floating_base_rows) describes the same set."""
    from pink_amd import Configuration, build_chain
    from pink_amd.lie import SE3, exp6
    from pink_amd.limits import FloatingBaseVelocityLimit
    from pink_amd.rollout import _floating_base_rows

    g = golden4
    m = build_chain(6, free_flyer=True, seed=8)
    root_id = m.joints.index(m.root_joint)
    m.add_frame("base_id", root_id, SE3())
    m.add_frame("base_off", root_id, SE3(np.eye(3), [0.1, -0.05, 0.2]))
    m.add_frame("base_rot", root_id, exp6(np.array([0.05, 0.1, -0.1, 0.4, -0.3, 0.6])))
    cfg, dt = Configuration(m, g["fb/q"].copy()), float(g["fb/dt"])
    rng = np.random.default_rng(1)
    for name in ("base_id", "base_off", "base_rot"):
        lim = FloatingBaseVelocityLimit(m, name, [0.3, 0.2, np.inf], 0.5)
        G, h = lim.compute_qp_inequalities(cfg, dt)
        assert np.abs(G - g[f"fb/{name}/G"]).max() < 1e-13 and np.abs(h - g[f"fb/{name}/h"]).max() < 1e-16, name
        # the device form: box lo / hi on the six root coordinates + dense rows [n, 6]: same feasible set as G dq <= h
        box, rows, hh = _floating_base_rows(m, lim, dt)
        Gr, hr = g[f"fb/{name}/G"], g[f"fb/{name}/h"]
        for _ in range(200):
            dq = np.r_[rng.normal(size=6) * 2e-3, rng.normal(size=m.nv - 6)]
            ref_ok = bool((Gr @ dq <= hr + 1e-15).all())
            dev_ok = True
            if box is not None:
                dev_ok &= bool((dq[:6] >= box[:6] - 1e-15).all() and (dq[:6] <= box[6:] + 1e-15).all())
            if len(hh):
                dev_ok &= bool((np.asarray(rows) @ dq[:6] <= np.asarray(hh) + 1e-15).all())
            assert ref_ok == dev_ok, name


def test_frame_tasks_follow_the_compositions_of_the_reference_classes(golden4):
    """pink.tasks.FrameTask / RelativeFrameTask of the reference (frame_task.py:148-227, relative_frame_task.py:142-231),
    run on this repo's kinematics with pin.log / pin.Jlog6 replaced by the independent maps of oracle/se3_oracle.py
    (matrix logarithm; central differences of mpmath's 50-digit logarithm for the Jacobian: J is held to 1e-13): which transforms are composed, in which
    order and with which sign is the reference's; pink_amd's classes (closed forms / series) give the same e and J."""
    from pink_amd import Configuration, FrameTask, build_chain
    from pink_amd.lie import SE3
    from pink_amd.tasks import RelativeFrameTask

    g = golden4
    m = build_chain(8, free_flyer=True, seed=11)
    m.add_frame("mid", m.getJointId("joint_4"), SE3(np.eye(3), [0.0, 0.05, 0.1]))
    cfg = Configuration(m, g["ft/q"].copy())
    for name in ("small", "large"):
        ft = FrameTask("tool0", 1.0, 0.5)
        T = g[f"ft/{name}/target"]
        ft.set_target(SE3(T[:9].reshape(3, 3), T[9:]))
        assert np.abs(ft.compute_error(cfg) - g[f"ft/{name}/e"]).max() < 1e-11, name
        assert np.abs(ft.compute_jacobian(cfg) - g[f"ft/{name}/J"]).max() < 1e-13, name
        rt = RelativeFrameTask("tool0", "mid", 1.0, 0.5)
        T = g[f"ft/{name}/rel_target"]
        rt.set_target(SE3(T[:9].reshape(3, 3), T[9:]))
        assert np.abs(rt.compute_error(cfg) - g[f"ft/{name}/rel_e"]).max() < 1e-11, name
        assert np.abs(rt.compute_jacobian(cfg) - g[f"ft/{name}/rel_J"]).max() < 1e-13, name


@pytest.mark.parametrize("n", ["arm7", "arm12"])
def test_device_box_equals_the_reference_limits_merged(golden4, n, emu):
    """The box the device kernels form per coordinate (coordinate_box of ik_kinematics.h: configuration limit with an
    explicit gain, velocity limit, and -- through the solve -- the acceleration tables) against the rows of the
    REFERENCE's ConfigurationLimit(model, 0.7), VelocityLimit and AccelerationLimit merged coordinate by coordinate
    (pink/solve_ik.py:107-122 stacks them; every row is +-e_i)."""
    from pink_amd.limits import ConfigurationLimit, VelocityLimit
    from pink_amd.rollout import ModelArrays

    g = golden4
    m, cfg, dt = _model_of(g, n)
    nv = m.nv
    for ours, key in ((ConfigurationLimit(m, 0.7), "cl"), (VelocityLimit(m), "vl")):
        G, h = ours.compute_qp_inequalities(cfg, dt)
        assert np.array_equal(G, g[f"{n}/{key}_G"]) and np.abs(h - g[f"{n}/{key}_h"]).max() < 1e-15, key
    # the reference's rows, merged into a box
    lo, hi = np.full(nv, -np.inf), np.full(nv, np.inf)
    for key in ("cl", "vl"):
        for row, bound in zip(g[f"{n}/{key}_G"], g[f"{n}/{key}_h"]):
            i = int(np.nonzero(row)[0][0])
            if row[i] > 0:
                hi[i] = min(hi[i], bound / row[i])
            else:
                lo[i] = max(lo[i], bound / row[i])
    arrays = ModelArrays(m, ["tool0"])
    dm = emu.model_create(arrays.desc)
    d_q, d_qt = emu.alloc(8 * m.nq), emu.alloc(8 * m.nq)
    d_lb, d_ub, d_e = emu.alloc(8 * nv), emu.alloc(8 * nv), emu.alloc(8 * nv)
    emu.put(d_q, cfg.q.reshape(1, -1))
    emu.put(d_qt, m.neutral())
    emu.limits_posture(dm, 1, dt, 0.7, d_q, d_qt, 0, d_lb, d_ub, d_e, nv, 0)
    emu.sync()
    lb, ub = np.zeros((1, nv)), np.zeros((1, nv))
    emu.get(lb, d_lb), emu.get(ub, d_ub)
    assert np.abs(lb[0] - lo).max() < 1e-15 and np.abs(ub[0] - hi).max() < 1e-15
    for p_ in (d_q, d_qt, d_lb, d_ub, d_e):
        emu.release(p_)
    emu.model_destroy(dm)


@pytest.mark.parametrize("case", ["cl_arm", "cl_humanoid"])
def test_check_limits_follows_the_reference_method(golden4, case, emu):
    """pink.Configuration.check_limits of the reference (configuration.py:166-201), called on stand-ins by
    make_golden_round4.py: inside, outside by less / more than the tolerance, two offending entries (the first one is
    reported), the root joint's coordinates (never checked) -- pink_amd's Configuration.check_limits, the vectorised check
    of a ConfigurationBatch and the device kernel (pinkhip_check_limits_device) give the same verdicts."""
    from pink_amd import Configuration, ConfigurationBatch, build_chain
    from pink_amd.exceptions import NotWithinConfigurationLimits
    from pink_amd.rollout import ModelArrays

    g = golden4
    m = build_chain(6, free_flyer=case == "cl_humanoid", seed=3, limit=1.0)
    arrays = ModelArrays(m, ["tool0"])
    dm = emu.model_create(arrays.desc)
    d_q = emu.alloc(8 * m.nq)
    for q, verdict in zip(g[f"{case}/q"], g[f"{case}/verdict"]):
        want = int(verdict[0])
        for check in (lambda: Configuration(m, q.copy()).check_limits(), lambda: ConfigurationBatch(m, q[None].copy()).check_limits()):
            if want < 0:
                check()
            else:
                with pytest.raises(NotWithinConfigurationLimits) as ei:
                    check()
                assert (ei.value.joint, ei.value.value, ei.value.lower, ei.value.upper) == (want, verdict[1], verdict[2], verdict[3])
        emu.put(d_q, q[None].copy())
        assert emu.check_limits(dm, 1, d_q) == want  # (b * nq + i with b = 0, or -1)
    emu.release(d_q)
    emu.model_destroy(dm)


@pytest.mark.parametrize("where", ["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_whole_stack_against_the_references_build_ik(golden4, where, request):
    """The reference's OWN pink.build_ik (pink/solve_ik.py:152-203) over its OWN FrameTask x 3, RelativeFrameTask,
    PostureTask, JointCouplingTask x 2, DampingTask, ConfigurationLimit (explicit gain), VelocityLimit, AccelerationLimit,
    FloatingBaseVelocityLimit and PositionBarrier objects on a floating-base robot (make_golden_round4.py: kinematics from
    this repo's stand-in, pin.log / pin.Jlog6 from the matrix logarithm): (P, q, G, h) of pink_amd.build_ik row for row,
    and the velocity of the device route -- every one of these terms formed on chip by the whole-step kernel -- as the
    minimiser of THAT QP (KKT residuals, and the C oracle's solution of it)."""
    import pink_amd
    from pink_amd import Configuration, ConfigurationBatch, DampingTask, FrameTask, PostureTask, build_chain, solve_ik_batch
    from pink_amd.barriers import PositionBarrier
    from pink_amd.lie import SE3, exp6
    from pink_amd.limits import AccelerationLimit, ConfigurationLimit, FloatingBaseVelocityLimit, VelocityLimit
    from pink_amd.runtime import set_default_solver
    from pink_amd.tasks import JointCouplingTask, RelativeFrameTask

    g = golden4
    m = build_chain(12, free_flyer=True, seed=21, limit=2.5, velocity=6.0)
    root_id = m.joints.index(m.root_joint)
    m.add_frame("base", root_id, exp6(np.array([0.02, 0.0, 0.05, 0.1, -0.2, 0.3])))
    m.add_frame("mid", m.getJointId("joint_5"), SE3(np.eye(3), [0.0, 0.05, 0.1]))
    cfg, dt = Configuration(m, g["full/q"].copy()), float(g["full/dt"])
    se3 = lambda v: SE3(v[:9].reshape(3, 3), v[9:])  # noqa: E731
    tasks = []
    for k, (frame, pc, oc, lm, gain) in enumerate((("tool0", 1.0, 1.0, 1e-3, 1.0), ("joint_4", [1.0, 2.0, 0.5], 0.0, 0.0, 0.85), ("joint_9", 4.0, 4.0, 1e-2, 0.5))):
        t = FrameTask(frame, pc, oc, lm_damping=lm, gain=gain)
        t.set_target(se3(g[f"full/target_frame{k}"]))
        tasks.append(t)
    rt = RelativeFrameTask("tool0", "mid", 0.8, 0.3, lm_damping=1e-3, gain=0.9)
    rt.set_target(se3(g["full/target_rel"]))
    posture = PostureTask(cost=1e-1)
    posture.set_target(m.neutral())
    tasks += [rt, posture, JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 100.0, cfg, lm_damping=5e-7),
              JointCouplingTask(["joint_7", "joint_8"], [1.0, -1.0], 100.0, cfg, lm_damping=5e-7), DampingTask(cost=1e-2)]
    acc = AccelerationLimit(m, g["full/a_max"].copy())
    acc.set_last_integration(g["full/v_prev"], dt)
    fb = FloatingBaseVelocityLimit(m, "base", [0.4, 0.3, 0.5], 0.8)
    m.ensure_limits()
    m.floating_base_velocity_limit = fb
    limits = [ConfigurationLimit(m, 0.6), VelocityLimit(m), acc, fb]
    bar = PositionBarrier("tool0", indices=[2], p_max=np.array([float(g["full/bar_pmax"])]), gain=np.array([50.0]), safe_displacement_gain=1.0)
    P, c, G, h = g["full/P"], g["full/c"], g["full/G"], g["full/h"]
    set_default_solver(request.getfixturevalue("emu" if where == "emu" else "gpu_solver"))
    try:
        pr = pink_amd.build_ik(cfg, tasks, dt, damping=1e-12, limits=limits, barriers=[bar])
        assert np.abs(pr.P - P).max() < 1e-13 * np.abs(P).max() and np.abs(pr.q - c).max() < 1e-13 * max(1.0, np.abs(c).max())
        assert pr.G.shape == G.shape and np.abs(pr.G - G).max() < 1e-11 * max(1.0, np.abs(G).max()) and np.abs(pr.h - h).max() < 1e-12
        # the device route: everything above formed on chip from q, the targets and tables
        cb = ConfigurationBatch(m, np.tile(cfg.q, (2, 1)))
        V = solve_ik_batch(cb, tasks, dt, limits=limits, barriers=[bar], device_kinematics=True)
        assert pink_amd.last_solve_stats()["route"] == "device"
        x = V[0] * dt
        stat, viol, _ = po.kkt_residuals(P, c, G, h, x)
        scale = max(1.0, float(np.abs(c).max()), float(np.abs(P).max() * np.abs(x).max()))
        assert viol < 1e-10 and stat < 1e-10 * scale, (stat, viol)  # (north_star's tolerance is 1e-8: two decimals inside it)
        x_ref, st, _, _ = c_oracle.gi_solve(P, c, G, h)  # the reference's QP through the C restatement of Goldfarb-Idnani
        assert st == 0 and np.abs(x - x_ref).max() < 1e-10 * max(1e-3, np.abs(x_ref).max()), np.abs(x - x_ref).max()
        assert np.abs(V[1] - V[0]).max() == 0.0
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)


@pytest.mark.parametrize("where", ["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_arm_stack_with_previous_step_state_against_the_references_build_ik(golden4, where, request):
    """The reference's build_ik over its own FrameTask x 2, PostureTask, LowAccelerationTask, JointVelocityTask (the class
    whose error sign this file's generator corrected) under ConfigurationLimit + VelocityLimit + AccelerationLimit on a
    7-joint arm: pink_amd.build_ik row for row, and the device route's velocity as the minimiser of that QP (the arm runs
    the whole-step kernel at NV = 12 because the stack needs it)."""
    import pink_amd
    from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik_batch
    from pink_amd.lie import SE3
    from pink_amd.limits import AccelerationLimit, ConfigurationLimit, VelocityLimit
    from pink_amd.runtime import set_default_solver
    from pink_amd.tasks import JointVelocityTask, LowAccelerationTask

    g = golden4
    m = build_chain(7, free_flyer=False, seed=31, limit=2.2, velocity=5.0)
    cfg, dt = Configuration(m, g["arm/q"].copy()), float(g["arm/dt"])
    se3 = lambda v: SE3(v[:9].reshape(3, 3), v[9:])  # noqa: E731
    ft = FrameTask("tool0", [1.0, 1.0, 2.0], 0.2, lm_damping=1e-2, gain=0.7)
    ft.set_target(se3(g["arm/target0"]))
    ft2 = FrameTask("joint_4", 0.5, 0.0)
    ft2.set_target(se3(g["arm/target1"]))
    posture = PostureTask(cost=5e-2, gain=0.8)
    posture.set_target(g["arm/q_star"])
    la = LowAccelerationTask(cost=0.05)
    la.set_last_integration(g["arm/v_prev"], dt)
    jv = JointVelocityTask(cost=0.08)
    jv.set_target(g["arm/v_t"], dt)
    acc = AccelerationLimit(m, g["arm/a_max"].copy())
    acc.set_last_integration(g["arm/v_prev"], dt)
    tasks, limits = [ft, ft2, posture, la, jv], [ConfigurationLimit(m), VelocityLimit(m), acc]
    P, c, G, h = g["arm/P"], g["arm/c"], g["arm/G"], g["arm/h"]
    set_default_solver(request.getfixturevalue("emu" if where == "emu" else "gpu_solver"))
    try:
        pr = pink_amd.build_ik(cfg, tasks, dt, damping=1e-12, limits=limits)
        assert np.abs(pr.P - P).max() < 1e-13 * np.abs(P).max() and np.abs(pr.q - c).max() < 1e-13 * max(1.0, np.abs(c).max())
        assert pr.G.shape == G.shape and np.abs(pr.G - G).max() < 1e-12 and np.abs(pr.h - h).max() < 1e-13
        V = solve_ik_batch(ConfigurationBatch(m, np.tile(cfg.q, (3, 1))), tasks, dt, limits=limits, device_kinematics=True)
        assert pink_amd.last_solve_stats()["route"] == "device"
        x = V[0] * dt
        x_ref, st, _, _ = c_oracle.gi_solve(P, c, G, h)
        assert st == 0 and np.abs(x - x_ref).max() < 1e-10 * max(1e-3, np.abs(x_ref).max()), np.abs(x - x_ref).max()
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)


@pytest.mark.parametrize("where", ["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_equality_constraints_against_the_references_build_ik(golden4, where, request):
    """pink.build_ik(..., constraints=[FrameTask]) of the reference (solve_ik.py:125-149) on the 7-joint arm: A, b of
    pink_amd.build_ik row for row; the hybrid route (frame rows and the constraint's rows formed on the device) and the
    device route (the whole-step kernel: the constraint's rows are its leading equality rows) return the minimiser of that
    equality-constrained QP."""
    import pink_amd
    from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik_batch
    from pink_amd.lie import SE3
    from pink_amd.limits import ConfigurationLimit, VelocityLimit
    from pink_amd.runtime import set_default_solver

    g = golden4
    m = build_chain(7, free_flyer=False, seed=31, limit=2.2, velocity=5.0)
    cfg, dt = Configuration(m, g["arm/q"].copy()), float(g["arm/dt"])
    se3 = lambda v: SE3(v[:9].reshape(3, 3), v[9:])  # noqa: E731
    ft = FrameTask("tool0", [1.0, 1.0, 2.0], 0.2, lm_damping=1e-2, gain=0.7)
    ft.set_target(se3(g["arm/target0"]))
    posture = PostureTask(cost=5e-2, gain=0.8)
    posture.set_target(g["arm/q_star"])
    hold = FrameTask("joint_6", 1.0, 1.0, gain=0.6)
    hold.set_target(se3(g["eq/hold_target"]))
    limits = [ConfigurationLimit(m), VelocityLimit(m)]
    P, c, G, h, A, b = (g[f"eq/{k}"] for k in ("P", "c", "G", "h", "A", "b"))
    set_default_solver(request.getfixturevalue("emu" if where == "emu" else "gpu_solver"))
    try:
        pr = pink_amd.build_ik(cfg, [ft, posture], dt, damping=1e-12, limits=limits, constraints=[hold])
        assert np.abs(pr.P - P).max() < 1e-13 * np.abs(P).max() and np.abs(pr.q - c).max() < 1e-13 * max(1.0, np.abs(c).max())
        assert np.abs(pr.G - G).max() < 1e-12 and np.abs(pr.h - h).max() < 1e-13
        assert np.abs(pr.A - A).max() < 1e-13 and np.abs(pr.b - b).max() < 1e-13
        x_ref, st, _, _ = c_oracle.gi_solve(P, c, np.vstack([A, G]), np.hstack([b, h]), meq=A.shape[0])
        assert st == 0
        B = 70  # (from 64 configurations on the device routes are the automatic choice)
        cb = ConfigurationBatch(m, np.tile(cfg.q, (B, 1)))
        V = solve_ik_batch(cb, [ft, posture], dt, limits=limits, constraints=[hold], device_kinematics="frame_rows")
        assert pink_amd.last_solve_stats()["route"] == "hybrid"
        assert np.abs(V[0] * dt - x_ref).max() < 1e-10 * max(1e-3, np.abs(x_ref).max()) and np.abs(V - V[0]).max() == 0.0
        # round 5: the whole-step kernel forms the constraint's rows on chip as its leading equality rows
        V = solve_ik_batch(cb, [ft, posture], dt, limits=limits, constraints=[hold])
        assert pink_amd.last_solve_stats()["route"] == "device"
        assert np.abs(V[0] * dt - x_ref).max() < 1e-10 * max(1e-3, np.abs(x_ref).max()) and np.abs(V - V[0]).max() == 0.0
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)
