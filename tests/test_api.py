"""The Pink-style Python surface (solve_ik, build_ik, Task, Limit, Barrier) on top
of the packed-batch path.  Runs on the CPU wave emulator here and, marked ``gpu``,
on the MI355X.  Each test names the reference test it restates."""
import numpy as np
import pytest

import pink_amd
from oracle import c_oracle
from oracle import pink_oracle as po
from pink_amd import (Configuration, DampingTask, FrameTask, NoSolutionFound, NotWithinConfigurationLimits, PostureTask,
                      TargetNotSet, build_chain, build_ik, solve_ik, solve_ik_batch)
from pink_amd.barriers import PositionBarrier
from pink_amd.lie import SE3, log6
from pink_amd.limits import ConfigurationLimit, VelocityLimit
from pink_amd.runtime import set_default_solver
from pink_amd.tasks import LowAccelerationTask

Q0 = np.array([0.3, -0.8, 1.2, 0.4, -0.3, 0.5])


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "emu":
        set_default_solver(request.getfixturevalue("emu"))
    else:
        set_default_solver(request.getfixturevalue("gpu_solver"))
    yield request.param
    set_default_solver(None)


def _arm():
    m = build_chain(6)
    return m, Configuration(m, Q0)


def test_frame_jacobians_by_finite_differences():
    """tests/test_jacobians.py:47-84 (h = 1e-6, tol 1e-5)."""
    m, cfg = _arm()
    task = FrameTask("tool0", 1.0, 1.0)
    tgt = cfg.get_transform_frame_to_world("tool0")
    tgt.translation += [0.05, -0.1, 0.02]
    task.set_target(tgt)
    J = task.compute_jacobian(cfg)
    for k in range(6):
        d = np.zeros(6)
        d[k] = 1e-6
        fd = (task.compute_error(Configuration(m, Q0 + d)) - task.compute_error(Configuration(m, Q0 - d))) / 2e-6
        assert np.abs(J[:, k] - fd).max() < 1e-5


def test_target_not_set_and_zero_error_at_target():
    """tests/test_frame_task.py:112-121, pink/tasks/frame_task.py:176-177."""
    m, cfg = _arm()
    task = FrameTask("tool0", 1.0, 1.0)
    with pytest.raises(TargetNotSet):
        task.compute_error(cfg)
    task.set_target_from_configuration(cfg)
    assert np.linalg.norm(task.compute_error(cfg)) < 1e-10
    assert np.allclose(task.compute_jacobian(cfg), -cfg.get_frame_jacobian("tool0"))  # Jlog6(I) = I


def test_task_qp_objective_known_answers(backend):
    """tests/test_frame_task.py:123-141; test_low_acceleration_task.py:34-42; test_damping_task.py:34-39."""
    m, cfg = _arm()
    task = FrameTask("tool0", 1.0, 1.0)
    tgt = cfg.get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.0, 0.01, 0.0])
    task.set_target(tgt)
    J, e = task.compute_jacobian(cfg), task.compute_error(cfg)
    H, c = task.compute_qp_objective(cfg)
    assert np.allclose(J.T @ J, H) and np.allclose(e.T @ J, c)
    la = LowAccelerationTask(cost=1.0)
    v_prev = np.array([1.0, 2.0, 3.0, 4.0, -3.0, -2.0])
    la.set_last_integration(v_prev, 1.234e-2)
    H, c = la.compute_qp_objective(cfg)
    assert np.linalg.norm(H - np.eye(6)) < 1e-10 and np.linalg.norm(c + v_prev * 1.234e-2) < 1e-10
    H, c = DampingTask(cost=1.0).compute_qp_objective(cfg)
    assert np.array_equal(H, np.eye(6)) and not c.any()


def test_limits_shapes_and_none():
    """tests/test_limits.py:22-66, tests/test_velocity_limit.py:46-55."""
    m, cfg = _arm()
    for lim in (ConfigurationLimit(m), VelocityLimit(m)):
        G, h = lim.compute_qp_inequalities(cfg, 1e-3)
        assert G.shape == (12, 6) and h.shape == (12,)
    G, h = VelocityLimit(m, np.full(6, 2.0)).compute_qp_inequalities(cfg, 1e-3)
    assert np.allclose(h, 2e-3)
    free = build_chain(3, limit=np.inf, velocity=np.inf)
    c3 = Configuration(free, np.zeros(3))
    assert ConfigurationLimit(free).compute_qp_inequalities(c3, 1e-3) is None
    assert VelocityLimit(free).compute_qp_inequalities(c3, 1e-3) is None
    assert build_ik(c3, [], dt=1.0).G is None  # tests/test_solve_ik.py:67-77


def test_check_limits_raises_or_warns(backend):
    """tests/test_solve_ik.py:39-65."""
    m = build_chain(6, limit=1.0)
    cfg = Configuration(m, np.array([0.0, 1.5, 0, 0, 0, 0]))
    with pytest.raises(NotWithinConfigurationLimits):
        solve_ik(cfg, [], dt=1e-3)
    solve_ik(cfg, [], dt=1e-3, safety_break=False, limits=[])  # warns only


def test_check_limits_warns_about_every_violated_joint(caplog):
    """pink/configuration.py:183-201: without the safety break every joint out of its limits is logged, not only the first."""
    import logging

    m = build_chain(6, limit=1.0)
    cfg = Configuration(m, np.array([0.0, 1.5, 0.0, -1.2, 0.0, 2.0]))
    with caplog.at_level(logging.WARNING):
        cfg.check_limits(safety_break=False)
    hits = [r for r in caplog.records if "out of limits" in r.getMessage()]
    assert len(hits) == 3 and [r.args[1] for r in hits] == [1, 3, 5]


def test_no_task_and_fulfilled_task_give_zero_velocity(backend):
    """tests/test_solve_ik.py:79-102."""
    m, cfg = _arm()
    assert np.allclose(solve_ik(cfg, [], dt=1e-3), 0.0)
    task = FrameTask("tool0", 1.0, 1.0)
    task.set_target_from_configuration(cfg)
    assert np.allclose(solve_ik(cfg, [task], dt=5e-3, damping=1e-8), 0.0)


def test_build_ik_matches_oracle_and_solve_matches(backend):
    m, cfg = _arm()
    task = FrameTask("tool0", [1.0, 2.0, 0.5], 0.3, lm_damping=0.1, gain=0.8)
    post = PostureTask(cost=1e-2)
    task.set_target(cfg.get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.1, 0.2, -0.1]))
    post.set_target(np.zeros(6))
    dt = 5e-3
    prob = build_ik(cfg, [task, post], dt)
    tasks = [(task.compute_jacobian(cfg), task.compute_error(cfg), task.cost, 0.8, 0.1),
             (post.compute_jacobian(cfg), post.compute_error(cfg), 1e-2, 1.0, 0.0)]
    P, q = po.qp_objective(6, tasks, 1e-12)
    assert np.allclose(prob.P, P, rtol=1e-13, atol=1e-15) and np.allclose(prob.q, q, rtol=1e-13, atol=1e-15)
    assert prob.G.shape == (24, 6)
    v = solve_ik(cfg, [task, post], dt)
    x, st, _, _ = c_oracle.gi_solve(P, q, prob.G, prob.h)
    assert st == 0 and np.abs(v * dt - x).max() < 1e-11


def test_closed_loop_convergence(backend):
    """tests/test_solve_ik.py:160-210 (error decreases monotonically and converges)."""
    m, cfg = _arm()
    task = FrameTask("tool0", 1.0, 1.0, lm_damping=1e-3)
    post = PostureTask(cost=1e-3)
    for t in (task, post):
        t.set_target_from_configuration(cfg)
    tgt = task.transform_target_to_world.copy()
    tgt.translation[1] += 0.1
    task.set_target(tgt)
    dt, errs = 5e-3, []
    for _ in range(120):
        cfg.integrate_inplace(solve_ik(cfg, [task, post], dt), dt)
        errs.append(np.linalg.norm(task.compute_error(cfg)))
    big = [e for e in errs if e > 1e-4]  # monotone until the posture task balances the residual
    assert errs[-1] < 1e-3 and all(b < a for a, b in zip(big, big[1:]))


def test_position_barrier(backend):
    """tests/test_solve_ik.py:104-158: satisfied barrier => v = 0, active barrier stops the motion."""
    m, cfg = _arm()
    y0 = cfg.get_transform_frame_to_world("tool0").translation[1]
    task = FrameTask("tool0", 1.0, 1.0)
    task.set_target_from_configuration(cfg)
    bar = PositionBarrier("tool0", indices=[1], p_max=np.array([y0 + 0.01]), gain=np.array([100.0]), safe_displacement_gain=1.0)
    assert bar.compute_qp_inequalities(cfg, 1e-3)[0].shape == (1, 6)
    assert np.allclose(solve_ik(cfg, [task], dt=5e-3, barriers=[bar]), 0.0, atol=1e-9)
    tgt = task.transform_target_to_world.copy()
    tgt.translation[1] += 0.2
    task.set_target(tgt)
    dt = 5e-3
    for _ in range(40):
        cfg.integrate_inplace(solve_ik(cfg, [task], dt, barriers=[bar]), dt)
    assert cfg.get_transform_frame_to_world("tool0").translation[1] <= y0 + 0.01 + 1e-4


def test_solve_ik_batch_equals_loop_and_reports_failures(backend):
    m = build_chain(6)
    rng = np.random.default_rng(3)
    cfgs = [Configuration(m, Q0 + 0.2 * rng.normal(size=6)) for _ in range(5)]
    task = FrameTask("tool0", 1.0, 1.0, lm_damping=0.5)
    post = PostureTask(cost=1e-3)
    task.set_target(cfgs[0].get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.0, 0.1, 0.05]))
    post.set_target(Q0)
    V = solve_ik_batch(cfgs, [task, post], 5e-3)
    for b, cfg in enumerate(cfgs):
        assert np.abs(V[b] - solve_ik(cfg, [task, post], 5e-3)).max() < 1e-9  # frame terms from the GPU kernel
    Vh = solve_ik_batch(cfgs, [task, post], 5e-3, gpu_frame_tasks=False)
    for b, cfg in enumerate(cfgs):
        # (host-evaluated terms: the batched evaluators of pink_amd/batch_eval.py sum in another order than the
        # per-configuration methods -- round-off apart, not bitwise)
        assert np.abs(Vh[b] - solve_ik(cfg, [task, post], 5e-3)).max() < 1e-10

    class Crossed(pink_amd.limits.Limit):
        def compute_qp_inequalities(self, configuration, dt):
            G = np.zeros((2, 6))
            G[0, 2], G[1, 2] = 1.0, -1.0
            return G, np.array([-1.0, -1.0])  # dq_2 <= -1 and dq_2 >= 1

    with pytest.raises(NoSolutionFound) as ei:
        solve_ik_batch(cfgs, [task, post], 5e-3, limits=[Crossed()])
    assert ei.value.indices.tolist() == [0, 1, 2, 3, 4] and (ei.value.status == 2).all()
    # the reference's solver strings (pink/solve_ik.py:210, README "quadprog", tests "daqp" / "proxqp") are aliases of
    # the MI355X solver: a Pink call runs unchanged; anything else is refused by name
    v_ref = solve_ik(cfgs[0], [task], 5e-3)
    for name in ("quadprog", "daqp", "proxqp", "osqp"):
        assert np.array_equal(solve_ik(cfgs[0], [task], 5e-3, solver=name), v_ref)
    assert np.array_equal(solve_ik_batch(cfgs, [task, post], 5e-3, solver="quadprog"), V)
    with pytest.raises(pink_amd.PinkError, match="not_a_solver"):
        solve_ik(cfgs[0], [task], 5e-3, solver="not_a_solver")
    # per-instance task lists of different lengths are an error (not silently cut to the first one's length)
    ragged = [[task, post] for _ in cfgs]
    ragged[3] = [task, post, PostureTask(cost=1e-2)]
    ragged[3][2].set_target_from_configuration(cfgs[3])
    for dk in (False, None):
        with pytest.raises(pink_amd.PinkError, match="same length"):
            solve_ik_batch(cfgs, ragged, 5e-3, device_kinematics=dk)
    ragged[3] = [task]
    with pytest.raises(pink_amd.PinkError, match="same length"):
        solve_ik_batch(cfgs, ragged, 5e-3)


def test_equality_constraints_via_constraints_argument(backend):
    """pink/solve_ik.py:125-149: a task passed in constraints= is enforced exactly: J dq = -gain e."""
    m, cfg = _arm()
    hold = FrameTask("tool0", 1.0, 0.0)  # keep the tool position ...
    hold.set_target_from_configuration(cfg)
    post = PostureTask(cost=1.0)  # ... while the posture task pulls the joints to zero
    post.set_target(np.zeros(6))
    dt = 5e-3

    class PositionOnly(pink_amd.Task):
        def __init__(self, inner):
            super().__init__(gain=1.0)
            self.inner = inner

        def compute_error(self, configuration):
            return self.inner.compute_error(configuration)[:3]

        def compute_jacobian(self, configuration):
            return self.inner.compute_jacobian(configuration)[:3]

        def __repr__(self):
            return "PositionOnly()"

    con = PositionOnly(hold)
    prob = build_ik(cfg, [post], dt, constraints=[con])
    assert prob.A.shape == (3, 6) and prob.b.shape == (3,) and prob.batch.n_eq == 3
    v = solve_ik(cfg, [post], dt, constraints=[con])
    assert np.abs(con.compute_jacobian(cfg) @ (v * dt) + con.compute_error(cfg)).max() < 1e-12
    assert np.linalg.norm(v) > 1e-3  # the posture task still moves the arm in the null space
    v_free = solve_ik(cfg, [post], dt)
    assert np.linalg.norm(con.compute_jacobian(cfg) @ v_free) > 1e-3  # without the constraint the tool moves


def test_frame_task_kernel_matches_host_lie_and_oracle(backend):
    """pink/tasks/frame_task.py:176-227 evaluated for a batch by the HIP frame-task kernel."""
    from oracle import se3_oracle
    from pink_amd.lie import Jlog6, exp6
    from pink_amd.runtime import default_solver
    from pink_amd.solve_ik import _pose12

    rng = np.random.default_rng(5)
    for nv in (6, 13, 30, 50):
        B = 9
        frames = [exp6(rng.normal(size=6)) for _ in range(B)]
        offs = [exp6(rng.normal(size=6) * [0.1, 0.1, 0.1, 0.6, 0.6, 0.6]) for _ in range(B)]
        offs[0] = SE3()  # zero error
        offs[1] = exp6(np.array([0.0, 0.0, 0.0, 0.0, 0.0, 3.14159]))  # rotation close to pi
        targets = [f * o for f, o in zip(frames, offs)]
        Jb = rng.normal(size=(B, 6, nv))
        Tf = np.array([_pose12(f) for f in frames])
        Tt = np.array([_pose12(t) for t in targets])
        e, J = default_solver().frame_task_terms(Tf, Tt, Jb)
        for b in range(B):  # host closed forms (pink_amd.lie, what FrameTask uses per instance)
            assert np.abs(e[b] - log6(frames[b].actInv(targets[b]))).max() < 1e-12
            assert np.abs(J[b] + Jlog6(targets[b].actInv(frames[b])) @ Jb[b]).max() < 1e-11
        eo, Jo = se3_oracle.frame_task_terms(Tf[2:], Tt[2:], Jb[2:])  # independent: logm + finite differences
        assert np.abs(e[2:] - eo).max() < 1e-12 and np.abs(J[2:] - Jo).max() < 1e-7
        assert np.array_equal(J[0], -Jb[0]) and not e[0].any()  # tests/test_frame_task.py:112-121
    # small orientation errors, where a tracking controller lives: the coefficient functions of log6 / Jlog6 are power
    # series there (tests/test_lie_accuracy.py holds the host's to 60-digit arithmetic); their closed forms gave 1e-10
    # between the GPU's sin / cos and NumPy's at 1e-3 rad, 1e-8 on dq
    nv = 30
    ths = np.r_[10.0 ** np.arange(-6, -0.2, 0.5), 0.45, 0.49, 0.51]
    frames = [exp6(rng.normal(size=6)) for _ in ths]
    offs = []
    for th in ths:
        w = rng.normal(size=3)
        offs.append(exp6(np.r_[rng.normal(size=3) * 0.05, w * th / np.linalg.norm(w)]))
    targets = [f * o for f, o in zip(frames, offs)]
    Jb = rng.normal(size=(len(ths), 6, nv))
    e, J = default_solver().frame_task_terms(np.array([_pose12(f) for f in frames]), np.array([_pose12(t) for t in targets]), Jb)
    for b in range(len(ths)):
        assert np.abs(e[b] - log6(frames[b].actInv(targets[b]))).max() < 1e-14, ths[b]
        assert np.abs(J[b] + Jlog6(targets[b].actInv(frames[b])) @ Jb[b]).max() < 2e-13, ths[b]


def test_batched_frame_tasks_on_gpu_equal_host_evaluation(backend):
    m = build_chain(6)
    rng = np.random.default_rng(7)
    cfgs = [Configuration(m, Q0 + 0.3 * rng.normal(size=6)) for _ in range(6)]
    task = FrameTask("tool0", [1.0, 2.0, 3.0], 0.5, lm_damping=0.2)
    post = PostureTask(cost=1e-2)
    task.set_target(cfgs[0].get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.05, 0.1, 0.0]))
    post.set_target(Q0)
    Vg = solve_ik_batch(cfgs, [task, post], 5e-3, gpu_frame_tasks=True)
    Vh = solve_ik_batch(cfgs, [task, post], 5e-3, gpu_frame_tasks=False)
    assert np.abs(Vg - Vh).max() < 1e-9
    # per-instance targets: one task list per configuration
    per = []
    for cfg in cfgs:
        t = FrameTask("tool0", 1.0, 1.0)
        t.set_target(cfg.get_transform_frame_to_world("tool0") * SE3(np.eye(3), 0.05 * rng.normal(size=3)))
        per.append([t, post])
    Vp = solve_ik_batch(cfgs, per, 5e-3)
    for b, cfg in enumerate(cfgs):
        assert np.abs(Vp[b] - solve_ik(cfg, per[b], 5e-3)).max() < 1e-9


def test_acceleration_limit(backend):
    """tests/test_acceleration_limit.py:34-94: shapes, None for limit-less models, self-consistency
    for joints without configuration limits, and the limit changes the IK solution."""
    from pink_amd.limits import AccelerationLimit

    m, cfg = _arm()
    lim = AccelerationLimit(m, np.full(6, 14.0))
    assert lim.projection_matrix.shape == (6, 6)
    G, h = lim.compute_qp_inequalities(cfg, 5e-3)
    assert G.shape == (12, 6) and h.shape == (12,)
    free = build_chain(1, limit=np.inf, velocity=np.inf)  # a "continuous" joint
    cf = Configuration(free, np.zeros(1))
    lf = AccelerationLimit(free, np.array([14.0]))
    lf.set_last_integration(np.array([3.0]), 5e-3)
    G, h = lf.compute_qp_inequalities(cf, 5e-3)
    assert (-h[1:] <= h[:1]).all()  # lower bound not above upper bound
    assert AccelerationLimit(free, np.array([np.inf])).compute_qp_inequalities(cf, 5e-3) is None
    task = FrameTask("tool0", 1.0, 1.0)
    task.set_target(cfg.get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.0, 0.3, 0.0]))
    dt = 5e-3
    v_free = solve_ik(cfg, [task], dt, damping=1e-6, limits=[])
    v_lim = solve_ik(cfg, [task], dt, damping=1e-6, limits=[lim])
    assert np.abs(v_lim).max() <= 14.0 * dt + 1e-9 and np.abs(v_free).max() > np.abs(v_lim).max()


def test_floating_base_velocity_limit(backend):
    """tests/test_floating_base_velocity_limit.py: rows only touch the root columns; the base twist obeys the bound."""
    from pink_amd.limits import FloatingBaseVelocityLimit

    m = build_chain(4, free_flyer=True)
    cfg = Configuration(m, m.neutral())
    lim = FloatingBaseVelocityLimit(m, None, max_linear_velocity=0.1, max_angular_velocity=[0.2, np.inf, 0.2])
    G, h = lim.compute_qp_inequalities(cfg, 1e-2)
    assert G.shape == (10, m.nv) and not G[:, 6:].any() and np.allclose(h[:3], 1e-3)
    with pytest.raises(ValueError):
        FloatingBaseVelocityLimit(build_chain(3), None, 1.0, 1.0)
    task = FrameTask("tool0", 1.0, 0.0)
    task.set_target(cfg.get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.5, 0.2, 0.1]))
    v = solve_ik(cfg, [task], 1e-2, damping=1e-6, limits=[lim])
    assert np.abs(v[:3]).max() <= 0.1 + 1e-9 and abs(v[3]) <= 0.2 + 1e-9 and abs(v[5]) <= 0.2 + 1e-9
    assert np.abs(v[:3]).max() > 0.09  # the bound is active: the target is far


def test_urdf_reader_on_reference_robots():
    import os

    path = "/root/reference/examples/robots/double_pendulum.urdf"
    if not os.path.exists(path):
        pytest.skip("reference checkout not mounted")
    m = pink_amd.load_urdf(path)
    assert m.nv == 2 and m.nq == 2
    cfg = Configuration(m, np.array([0.3, -0.4]))
    J = cfg.get_frame_jacobian(m.frames[-1].name)
    assert J.shape == (6, 2) and np.abs(J).max() > 0


def test_solve_ik_batch_ragged_dense_rows_safe_displacement_and_constraints(backend):
    """pack_configurations must not key the batch's dense rows / safe displacements on instance 0:
    a limit that yields dense rows only at some configurations, a barrier whose safe displacement is zero at
    others, and constraints= all have to match the per-instance solve_ik."""
    m = build_chain(6)
    rng = np.random.default_rng(8)
    cfgs = [Configuration(m, Q0 + 0.2 * rng.normal(size=6)) for _ in range(5)]
    task = FrameTask("tool0", 1.0, 1.0, lm_damping=0.1)
    task.set_target(cfgs[0].get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.0, 0.004, 0.002]))
    post = PostureTask(cost=1e-2)
    post.set_target(Q0)
    free0 = solve_ik_batch(cfgs, [task, post], 5e-3, gpu_frame_tasks=False) * 5e-3  # dq without the extra limit

    class SometimesDense(pink_amd.limits.Limit):
        """No row at instance 0, an axis-aligned row at instance 1 (goes into the box), k - 1 dense rows at
        instance k >= 2; the first dense row cuts the unconstrained step in half along its own direction."""

        def compute_qp_inequalities(self, configuration, dt):
            k = int(np.argmin([np.abs(configuration.q - c.q).max() for c in cfgs]))
            if k == 0:
                return None
            if k == 1:
                G = np.zeros((1, 6))
                G[0, 3] = 2.0
                return G, np.array([1e-3])
            G = np.ones((k - 1, 6)) + 0.1 * np.arange(6)[None] * np.arange(1, k)[:, None]
            h = np.full(k - 1, 1.0)
            G[0] = free0[k] / np.linalg.norm(free0[k])
            h[0] = 0.5 * np.linalg.norm(free0[k])
            return G, h

    class SometimesSafe(PositionBarrier):
        def compute_safe_displacement(self, configuration):
            k = int(np.argmin([np.abs(configuration.q - c.q).max() for c in cfgs]))
            return np.zeros(6) if k in (0, 3) else 1e-3 * (k + 1) * np.arange(1.0, 7.0)

    bar = SometimesSafe("tool0", indices=[1], p_max=np.array([10.0]), gain=np.array([100.0]), safe_displacement_gain=2.0)
    lims = [m.configuration_limit, m.velocity_limit, SometimesDense()]
    V = solve_ik_batch(cfgs, [task, post], 5e-3, limits=lims, barriers=[bar], gpu_frame_tasks=False)
    for b, cfg in enumerate(cfgs):
        assert np.abs(V[b] - solve_ik(cfg, [task, post], 5e-3, limits=lims, barriers=[bar])).max() < 1e-11, b
    # the limit really constrains the later instances (the test would be vacuous otherwise)
    free = solve_ik_batch(cfgs, [task, post], 5e-3, barriers=[bar], gpu_frame_tasks=False)
    assert np.abs(free[3] - V[3]).max() > 1e-6
    # constraints= on the batched entry point (pink/solve_ik.py:125-149)
    # (one constraint list per instance: each robot moves its own tool position by 1 mm, exactly, while the
    # posture task pulls; position rows only -- six equalities on nv = 6 would leave no freedom for the box)
    class PositionOf(pink_amd.tasks.Task):
        def __init__(self, frame_task):
            super().__init__(cost=1.0, gain=frame_task.gain)
            self.t = frame_task

        def compute_error(self, configuration):
            return self.t.compute_error(configuration)[:3]

        def compute_jacobian(self, configuration):
            return self.t.compute_jacobian(configuration)[:3]

        def __repr__(self):
            return "PositionOf()"

    holds = []
    for cfg in cfgs:
        ft = FrameTask("tool0", 1.0, 1.0)
        ft.set_target(cfg.get_transform_frame_to_world("tool0") * SE3(np.eye(3), [0.0, 1e-3, 0.0]))
        holds.append([PositionOf(ft)])
    Vc = solve_ik_batch(cfgs, [post], 5e-3, constraints=holds, gpu_frame_tasks=False)
    for b, cfg in enumerate(cfgs):
        hold = holds[b][0]
        assert np.abs(Vc[b] - solve_ik(cfg, [post], 5e-3, constraints=[hold])).max() < 1e-10
        J, e = hold.compute_jacobian(cfg), hold.compute_error(cfg)
        assert np.abs(J @ (Vc[b] * 5e-3) + hold.gain * e).max() < 1e-12
    assert np.abs(Vc).max() > 1e-3


def test_solve_ik_batch_device_kinematics_equals_host_path(backend):
    """FrameTasks + PostureTask under the default limits: FK, task rows and limits evaluated by the device kernels
    from q alone must give the velocities of the host-evaluated path (same QP, rows computed in another order)."""
    for m, frames in ((build_chain(6), ["tool0"]), (build_chain(9, free_flyer=True, seed=3), ["tool0", "joint_4"])):
        rng = np.random.default_rng(31)
        B = 7
        cfgs = []
        for _ in range(B):
            q = m.neutral()
            for j in m.joints:
                if j.kind != "free_flyer":
                    q[j.idx_q] = rng.uniform(-0.9, 0.9)
            cfgs.append(Configuration(m, q))
        per_instance = []
        for cfg in cfgs:
            tl = []
            for k, f in enumerate(frames):
                t = FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9)
                t.set_target(cfg.get_transform_frame_to_world(f) * SE3(np.eye(3), 0.03 * rng.normal(size=3)))
                tl.append(t)
            p = PostureTask(cost=1e-2, gain=0.7)
            p.set_target(m.neutral())
            tl.append(p)
            per_instance.append(tl)
        V_host = solve_ik_batch(cfgs, per_instance, 5e-3, device_kinematics=False, gpu_frame_tasks=False)
        V_dev = solve_ik_batch(cfgs, per_instance, 5e-3, device_kinematics=True)
        assert np.abs(V_dev - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
        assert np.abs(V_host).max() > 1e-3
    # a PositionBarrier on a task frame stays on the device (its rows are formed on chip) ...
    bar = PositionBarrier("tool0", indices=[1], p_max=np.array([10.0]), gain=np.array([100.0]))
    V_bar = solve_ik_batch(cfgs, per_instance, 5e-3, barriers=[bar], device_kinematics=True)
    assert np.abs(V_bar - solve_ik_batch(cfgs, per_instance, 5e-3, barriers=[bar], device_kinematics=False, gpu_frame_tasks=False)).max() < 1e-8
    # ... so does one on a frame that carries no task (round 5: a zero-cost slot of the device model) ...
    other = PositionBarrier("joint_2", indices=[1], p_max=np.array([10.0]), gain=np.array([100.0]))
    V_other = solve_ik_batch(cfgs, per_instance, 5e-3, barriers=[other], device_kinematics=True)
    assert np.abs(V_other - solve_ik_batch(cfgs, per_instance, 5e-3, barriers=[other], device_kinematics=False, gpu_frame_tasks=False)).max() < 1e-8
    # ... not eligible: a barrier with its own class-K function
    odd = PositionBarrier("tool0", indices=[1], p_max=np.array([10.0]), gain=np.array([100.0]))
    odd.gain_function, odd.identity_gain_function = (lambda h: 2.0 * h), False
    with pytest.raises(pink_amd.PinkError):
        solve_ik_batch(cfgs, per_instance, 5e-3, barriers=[odd], device_kinematics=True)


def test_solve_ik_batch_on_arrays_equals_the_list_of_configurations(backend):
    """``ConfigurationBatch`` (q as one array) + per-instance targets as arrays: same velocities as the list of
    Configuration objects with per-instance task objects; a second call with other q re-uses the device state; the
    in-process sharder over two solver handles (``device_ids=`` / ``MultiDeviceSolver``) returns the same batch."""
    from pink_amd import ConfigurationBatch, MultiDeviceSolver
    from pink_amd.runtime import default_solver

    m, frames = build_chain(9, free_flyer=True, seed=3), ["tool0", "joint_4"]
    rng = np.random.default_rng(5)
    B = 9
    pink_amd.clear_device_cache()  # (the solver of the test session may hold the states of earlier tests)

    def draw():
        q = np.tile(m.neutral(), (B, 1))
        for j in m.joints:
            if j.kind != "free_flyer":
                q[:, j.idx_q] = rng.uniform(-0.9, 0.9, size=B)
        return q

    for trial in range(2):  # the second trial hits the cached device state with new configurations and targets
        q = draw()
        cfgs = [Configuration(m, q[b]) for b in range(B)]
        shared = [FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9) for k, f in enumerate(frames)]
        post = PostureTask(cost=1e-2, gain=0.7)
        qp = draw()
        post.set_target_batch(qp)
        per_instance = [[] for _ in range(B)]
        for k, f in enumerate(frames):
            R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
            for b, cfg in enumerate(cfgs):
                T = cfg.get_transform_frame_to_world(f) * SE3(np.eye(3), 0.03 * rng.normal(size=3))
                R[b], t[b] = T.rotation, T.translation
                ft = FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9)
                ft.set_target(T)
                per_instance[b].append(ft)
            shared[k].set_target_poses(R, t)
        for b in range(B):
            p = PostureTask(cost=1e-2, gain=0.7)
            p.set_target(qp[b])
            per_instance[b].append(p)
        V_list = solve_ik_batch(cfgs, per_instance, 5e-3, device_kinematics=True)
        V_arr = solve_ik_batch(ConfigurationBatch(m, q), shared + [post], 5e-3, device_kinematics=True)
        assert np.array_equal(V_arr, V_list) and np.abs(V_arr).max() > 1e-3
        V_host = solve_ik_batch(cfgs, per_instance, 5e-3, device_kinematics=False, gpu_frame_tasks=False)
        assert np.abs(V_arr - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
        # the array form through the host-evaluated path: per-instance targets are expanded to task objects
        V_arr_host = solve_ik_batch(ConfigurationBatch(m, q), shared + [post], 5e-3, device_kinematics=False, gpu_frame_tasks=False)
        assert np.array_equal(V_arr_host, V_host)
    assert len(default_solver()._pinkhip_rollouts) == 1  # one call shape: one cached device state
    # two handles driven from this process, contiguous shards (on the GPU box both handles sit on device 0)
    class Serialised:  # the CPU wave emulator is one global wavefront: its calls take turns
        import threading

        lock = threading.Lock()

        def __init__(self, inner):
            object.__setattr__(self, "_inner", inner)

        def __getattr__(self, name):
            attr = getattr(self._inner, name)
            if not callable(attr):
                return attr

            def call(*a, **k):
                with Serialised.lock:
                    return attr(*a, **k)

            return call

    mk = (lambda d: Serialised(default_solver())) if backend == "emu" else (lambda d: pink_amd.BatchSolver(device_id=0))
    pool = MultiDeviceSolver([0, 1], solver_factory=mk)
    try:
        V_pool = solve_ik_batch(ConfigurationBatch(m, q), shared + [post], 5e-3, device_kinematics=True, solver_handle=pool)
        assert np.array_equal(V_pool, V_arr)
        V_pool_host = solve_ik_batch(cfgs, per_instance, 5e-3, device_kinematics=False, gpu_frame_tasks=False, solver_handle=pool)
        assert np.array_equal(V_pool_host, V_host)
        with pytest.raises(pink_amd.PinkError):
            solve_ik_batch(cfgs, per_instance, 5e-3, solver_handle=pool, device_ids=[0])
        # the tasks the kernel forms from tables, a relative slot and an acceleration limit travel with every shard
        from pink_amd import DampingTask
        from pink_amd.limits import AccelerationLimit, ConfigurationLimit, VelocityLimit
        from pink_amd.tasks import JointCouplingTask, RelativeFrameTask

        jc = JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 10.0, cfgs[0], lm_damping=5e-7)
        rel = RelativeFrameTask("joint_8", "joint_4", 1.0, 0.3, lm_damping=1e-3)
        rel.set_target(cfgs[0].get_transform("joint_8", "joint_4") * SE3(np.eye(3), [0.01, 0.0, 0.02]))
        acc = AccelerationLimit(m, np.r_[np.full(6, np.inf), np.full(9, 300.0)])
        acc.set_last_integration(np.r_[np.zeros(6), 0.2 * rng.normal(size=9)], 5e-3)
        stack, lim = shared + [jc, rel, post, DampingTask(cost=1e-2)], [ConfigurationLimit(m), VelocityLimit(m), acc]
        V_one = solve_ik_batch(ConfigurationBatch(m, q), stack, 5e-3, device_kinematics=True, limits=lim)
        V_two = solve_ik_batch(ConfigurationBatch(m, q), stack, 5e-3, device_kinematics=True, limits=lim, solver_handle=pool)
        assert np.array_equal(V_two, V_one)
        V_ref = solve_ik_batch(ConfigurationBatch(m, q), stack, 5e-3, device_kinematics=False, gpu_frame_tasks=False, limits=lim)
        assert np.abs(V_one - V_ref).max() < 1e-8 * max(1.0, np.abs(V_ref).max())
    finally:
        if backend != "emu":
            pool.close()
        pink_amd.clear_device_cache()
    # limits are checked on the array form too
    q_bad = q.copy()
    q_bad[4, 7 + 2] = 9.0
    with pytest.raises(NotWithinConfigurationLimits):
        solve_ik_batch(ConfigurationBatch(m, q_bad), shared + [post], 5e-3, device_kinematics=True)


def test_solve_ik_batch_pipelined_ranges_equal_the_single_launch(backend, monkeypatch):
    """Large array batches are uploaded and solved in four overlapping ranges (``DeviceRollout.solve_pipelined``,
    ``pinkhip_memcpy_h2d_overlapped``): same velocities, bit for bit, as the single launch; per-robot and shared
    posture targets; the limit check still covers the whole batch."""
    import sys

    from pink_amd import ConfigurationBatch
    from pink_amd.rollout import DeviceRollout

    sik = sys.modules["pink_amd.solve_ik"]  # (the package re-exports the function under the module's name)

    m, frames = build_chain(9, free_flyer=True, seed=3), ["tool0", "joint_4"]
    rng = np.random.default_rng(15)
    B = 11  # (ranges of 3, 3, 3, 2)
    q = np.tile(m.neutral(), (B, 1))
    for j in m.joints:
        if j.kind != "free_flyer":
            q[:, j.idx_q] = rng.uniform(-0.9, 0.9, size=B)
    tasks = []
    for k, f in enumerate(frames):
        ft = FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9)
        R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
        for b in range(B):
            T = Configuration(m, q[b]).get_transform_frame_to_world(f) * SE3(np.eye(3), 0.03 * rng.normal(size=3))
            R[b], t[b] = T.rotation, T.translation
        ft.set_target_poses(R, t)
        tasks.append(ft)
    calls = []
    orig = DeviceRollout.solve_pipelined
    monkeypatch.setattr(DeviceRollout, "solve_pipelined", lambda self, *a, **k: calls.append(1) or orig(self, *a, **k))
    for batched_posture in (False, True):
        post = PostureTask(cost=1e-2, gain=0.7)
        if batched_posture:
            post.set_target_batch(q + 0.05 * rng.normal(size=q.shape) * (np.arange(m.nq) >= 7))
        else:
            post.set_target(m.neutral())
        pink_amd.clear_device_cache()
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 1 << 30)
        V_one = solve_ik_batch(ConfigurationBatch(m, q), tasks + [post], 5e-3, device_kinematics=True)
        assert not calls
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 4)
        for _ in range(2):  # fresh device state, then the cached one
            V_pipe = solve_ik_batch(ConfigurationBatch(m, q), tasks + [post], 5e-3, device_kinematics=True)
            assert np.array_equal(V_pipe, V_one) and np.abs(V_one).max() > 1e-3
        assert len(calls) == 2
        calls.clear()
    # round 5: ranges of a batch WITH dense rows (barrier rows and an equality constraint formed on chip): the tables
    # behind them are the same for every robot
    from pink_amd.barriers import BodySphericalBarrier

    p_tool = np.array([Configuration(m, q[b]).get_transform_frame_to_world("tool0").translation for b in range(B)])
    bars = [PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.01]), gain=np.array([50.0]), safe_displacement_gain=1.0),
            BodySphericalBarrier(("tool0", "joint_2"), d_min=0.01, gain=10.0)]
    hold = FrameTask("joint_8", 1.0, 1.0, gain=0.5)
    Rh, th = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for b in range(B):
        T = Configuration(m, q[b]).get_transform_frame_to_world("joint_8")
        Rh[b], th[b] = T.rotation, T.translation + 1e-4 * rng.normal(size=3)
    hold.set_target_poses(Rh, th)
    for kw in (dict(barriers=bars), dict(constraints=[hold])):
        pink_amd.clear_device_cache()
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 1 << 30)
        V_one = solve_ik_batch(ConfigurationBatch(m, q), tasks + [post], 5e-3, device_kinematics=True, **kw)
        assert not calls
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 4)
        V_pipe = solve_ik_batch(ConfigurationBatch(m, q), tasks + [post], 5e-3, device_kinematics=True, **kw)
        assert len(calls) == 1 and np.array_equal(V_pipe, V_one) and np.abs(V_one).max() > 1e-3
        calls.clear()
    q_bad = q.copy()
    q_bad[9, 7 + 2] = 9.0
    with pytest.raises(NotWithinConfigurationLimits):
        solve_ik_batch(ConfigurationBatch(m, q_bad), tasks + [post], 5e-3, device_kinematics=True)
    pink_amd.clear_device_cache()


def _biped():
    """Floating base with two 6-joint legs (a tree, not a chain) and frames at the pelvis and the ankles: the
    smallest model with the structure of the reference's JVRC / Upkie fixtures."""
    m = pink_amd.Model()
    root = m.add_joint("root_joint", "free_flyer")
    m.add_frame("pelvis", root, SE3(np.eye(3), [0.0, 0.0, 0.05]))
    axes = [[0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 1, 0], [0, 1, 0], [1, 0, 0]]
    for side, y in (("l", 0.1), ("r", -0.1)):
        parent = root
        for i, ax in enumerate(axes):
            off = SE3(np.eye(3), [0.0, y if i == 0 else 0.0, -0.05 if i < 3 else -0.35 if i in (3, 4) else -0.05])
            parent = m.add_joint(f"{side}_leg_{i}", "revolute", parent, off, ax, -2.0, 2.0, 10.0)
        m.add_frame(f"{side}_ankle", parent, SE3(np.eye(3), [0.0, 0.0, -0.05]))
    m.finalize() if hasattr(m, "finalize") else None
    return m


def test_single_task_translation_gives_pure_linear_velocity(backend):
    """tests/test_solve_ik.py:212-247: translating the target of a frame yields a linear velocity of that frame
    along the translation and no angular velocity."""
    m = _biped()
    q = m.neutral()
    q[7 + 3] = 0.4  # bend the knees a little, away from the singular straight leg
    q[7 + 4] = -0.2
    q[7 + 9] = 0.4
    q[7 + 10] = -0.2
    cfg = Configuration(m, q)
    task = FrameTask("r_ankle", position_cost=1.0, orientation_cost=1.0)
    tgt = cfg.get_transform_frame_to_world("r_ankle").copy()
    tgt.translation[1] -= 0.1
    task.set_target(tgt)
    task.lm_damping = 0.0  # only Tikhonov damping for this test
    v = solve_ik(cfg, [task], dt=1e-3, damping=1e-12)
    twist = cfg.get_frame_jacobian("r_ankle") @ v  # body twist of the frame, [linear; angular]
    R = cfg.get_transform_frame_to_world("r_ankle").rotation
    lin_world = R @ twist[:3]
    assert np.allclose(twist[3:], 0.0, atol=1e-7)
    assert abs(lin_world[0]) < 1e-6 and abs(lin_world[2]) < 1e-6 and lin_world[1] < 0.0


def test_three_tasks_convergence(backend):
    """tests/test_solve_ik.py:279-339: three simultaneously feasible FrameTasks on a floating-base biped converge in
    fewer than 42 closed-loop steps (velocity norm below 1e-6, small residual errors)."""
    m = _biped()
    q = m.neutral()
    for k in (3, 9):
        q[7 + k] = 0.5
        q[7 + k + 1] = -0.25
    cfg = Configuration(m, q)
    l_task = FrameTask("l_ankle", position_cost=1.0, orientation_cost=3.0)
    r_task = FrameTask("r_ankle", position_cost=1.0, orientation_cost=3.0)
    p_task = FrameTask("pelvis", position_cost=1.0, orientation_cost=0.0)
    tasks = [p_task, l_task, r_task]
    l_task.set_target(cfg.get_transform_frame_to_world("l_ankle") * SE3(np.eye(3), [0.1, 0.0, 0.0]))
    r_task.set_target(cfg.get_transform_frame_to_world("r_ankle") * SE3(np.eye(3), [-0.1, 0.0, 0.0]))
    p_task.set_target(cfg.get_transform_frame_to_world("pelvis"))
    dt, max_iter = 4e-3, 42
    for nb_iter in range(max_iter):
        v = solve_ik(cfg, tasks, dt)
        if np.linalg.norm(v) < 1e-6:
            break
        cfg = Configuration(m, cfg.integrate(v, dt))
    assert nb_iter < max_iter and np.linalg.norm(v) < 1e-6
    assert max(np.linalg.norm(t.compute_error(cfg)) for t in tasks) < 0.5
