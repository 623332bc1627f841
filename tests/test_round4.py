"""Round 4: the host-evaluated route of solve_ik_batch for every task / limit / barrier class the package ships
(vectorised over the batch) against Pink's calling pattern -- one solve_ik per configuration (pink/solve_ik.py:206-275);
which solver path an instance took (include/pinkhip.h PINKHIP_PATH_*); explicit default limits on the device route;
the regressions the round-3 advisor reproduced.  Emulator here, MI355X under -m gpu."""
import numpy as np
import pytest

import pink_amd
from pink_amd import (Configuration, ConfigurationBatch, DampingTask, FrameTask, PostureTask, build_chain, solve_ik,
                      solve_ik_batch)
from pink_amd.barriers import BodySphericalBarrier, PositionBarrier
from pink_amd.batch import DenseTaskTerm, DiagonalTaskTerm, pack_terms
from pink_amd.lie import SE3, exp6
from pink_amd.limits import AccelerationLimit, ConfigurationLimit, VelocityLimit
from pink_amd.runtime import set_default_solver
from pink_amd.tasks import JointCouplingTask, JointVelocityTask, LowAccelerationTask, RelativeFrameTask


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    s = request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")
    set_default_solver(s)
    yield request.param
    pink_amd.clear_device_cache()
    set_default_solver(None)


def _draw_q(m, B, rng, spread=0.8):
    q = np.tile(m.neutral(), (B, 1))
    for j in m.joints:
        if j.kind == "free_flyer":
            from pink_amd.configuration import _rot_to_quat

            for b in range(B):
                M = exp6(0.4 * rng.normal(size=6))
                q[b, j.idx_q:j.idx_q + 3], q[b, j.idx_q + 3:j.idx_q + 7] = M.translation, _rot_to_quat(M.rotation)
        else:
            q[:, j.idx_q] = rng.uniform(-spread, spread, size=B)
    return q


@pytest.mark.parametrize("free_flyer", [False, True])
def test_host_evaluated_stacks_equal_one_solve_ik_per_configuration(backend, free_flyer):
    """Task stacks the device-resident route does not form on chip -- RelativeFrameTask, DampingTask, LowAccelerationTask,
    JointVelocityTask, JointCouplingTask next to FrameTask + PostureTask; an explicit limit list with an AccelerationLimit;
    a BodySphericalBarrier next to a PositionBarrier; an equality constraint -- as ONE batched call (ConfigurationBatch
    and list of Configuration objects) against Pink's loop: one solve_ik per configuration."""
    m = build_chain(8, free_flyer=free_flyer, seed=7, limit=2.6, velocity=4.0)
    m.add_frame("mid", m.getJointId("joint_4"), SE3(np.eye(3), [0.0, 0.05, 0.1]))
    rng = np.random.default_rng(11)
    B, dt = 70, 5e-3  # (>= 64: the automatic choice of route is exercised too)
    q = _draw_q(m, B, rng)
    j5, j6 = (m.joints[m.getJointId(n)].idx_q for n in ("joint_5", "joint_6"))
    q[:, j6] = -2.0 * q[:, j5] + 0.002 * rng.normal(size=B)  # (the equality below is then within reach of one step)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    ft = FrameTask("tool0", 1.0, 0.6, lm_damping=1e-3)
    rt = RelativeFrameTask("tool0", "mid", 0.8, 0.2, lm_damping=1e-3, gain=0.8)
    R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for b, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6))
        R[b], t[b] = T.rotation, T.translation
    ft.set_target_poses(R, t)
    rt.set_target(cfgs[0].get_transform("tool0", "mid") * exp6(0.05 * rng.normal(size=6)))
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    la = LowAccelerationTask(cost=0.05)
    la.set_last_integration(0.2 * rng.normal(size=m.nv), dt)
    jv = JointVelocityTask(cost=0.05)
    jv.set_target(0.3 * rng.normal(size=m.nv - (6 if free_flyer else 0)), dt)
    jc = JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 2.0, cfgs[0])
    eq = JointCouplingTask(["joint_5", "joint_6"], [1.0, 0.5], 1.0, cfgs[0])
    tasks = [ft, rt, po, DampingTask(cost=1e-2), la, jv, jc]
    acc = AccelerationLimit(m, np.r_[np.full(6 if free_flyer else 0, np.inf), np.full(8, 400.0)])
    limits = [ConfigurationLimit(m, 0.6), VelocityLimit(m), acc]
    p_tool = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs])
    bars = [PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.02]), gain=np.array([50.0]), safe_displacement_gain=1.0),
            BodySphericalBarrier(("tool0", "joint_2"), d_min=0.02, gain=10.0)]
    kw = dict(limits=limits, barriers=bars, constraints=[eq])
    V_arr = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt, **kw)
    assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
    V_list = solve_ik_batch(cfgs, tasks, dt, **kw)
    V_np = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt, gpu_frame_tasks=False, **kw)
    assert np.array_equal(V_arr, V_list) and np.abs(V_arr - V_np).max() < 1e-9 * max(1.0, np.abs(V_np).max())
    n = 12 if backend == "emu" else B
    for b in range(n):
        fb = FrameTask("tool0", 1.0, 0.6, lm_damping=1e-3)
        fb.set_target(SE3(R[b], t[b]))
        v = solve_ik(cfgs[b], [fb] + tasks[1:], dt, **kw)
        assert np.abs(V_arr[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
    assert np.abs(V_arr).max() > 1e-3


def test_explicit_default_limits_take_the_device_route(backend):
    """limits=[ConfigurationLimit(model, gain), VelocityLimit(model)] is what limits=None installs (pink/solve_ik.py:94-105):
    the device kernels form both from the model tables -- same velocities as the host-evaluated route, the limit's own
    gain honoured; a list that is NOT the defaults (one limit missing, another velocity vector) stays host-evaluated."""
    m = build_chain(7, seed=2, limit=1.2, velocity=50.0)
    rng = np.random.default_rng(3)
    B, dt = 66, 1e-2
    q = _draw_q(m, B, rng, spread=1.15)  # close to the joint limits: the configuration limit binds
    ft = FrameTask("tool0", 1.0, 0.3, lm_damping=1e-2)
    ft.set_target(Configuration(m, q[0]).get_transform_frame_to_world("tool0") * exp6(0.5 * rng.normal(size=6)))
    po = PostureTask(cost=1e-2)
    po.set_target(m.neutral())
    batch = ConfigurationBatch(m, q)
    for gain in (0.5, 0.9):
        lim = [ConfigurationLimit(m, gain), VelocityLimit(m)]
        V_dev = solve_ik_batch(batch, [ft, po], dt, limits=lim)
        assert pink_amd.last_solve_stats()["route"] == "device"
        V_host = solve_ik_batch(batch, [ft, po], dt, limits=lim, device_kinematics=False)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
        assert np.abs(V_dev - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
    V_none = solve_ik_batch(batch, [ft, po], dt)
    V_half = solve_ik_batch(batch, [ft, po], dt, limits=[ConfigurationLimit(m, 0.5), VelocityLimit(m)])
    assert np.array_equal(V_none, V_half)
    assert np.abs(V_none - V_dev).max() > 1e-6  # (the gain matters on this batch)
    # a VelocityLimit with its own vector is the same table with other numbers: still the whole-step kernel
    lim = [ConfigurationLimit(m), VelocityLimit(m, np.full(m.nv, 0.7))]
    v_d = solve_ik_batch(batch, [ft, po], dt, limits=lim)
    assert pink_amd.last_solve_stats()["route"] == "device"
    v_a = solve_ik_batch(batch, [ft, po], dt, limits=lim, device_kinematics=False)
    assert np.abs(v_d - v_a).max() < 1e-8 * max(1.0, np.abs(v_a).max())
    # (round 5: a list without a VelocityLimit is served too -- tests/test_round5_regressions.py)
    for lim in ([VelocityLimit(m)],):  # (a list the device tables cannot hold: no ConfigurationLimit)
        v_h = solve_ik_batch(batch, [ft, po], dt, limits=lim)
        # (not the whole-step kernel: the limits are evaluated on the host, the FrameTask rows formed on the device)
        assert pink_amd.last_solve_stats()["route"] == "hybrid"
        v_a = solve_ik_batch(batch, [ft, po], dt, limits=lim, device_kinematics=False)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
        assert np.abs(v_h - v_a).max() < 1e-8 * max(1.0, np.abs(v_a).max())
        with pytest.raises(pink_amd.PinkError):
            solve_ik_batch(batch, [ft, po], dt, limits=lim, device_kinematics=True)


def test_one_target_source_is_live_and_both_routes_agree(backend):
    """Round-3 advisor: set_target_from_configuration followed by set_target_poses -- the device route used the single
    target, the host route the per-instance ones (max |dv| = 475).  The setters now replace each other, and a list of
    Configuration objects with shared tasks follows the same rules as the array form."""
    m = build_chain(6, seed=1)
    rng = np.random.default_rng(8)
    B, dt = 65, 5e-3
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    ft = FrameTask("tool0", 1.0, 1.0, lm_damping=1e-2)
    ft.set_target_from_configuration(cfgs[0])
    R = np.array([c.get_transform_frame_to_world("tool0").rotation for c in cfgs])
    t = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs]) + 0.05 * rng.normal(size=(B, 3))
    ft.set_target_poses(R, t)
    assert ft.transform_target_to_world is None
    po = PostureTask(cost=1e-2)
    po.set_target_from_configuration(cfgs[0])
    po.set_target_batch(q + 0.1 * rng.normal(size=q.shape))
    assert po.target_q is None
    V = {(src, dev): solve_ik_batch(cfgs if src == "list" else ConfigurationBatch(m, q), [ft, po], dt, device_kinematics=dev)
         for src in ("list", "array") for dev in (True, False)}
    ref = V["array", False]
    for key, v in V.items():
        assert np.abs(v - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), key
    ft.set_target(cfgs[0].get_transform_frame_to_world("tool0"))
    assert ft.target_poses is None
    po.set_target(q[0])
    assert po.target_q_batch is None


def test_auto_route_falls_back_when_no_whole_step_kernel_fits(backend):
    """Round-3 advisor: a 9-dof arm with three PositionBarriers on p_min and p_max (18 rows) at B = 64 raised in auto
    mode (no whole-step instantiation holds 18 barrier rows at nv = 9) while device_kinematics=False solved it."""
    m = build_chain(9, seed=4)
    rng = np.random.default_rng(5)
    B, dt = 64, 5e-3
    q = _draw_q(m, B, rng, spread=0.5)
    frames = ["tool0", "joint_6", "joint_3"]
    tasks = []
    for f in frames:
        ft = FrameTask(f, 1.0, 0.0, lm_damping=1e-2)
        ft.set_target(Configuration(m, q[0]).get_transform_frame_to_world(f))
        tasks.append(ft)
    po = PostureTask(cost=1e-2)
    po.set_target(m.neutral())
    bars = [PositionBarrier(f, p_min=np.full(3, -5.0), p_max=np.full(3, 5.0), gain=np.full(6, 10.0)) for f in frames]
    batch = ConfigurationBatch(m, q)
    V_auto = solve_ik_batch(batch, tasks + [po], dt, barriers=bars)
    assert pink_amd.last_solve_stats()["route"] == "hybrid"  # (barrier rows from the host, FrameTask rows from the device)
    V_host = solve_ik_batch(batch, tasks + [po], dt, barriers=bars, device_kinematics=False)
    assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
    assert np.abs(V_auto - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
    with pytest.raises(pink_amd.PinkError):
        solve_ik_batch(batch, tasks + [po], dt, barriers=bars, device_kinematics=True)


def test_configuration_batch_on_a_model_that_never_built_a_configuration(backend):
    """Round-3 advisor: the array route read model.configuration_limit, which only Configuration.__init__ attached."""
    m = build_chain(6, seed=9)
    assert not hasattr(m, "configuration_limit")
    q = _draw_q(m, 64, np.random.default_rng(1))
    ft = FrameTask("tool0", 1.0, 1.0, lm_damping=1e-2)
    ft.set_target(SE3(np.eye(3), [0.3, 0.1, 0.4]))
    for dev in (True, False, None):
        v = solve_ik_batch(ConfigurationBatch(m, q), [ft], 5e-3, device_kinematics=dev)
        assert v.shape == (64, 6) and np.isfinite(v).all()


def test_a_model_edit_is_not_served_from_the_cached_device_state(backend):
    """Round-3 advisor (low): the cache key held id(model) only; joint limits edited between two calls were ignored."""
    m = build_chain(6, seed=3, limit=2.0, velocity=80.0)
    q = _draw_q(m, 64, np.random.default_rng(2), spread=1.0)
    ft = FrameTask("tool0", 1.0, 1.0, lm_damping=1e-2)
    ft.set_target(SE3(np.eye(3), [0.5, 0.2, 0.3]))
    batch = ConfigurationBatch(m, q)
    v1 = solve_ik_batch(batch, [ft], 1e-2, device_kinematics=True)
    m._upper = [1.05] * m.nq  # tighter upper limits: the configuration-limit rows change
    v2 = solve_ik_batch(batch, [ft], 1e-2, device_kinematics=True)
    v2_host = solve_ik_batch(batch, [ft], 1e-2, device_kinematics=False)
    assert np.abs(v2 - v2_host).max() < 1e-8 * max(1.0, np.abs(v2_host).max())
    assert np.abs(v1 - v2).max() > 1e-6


def _solver_of(backend, request):
    return request.getfixturevalue("emu" if backend == "emu" else "gpu_solver")


def test_solver_path_is_reported_per_instance(backend, request, monkeypatch):
    """iters[b] carries the code that solved the instance (PINKHIP_ITERS_PATH): the sweep tableau on a well conditioned
    batch; the Goldfarb-Idnani kernel by dispatch where the stack is rank deficient by construction (fewer task rows
    than coordinates, no LM term: examples/humanoid_jvrc.py:69-81) or when it is forced; `routed` where the stack has
    rows enough but `damping` alone makes H positive definite (a posture task of cost 1e-7); the iteration counts stay
    what they were."""
    s = _solver_of(backend, request)
    rng = np.random.default_rng(4)
    nv, B = 12, 6
    J = rng.normal(0, 0.5, size=(B, 6, nv))
    e = 0.1 * rng.normal(size=(B, 6))
    box = [(-0.05 * np.ones((B, nv)), 0.05 * np.ones((B, nv)))]
    good = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=1.0), DiagonalTaskTerm(col0=0, e=0.1 * rng.normal(size=(B, nv)), cost=0.1)],
                      5e-3, 1e-12, boxes=box, batch_size=B)
    deficient = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=1.0)], 5e-3, 1e-12, boxes=box, batch_size=B)
    weak = pack_terms(nv, [DenseTaskTerm(J=J, e=e, cost=1.0), DiagonalTaskTerm(col0=0, e=0.1 * rng.normal(size=(B, nv)), cost=1e-7)],
                      5e-3, 1e-12, boxes=box, batch_size=B)
    r = s.solve(good)
    assert (r.status == 0).all() and (r.path == 0).all() and r.iters.max() < 100 and r.path_fractions()["tableau"] == 1.0
    rd = s.solve(deficient)
    assert (rd.status == 0).all() and (rd.path == 3).all()
    rw = s.solve(weak)
    # (round 6: the conditioning estimate behind `routed` is taken on the coordinates the start sweeps in -- the guessed
    # free set -- so an instance whose ill-conditioned directions are held by bounds stays on the tableau, and the KKT
    # certificate decides; the packed run below is the check of the numbers either way)
    assert (rw.status == 0).all() and np.isin(rw.path, (0, 1, 2)).all() and rw.iters.max() < 200
    fr = rw.path_fractions()
    assert fr["goldfarb_idnani"] == 0.0 and abs(sum(fr.values()) - 1.0) < 1e-12
    monkeypatch.setenv("PINKHIP_SOLVER", "packed")
    rp = s.solve(good)
    assert (rp.path == 3).all() and np.abs(rp.dq - r.dq).max() < 1e-10
    rwp = s.solve(weak)
    # the routed instances ran the same Goldfarb-Idnani code on the same stacked problem
    assert np.abs(rwp.dq - rw.dq).max() < 1e-9 * max(1.0, np.abs(rw.dq).max())


@pytest.mark.parametrize("free_flyer", [False, True])
def test_hybrid_route_forms_frame_task_rows_on_the_device(backend, free_flyer):
    """FrameTasks next to identity-Jacobian tasks (posture, damping, low acceleration, joint velocity) under an explicit
    limit list with an AccelerationLimit and a barrier: in automatic mode the FrameTask rows are formed by the device
    from q (pinkhip_fk_frame_tasks_device writes them into the packed streams), everything else is evaluated on the host
    for the whole batch -- same velocities as the all-host evaluation and as one solve_ik per configuration; shared
    tasks with batched targets and per-instance task lists alike; the device state is kept between calls."""
    m = build_chain(9, free_flyer=free_flyer, seed=5, limit=2.8, velocity=6.0)
    rng = np.random.default_rng(21)
    B, dt = 67, 5e-3
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    frames = ["tool0", "joint_5"]
    shared, per_instance = [], [[] for _ in range(B)]
    for k, f in enumerate(frames):
        R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
        ft = FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9)
        for b, c in enumerate(cfgs):
            T = c.get_transform_frame_to_world(f) * exp6(0.05 * rng.normal(size=6))
            R[b], t[b] = T.rotation, T.translation
            fb = FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9)
            fb.set_target(T)
            per_instance[b].append(fb)
        ft.set_target_poses(R, t)
        shared.append(ft)
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    la = LowAccelerationTask(cost=0.05)
    la.set_last_integration(0.2 * rng.normal(size=m.nv), dt)
    jv = JointVelocityTask(cost=0.05)
    jv.set_target(0.3 * rng.normal(size=m.nv - (6 if free_flyer else 0)), dt)
    rest = [po, DampingTask(cost=1e-2), la, jv]
    for b in range(B):
        per_instance[b] += rest
    acc = AccelerationLimit(m, np.r_[np.full(6 if free_flyer else 0, np.inf), np.full(9, 500.0)])
    # (one AccelerationLimit next to the defaults is folded into the box by the whole-step kernel; two are not)
    limits = [ConfigurationLimit(m, 0.7), VelocityLimit(m), acc, AccelerationLimit(m, np.r_[np.full(6 if free_flyer else 0, np.inf), np.full(9, 800.0)])]
    p_tool = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs])
    bars = [PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.02]), gain=np.array([50.0]), safe_displacement_gain=1.0)]
    kw = dict(limits=limits, barriers=bars)
    V = solve_ik_batch(ConfigurationBatch(m, q), shared + rest, dt, **kw)
    assert pink_amd.last_solve_stats()["route"] == "hybrid"
    V_host = solve_ik_batch(ConfigurationBatch(m, q), shared + rest, dt, device_kinematics=False, gpu_frame_tasks=False, **kw)
    assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
    scale = max(1.0, np.abs(V_host).max())
    assert np.abs(V - V_host).max() < 1e-8 * scale and np.abs(V).max() > 1e-3
    V_list = solve_ik_batch(cfgs, per_instance, dt, **kw)  # per-instance task objects, the cached device state
    assert pink_amd.last_solve_stats()["route"] == "hybrid"
    assert np.abs(V_list - V_host).max() < 1e-8 * scale
    for b in range(6 if backend == "emu" else B):
        v = solve_ik(cfgs[b], per_instance[b], dt, **kw)
        assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
    # a RelativeFrameTask is a relative slot of the same kernel ...
    rt = RelativeFrameTask("tool0", "joint_5", 1.0, 0.0)
    rt.set_target(cfgs[0].get_transform("tool0", "joint_5"))
    V_rel = solve_ik_batch(ConfigurationBatch(m, q), shared + rest + [rt], dt, **kw)
    assert pink_amd.last_solve_stats()["route"] == "hybrid"
    V_rel_host = solve_ik_batch(ConfigurationBatch(m, q), shared + rest + [rt], dt, device_kinematics=False, gpu_frame_tasks=False, **kw)
    assert np.abs(V_rel - V_rel_host).max() < 1e-8 * max(1.0, np.abs(V_rel_host).max())
    # ... any other dense task next to explicit limits: the all-host evaluation serves the call
    jc = JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 10.0, cfgs[0])
    solve_ik_batch(ConfigurationBatch(m, q), shared + rest + [jc], dt, **kw)
    assert pink_amd.last_solve_stats()["route"] == "host-evaluated"


@pytest.mark.parametrize("free_flyer", [False, True])
def test_whole_step_kernel_forms_coupling_and_identity_tasks_on_chip(backend, free_flyer):
    """The task stack of the reference's own humanoid example (examples/humanoid_draco3.py:34-71: FrameTasks, a
    PostureTask and two JointCouplingTasks) and the identity-Jacobian tasks next to it (DampingTask, LowAccelerationTask,
    JointVelocityTask) take the device-resident route: the kernel forms the coupling rows from a constant table and
    e = A (q (-) q_0) - b from the configuration it already holds (pink/tasks/linear_holonomic_task.py:103-148), the
    identity tasks from a batch-constant error.  Same velocities as the all-host evaluation and as one solve_ik per
    configuration, in any task order, with barriers, through the Goldfarb-Idnani code for a stack that is rank deficient
    by construction; a LowAccelerationTask that moves between two calls is served by the cached device state."""
    m = build_chain(9, free_flyer=free_flyer, seed=5, limit=2.8, velocity=6.0)
    rng = np.random.default_rng(33)
    B, dt = 9, 5e-3
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    frames = []
    for k, f in enumerate(["tool0", "joint_5"]):
        R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
        ft = FrameTask(f, 1.0, 0.5 if k == 0 else 0.0, lm_damping=1e-3, gain=0.9)
        for b, c in enumerate(cfgs):
            T = c.get_transform_frame_to_world(f) * exp6(0.05 * rng.normal(size=6))
            R[b], t[b] = T.rotation, T.translation
        ft.set_target_poses(R, t)
        frames.append(ft)
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    jc1 = JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 100.0, cfgs[0], lm_damping=5e-7)
    jc2 = JointCouplingTask(["joint_6", "joint_7", "joint_8"], [1.0, 0.5, -2.0], 50.0, cfgs[0], gain=0.7)
    la = LowAccelerationTask(cost=0.05)
    la.set_last_integration(0.2 * rng.normal(size=m.nv), dt)
    jv = JointVelocityTask(cost=0.05)
    jv.set_target(0.3 * rng.normal(size=m.nv - (6 if free_flyer else 0)), dt)
    p_tool = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs])
    bars = [PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.02]), gain=np.array([50.0]), safe_displacement_gain=1.0)]
    cb = ConfigurationBatch(m, q)
    stacks = {
        "reference example": (frames + [po, jc1, jc2], {}),
        "interleaved": ([frames[0], jc1, frames[1], jc2, po, DampingTask(cost=1e-2)], {}),
        "every identity task": (frames + [jc1, po, DampingTask(cost=1e-2), la, jv], {}),
        "no posture": (frames + [jc2, DampingTask(cost=1e-2)], {}),
        "barrier": (frames + [po, jc1, jc2], dict(barriers=bars)),
    }
    for name, (ts, kw) in stacks.items():
        V = solve_ik_batch(cb, ts, dt, device_kinematics=True, **kw)
        assert pink_amd.last_solve_stats()["route"] == "device", name
        V_host = solve_ik_batch(cb, ts, dt, device_kinematics=False, gpu_frame_tasks=False, **kw)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated", name
        scale = max(1.0, np.abs(V_host).max())
        assert np.abs(V - V_host).max() < 1e-8 * scale and np.abs(V).max() > 1e-3, name
        for b in range(3):  # Pink's calling pattern: one target per task object, one solve_ik per configuration
            own = {id(ft): FrameTask(ft.frame, ft.cost[0], ft.cost[3], lm_damping=ft.lm_damping, gain=ft.gain) for ft in frames}
            for ft in frames:
                own[id(ft)].set_target(SE3(ft.target_poses[b, :9].reshape(3, 3), ft.target_poses[b, 9:]))
            v = solve_ik(cfgs[b], [own.get(id(t_), t_) for t_ in ts], dt, **kw)
            assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), (name, b)
    # FrameTasks and couplings alone on more coordinates than rows, no damping: rank deficient by construction, the
    # Goldfarb-Idnani code forms the same rows
    for ft in frames:
        ft.lm_damping = 0.0
    ts = frames + [JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 1.0, cfgs[0])]
    V = solve_ik_batch(cb, ts, dt, damping=1e-12, device_kinematics=True)
    st = pink_amd.last_solve_stats()
    assert st["route"] == "device" and (st["paths"]["goldfarb_idnani"] == 1.0) == (m.nv > 13)
    V_host = solve_ik_batch(cb, ts, dt, damping=1e-12, device_kinematics=False, gpu_frame_tasks=False)
    # (H = J^T W J + 1e-12 I has a null space but for the damping: its component of the velocity is fixed to ~1e-4 only,
    # in either route -- tests/test_gpu_parity.py compares such problems through their KKT certificates)
    assert np.abs(V - V_host).max() < 1e-3 * max(1.0, np.abs(V_host).max())
    # the low-acceleration error moves from one control step to the next: same device state, new error table
    ts = frames + [jc1, po, la]
    for ft in frames:
        ft.lm_damping = 1e-3
    for _ in range(2):
        la.set_last_integration(0.2 * rng.normal(size=m.nv), dt)
        V = solve_ik_batch(cb, ts, dt, device_kinematics=True)
        assert pink_amd.last_solve_stats()["route"] == "device"
        V_host = solve_ik_batch(cb, ts, dt, device_kinematics=False, gpu_frame_tasks=False)
        assert np.abs(V - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
    # a coupling that touches the floating base is not a constant row: the call is served by another route
    if free_flyer:
        A = np.zeros((1, m.nv))
        A[0, 2], A[0, 8] = 1.0, -1.0
        from pink_amd.tasks import LinearHolonomicTask

        lh = LinearHolonomicTask(A, np.zeros(1), m.neutral(), cost=1.0)
        with pytest.raises(pink_amd.PinkError):
            solve_ik_batch(cb, frames + [po, lh], dt, device_kinematics=True)
        V = solve_ik_batch(cb, frames + [po, lh], dt)
        assert pink_amd.last_solve_stats()["route"] != "device"
        V_host = solve_ik_batch(cb, frames + [po, lh], dt, device_kinematics=False, gpu_frame_tasks=False)
        assert np.abs(V - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())


@pytest.mark.parametrize("free_flyer", [False, True])
def test_relative_frame_tasks_are_formed_on_the_device(backend, free_flyer):
    """RelativeFrameTask (pink/tasks/relative_frame_task.py:142-231) next to FrameTasks and a PostureTask: a relative
    slot of the device model -- the target carried into the world by the root frame's current pose, a signed ancestor
    indicator on the columns -- in the whole-step kernel (device route) and in the frame-row kernel (hybrid route).
    Same velocities as the all-host evaluation and as one solve_ik per configuration; root frame below or above the
    task frame in the tree; per-instance task objects with their own targets."""
    m = build_chain(10, free_flyer=free_flyer, seed=9, limit=2.8, velocity=6.0)
    m.add_frame("mid", m.getJointId("joint_4"), SE3(np.eye(3), [0.0, 0.05, 0.1]))
    rng = np.random.default_rng(44)
    B, dt = 7, 5e-3
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3, gain=0.9)
    R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for b, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6))
        R[b], t[b] = T.rotation, T.translation
    ft.set_target_poses(R, t)
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    r1 = RelativeFrameTask("joint_8", "mid", 1.0, 0.4, lm_damping=1e-3, gain=0.8)  # root above the frame
    r1.set_target(cfgs[0].get_transform("joint_8", "mid") * exp6(0.05 * rng.normal(size=6)))
    r2 = RelativeFrameTask("joint_3", "tool0", [1.0, 0.5, 2.0], 0.3)  # root below the frame
    r2.set_target(cfgs[1].get_transform("joint_3", "tool0") * exp6(0.05 * rng.normal(size=6)))
    cb = ConfigurationBatch(m, q)
    for name, ts in (("one", [ft, r1, po]), ("two, interleaved", [r2, ft, r1, po, DampingTask(cost=1e-2)])):
        V = solve_ik_batch(cb, ts, dt, device_kinematics=True)
        assert pink_amd.last_solve_stats()["route"] == "device", name
        V_rows = solve_ik_batch(cb, ts, dt, device_kinematics="frame_rows")
        assert pink_amd.last_solve_stats()["route"] == "hybrid", name
        V_host = solve_ik_batch(cb, ts, dt, device_kinematics=False, gpu_frame_tasks=False)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated", name
        scale = max(1.0, np.abs(V_host).max())
        assert np.abs(V - V_host).max() < 1e-8 * scale and np.abs(V_rows - V_host).max() < 1e-8 * scale and np.abs(V).max() > 1e-3, name
        for b in range(3):  # Pink's calling pattern
            own = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3, gain=0.9)
            own.set_target(SE3(R[b], t[b]))
            v = solve_ik(cfgs[b], [own if t_ is ft else t_ for t_ in ts], dt)
            assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), (name, b)
    # per-instance task objects, each with its own relative target
    per = []
    for b, c in enumerate(cfgs):
        fb = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3, gain=0.9)
        fb.set_target(SE3(R[b], t[b]))
        rb = RelativeFrameTask("joint_8", "mid", 1.0, 0.4, lm_damping=1e-3, gain=0.8)
        rb.set_target(c.get_transform("joint_8", "mid") * exp6(0.05 * rng.normal(size=6)))
        per.append([fb, rb, po])
    V = solve_ik_batch(cfgs, per, dt, device_kinematics=True)
    assert pink_amd.last_solve_stats()["route"] == "device"
    for b in range(B):
        v = solve_ik(cfgs[b], per[b], dt)
        assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
    # a position barrier needs the world pose of its frame: round 5 gives a frame that only a relative slot carries an
    # ordinary slot of its own, with zero cost (round 4 refused the device route here)
    bar = PositionBarrier("joint_8", indices=[2], p_max=np.array([10.0]), gain=np.array([50.0]), safe_displacement_gain=1.0)
    V_bar = solve_ik_batch(cb, [ft, r1, po], dt, barriers=[bar], device_kinematics=True)
    assert pink_amd.last_solve_stats()["route"] == "device"
    V_bar_host = solve_ik_batch(cb, [ft, r1, po], dt, barriers=[bar], device_kinematics=False, gpu_frame_tasks=False)
    assert np.abs(V_bar - V_bar_host).max() < 1e-8 * max(1.0, np.abs(V_bar_host).max())


@pytest.mark.parametrize("free_flyer", [False, True])
def test_acceleration_limit_is_folded_into_the_box_on_chip(backend, free_flyer):
    """An explicit limit list with an AccelerationLimit on the joints behind the root
    (pink/limits/acceleration_limit.py:158-199) stays on the whole-step kernel: the box of a coordinate is formed from q,
    the previous displacement and three per-coordinate tables.  Same velocities as the all-host evaluation and as one
    solve_ik per configuration; the limit binds (the velocities differ from those without it); a new
    set_last_integration between two calls is served by the cached device state; small robots too (the kernel is then
    needed, whatever their size)."""
    for n in (9, 5):
        m = build_chain(n, free_flyer=free_flyer, seed=5, limit=2.8, velocity=60.0)
        rng = np.random.default_rng(70 + n)
        B, dt = 8, 5e-3
        q = _draw_q(m, B, rng, spread=2.5)  # (some joints close to their limits: the braking-distance term binds)
        cfgs = [Configuration(m, q[b]) for b in range(B)]
        ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
        R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
        for b, c in enumerate(cfgs):
            T = c.get_transform_frame_to_world("tool0") * exp6(0.3 * rng.normal(size=6))
            R[b], t[b] = T.rotation, T.translation
        ft.set_target_poses(R, t)
        po = PostureTask(cost=5e-2)
        po.set_target(m.neutral())
        a_max = np.r_[np.full(6 if free_flyer else 0, np.inf), rng.uniform(20.0, 400.0, size=n)]
        a_max[-2] = np.inf  # (one joint without an acceleration bound)
        acc = AccelerationLimit(m, a_max)
        acc.set_last_integration(np.r_[np.zeros(6 if free_flyer else 0), rng.normal(size=n)] * 0.5, dt)
        limits = [ConfigurationLimit(m, 0.7), VelocityLimit(m), acc]
        cb = ConfigurationBatch(m, q)
        V_free = solve_ik_batch(cb, [ft, po], dt, device_kinematics=True, limits=limits[:2])
        for _ in range(2):
            V = solve_ik_batch(cb, [ft, po], dt, device_kinematics=True, limits=limits)
            assert pink_amd.last_solve_stats()["route"] == "device"
            V_host = solve_ik_batch(cb, [ft, po], dt, device_kinematics=False, gpu_frame_tasks=False, limits=limits)
            scale = max(1.0, np.abs(V_host).max())
            assert np.abs(V - V_host).max() < 1e-8 * scale
            assert np.abs(V - V_free).max() > 1e-3 * scale
            for b in range(3):
                own = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
                own.set_target(SE3(R[b], t[b]))
                v = solve_ik(cfgs[b], [own, po], dt, limits=limits)
                assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
            acc.set_last_integration(V[0], dt)  # (the next control step: another previous displacement)
    # a bound on a floating-base coordinate is not a table entry: another route
    if free_flyer:
        acc6 = AccelerationLimit(m, np.full(m.nv, 50.0))
        solve_ik_batch(cb, [ft, po], dt, limits=[ConfigurationLimit(m, 0.7), VelocityLimit(m), acc6])
        assert pink_amd.last_solve_stats()["route"] != "device"


@pytest.mark.parametrize("free_flyer", [False, True])
def test_equality_constraints_made_of_frame_tasks_on_the_hybrid_route(backend, free_flyer):
    """solve_ik(..., constraints=[FrameTask / RelativeFrameTask]) (pink/solve_ik.py:125-149: A = J, b = -gain e): the
    frame-row kernel writes the constraint's Jacobian into the leading dense rows of the QP and its error into their
    right-hand sides on the device; only -gain is applied on the host.  Same velocities as the all-host evaluation and as
    one solve_ik per configuration; the equality holds for the returned velocity."""
    m = build_chain(10, free_flyer=free_flyer, seed=12, limit=2.8, velocity=8.0)
    rng = np.random.default_rng(55)
    B, dt = 70, 5e-3
    q = _draw_q(m, B, rng, spread=0.6)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    hold_R, hold_t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for b, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6))
        R[b], t[b] = T.rotation, T.translation
        H = c.get_transform_frame_to_world("joint_9")  # (nine joints upstream: six equations are within reach of the box)
        hold_R[b], hold_t[b] = H.rotation, H.translation
    ft.set_target_poses(R, t)
    hold = FrameTask("joint_9", 1.0, 1.0, gain=0.7)  # enforced strictly: the frame goes where its target is
    hold.set_target_poses(hold_R, hold_t + 2e-4 * rng.normal(size=(B, 3)))
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    cb = ConfigurationBatch(m, q)
    stacks = [("one", [hold])]
    if free_flyer:  # (twelve equations need the floating base's six coordinates); per-instance constraint objects
        per = []
        for b, c in enumerate(cfgs):
            hb = FrameTask("joint_9", 1.0, 1.0, gain=0.7)
            hb.set_target(SE3(hold.target_poses[b, :9].reshape(3, 3), hold.target_poses[b, 9:]))
            rb = RelativeFrameTask("tool0", "joint_2", 1.0, 1.0, gain=0.5)
            rb.set_target(c.get_transform("tool0", "joint_2") * exp6(1e-3 * rng.normal(size=6)))
            per.append([hb, rb])
        stacks.append(("frame + relative", per))
    for name, cons in stacks:
        V = solve_ik_batch(cb, [ft, po], dt, constraints=cons, device_kinematics="frame_rows")
        assert pink_amd.last_solve_stats()["route"] == "hybrid", name
        # (round 5: constraints made of FrameTasks alone are the leading equality rows of the whole-step kernel)
        V_auto = solve_ik_batch(cb, [ft, po], dt, constraints=cons)
        assert pink_amd.last_solve_stats()["route"] == "device", name  # (a RelativeFrameTask constraint: a relative slot)
        assert np.abs(V_auto - V).max() < 1e-8 * max(1.0, np.abs(V).max()), name
        V_host = solve_ik_batch(cb, [ft, po], dt, constraints=cons, device_kinematics=False, gpu_frame_tasks=False)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated", name
        scale = max(1.0, np.abs(V_host).max())
        assert np.abs(V - V_host).max() < 1e-7 * scale and np.abs(V).max() > 1e-3, name
        for b in range(3):
            own = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
            own.set_target(SE3(R[b], t[b]))
            oh = FrameTask("joint_9", 1.0, 1.0, gain=0.7)
            oh.set_target(SE3(hold.target_poses[b, :9].reshape(3, 3), hold.target_poses[b, 9:]))
            mine = [oh] if name == "one" else cons[b]
            v = solve_ik(cfgs[b], [own, po], dt, constraints=mine)
            assert np.abs(V[b] - v).max() < 1e-7 * max(1.0, np.abs(v).max()), (name, b)
            for c_ in mine:  # J dq = -gain e
                assert np.abs(c_.compute_jacobian(cfgs[b]) @ (V[b] * dt) + c_.gain * c_.compute_error(cfgs[b])).max() < 1e-9, (name, b)


@pytest.mark.parametrize("free_flyer", [False, True])
def test_velocity_limit_with_its_own_vector_stays_on_the_device_route(backend, free_flyer):
    """VelocityLimit(model, velocity_limit=...) (pink/limits/velocity_limit.py:46-58: how joints without a model limit get
    one): the device model of the call carries that vector; a joint it leaves unbounded is unbounded; same velocities as
    the all-host evaluation and as solve_ik per configuration; the limit binds."""
    m = build_chain(8, free_flyer=free_flyer, seed=6, limit=2.8, velocity=50.0)
    rng = np.random.default_rng(81)
    B, dt = 9, 5e-3
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for b, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("tool0") * exp6(0.3 * rng.normal(size=6))
        R[b], t[b] = T.rotation, T.translation
    ft.set_target_poses(R, t)
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    r = 6 if free_flyer else 0
    vec = np.r_[np.full(r, np.inf), rng.uniform(0.5, 3.0, size=8)]
    vec[r + 3] = np.inf  # (no velocity bound on this joint)
    limits = [ConfigurationLimit(m), VelocityLimit(m, vec)]
    cb = ConfigurationBatch(m, q)
    V = solve_ik_batch(cb, [ft, po], dt, limits=limits, device_kinematics=True)
    assert pink_amd.last_solve_stats()["route"] == "device"
    V_host = solve_ik_batch(cb, [ft, po], dt, limits=limits, device_kinematics=False, gpu_frame_tasks=False)
    V_model = solve_ik_batch(cb, [ft, po], dt, device_kinematics=True)  # the model's own (loose) velocity limits
    scale = max(1.0, np.abs(V_host).max())
    assert np.abs(V - V_host).max() < 1e-8 * scale and np.abs(V - V_model).max() > 1e-2
    assert (np.abs(V[:, r:][:, np.isfinite(vec[r:])]) <= vec[r:][np.isfinite(vec[r:])] + 1e-9).all()
    for b in range(3):
        own = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
        own.set_target(SE3(R[b], t[b]))
        v = solve_ik(cfgs[b], [own, po], dt, limits=limits)
        assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
