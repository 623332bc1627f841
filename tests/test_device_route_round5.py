"""Round 5: what the whole-step kernel learnt to form on chip -- equality constraints made of FrameTasks
(pink/solve_ik.py:125-149: the leading equality rows of its QP), BodySphericalBarrier rows
(pink/barriers/body_spherical_barrier.py:73-143) and frames that only a barrier or a constraint needs (slots of the device
model with zero cost: pink/tasks/task.py:148-166 adds nothing for them) -- against the host-evaluated route and against
Pink's calling pattern, one solve_ik per configuration.  Emulator here, MI355X under -m gpu."""
import numpy as np
import pytest

import pink_amd
from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik, solve_ik_batch
from pink_amd.barriers import BodySphericalBarrier, PositionBarrier
from pink_amd.lie import SE3, exp6
from pink_amd.runtime import set_default_solver

from tests.test_round4 import _draw_q


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    s = request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")
    set_default_solver(s)
    yield request.param
    pink_amd.clear_device_cache()
    set_default_solver(None)


def _stack(free_flyer, seed, B=66, joints=8):
    m = build_chain(joints, free_flyer=free_flyer, seed=7, limit=2.6, velocity=4.0)
    rng = np.random.default_rng(seed)
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    ft = FrameTask("tool0", 1.0, 0.6, lm_damping=1e-3)
    R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for k, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6))
        R[k], t[k] = T.rotation, T.translation
    ft.set_target_poses(R, t)
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    return m, rng, q, cfgs, ft, po, R, t


def _own_frame_task(R, t, b):
    fb = FrameTask("tool0", 1.0, 0.6, lm_damping=1e-3)
    fb.set_target(SE3(R[b], t[b]))
    return fb


@pytest.mark.parametrize("free_flyer", [False, True])
def test_spherical_and_position_barriers_are_formed_on_chip(backend, free_flyer):
    """A BodySphericalBarrier between the tool and a frame that carries no task (a zero-cost slot of the device model)
    next to a PositionBarrier: route "device", velocities of the host-evaluated route and of solve_ik per configuration;
    the barrier binds on part of the batch (the test would pass trivially otherwise)."""
    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(free_flyer, 3)
    B = len(cfgs)
    p_tool = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs])
    d12 = np.array([np.linalg.norm(c.get_transform_frame_to_world("tool0").translation - c.get_transform_frame_to_world("joint_2").translation)
                    for c in cfgs])
    # the targets pull the tool towards joint_2, the sphere around it starts just inside the closest tool: the row binds
    p_j2 = np.array([c.get_transform_frame_to_world("joint_2").translation for c in cfgs])
    t_pull = p_tool + 0.5 * (p_j2 - p_tool)
    ft.set_target_poses(R, t_pull)
    t = t_pull
    bars = [BodySphericalBarrier(("tool0", "joint_2"), d_min=float(0.9995 * np.quantile(d12, 0.3)), gain=10.0, safe_displacement_gain=2.0),
            PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.002]), gain=np.array([50.0]), safe_displacement_gain=1.0)]
    cb = ConfigurationBatch(m, q)
    keep = d12 >= np.quantile(d12, 0.3)  # (robots that start inside the sphere are not part of this test)
    q, cfgs, R, t, d12 = q[keep], [c for c, k in zip(cfgs, keep) if k], R[keep], t[keep], d12[keep]
    B = len(cfgs)
    ft.set_target_poses(R, t)
    cb = ConfigurationBatch(m, q)
    V = solve_ik_batch(cb, [ft, po], dt, barriers=bars, device_kinematics=True)
    assert pink_amd.last_solve_stats()["route"] == "device"
    V_free = solve_ik_batch(cb, [ft, po], dt, device_kinematics=True)
    assert (np.abs(V - V_free).max(axis=1) > 1e-6).sum() >= 3  # the barriers bind
    V_host = solve_ik_batch(cb, [ft, po], dt, barriers=bars, device_kinematics=False, gpu_frame_tasks=False)
    assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
    assert np.abs(V - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
    for b in range(6 if backend == "emu" else B):
        v = solve_ik(cfgs[b], [_own_frame_task(R, t, b), po], dt, barriers=bars)
        assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
    # a list of Configuration objects with per-instance task lists takes the same route
    per = [[_own_frame_task(R, t, b), po] for b in range(B)]
    V_list = solve_ik_batch(cfgs, per, dt, barriers=bars, device_kinematics=True)
    assert pink_amd.last_solve_stats()["route"] == "device" and np.abs(V_list - V).max() < 1e-12 * max(1.0, np.abs(V).max())


@pytest.mark.parametrize("free_flyer", [False, True])
def test_equality_constraints_are_the_leading_rows_of_the_whole_step_kernel(backend, free_flyer):
    """constraints=[FrameTask] with one target per instance (and, second, next to a PositionBarrier: seven dense rows):
    route "device"; J dq = -gain e holds for the returned velocity; host-evaluated route and solve_ik agree."""
    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(free_flyer, 5)
    B = len(cfgs)
    hold = FrameTask("joint_7", 1.0, 1.0, gain=0.7)  # (seven joints upstream: six equations are within reach)
    Rh, th = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for k, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("joint_7") * exp6(2e-4 * rng.normal(size=6))
        Rh[k], th[k] = T.rotation, T.translation
    hold.set_target_poses(Rh, th)
    p_tool = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs])
    bar = PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.02]), gain=np.array([50.0]), safe_displacement_gain=1.0)
    cb = ConfigurationBatch(m, q)
    for kw in (dict(constraints=[hold]), dict(constraints=[hold], barriers=[bar])):
        V = solve_ik_batch(cb, [ft, po], dt, **kw)
        assert pink_amd.last_solve_stats()["route"] == "device", kw.keys()
        V_host = solve_ik_batch(cb, [ft, po], dt, device_kinematics=False, gpu_frame_tasks=False, **kw)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
        assert np.abs(V - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max()) and np.abs(V).max() > 1e-3
        for b in range(4 if backend == "emu" else B):
            oh = FrameTask("joint_7", 1.0, 1.0, gain=0.7)
            oh.set_target(SE3(Rh[b], th[b]))
            own = dict(kw, constraints=[oh])
            v = solve_ik(cfgs[b], [_own_frame_task(R, t, b), po], dt, **own)
            assert np.abs(V[b] - v).max() < 1e-8 * max(1.0, np.abs(v).max()), b
            assert np.abs(oh.compute_jacobian(cfgs[b]) @ (V[b] * dt) + oh.gain * oh.compute_error(cfgs[b])).max() < 1e-9, b
    # the constrained frame may carry a task of the objective too (another slot, another target, its own costs)
    soft = FrameTask("joint_7", 0.3, 0.1, lm_damping=1e-3)
    soft.set_target(cfgs[0].get_transform_frame_to_world("joint_7") * exp6(0.02 * rng.normal(size=6)))
    V = solve_ik_batch(cb, [ft, soft, po], dt, constraints=[hold])
    assert pink_amd.last_solve_stats()["route"] == "device"
    V_host = solve_ik_batch(cb, [ft, soft, po], dt, constraints=[hold], device_kinematics=False, gpu_frame_tasks=False)
    assert np.abs(V - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())


def test_inconsistent_equality_constraints_are_reported_per_instance(backend):
    """Six equations on a frame with five joints upstream: quadprog's "constraints are inconsistent" for the instances
    whose target is out of reach -- the device route reports them as the host route does (NoSolutionFound lists them)."""
    from pink_amd.exceptions import NoSolutionFound

    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(False, 9, B=64)
    hold = FrameTask("joint_5", 1.0, 1.0)
    Rh, th = np.zeros((64, 3, 3)), np.zeros((64, 3))
    for k, c in enumerate(cfgs):
        T = c.get_transform_frame_to_world("joint_5")
        if k % 2:
            T = T * exp6(1e-2 * rng.normal(size=6))  # out of reach of five joints
        Rh[k], th[k] = T.rotation, T.translation
    hold.set_target_poses(Rh, th)
    with pytest.raises(NoSolutionFound) as dev:
        solve_ik_batch(ConfigurationBatch(m, q), [ft, po], dt, constraints=[hold])
    assert pink_amd.last_solve_stats()["route"] == "device"
    with pytest.raises(NoSolutionFound) as host:
        solve_ik_batch(ConfigurationBatch(m, q), [ft, po], dt, constraints=[hold], device_kinematics=False, gpu_frame_tasks=False)
    assert list(dev.value.indices) == list(host.value.indices) == list(range(1, 64, 2))


def test_frozen_targets_are_uploaded_once_and_new_targets_replace_them(backend):
    """FrameTask.freeze_targets: the per-instance target array becomes read-only and the cached device state keeps its
    copy across calls (nothing but q goes up); set_target_poses afterwards is seen by the next call; an array that was
    NOT frozen may be refilled in place between two calls and the second call sees the new numbers."""
    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(False, 11, B=64, joints=14)  # (14 joints: the whole-step kernel serves the call)
    cb = ConfigurationBatch(m, q)
    V0 = solve_ik_batch(cb, [ft, po], dt).copy()
    # refilled in place, not frozen: the next call uploads what the array holds then
    ft.target_poses[:, 9:] += 0.01
    V1 = solve_ik_batch(cb, [ft, po], dt).copy()
    assert np.abs(V1 - V0).max() > 1e-3
    ft.freeze_targets()
    assert not ft.target_poses.flags.writeable
    with pytest.raises(ValueError):
        ft.target_poses[0, 9] = 0.0
    V2 = solve_ik_batch(cb, [ft, po], dt).copy()
    assert np.array_equal(V2, V1)
    puts = []
    api = pink_amd.runtime.default_solver()
    orig = type(api).put
    try:
        type(api).put = lambda self, ptr, arr: (puts.append(np.asarray(arr).nbytes), orig(self, ptr, arr))[1]
        V3 = solve_ik_batch(cb, [ft, po], dt).copy()
    finally:
        type(api).put = orig
    assert np.array_equal(V3, V1)
    assert pink_amd.last_solve_stats()["route"] == "device"
    assert 64 * 12 * 8 not in puts, puts  # (the frozen [64, 12] target array did not go up again)
    # new targets: a fresh (writeable, unfrozen) array replaces the frozen one
    ft.set_target_poses(R, t + 0.02)
    V4 = solve_ik_batch(cb, [ft, po], dt)
    ref = solve_ik(cfgs[3], [_own_frame_task(R, t + 0.02, 3), po], dt)
    assert np.abs(V4[3] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()) and np.abs(V4 - V1).max() > 1e-3


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def async_backend(request):
    """(solver, batch size): the emulator with the asynchronous surface (host logic of the pipelined call) / the MI355X."""
    if request.param == "emu":
        return request.getfixturevalue("emu_async"), 67
    return request.getfixturevalue("gpu_solver"), 4099


def test_page_locked_pipelined_call_on_two_compute_streams(async_backend, monkeypatch):
    """The pipelined array call as a control loop makes it -- q, targets and out= in page-locked memory, so uploads,
    range kernels (alternating between the handle's two compute streams) and downloads are all in flight at once -- with
    frozen targets, with dense rows (a constraint; barriers) and with one compute stream: bit for bit the single launch."""
    import sys

    sik = sys.modules["pink_amd.solve_ik"]
    gpu_solver, B = async_backend
    set_default_solver(gpu_solver)
    try:
        dt = 5e-3  # (B: ranges of unequal size)
        m = build_chain(14, free_flyer=True, seed=3, limit=2.6, velocity=4.0)
        rng = np.random.default_rng(7)
        q = pink_amd.pinned_empty((B, m.nq))
        q[:] = _draw_q(m, B, rng)
        c0 = Configuration(m, q[0])
        tasks = []
        for f in ("tool0", "joint_6"):
            ft = FrameTask(f, 1.0, 0.5, lm_damping=1e-3)
            T0 = c0.get_transform_frame_to_world(f)
            ft.set_target_poses(np.broadcast_to(T0.rotation, (B, 3, 3)), T0.translation + 0.05 * rng.normal(size=(B, 3)), out=pink_amd.pinned_empty((B, 12)))
            tasks.append(ft)
        po = PostureTask(cost=5e-2)
        po.set_target(m.neutral())
        tasks.append(po)
        hold = FrameTask("joint_12", 1.0, 1.0, gain=0.5)
        hold.set_target(c0.get_transform_frame_to_world("joint_12"))
        q[:] = q[0]  # (the constraint is within reach everywhere)
        p_tool = c0.get_transform_frame_to_world("tool0").translation
        bars = [PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[2] + 0.01]), gain=np.array([50.0]), safe_displacement_gain=1.0),
                BodySphericalBarrier(("tool0", "joint_3"), d_min=0.01, gain=10.0)]
        cb = ConfigurationBatch(m, q)
        for kw in (dict(), dict(constraints=[hold]), dict(barriers=bars)):
            monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 1 << 30)
            pink_amd.clear_device_cache()
            V_one = solve_ik_batch(cb, tasks, dt, **kw).copy()
            assert pink_amd.last_solve_stats()["route"] == "device" and np.abs(V_one).max() > 1e-3
            monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 64)
            out = pink_amd.pinned_empty((B, m.nv))
            for env in ("0", "1"):
                monkeypatch.setenv("PINKHIP_ONE_COMPUTE_STREAM", env)
                pink_amd.clear_device_cache()
                for _ in range(2):  # fresh device state, then the cached one
                    out[:] = np.nan
                    V = solve_ik_batch(cb, tasks, dt, out=out, **kw)
                    assert V is out and np.array_equal(V, V_one), (kw.keys(), env)
        # frozen targets: the second call uploads q only, same result
        for ft in tasks[:2]:
            ft.freeze_targets()
        monkeypatch.setenv("PINKHIP_ONE_COMPUTE_STREAM", "0")
        pink_amd.clear_device_cache()
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 1 << 30)
        V_one = solve_ik_batch(cb, tasks, dt).copy()
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 64)
        pink_amd.clear_device_cache()
        out = pink_amd.pinned_empty((B, m.nv))
        for _ in range(3):
            out[:] = np.nan
            assert np.array_equal(solve_ik_batch(cb, tasks, dt, out=out), V_one)
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)


def test_page_locked_results_written_by_the_kernel(async_backend, monkeypatch):
    """A page-locked ``out`` (and the small status / iteration arrays behind it) is written by the whole-step kernel itself
    (mapped host memory) instead of being copied home range by range: same bits as the copies (PINKHIP_RESULT_COPIES=1)
    and as the single launch, also after ``q`` and the targets are refilled in place."""
    import sys

    sik = sys.modules["pink_amd.solve_ik"]
    gpu_solver, B = async_backend
    set_default_solver(gpu_solver)
    try:
        dt = 5e-3
        m = build_chain(14, free_flyer=True, seed=3, limit=2.6, velocity=4.0)
        rng = np.random.default_rng(11)
        q = pink_amd.pinned_empty((B, m.nq))
        q[:] = _draw_q(m, B, rng)
        c0 = Configuration(m, q[0])
        block = pink_amd.pinned_empty((2, B, 12))
        tasks = []
        for k, f in enumerate(("tool0", "joint_6")):
            ft = FrameTask(f, 1.0, 0.5, lm_damping=1e-3)
            T0 = c0.get_transform_frame_to_world(f)
            ft.set_target_poses(np.broadcast_to(T0.rotation, (B, 3, 3)), T0.translation + 0.05 * rng.normal(size=(B, 3)), out=block[k])
            tasks.append(ft)
        po = PostureTask(cost=5e-2)
        po.set_target(m.neutral())
        tasks.append(po)
        cb = ConfigurationBatch(m, q)
        out = pink_amd.pinned_empty((B, m.nv))
        for step in range(2):
            monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 1 << 30)
            pink_amd.clear_device_cache()
            V_one = solve_ik_batch(cb, tasks, dt).copy()
            assert pink_amd.last_solve_stats()["route"] == "device" and np.abs(V_one).max() > 1e-3
            monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 64)
            for rc in ("0", "1"):
                monkeypatch.setenv("PINKHIP_RESULT_COPIES", rc)
                pink_amd.clear_device_cache()
                gets = getattr(gpu_solver, "async_gets", None)
                for _ in range(2):
                    out[:] = np.nan
                    V = solve_ik_batch(cb, tasks, dt, out=out)
                    assert V is out and np.array_equal(V, V_one), (step, rc)
                if gets is not None:  # (emulator: the result stream is used exactly when the copies are asked for)
                    assert (gpu_solver.async_gets > gets) == (rc == "1") and set(gpu_solver.streams_selected) == {0, 1}
            # the next control step: configurations and targets refilled in place
            q[:] = _draw_q(m, B, rng)
            block[:, :, 9:] += 0.01 * rng.normal(size=(2, B, 3))
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)


def test_a_batch_outside_its_limits_never_hands_out_velocities(async_backend, monkeypatch):
    """``Configuration.check_limits`` (``pink/solve_ik.py:260``) runs on the whole batch; in the pipelined call the range
    kernels are already writing into a page-locked ``out`` when the check fails: the exception propagates, the array is
    blanked (NaN) and the next call with repaired configurations is served normally."""
    import sys

    from pink_amd.exceptions import NotWithinConfigurationLimits

    sik = sys.modules["pink_amd.solve_ik"]
    solver, B = async_backend
    set_default_solver(solver)
    try:
        dt = 5e-3
        m = build_chain(14, free_flyer=True, seed=3, limit=2.6, velocity=4.0)
        rng = np.random.default_rng(13)
        q = pink_amd.pinned_empty((B, m.nq))
        q[:] = _draw_q(m, B, rng)
        c0 = Configuration(m, q[0])
        ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
        T0 = c0.get_transform_frame_to_world("tool0")
        ft.set_target_poses(np.broadcast_to(T0.rotation, (B, 3, 3)), T0.translation + 0.05 * rng.normal(size=(B, 3)), out=pink_amd.pinned_empty((B, 12)))
        po = PostureTask(cost=5e-2)
        po.set_target(m.neutral())
        tasks = [ft, po]
        cb = ConfigurationBatch(m, q)
        out = pink_amd.pinned_empty((B, m.nv))
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 64)
        good = solve_ik_batch(cb, tasks, dt, out=out).copy()
        bad_b, bad_i = B - 3, 7 + 4  # (a joint behind the free flyer, in the last range)
        keep = q[bad_b, bad_i]
        q[bad_b, bad_i] = 2.6 + 0.5
        out[:] = 0.0
        with pytest.raises(NotWithinConfigurationLimits):
            solve_ik_batch(cb, tasks, dt, out=out)
        assert np.isnan(out).all()
        q[bad_b, bad_i] = keep
        assert np.array_equal(solve_ik_batch(cb, tasks, dt, out=out), good)
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)
