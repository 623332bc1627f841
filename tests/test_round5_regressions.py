"""Round 5: the regressions the round-4 advisor reproduced, each against Pink's calling pattern (one solve_ik per
configuration, pink/solve_ik.py:206-275).  Emulator here, MI355X under -m gpu."""
import numpy as np
import pytest

import pink_amd
from pink_amd import (Configuration, ConfigurationBatch, DampingTask, FrameTask, PostureTask, build_chain, solve_ik,
                      solve_ik_batch)
from pink_amd.barriers import PositionBarrier
from pink_amd.exceptions import PinkError
from pink_amd.lie import exp6
from pink_amd.limits import AccelerationLimit, ConfigurationLimit, VelocityLimit
from pink_amd.runtime import set_default_solver
from pink_amd.tasks import JointCouplingTask, LinearHolonomicTask, RelativeFrameTask

from tests.test_round4 import _draw_q


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    s = request.getfixturevalue("emu" if request.param == "emu" else "gpu_solver")
    set_default_solver(s)
    yield request.param
    pink_amd.clear_device_cache()
    set_default_solver(None)


def _per_configuration(m, q, make_tasks, dt, n, **kw):
    return np.array([solve_ik(Configuration(m, q[b]), make_tasks(b), dt, **kw) for b in range(n)])


@pytest.mark.parametrize("route", ["hybrid", "host-evaluated"])
def test_a_configuration_batch_refilled_in_place_is_evaluated_at_its_new_q(backend, route):
    """``ConfigurationBatch.q`` aliases the caller's array (``pinned_empty`` recommends refilling it in place): the
    second call must see the second state on every route -- the forward kinematics of the first call are not kept."""
    m = build_chain(7, seed=3, limit=2.8, velocity=6.0)
    rng = np.random.default_rng(5)
    B, dt = 66, 5e-3
    q = _draw_q(m, B, rng)
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    T0 = Configuration(m, m.neutral()).get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6))
    ft.set_target(T0)
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    tasks = [ft, po, DampingTask(cost=1e-2)]
    # (two acceleration limits: not the device route; a barrier keeps the host's forward kinematics in play)
    limits = [ConfigurationLimit(m, 0.7), VelocityLimit(m), AccelerationLimit(m, np.full(7, 500.0)), AccelerationLimit(m, np.full(7, 800.0))]
    bars = [PositionBarrier("tool0", indices=[2], p_max=np.array([10.0]), gain=np.array([50.0]), safe_displacement_gain=1.0)]
    kw = dict(limits=limits, barriers=bars)
    if route == "host-evaluated":
        kw.update(device_kinematics=False, gpu_frame_tasks=False)
    cb = ConfigurationBatch(m, q)
    assert cb.q is q  # the alias the docstring of pinned_empty relies on
    V1 = solve_ik_batch(cb, tasks, dt, **kw).copy()
    assert pink_amd.last_solve_stats()["route"] == route
    q[...] = _draw_q(m, B, rng)  # the next control step's configurations, in place
    V2 = solve_ik_batch(cb, tasks, dt, **kw)
    fresh = solve_ik_batch(ConfigurationBatch(m, q.copy()), tasks, dt, **kw)
    assert np.array_equal(V2, fresh) and np.abs(V2 - V1).max() > 1e-2
    ref = _per_configuration(m, q, lambda b: tasks, dt, 6, limits=limits, barriers=bars)
    assert np.abs(V2[:6] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())


def test_hybrid_route_refuses_frame_slots_with_different_gains(backend):
    """Per-instance FrameTask objects whose gains differ: the device plan declines them, the hybrid route read the gain
    off instance 0 for the whole batch (silently wrong velocities); now it refuses like the host route."""
    m = build_chain(7, seed=3, limit=2.8, velocity=6.0)
    rng = np.random.default_rng(9)
    B, dt = 65, 5e-3
    q = _draw_q(m, B, rng)
    cfgs = [Configuration(m, q[b]) for b in range(B)]
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    per_instance = []
    for b, c in enumerate(cfgs):
        fb = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3, gain=1.0 if b != 1 else 0.4)
        fb.set_target(c.get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6)))
        per_instance.append([fb, po])
    with pytest.raises(PinkError, match="gain"):
        solve_ik_batch(cfgs, per_instance, dt)
    with pytest.raises(PinkError, match="gain"):
        solve_ik_batch(cfgs, per_instance, dt, device_kinematics=False, gpu_frame_tasks=False)


@pytest.mark.parametrize("free_flyer", [False, True])
def test_constant_row_tasks_with_different_reference_configurations(backend, free_flyer):
    """Two JointCouplingTasks built at different configurations (different q_0): the device route brings them to one
    reference (b absorbs A (q0_i - q0_0)) instead of aborting the call in automatic mode."""
    m = build_chain(8, free_flyer=free_flyer, seed=7, limit=2.6, velocity=4.0)
    rng = np.random.default_rng(13)
    B, dt = 64, 5e-3
    q = _draw_q(m, B, rng)
    ca, cbb = Configuration(m, q[0]), Configuration(m, q[1])
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    ft.set_target(ca.get_transform_frame_to_world("tool0") * exp6(0.05 * rng.normal(size=6)))
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    j1 = JointCouplingTask(["joint_2", "joint_3"], [1.0, -1.0], 2.0, ca)  # q_0 = neutral
    A = np.zeros((2, m.nv))
    o = 6 if free_flyer else 0
    A[0, o + 4], A[0, o + 5], A[1, o + 6] = 1.0, 0.5, 1.0
    j2 = LinearHolonomicTask(A, np.array([0.05, -0.02]), cbb.q.copy(), cost=[1.5, 0.7], gain=0.9)  # q_0 = a drawn configuration
    assert not np.array_equal(j1.q_0, j2.q_0)
    tasks = [ft, po, j1, j2]
    V = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt)
    assert pink_amd.last_solve_stats()["route"] == "device"
    V_host = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt, device_kinematics=False, gpu_frame_tasks=False)
    assert np.abs(V - V_host).max() < 1e-8 * max(1.0, np.abs(V_host).max())
    ref = _per_configuration(m, q, lambda b: tasks, dt, 6)
    assert np.abs(V[:6] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())


def test_a_relative_slot_beyond_the_sixteenth_frame_leaves_the_device_route(backend):
    """The device model keeps relative slots among its first 16 frames (csrc/model_tables.h): a stack with a
    RelativeFrameTask at slot 16 is served by another route instead of failing in model_create."""
    m = build_chain(6, seed=3, limit=2.8, velocity=6.0)
    rng = np.random.default_rng(17)
    B, dt = 64, 5e-3
    q = _draw_q(m, B, rng)
    c0 = Configuration(m, q[0])
    tasks = []
    for k in range(16):
        ft = FrameTask("tool0" if k % 2 == 0 else "joint_4", 1.0 / 16.0, 0.1, lm_damping=1e-3)
        ft.set_target(c0.get_transform_frame_to_world(ft.frame) * exp6(0.02 * rng.normal(size=6)))
        tasks.append(ft)
    rt = RelativeFrameTask("tool0", "joint_3", 0.5, 0.1)
    rt.set_target(c0.get_transform("tool0", "joint_3"))
    po = PostureTask(cost=5e-2)
    po.set_target(m.neutral())
    tasks += [rt, po]
    V = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt)
    assert pink_amd.last_solve_stats()["route"] != "device"
    ref = _per_configuration(m, q, lambda b: tasks, dt, 4)
    assert np.abs(V[:4] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("partly", [False, True])
def test_velocity_bounds_on_the_floating_base_coordinates(backend, partly):
    """A VelocityLimit built with its own vector (``pink/limits/velocity_limit.py:46-73``) whose entries on the free-flyer's
    six tangent coordinates are finite bounds those coordinates like any joint (rows ``+-e_i dq <= dt v_i``, ``:118-121``):
    the device route carries them in the box of the root coordinates (next to a FloatingBaseVelocityLimit's, if any)
    instead of declining the stack."""
    m = build_chain(8, free_flyer=True, seed=7, limit=2.6, velocity=4.0)
    rng = np.random.default_rng(23)
    B, dt = 64, 5e-3
    q = _draw_q(m, B, rng)
    vroot = np.array([0.3, 0.5, 0.4, 0.6, 0.9, 0.7])
    if partly:  # (one entry missing: the free-flyer is then not a velocity-limited joint at all, velocity_limit.py:66-74)
        vroot[4] = np.inf
    v = np.asarray(m.velocityLimit, dtype=float).copy()
    v[:6] = vroot
    limits = [ConfigurationLimit(m), VelocityLimit(m, v)]
    c0 = Configuration(m, q[0])
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    ft.set_target(c0.get_transform_frame_to_world("tool0") * exp6(0.3 * rng.normal(size=6)))  # far: the root saturates
    po = PostureTask(cost=1e-3)
    po.set_target(m.neutral())
    tasks = [ft, po]
    V = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt, limits=limits)
    assert pink_amd.last_solve_stats()["route"] == "device"
    ref = _per_configuration(m, q, lambda b: tasks, dt, 8, limits=limits)
    assert np.abs(V[:8] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
    unbounded = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt)
    if partly:
        assert np.abs(unbounded - V).max() < 1e-8 * max(1.0, np.abs(V).max())
    else:
        assert (np.abs(np.abs(ref[:, :6]) - vroot) < 1e-9).any() and np.abs(unbounded - V).max() > 1e-3  # some root bound is active


@pytest.mark.parametrize("attached", [False, True])
def test_a_floating_base_velocity_limit_written_into_the_limits_list(backend, attached):
    """``limits=[ConfigurationLimit, VelocityLimit, FloatingBaseVelocityLimit]`` with a floating-base limit that is NOT the
    model's attribute (``pink/solve_ik.py:94-105`` reads the attribute only for ``limits=None``): the plan carries the
    limit of the call to the device model; a model that has one attached but is called with a list that omits it runs
    without it."""
    from pink_amd.limits import FloatingBaseVelocityLimit

    m = build_chain(8, free_flyer=True, seed=7, limit=2.6, velocity=4.0)
    rng = np.random.default_rng(29)
    B, dt = 64, 5e-3
    q = _draw_q(m, B, rng)
    fbl = FloatingBaseVelocityLimit(m, None, max_linear_velocity=[0.2, 0.3, 0.25], max_angular_velocity=[0.4, np.inf, 0.5])
    c0 = Configuration(m, q[0])
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    ft.set_target(c0.get_transform_frame_to_world("tool0") * exp6(0.3 * rng.normal(size=6)))
    po = PostureTask(cost=1e-2)
    po.set_target(m.neutral())
    tasks = [ft, po]
    if attached:
        m.floating_base_velocity_limit = fbl
        limits = [ConfigurationLimit(m), VelocityLimit(m)]  # (the list omits the attached limit: the call runs without it)
    else:
        limits = [ConfigurationLimit(m), VelocityLimit(m), fbl]
    try:
        V = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt, limits=limits)
        assert pink_amd.last_solve_stats()["route"] == "device"
        ref = _per_configuration(m, q, lambda b: tasks, dt, 8, limits=limits)
        assert np.abs(V[:8] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
        bounded = np.abs(V[:, :3]).max() <= 0.3 + 1e-9
        assert bounded == (not attached)
    finally:
        m.floating_base_velocity_limit = None


def test_a_limits_list_without_a_velocity_limit(backend):
    """``limits=[ConfigurationLimit(model)]``: Pink stacks the rows of the limits it is given and nothing else
    (``pink/solve_ik.py:107-113``) -- the device route serves the call with a velocity table that bounds no coordinate."""
    m = build_chain(7, seed=3, limit=2.8, velocity=0.5)
    rng = np.random.default_rng(31)
    B, dt = 64, 5e-3
    q = _draw_q(m, B, rng)
    ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
    ft.set_target(Configuration(m, q[0]).get_transform_frame_to_world("tool0") * exp6(0.3 * rng.normal(size=6)))
    po = PostureTask(cost=1e-2)
    po.set_target(m.neutral())
    tasks = [ft, po]
    limits = [ConfigurationLimit(m, 0.6)]
    V = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt, limits=limits)
    assert pink_amd.last_solve_stats()["route"] == "device"
    ref = _per_configuration(m, q, lambda b: tasks, dt, 8, limits=limits)
    assert np.abs(V[:8] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
    assert np.abs(V).max() > 0.5 + 1e-3  # (faster than the model's velocity limit would allow)
    bounded = solve_ik_batch(ConfigurationBatch(m, q), tasks, dt)
    assert np.abs(bounded).max() <= 0.5 + 1e-9
