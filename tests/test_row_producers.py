"""The remaining generic row producers of SURVEY.md 8(f-4): RelativeFrameTask, LinearHolonomicTask,
JointCouplingTask, JointVelocityTask, BodySphericalBarrier.  Jacobians are checked against central
finite differences of the errors (the reference's own method, tests/test_jacobians.py:47-84, h = 1e-6,
tol 1e-5), then each producer is driven through the stack + solve kernel (emulator here, MI355X when
marked gpu) and compared with the oracle on the same rows."""
import numpy as np
import pytest

from oracle import c_oracle
from pink_amd import Configuration, FrameTask, PostureTask, build_chain, build_ik, solve_ik
from pink_amd.barriers import BodySphericalBarrier
from pink_amd.exceptions import NegativeMinimumDistance, TargetNotSet, TaskDefinitionError, TaskJacobianNotSet
from pink_amd.lie import SE3, exp6
from pink_amd.runtime import set_default_solver
from pink_amd.tasks import JointCouplingTask, JointVelocityTask, LinearHolonomicTask, RelativeFrameTask

Q6 = np.array([0.3, -0.8, 1.2, 0.4, -0.3, 0.5])


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "emu":
        set_default_solver(request.getfixturevalue("emu"))
    else:
        set_default_solver(request.getfixturevalue("gpu_solver"))
    yield request.param
    set_default_solver(None)


def _floating_arm():
    m = build_chain(5, free_flyer=True)
    q = m.neutral()
    q = m.integrate(q, np.array([0.1, -0.2, 0.3, 0.2, -0.1, 0.4, 0.3, -0.5, 0.8, 0.2, -0.4]))
    return m, q


def _fd_jacobian(m, q, fn, h=1e-6):
    cols = []
    for k in range(m.nv):
        d = np.zeros(m.nv)
        d[k] = h
        cols.append((fn(Configuration(m, m.integrate(q, d))) - fn(Configuration(m, m.integrate(q, -d)))) / (2 * h))
    return np.stack(cols, axis=1)


def test_relative_frame_task_jacobian_and_target():
    m, q = _floating_arm()
    cfg = Configuration(m, q)
    task = RelativeFrameTask("tool0", "joint_2", position_cost=1.0, orientation_cost=0.5)
    with pytest.raises(TargetNotSet):
        task.compute_error(cfg)
    task.set_target_from_configuration(cfg)
    assert np.linalg.norm(task.compute_error(cfg)) < 1e-12
    task.set_target(task.transform_target_to_root * exp6(np.array([0.05, -0.02, 0.03, 0.2, -0.1, 0.15])))
    J = task.compute_jacobian(cfg)
    fd = _fd_jacobian(m, q, task.compute_error)
    assert J.shape == (6, m.nv) and np.abs(J - fd).max() < 1e-5
    assert np.abs(J[:, :6]).max() < 1e-12  # a relative pose does not depend on the floating base
    with pytest.raises(TaskDefinitionError):
        task.set_position_cost(-1.0)


def test_linear_holonomic_and_joint_coupling():
    m, q = _floating_arm()
    cfg = Configuration(m, q)
    rng = np.random.default_rng(0)
    A, b = rng.normal(size=(3, m.nv)), rng.normal(size=3)
    q0 = m.integrate(m.neutral(), 0.3 * rng.normal(size=m.nv))
    task = LinearHolonomicTask(A, b, q0, cost=2.0)
    fd = _fd_jacobian(m, q, task.compute_error)
    assert np.abs(task.compute_jacobian(cfg) - fd).max() < 1e-5
    with pytest.raises(TaskDefinitionError):
        LinearHolonomicTask(A, np.zeros(2), q0)
    with pytest.raises(TaskJacobianNotSet):
        LinearHolonomicTask(A[:, :4], b, q0).compute_error(cfg)
    coupling = JointCouplingTask(["joint_2", "joint_3"], [1.0, -2.0], 100.0, cfg)
    j2, j3 = (m.joints[m.getJointId(n)] for n in ("joint_2", "joint_3"))
    assert np.isclose(coupling.compute_error(cfg)[0], q[j2.idx_q] - 2.0 * q[j3.idx_q])
    row = coupling.compute_jacobian(cfg)
    assert row.shape == (1, m.nv) and row[0, j2.idx_v] == 1.0 and row[0, j3.idx_v] == -2.0


def test_joint_coupling_holds_in_closed_loop(backend):
    """Same scenario as the reference's joint-coupling unit test: a heavily weighted coupling keeps
    ratio_2 q_2 + ratio_3 q_3 at zero while a frame task moves the arm."""
    m = build_chain(6)
    q = np.array([0.3, 0.4, 0.2, 0.4, -0.3, 0.5])  # q_2 - 2 q_3 = 0
    cfg = Configuration(m, q)
    frame = FrameTask("tool0", 1.0, 0.1, lm_damping=1e-3)
    frame.set_target_from_configuration(cfg)
    tgt = frame.transform_target_to_world.copy()
    tgt.translation[2] -= 0.1
    frame.set_target(tgt)
    coupling = JointCouplingTask(["joint_2", "joint_3"], [1.0, -2.0], 100.0, cfg)
    e0 = np.linalg.norm(frame.compute_error(cfg)[:3])
    for _ in range(30):
        cfg.integrate_inplace(solve_ik(cfg, [frame, coupling], 5e-3), 5e-3)
    assert abs(coupling.compute_error(cfg)[0]) < 1e-4
    assert np.linalg.norm(frame.compute_error(cfg)[:3]) < 0.5 * e0  # position rows carry the weight


def test_joint_velocity_task_is_diagonal_and_follows_the_reference(backend):
    m, q = _floating_arm()
    cfg = Configuration(m, q)
    task = JointVelocityTask(cost=1.0)
    with pytest.raises(TargetNotSet):
        task.compute_error(cfg)
    v_t, dt = np.array([0.1, -0.2, 0.3, 0.0, 0.05]), 1e-2
    task.set_target(v_t, dt)
    assert task.diagonal_col0(cfg) == 6 and np.array_equal(task.compute_jacobian(cfg), np.eye(m.nv)[6:])
    # the reference's compute_error returns the target displacement itself (joint_velocity_task.py:59-80; its own test
    # asserts e[0] == dt for a unit target, tests/test_joint_velocity_task.py:72,77; tests/golden/pink_round4.npz holds the
    # class's output), and every task is regulated by J dq = -gain e (pink/tasks/task.py:145-167): identical results mean
    # the velocity that comes out is -v_target
    assert np.allclose(task.compute_error(cfg), dt * v_t, atol=0.0)
    v = solve_ik(cfg, [task], dt, damping=1e-12, limits=[])
    assert np.allclose(v[6:], -v_t, atol=1e-8) and np.allclose(v[:6], 0.0, atol=1e-8)


def test_body_spherical_barrier(backend):
    m = build_chain(6)
    cfg = Configuration(m, Q6)
    with pytest.raises(NegativeMinimumDistance):
        BodySphericalBarrier(("tool0", "joint_2"), d_min=-0.1)
    p1 = cfg.get_transform_frame_to_world("tool0").translation
    p2 = cfg.get_transform_frame_to_world("joint_2").translation
    dist = np.linalg.norm(p1 - p2)
    bar = BodySphericalBarrier(("tool0", "joint_2"), d_min=0.9 * dist, gain=100.0, safe_displacement_gain=1.0)
    assert np.isclose(bar.compute_barrier(cfg)[0], dist ** 2 - (0.9 * dist) ** 2)
    fd = _fd_jacobian(m, Q6, bar.compute_barrier)
    assert np.abs(bar.compute_jacobian(cfg) - fd).max() < 1e-5
    # the QP the kernel solves is the one the oracle solves on the same rows (barrier.py:205-245)
    task = FrameTask("tool0", 1.0, 1.0)
    tgt = cfg.get_transform_frame_to_world("joint_2").copy()  # pull the tool onto joint_2
    task.set_target(tgt)
    prob = build_ik(cfg, [task], dt=5e-3, barriers=[bar])
    x, st, _, _ = c_oracle.gi_solve(prob.P, prob.q, prob.G, prob.h)
    v = solve_ik(cfg, [task], dt=5e-3, barriers=[bar])
    assert st == 0 and np.abs(v * 5e-3 - x).max() < 1e-10
    # closed loop: the distance never drops below d_min (up to the discretisation of the barrier)
    for _ in range(60):
        cfg.integrate_inplace(solve_ik(cfg, [task], 5e-3, barriers=[bar]), 5e-3)
    p1 = cfg.get_transform_frame_to_world("tool0").translation
    p2 = cfg.get_transform_frame_to_world("joint_2").translation
    assert np.linalg.norm(p1 - p2) >= 0.9 * dist - 1e-3


def test_self_collision_barrier_rows_and_solve(backend):
    """pink/barriers/self_collision_barrier.py:95-224 with sphere pairs as the distance query: barrier values
    are the `dim` smallest distances minus d_min, the Jacobian rows match finite differences of those
    distances, and solve_ik keeps the spheres from approaching faster than the barrier allows."""
    from pink_amd.barriers import SelfCollisionBarrier, SpherePairs
    from pink_amd.exceptions import InvalidCollisionPairs, NegativeMinimumDistance

    m = build_chain(7, free_flyer=True, seed=5)
    rng = np.random.default_rng(12)
    q = m.neutral()
    for j in m.joints:
        if j.kind != "free_flyer":
            q[j.idx_q] = rng.uniform(-0.9, 0.9)
    cfg = Configuration(m, q)
    nj = len(m.joints)
    pairs = SpherePairs([(1, [0.02, 0, 0.01], 0.03, nj - 1, [0, 0.01, 0.02], 0.04),
                         (2, [0, 0, 0], 0.02, nj - 2, [0.01, 0, 0], 0.03),
                         (0, [0.1, 0, 0], 0.05, nj - 1, [0, 0, 0.05], 0.02)])
    bar = SelfCollisionBarrier(2, gain=50.0, safe_displacement_gain=1.0, d_min=0.01, distance_query=pairs)
    dists = np.array([p.min_distance for p in pairs(cfg)])
    h = bar.compute_barrier(cfg)
    assert np.allclose(np.sort(h), np.sort(dists)[:2] - 0.01)
    J = bar.compute_jacobian(cfg)
    assert J.shape == (2, m.nv)
    # finite differences of the distances along tangent directions (tests/test_jacobians.py:47-75 pattern)
    order = np.argpartition(-dists, -2)[-2:]
    eps = 1e-6
    for i in range(m.nv):
        dv = np.zeros(m.nv)
        dv[i] = eps
        d_plus = np.array([p.min_distance for p in pairs(Configuration(m, m.integrate(q, dv)))])
        d_minus = np.array([p.min_distance for p in pairs(Configuration(m, m.integrate(q, -dv)))])
        assert np.abs((d_plus - d_minus)[order] / (2 * eps) - J[:, i]).max() < 1e-6
    G, hh = bar.compute_qp_inequalities(cfg, 5e-3)
    assert G.shape == (2, m.nv) and np.allclose(G, -J / 5e-3) and np.allclose(hh, 50.0 * h)
    # in the QP: a posture task that would fold the chain is slowed down by the barrier rows
    post = PostureTask(cost=1.0)
    post.set_target(m.neutral())
    v_free = solve_ik(cfg, [post], 5e-3)
    tight = SelfCollisionBarrier(3, gain=1.0, safe_displacement_gain=0.0, d_min=float(dists.min()) - 1e-4, distance_query=pairs)
    v_bar = solve_ik(cfg, [post], 5e-3, barriers=[tight])
    Gt, ht = tight.compute_qp_inequalities(cfg, 5e-3)
    assert (Gt @ (v_bar * 5e-3) <= ht + 1e-12).all()
    if (Gt @ (v_free * 5e-3) > ht + 1e-9).any():
        assert np.abs(v_bar - v_free).max() > 1e-6
    with pytest.raises(NegativeMinimumDistance):
        SelfCollisionBarrier(1, d_min=-1.0)
    with pytest.raises(InvalidCollisionPairs):
        SelfCollisionBarrier(5, distance_query=pairs).compute_barrier(cfg)
    with pytest.raises(InvalidCollisionPairs):
        SelfCollisionBarrier(1).compute_barrier(cfg)
