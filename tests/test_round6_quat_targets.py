"""Round 6 (round-5 review, item 9): FrameTask targets handed over as translation + unit quaternion, 7 numbers per pose in
Pinocchio's `(x, y, z, w)` order, instead of 12 -- a moving-target call sends 56 B per frame task and robot instead of 96 B,
`pinkhip_pose_targets_device` writes the poses the kernels read.  Instance b plays
`FrameTask.set_target(pin.XYZQUATToSE3(...))` (pink/tasks/frame_task.py:129-137): same velocities as the same targets
handed over as rotation matrices, on every route.  Emulator here, MI355X under -m gpu."""
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import pink_amd
from pink_amd import Configuration, ConfigurationBatch, FrameTask, PostureTask, build_chain, solve_ik, solve_ik_batch
from pink_amd.exceptions import TaskDefinitionError
from pink_amd.lie import SE3, exp6
from pink_amd.runtime import set_default_solver
from pink_amd.tasks.frame_task import poses_from_pq

from tests.test_round4 import _draw_q


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def async_backend(request):
    if request.param == "emu":
        return request.getfixturevalue("emu_async"), 67
    return request.getfixturevalue("gpu_solver"), 4099


def _targets(m, q, frames, rng):
    """Per robot and frame: the frame's pose moved by a random twist; as rotation matrices and as quaternions (x, y, z, w)."""
    B = q.shape[0]
    out = {}
    for f in frames:
        R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
        for b in range(B):
            T = Configuration(m, q[b]).get_transform_frame_to_world(f) * exp6(0.05 * rng.normal(size=6))
            R[b], t[b] = T.rotation, T.translation
        out[f] = (R, t, Rotation.from_matrix(R).as_quat())  # (scipy: scalar last, Pinocchio's order)
    return out


def test_host_statement_of_the_conversion():
    rng = np.random.default_rng(0)
    quat = rng.normal(size=(50, 4))
    quat /= np.linalg.norm(quat, axis=1)[:, None]
    t = rng.normal(size=(50, 3))
    P = poses_from_pq(np.hstack([t, quat]))
    assert np.abs(P[:, :9].reshape(-1, 3, 3) - Rotation.from_quat(quat).as_matrix()).max() < 1e-15
    assert np.array_equal(P[:, 9:], t)
    ft = FrameTask("tool0", 1.0, 1.0)
    with pytest.raises(TaskDefinitionError):
        ft.set_target_poses_quat(t, 2.0 * quat)  # not unit
    with pytest.raises(TaskDefinitionError):
        ft.set_target_poses_quat(t, quat[:, :3])
    ft.set_target_poses_quat(t, quat)
    assert ft.target_poses is None and ft.target_array().shape == (50, 7) and np.array_equal(ft.poses12(), P)
    ft.set_target(SE3(np.eye(3), np.zeros(3)))  # one target source is live at a time
    assert ft.target_array() is None


def test_device_kernel_writes_the_poses(async_backend):
    solver, B = async_backend
    rng = np.random.default_rng(1)
    quat = rng.normal(size=(B, 4))
    quat /= np.linalg.norm(quat, axis=1)[:, None]
    pq = np.hstack([rng.normal(size=(B, 3)), quat * (1.0 + 1e-9 * rng.normal(size=(B, 1)))])  # (normalised on the device)
    d_pq, d_T = solver.alloc(pq.nbytes), solver.alloc(8 * 12 * B)
    try:
        solver.put(d_pq, pq)
        solver.pose_targets(B, d_pq, d_T)
        T = np.empty((B, 12))
        solver.get(T, d_T)
        solver.sync()
    finally:
        solver.release(d_pq)
        solver.release(d_T)
    assert np.abs(T - poses_from_pq(pq)).max() < 4e-15  # (the kernel contracts into FMAs)
    R = T[:, :9].reshape(B, 3, 3)
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 4e-15


@pytest.mark.parametrize("pipelined", [False, True])
def test_quaternion_targets_give_the_velocities_of_matrix_targets(async_backend, monkeypatch, pipelined):
    """Two FrameTasks + posture on a floating-base chain: device route with [B, 7] targets (single launch and the pipelined
    page-locked call, moving and frozen) against the same call with [B, 12] targets, against the host-evaluated route, and
    against solve_ik per configuration with SE3 targets built from the quaternions."""
    solver, B = async_backend
    sik = sys.modules["pink_amd.solve_ik"]
    set_default_solver(solver)
    try:
        dt = 5e-3
        m = build_chain(14, free_flyer=True, seed=3, limit=2.6, velocity=4.0)
        rng = np.random.default_rng(11)
        q = pink_amd.pinned_empty((B, m.nq))
        q[:] = _draw_q(m, B, rng)
        frames = ("tool0", "joint_6")
        tg = _targets(m, q, frames, rng)
        po = PostureTask(cost=5e-2)
        po.set_target(m.neutral())
        mat, qua = [], []
        for f in frames:
            R, t, quat = tg[f]
            a = FrameTask(f, 1.0, 0.5, lm_damping=1e-3)
            a.set_target_poses(R, t, out=pink_amd.pinned_empty((B, 12)))
            b = FrameTask(f, 1.0, 0.5, lm_damping=1e-3)
            b.set_target_poses_quat(t, quat, out=pink_amd.pinned_empty((B, 7)))
            mat.append(a), qua.append(b)
        cb = ConfigurationBatch(m, q)
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 64 if pipelined else 1 << 30)
        pink_amd.clear_device_cache()
        V_mat = solve_ik_batch(cb, mat + [po], dt).copy()
        assert pink_amd.last_solve_stats()["route"] == "device" and np.abs(V_mat).max() > 1e-3
        pink_amd.clear_device_cache()
        out = pink_amd.pinned_empty((B, m.nv))
        for _ in range(2):  # fresh device state, then the cached one
            out[:] = np.nan
            V = solve_ik_batch(cb, qua + [po], dt, out=out)
            assert pink_amd.last_solve_stats()["route"] == "device"
            assert np.abs(V - V_mat).max() < 1e-9 * max(1.0, np.abs(V_mat).max())
        # the targets move in place between two calls (a control loop refills its page-locked arrays)
        for a, b, f in zip(mat, qua, frames):
            R, t, quat = tg[f]
            a.target_poses[:, 9:] += 0.01
            b.target_pq[:, :3] += 0.01
        V_mat2 = solve_ik_batch(cb, mat + [po], dt).copy()
        V2 = solve_ik_batch(cb, qua + [po], dt).copy()
        assert np.abs(V_mat2 - V_mat).max() > 1e-4 and np.abs(V2 - V_mat2).max() < 1e-9 * max(1.0, np.abs(V_mat2).max())
        # frozen: uploaded and converted once per device state
        for b in qua:
            b.freeze_targets()
            assert not b.target_pq.flags.writeable
        for _ in range(3):
            assert np.abs(solve_ik_batch(cb, qua + [po], dt) - V_mat2).max() < 1e-9 * max(1.0, np.abs(V_mat2).max())
        # the host-evaluated route reads the same targets
        n = 9
        cbn = ConfigurationBatch(m, q[:n].copy())
        small = []
        for f in frames:
            R, t, quat = tg[f]
            s = FrameTask(f, 1.0, 0.5, lm_damping=1e-3)
            s.set_target_poses_quat(t[:n] + 0.01, quat[:n])
            small.append(s)
        V_host = solve_ik_batch(cbn, small + [po], dt, device_kinematics=False)
        assert np.abs(V_host - V_mat2[:n]).max() < 1e-8 * max(1.0, np.abs(V_mat2).max())
        # ... and Pink's calling pattern: one solve_ik per configuration with SE3 targets built from the quaternions
        for b in range(3):
            own = []
            for f in frames:
                R, t, quat = tg[f]
                o = FrameTask(f, 1.0, 0.5, lm_damping=1e-3)
                o.set_target(SE3(Rotation.from_quat(quat[b]).as_matrix(), t[b] + 0.01))
                own.append(o)
            v = solve_ik(Configuration(m, q[b].copy()), own + [po], dt)
            assert np.abs(v - V_mat2[b]).max() < 1e-8 * max(1.0, np.abs(V_mat2).max())
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)


def test_a_moving_target_call_sends_fewer_bytes(async_backend, monkeypatch):
    solver, B = async_backend
    sik = sys.modules["pink_amd.solve_ik"]
    set_default_solver(solver)
    try:
        m = build_chain(14, free_flyer=True, seed=3, limit=2.6, velocity=4.0)
        rng = np.random.default_rng(5)
        q = pink_amd.pinned_empty((B, m.nq))
        q[:] = _draw_q(m, B, rng)
        tg = _targets(m, q, ("tool0",), rng)["tool0"]
        po = PostureTask(cost=5e-2)
        po.set_target(m.neutral())
        monkeypatch.setattr(sik, "_PIPELINE_MIN_B", 64)
        sent = {}
        for kind in ("matrix", "quaternion"):
            ft = FrameTask("tool0", 1.0, 0.5, lm_damping=1e-3)
            if kind == "matrix":
                ft.set_target_poses(tg[0], tg[1], out=pink_amd.pinned_empty((B, 12)))
            else:
                ft.set_target_poses_quat(tg[1], tg[2], out=pink_amd.pinned_empty((B, 7)))
            pink_amd.clear_device_cache()
            solve_ik_batch(ConfigurationBatch(m, q), [ft, po], 5e-3, out=pink_amd.pinned_empty((B, m.nv)))
            ro = next(reversed(sik._rollout_cache(solver).values()))
            sent[kind] = ro.bytes_in_last_call
        assert sent["matrix"] - sent["quaternion"] == 8 * 5 * B
    finally:
        pink_amd.clear_device_cache()
        set_default_solver(None)
