"""Accuracy of the SE(3) logarithm's coefficient functions (host NumPy, per instance and batched) against 60-digit
arithmetic: alpha = (th/2) cot(th/2), beta = (1 - alpha) / th^2, beta_dot = beta'(th) / th (pin.log6 / pin.Jlog6 at the
call sites pink/tasks/frame_task.py:181-193,222-227).  Their closed forms subtract numbers of order 1/th^2 and 1/th^4:
at th = 1e-3 -- a tracking controller's orientation error -- they are wrong by 2e-4 and 1e4 relative in double precision,
which one ulp of difference in sin / cos between the GPU and NumPy turned into 1e-8 relative on dq
(scripts/gpu_fuzz_rollout.py, round 4).  The device code (pink_amd/csrc/ik_frame_task.h) uses the same series; it is held to
the host functions by tests/test_api.py::test_frame_task_kernel_matches_host_lie_and_oracle at small angles."""
import numpy as np
import pytest

from pink_amd import lie, lie_batch

mp = pytest.importorskip("mpmath")


def _exact(th):
    mp.mp.dps = 60
    th = mp.mpf(float(th))
    s, c = mp.sin(th), mp.cos(th)
    return th * s / (2 * (1 - c)), 1 / (th * th) - s / (2 * th * (1 - c)), -2 / th**4 + (1 + s / th) / (2 * th * th * (1 - c))


def test_coefficients_of_log6_and_jlog6_to_double_precision():
    ths = np.r_[10.0 ** np.arange(-7, -0.5, 0.25), 0.3, 0.49, 0.4999, 0.5, 0.51, 0.8, 1.5, 2.5, 3.1]
    ab = lie_batch._alpha_beta(ths)
    for k, th in enumerate(ths):
        alpha, beta, beta_dot = _exact(th)
        a, b = lie._alpha_beta(float(th))
        bd = lie._beta_dot(float(th))
        tol_b, tol_bd = (2e-15, 2e-15) if th < lie.SERIES_TH else (1e-13, 2e-11)
        assert abs((mp.mpf(a) - alpha) / alpha) < 2e-15, th
        assert abs((mp.mpf(b) - beta) / beta) < tol_b, th
        assert abs((mp.mpf(bd) - beta_dot) / beta_dot) < tol_bd, th
        assert ab[0][k] == a and ab[1][k] == b  # (the batched functions evaluate the same expressions)


def test_log6_inverts_exp6_at_small_angles():
    rng = np.random.default_rng(3)
    for th in (1e-7, 1e-5, 1e-3, 3e-2, 0.1, 0.3, 0.6, 2.0):
        w = rng.normal(size=3)
        xi = np.r_[rng.normal(size=3), w * th / np.linalg.norm(w)]
        M = lie.exp6(xi)
        assert np.abs(lie.log6(M) - xi).max() < 2e-15 * max(1.0, np.abs(xi).max())
        # Jlog6 against a central difference of log6(M exp6(d)) (h = 1e-5: truncation 1e-10, round-off 1e-11)
        J = lie.Jlog6(M)
        h = 1e-5
        for j in range(6):
            d = np.zeros(6)
            d[j] = h
            col = (lie.log6(M * lie.exp6(d)) - lie.log6(M * lie.exp6(-d))) / (2 * h)
            assert np.abs(J[:, j] - col).max() < 5e-9
