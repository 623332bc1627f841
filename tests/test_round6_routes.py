"""Round 6: a batch that leaves the device route says so.  ``solve_ik_batch`` logs ONE warning per (stack signature,
route) when a batch of 64 and more is served by the hybrid or the host-evaluated route (the reference's one-shot warning
style: pink/configuration.py:188-201), naming the first term that declined the device route; ``strict_route=`` makes the
downgrade an error before any work is done."""
import importlib
import logging

import numpy as np
import pytest

import pink_amd
from pink_amd import ConfigurationBatch, FrameTask, PostureTask, solve_ik_batch
from pink_amd.barriers import PositionBarrier
from pink_amd.exceptions import PinkError
from pink_amd.runtime import set_default_solver

from tests.test_device_route_round5 import _stack

solve_ik_module = importlib.import_module("pink_amd.solve_ik")


@pytest.fixture
def on_emu(emu):
    set_default_solver(emu)
    solve_ik_module._ROUTE_WARNED.clear()
    yield emu
    pink_amd.clear_device_cache()
    set_default_solver(None)


def _custom_gain_barrier(cfgs):
    p_tool = np.array([c.get_transform_frame_to_world("tool0").translation for c in cfgs])
    # a class-K function of its own: the whole-step kernel forms the default one only (position_barrier.py:95-153)
    bar = PositionBarrier("tool0", indices=[2], p_max=np.array([p_tool[:, 2].max() + 0.02]), gain=np.array([50.0]), safe_displacement_gain=1.0)
    bar.gain_function, bar.identity_gain_function = (lambda h: 2.0 * h), False  # (pink/barriers/barrier.py:49-60: gain_function=)
    return bar


def test_a_downgrade_is_logged_once_and_names_the_term(on_emu, caplog):
    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(False, 11)
    cb = ConfigurationBatch(m, q)
    with caplog.at_level(logging.WARNING, logger="pink_amd"):
        solve_ik_batch(cb, [ft, po], dt)
        assert pink_amd.last_solve_stats()["route"] == "device" and not caplog.records
        bar = _custom_gain_barrier(cfgs)
        V = solve_ik_batch(cb, [ft, po], dt, barriers=[bar])
        route = pink_amd.last_solve_stats()["route"]
        assert route in ("hybrid", "host-evaluated")
        hits = [r for r in caplog.records if "leaves the device route" in r.getMessage()]
        assert len(hits) == 1 and repr(route) in hits[0].getMessage() and "class-K" in hits[0].getMessage()
        # the same stack again: no second warning; a batch under 64: none at all
        solve_ik_batch(cb, [ft, po], dt, barriers=[bar])
        assert len([r for r in caplog.records if "leaves the device route" in r.getMessage()]) == 1
    assert np.isfinite(V).all()


def test_small_batches_and_explicit_host_route_are_silent(on_emu, caplog):
    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(False, 12, B=20)
    with caplog.at_level(logging.WARNING, logger="pink_amd"):
        solve_ik_batch(ConfigurationBatch(m, q), [ft, po], dt, barriers=[_custom_gain_barrier(cfgs)])
        m2, rng2, q2, cfgs2, ft2, po2, R2, t2 = _stack(False, 13)
        solve_ik_batch(ConfigurationBatch(m2, q2), [ft2, po2], dt, device_kinematics=False, gpu_frame_tasks=False)
        assert pink_amd.last_solve_stats()["route"] == "host-evaluated"
    assert not [r for r in caplog.records if "leaves the device route" in r.getMessage()]


def test_strict_route_raises_before_any_work(on_emu):
    dt = 5e-3
    m, rng, q, cfgs, ft, po, R, t = _stack(False, 14)
    cb = ConfigurationBatch(m, q)
    V = solve_ik_batch(cb, [ft, po], dt, strict_route="device")
    assert pink_amd.last_solve_stats()["route"] == "device" and np.isfinite(V).all()
    with pytest.raises(PinkError, match="strict_route='device'.*class-K"):
        solve_ik_batch(cb, [ft, po], dt, barriers=[_custom_gain_barrier(cfgs)], strict_route="device")
    with pytest.raises(PinkError, match="strict_route='host-evaluated'"):
        solve_ik_batch(cb, [ft, po], dt, strict_route="host-evaluated")
    with pytest.raises(PinkError, match="strict_route="):
        solve_ik_batch(cb, [ft, po], dt, strict_route="gpu")
    # the host-evaluated route on request
    Vh = solve_ik_batch(cb, [ft, po], dt, device_kinematics=False, gpu_frame_tasks=False, strict_route="host-evaluated")
    assert np.abs(V - Vh).max() < 1e-8 * max(1.0, np.abs(Vh).max())
