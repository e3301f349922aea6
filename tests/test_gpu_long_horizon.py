"""Horizons beyond 128 knots (OH_MAX_T 128 -> 256 in round 6; the reference's builder takes any T: builder.py:14-99).  T = 200 on both trajectory families --
the orientation-locked figure-eight and position tracking -- runs in the batched launches (the persistent kernels hold 64 / 128 free knots) and is compared
with the numpy ports of the state machines (oracle/structured.py) and with the literal constraints."""
import numpy as np
import pytest

import bench
from conftest import KUKA_KIN
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
from oracle.robot import OracleRobot
from oracle.structured import FoldedChain, StructuredFigureEight, solve_free_lm, solve_structured_lm

pytestmark = pytest.mark.gpu
LINK = "end_effector_ball"


def test_figure_eight_at_200_knots(hip_lib, monkeypatch):
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    T, B = 200, 96
    assert _lib.OH_MAX_T >= 256
    orc = OracleRobot(KUKA_KIN)
    prob = StructuredFigureEight(orc, LINK, T=T, Tmax=bench.TMAX)
    t = np.linspace(0.0, bench.TMAX, T)
    lp = np.zeros((T, 3))
    lp[:, 0], lp[:, 1] = 0.2 * np.sin(t * np.pi * 0.5), 0.1 * np.sin(t * np.pi)
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK)
    be = FigureEightBackend(chain, T, float(t[1] - t[0]), lp, max_iter=400, tol=1e-8, hessian=2)
    qc = np.deg2rad(bench.QC0_DEG)[None, :] + np.random.default_rng(200).uniform(-0.1, 0.1, (B, 7))
    x0 = np.concatenate([np.repeat(qc, T, axis=0).reshape(B, 7 * T), np.zeros((B, 7 * (T - 1)))], axis=1)
    r = be.solve(x0, qc)
    assert (r.status == 0).all() and (r.kkt[:, 0] <= 1e-8).all() and (r.kkt[:, 1] <= 1e-9).all()
    Q = r.x[:, : 7 * T].reshape(B, T, 7)
    dQ = r.x[:, 7 * T :].reshape(B, T - 1, 7)
    assert np.array_equal(Q[:, 0], qc) and not dQ[:, 0].any() and np.abs(Q[:, 1:] - (Q[:, :-1] + float(t[1] - t[0]) * dQ)).max() <= 1e-12
    quat_c = orc.quaternion_batch(LINK, qc)
    qt = orc.quaternion_batch(LINK, Q.reshape(-1, 7)).reshape(B, T, 4)
    assert np.abs(quat_c[:, None, :] - qt).max() <= 1e-9  # the literal orientation rows on every knot
    same = 0
    for b in range(6):
        s = solve_structured_lm(prob, qc[b], max_iter=400, tol=1e-8)
        assert s["status"] == 0
        same += abs(s["f"] - r.f[b]) <= 1e-9 * abs(s["f"])
        assert abs(int(r.iters[b]) - s["iters"]) <= max(2, s["iters"] // 4), (b, r.iters[b], s["iters"])
    assert same >= 5, same  # (a fork between two local minima is possible on any one instance)
    alone = be.solve(x0[3], qc[3])
    assert abs(alone.f[0] - r.f[3]) <= 1e-9 * abs(alone.f[0])
    be.close()


def test_position_tracking_at_200_knots(hip_lib, monkeypatch):
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    from examples.dual_arm import path_offsets

    T, B = 200, 48
    offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
    dt = 10.0 / (T - 1)
    arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
    arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    be = FigureEightBackend(arm.kinematic_chain(LINK), T, dt, offs.T, w_path=1.0, w_vel=0.01, max_iter=400, tol=1e-8, hessian=0, lock_orientation=False, fix_dq0=False,
                            path_in_frame=False)
    qc = np.deg2rad([0, -30, 0, 90, 0, 30, 0])[None, :] + np.random.default_rng(201).uniform(-0.1, 0.1, (B, 7))
    x0 = np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1)
    r = be.solve(x0, qc)
    assert (r.status == 0).all() and (r.kkt[:, 0] <= 1e-8).all()
    rob = OracleRobot(KUKA_KIN, name="kukal")
    rob.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    ch = FoldedChain(rob, LINK)
    for b in range(4):
        s = solve_free_lm(ch, T, dt, offs.T, qc[b], Q0=np.tile(qc[b], (T, 1)), max_iter=400, tol=1e-8)
        assert abs(s["f"] - r.f[b]) <= 1e-9 * max(1e-3, abs(s["f"])), (b, s["f"], r.f[b])
        assert np.abs(s["Q"] - r.x[b, : 7 * T].reshape(T, 7)).max() <= 1e-6
    be.close()
