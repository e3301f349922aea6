"""Warm-started, device-resident receding horizon of the torque-MPC family (oh_tq_rollout; SURVEY 8(f) rank 2 beyond the point mass; round-4 verdict,
Missing 2).  The reference's pattern is example/point_mass_mpc.py:156-175: the seed of a tick is the previous solution.  Checked here:
  * the device loop equals the numpy port's loop (oracle/torque_ipm.py:rollout_torque_ipm) on a few plants -- states, applied torques, objectives;
  * it equals the same loop driven from the host through oh_solve (seed shifted in numpy), whose per-tick x and multipliers are then graded on the
    literal NLP by oracle/solvers.py:kkt_reference_form;
  * every warm tick ends at the optimum the COLD solve from the same plant state finds (objective 1e-6 relative), in a fraction of its steps."""
import os
import sys

import numpy as np
import pytest

from conftest import MED7_KIN, SEED
from optas_amd import _lib
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
from oracle.problems import TorqueMPCNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.torque import TorqueProblem
from oracle.torque_ipm import rollout_torque_ipm

pytestmark = pytest.mark.gpu
LINK = "lbr_link_ee"
W = dict(w_path=1000.0, w_vel=0.1, w_tau=1e-4)
T, DT, LIM = 30, 0.1, 58.0
QN = np.deg2rad([0, 30, 0, -90, 0, -30, 0])


def _goal_tables(robot, qc, rows):
    """Figure of eight in the end-effector frame at each plant's initial configuration (figure_eight_plan.py:90-96 pattern), `rows` knots of it."""
    pose, _ = robot._kin(LINK).fk_jac(qc, want_jac=False)
    x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
    Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                   np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                   np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
    ts = np.arange(rows) * DT
    loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(rows)])
    return pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)


def _backend(robot, **kw):
    return TorqueBackend(robot.kinematic_chain(LINK), robot.dynamics_tables(), T=T, dt=DT, tau_lo=-LIM, tau_up=LIM, max_iter=600, **W, **kw)


def test_device_loop_equals_the_numpy_port_loop(hip_lib):
    robot = RobotModel.builtin("med7")
    med7 = OracleRobot(MED7_KIN)
    prob = TorqueProblem(med7, LINK, T=T, dt=DT, tau_lim=LIM, **W)
    rng = np.random.default_rng(SEED + 61)
    B, n_ticks = 3, 6
    qc = QN + rng.uniform(-0.1, 0.1, (B, 7))
    goals = _goal_tables(robot, qc, n_ticks + T)
    be = _backend(robot)
    states, tau0, f, iters, status = be.rollout(np.concatenate([qc, np.zeros((B, 7))], 1), goals, n_ticks)
    be.close()
    assert _lib.status_ok(status).all() and states.shape == (n_ticks + 1, B, 14) and np.array_equal(states[0, :, :7], qc)
    for b in range(B):
        ref = rollout_torque_ipm(prob, qc[b], np.zeros(7), goals[b], n_ticks, max_iter=600)
        assert (np.isin(ref["status"], (0, 4))).all()
        assert np.abs(ref["f"] - f[:, b]).max() <= 1e-7 * ref["f"].max(), (b, ref["f"], f[:, b])
        assert np.abs(ref["states"] - states[:, b]).max() <= 1e-5 and np.abs(ref["tau0"] - tau0[:, b]).max() <= 1e-3
        assert np.abs(ref["iters"][1:] - iters[1:, b]).max() <= 3, (ref["iters"], iters[:, b])
        # the plan's torque of knot 0 is the inverse dynamics at the plant's state with the acceleration that carries it to the next state
        ddq0 = (states[1:, b, 7:] - states[:-1, b, 7:]) / DT
        from oracle.torque import rnea_batch

        assert np.abs(rnea_batch(prob.tb, states[:-1, b, :7], states[:-1, b, 7:], ddq0) - tau0[:, b]).max() <= 1e-9


def test_batch_of_plants_warm_ticks_reach_the_cold_optimum_in_a_fraction_of_the_steps(hip_lib):
    robot = RobotModel.builtin("med7")
    med7 = OracleRobot(MED7_KIN)
    prob = TorqueProblem(med7, LINK, T=T, dt=DT, tau_lim=LIM, **W)
    nlp = TorqueMPCNLP(prob)
    rng = np.random.default_rng(SEED + 62)
    B, n_ticks, mu_warm = 2048, 20, 1e-6
    qc = QN + rng.uniform(-0.1, 0.1, (B, 7))
    goals = _goal_tables(robot, qc, n_ticks + T)
    state0 = np.concatenate([qc, np.zeros((B, 7))], 1)
    be = _backend(robot)
    states, tau0, f, iters, status = be.rollout(state0, goals, n_ticks, mu_warm=mu_warm)
    tm = be.timing()
    assert _lib.status_ok(status).all(), np.bincount(status.reshape(-1))
    cold_p50, warm_p50, warm_max = np.median(iters[0]), np.median(iters[1:]), iters[1:].max()
    print("torque MPC closed loop: %d plants x %d ticks in %.1f ms device = %.0f ticks/s; steps per tick: cold p50 %d, warm p50 %d p90 %d max %d"
          % (B, n_ticks, tm["solve_ms"], B * n_ticks / tm["solve_ms"] * 1e3, cold_p50, warm_p50, np.percentile(iters[1:], 90), warm_max))
    assert warm_p50 <= 0.5 * cold_p50 and warm_p50 <= 12
    # (a) the same loop from the host for 64 of the plants: seed shifted in numpy, warm ticks on a handle whose initial barrier parameter is mu_warm
    idx = np.sort(rng.choice(B, 64, replace=False))
    warm = _backend(robot, mu_barrier0=mu_warm)
    warm.set_option("tq_mu_dec", 0.1)  # what oh_tq_rollout gives its warm ticks (tq_mu_dec_warm): next to the optimum the damping comes down faster
    st = state0[idx].copy()
    x_prev = None
    worst = np.zeros(3)
    for k in range(n_ticks):
        p = np.concatenate([st, goals[idx, k : k + T].reshape(len(idx), -1)], 1)
        x0 = np.zeros((len(idx), 4 * 7 * T))
        if x_prev is not None:
            U = x_prev[:, 2 * 7 * T : 3 * 7 * T].reshape(len(idx), T, 7)
            x0[:, 2 * 7 * T : 3 * 7 * T] = np.concatenate([U[:, 1:], U[:, -1:]], 1).reshape(len(idx), -1)
        h = be if k == 0 else warm
        r = h.solve(x0, p)
        assert np.array_equal(r.f, f[k, idx]) and np.array_equal(r.iters, iters[k, idx])  # the device loop IS this loop
        if k in (1, 7, n_ticks - 1):  # literal KKT of sampled warm ticks with the multipliers the library returns
            lam = h.multipliers(len(idx))
            for i in range(0, len(idx), 8):
                x = r.x[i]
                assert np.abs(nlp.a(x, p[i])).max() <= 1e-12 and np.abs(nlp.h(x, p[i])).max() <= 1e-10 and nlp.k(x, p[i]).min() > 0.0
                lk = np.concatenate([lam[i][:, :7].reshape(-1), lam[i][:, 7:].reshape(-1)])
                kk = kkt_reference_form(nlp, x, p[i], lam_kg=lk)
                worst = np.maximum(worst, [kk["stationarity"], kk["feasibility"], kk["complementarity"]])
        x_prev = r.x
        st = np.concatenate([r.x[:, 7:14], r.x[:, 7 * T + 7 : 7 * T + 14]], 1)
        assert np.array_equal(st, states[k + 1, idx])
    assert worst[0] <= 1e-5 and worst[1] <= 1e-10 and worst[2] <= 1e-7, worst
    # (b) every sampled warm tick against a COLD solve from the same plant state (barrier parameter 0.1, no plan to start from: the seed brakes to
    # rest in the first knot and holds still -- zero accelerations would let a moving arm drift for 3 s): the same optimum, in a fraction of the steps
    for k in (1, 5, 12, n_ticks - 1):
        p = np.concatenate([states[k, idx], goals[idx, k : k + T].reshape(len(idx), -1)], 1)
        x0 = np.zeros((len(idx), 4 * 7 * T))
        x0[:, 2 * 7 * T : 2 * 7 * T + 7] = -states[k, idx, 7:] / DT
        c = be.solve(x0, p)
        ok = _lib.status_ok(c.status)
        assert ok.mean() >= 0.9, (k, np.bincount(c.status))
        rel = np.abs(c.f - f[k, idx])[ok] / np.abs(c.f[ok])
        assert rel.max() <= 1e-6, (k, rel.max())
        # (0.6 until the cold solve itself got faster in round 5 -- barrier factor 0.4, Nielsen's 1/3, quartered Newton steps: 15-16 where it took 18+; the warm ticks stay at 10)
        assert np.median(iters[k, idx]) <= 0.8 * np.median(c.iters[ok]), (np.median(iters[k, idx]), np.median(c.iters[ok]))
    be.close()
    warm.close()
