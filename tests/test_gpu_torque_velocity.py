"""Joint-velocity limits on the torque-MPC family (round-2 verdict, Missing 3): ``builder.enforce_model_limits(name, time_deriv=1)``
(builder.py:471-509) on the problem of examples/torque_mpc.py.  The velocities are states of this family, so the rows dq_t - vlo >= 0,
vup - dq_t >= 0 are stage-local: they join the effort rows under the barrier of k_tq_eval3 (oh_torque_desc.vel_limits / dq_lo / dq_up).
Checked against the numpy port (oracle/torque_ipm.py:solve_torque_ipm(vlimits=...)), the augmented-Lagrangian machine of rounds 1-3
(oracle/torque.py:solve_torque_lm(vlimits=...), an independent second algorithm), the reference-form KKT conditions on the literal layout
(oracle/problems.py:TorqueMPCNLP(vlimits=...): 840 variables, 1680 + 420 rows at T = 30), through HIPSolver and on a batch."""
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
from oracle.torque import TorqueProblem, solve_torque_lm
from oracle.torque_ipm import solve_torque_ipm

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.torque_mpc import build_problem, figure_eight_goal  # noqa: E402

pytestmark = pytest.mark.gpu
LINK = "lbr_link_ee"
W = dict(w_path=1000.0, w_vel=0.1, w_tau=1e-4)
QC = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
VMAX = 0.3  # rad/s: the unconstrained optimum runs joints 0, 1 and 5 at 0.6 - 0.7


def test_velocity_limited_torque_mpc_through_hipsolver_port_and_literal_kkt(hip_lib):
    import optas_amd as optas

    T, dt, eff = 12, 0.1, 60.0
    vl = np.full(7, VMAX)
    robot, link, opt = build_problem(T, dt, effort=eff, velocity_limits=(-vl, vl))
    assert opt.nk == 2 * 7 * T + 2 * 7 * T
    solver = optas.HIPSolver(opt).setup("hip_sqp", {"max_iter": 600})
    goal = figure_eight_goal(robot, link, QC, T, dt)
    pd = {"qc": QC, "dqc": np.zeros(7), "goal": goal}
    solver.reset_parameters(pd)
    solver.reset_initial_seed({"med7/q/x": np.tile(QC[:, None], (1, T))})
    sol = solver.solve()
    st = solver.stats()
    assert solver.did_solve(), st
    dQ = np.asarray(sol["med7/dq"])
    assert np.abs(dQ).max() < VMAX and np.abs(dQ).max() >= VMAX - 1e-5 and np.abs(sol["tau/y"]).max() < eff  # interior
    med7 = OracleRobot(MED7_KIN)
    prob = TorqueProblem(med7, LINK, T=T, dt=dt, tau_lim=eff, **W)
    nlp = TorqueMPCNLP(prob, vlimits=(-vl, vl))
    x = np.asarray(opt.decision_variables.dict2vec({k: v for k, v in sol.items() if k.endswith("/x")})).reshape(-1)
    p = np.asarray(opt.parameters.dict2vec(pd)).reshape(-1)
    po = nlp.pack_p(QC, np.zeros(7), goal.T)
    assert (nlp.nx, nlp.nk, nlp.na, nlp.nh) == (opt.nx, opt.nk, opt.na, opt.nh) and np.abs(p - po).max() == 0.0
    rng = np.random.default_rng(SEED)
    xr = x + rng.normal(0, 0.05, x.shape)
    assert np.abs(opt.k(xr, p) - nlp.k(xr, po)).max() <= 1e-12 and np.array_equal(opt.dk(xr, p), nlp.dk(xr, po))  # same rows, same order
    assert abs(nlp.f(x, po) - st["f"][0]) <= 1e-9 * st["f"][0] and np.abs(nlp.a(x, po)).max() <= 1e-12 and np.abs(nlp.h(x, po)).max() <= 1e-10
    assert nlp.k(x, po).min() > 0.0
    lam = solver.backend.multipliers(1)[0]
    assert lam.shape == (T, 28) and lam.min() > 0.0
    k = kkt_reference_form(nlp, x, po, lam_kg=np.concatenate([lam[:, 7 * i:7 * i + 7].reshape(-1) for i in range(4)]))  # the multipliers that came with x
    assert k["stationarity"] <= 1e-6 and k["feasibility"] <= 1e-10 and k["complementarity"] <= 1e-8, k
    s = solve_torque_ipm(prob, QC, np.zeros(7), goal.T, vlimits=(-vl, vl), max_iter=600)  # the port: same state machine
    assert s["status"] == 0 and abs(s["f"] - st["f"][0]) <= 1e-9 * s["f"] and abs(int(st["iterations"][0]) - s["iters"]) <= 2, (st["iterations"][0], s["iters"])
    assert np.abs(lam - s["lam"]).max() <= 1e-5 * max(1.0, s["lam"].max())
    al = solve_torque_lm(prob, QC, np.zeros(7), goal.T, vlimits=(-vl, vl), max_iter=600)  # the independent second machine
    free = solve_torque_lm(prob, QC, np.zeros(7), goal.T)
    assert al["status"] == 0 and abs(al["f"] - st["f"][0]) <= 1e-7 * al["f"] and al["f"] > 1.5 * free["f"] and np.abs(free["dQ"]).max() > 2 * VMAX
    act = al["lam_v"] > 1e-6  # rows the augmented Lagrangian holds active carry the same multipliers
    assert act.sum() >= 5 and np.abs(lam[:, 14:][act] - al["lam_v"][act]).max() <= 1e-3 * max(1.0, al["lam_v"].max())


def test_batch_with_velocity_limits_properties_and_scalar_equivalence(hip_lib):
    T, B = 30, 512
    robot = RobotModel.builtin("med7")
    be = TorqueBackend(robot.kinematic_chain(LINK), robot.dynamics_tables(), T=T, dt=0.1, tau_lo=-58.0, tau_up=58.0, dq_lo=-0.5, dq_up=0.5, max_iter=1000, **W)
    rng = np.random.default_rng(SEED + 3)
    qc = QC + rng.uniform(-0.1, 0.1, (B, 7))
    goal = np.stack([figure_eight_goal(robot, LINK, q, T, 0.1).T for q in qc])
    p = np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1)
    x0 = np.zeros((B, 4 * 7 * T))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    r = be.solve(x0, p)
    ok = _lib.status_ok(r.status)
    # (round 3, augmented Lagrangian: p50 134 steps and an instance in 500 not through after 1000; the interior point brings every one home)
    assert ok.mean() >= 0.99, ok.mean()
    dQ = r.x[:, 7 * T : 14 * T]
    tau = r.x[:, 21 * T :]
    assert np.abs(dQ[ok]).max() < 0.5 and np.abs(tau[ok]).max() < 58.0 and (np.abs(dQ[ok]).max(1) >= 0.5 - 1e-5).mean() > 0.8
    assert (r.kkt[ok, 0] <= 1e-6).all() and (r.kkt[ok, 1] == 0.0).all() and (r.kkt[ok, 2] <= 1e-8).all()
    lam = be.multipliers(B)
    assert lam.shape == (B, T, 28) and lam.min() > 0.0 and (lam[ok][:, :, 14:].max((1, 2)) > 1e-3).mean() > 0.8
    for b in (0, 17, 300):  # an instance alone = the same instance in the batch, bit for bit
        a = be.solve(x0[b : b + 1], p[b : b + 1])
        assert np.array_equal(a.x[0], r.x[b]) and a.iters[0] == r.iters[b] and a.f[0] == r.f[b]
    # the literal rows on three instances
    med7 = OracleRobot(MED7_KIN)
    nlp = TorqueMPCNLP(TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W), vlimits=(-0.5, 0.5))
    for b in np.flatnonzero(ok)[:3]:
        assert np.abs(nlp.a(r.x[b], p[b])).max() <= 1e-12 and np.abs(nlp.h(r.x[b], p[b])).max() <= 1e-10 and nlp.k(r.x[b], p[b]).min() > 0.0
        assert abs(nlp.f(r.x[b], p[b]) - r.f[b]) <= 1e-9 * r.f[b]
    be.close()


def test_initial_velocity_outside_its_limits_is_reported_infeasible(hip_lib):
    """dq_0 = dqc is pinned by fix_configuration on the velocity state: its velocity-limit rows are constants of the instance.  Outside them the NLP has no
    feasible point (the literal rows say so) -- IPOPT would report an infeasible problem (solver.py:407-412); here OH_STATUS_INFEASIBLE, the rest of the batch
    untouched."""
    T, B = 12, 6
    robot = RobotModel.builtin("med7")
    be = TorqueBackend(robot.kinematic_chain(LINK), robot.dynamics_tables(), T=T, dt=0.1, tau_lo=-58.0, tau_up=58.0, dq_lo=-0.5, dq_up=0.5, max_iter=300, **W)
    qc = np.tile(QC, (B, 1))
    dqc = np.zeros((B, 7))
    dqc[2, 3], dqc[4, 0] = 0.8, -0.51
    goal = np.stack([figure_eight_goal(robot, LINK, q, T, 0.1).T for q in qc])
    p = np.concatenate([qc, dqc, goal.reshape(B, -1)], 1)
    r = be.solve(np.zeros((B, 4 * 7 * T)), p)
    nlp = TorqueMPCNLP(TorqueProblem(OracleRobot(MED7_KIN), LINK, T=T, dt=0.1, tau_lim=58.0, **W), vlimits=(-0.5, 0.5))
    for b in range(B):
        bad = np.abs(dqc[b]).max() > 0.5
        assert (r.status[b] == _lib.OH_STATUS_INFEASIBLE) == bad, (b, r.status[b])
        if bad:
            assert r.kkt[b, 1] >= np.abs(dqc[b]).max() - 0.5 - 1e-12 and not _lib.status_ok(r.status[b])
        else:
            assert _lib.status_ok(r.status[b]) and nlp.k(r.x[b], p[b]).min() > 0.0
    be.close()


def test_hard_pressed_velocity_rows_leave_the_relaxed_zone(hip_lib):
    """Round 6: a velocity limit the tracking cost pushes hard against (|dq| <= 0.2 where the free plan runs at 0.95) carries multipliers beyond 1 / theta, i.e. its
    slack at the stationary point of a barrier parameter lies inside the relaxed zone (below theta mu_b).  The barrier update used to wait for every row to be in
    the logarithmic regime -- and the instance sat at that point until the iteration cap (null steps, damping doubling).  Now the barrier parameter is lowered and
    the point evaluated again under it (k_tq_step: D.first; oracle/torque_ipm.py alike): the instances converge to the port's optima, inside every row."""
    T, vl = 30, 0.2
    robot = RobotModel.builtin("med7")
    med7 = OracleRobot(MED7_KIN)
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    rng = np.random.default_rng(SEED + 9)
    qc = QC[None] + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.05, 0.05, (5, 7))])
    goal = np.stack([prob.goal_figure_eight(q) for q in qc])
    be = TorqueBackend(robot.kinematic_chain(LINK), robot.dynamics_tables(), T=T, dt=0.1, tau_lo=-58.0, tau_up=58.0, dq_lo=-vl, dq_up=vl, max_iter=600, **W)
    p = np.ascontiguousarray(np.concatenate([qc, np.zeros((len(qc), 7)), goal.reshape(len(qc), -1)], 1))
    x0 = np.zeros((len(qc), 4 * 7 * T))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    r = be.solve(x0, p)
    be.close()
    ok = _lib.status_ok(r.status)
    # (rows with slacks of 1e-10 and multipliers of 1e2: the end game sits at the arithmetic floor of the reduced gradient, where the paths of device and port part
    #  -- step counts 29-47 on either side, not instance by instance -- and one instance in six may end NUMERICAL there; before round 6 all six ended at the cap)
    assert ok.sum() >= len(qc) - 1, (r.status, r.iters)
    dQ = r.x[:, 7 * T : 2 * 7 * T].reshape(len(qc), T, 7)
    assert np.abs(dQ[ok]).max() < vl and np.abs(dQ[ok]).max() > vl - 1e-6 and (r.iters[ok] < 80).all()
    for b in np.flatnonzero(ok)[:3]:
        s = solve_torque_ipm(prob, qc[b], np.zeros(7), goal[b], vlimits=(-vl, vl), max_iter=600)
        assert s["status"] in (0, 4) and abs(s["f"] - r.f[b]) <= 1e-6 * s["f"], (b, s["status"], s["f"], r.f[b], r.iters[b], s["iters"])
        assert s["f"] > 2.0 * 9.7  # (the free plan's objective is 9.73: the rows cost more than the plan)
