"""Torque-MPC family (BASELINE configs[4]) on chains other than seven joints (round-4 verdict, Missing 4 / Next 7: RobotModel takes any URDF, models.py:233-321;
RobotModel.rnea any chain of revolute joints behind a fixed one, models.py:1731-1884).  Until the end of round 5 `oh_create_torque` refused ndof != 7; the kernels
(k_tq_setup / k_tq_eval3 / k_tq_curv / k_tq_step / k_tq_finalize) are templates in the chain length and are now instantiated for 2 ... 7 joints (k_tq_step's lane
layout is 8 x 8: up to seven joints and the vector column).  Robots: tests/golden/tester_robot_revolute.kin.json (2 revolute joints, the reference's own test robot)
and the KUKA med7 cut after its 3rd ... 6th joint with a 0.4 kg tool.  Every answer is compared with the numpy port of the same state machine
(oracle/torque_ipm.py: objective 1e-9 relative, step counts +-2) and graded on the literal NLP (oracle/problems.py:TorqueMPCNLP) with the returned multipliers:
stationarity <= 1e-6, linear rows <= 1e-12, dynamics rows <= 1e-10, inequality rows strictly inside, complementarity <= 1e-8."""
import copy
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, MED7_KIN, SEED
from optas_amd import _lib
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
from oracle.problems import TorqueMPCNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.torque import TorqueProblem, rnea_batch
from oracle.torque_ipm import rollout_torque_ipm, solve_torque_ipm

pytestmark = pytest.mark.gpu
TESTER_REV_KIN = os.path.join(GOLDEN, "tester_robot_revolute.kin.json")
W = dict(w_path=1000.0, w_vel=0.1, w_tau=1e-4)
T = 12


def _med7_cut(tmp_path, n):
    """med7.kin.json with n actuated joints: cut after joint n, a tool (0.4 kg) on a fixed joint behind it -- every body RobotModel.rnea counts carries <inertial>."""
    d = json.load(open(MED7_KIN))
    joints = {j["name"]: j for j in d["joints"]}
    links = {l["name"]: l for l in d["links"]}
    out = copy.deepcopy(d)
    out["name"] = f"med{n}"
    keep = ["world_lbr_joint"] + [f"lbr_joint_{i}" for i in range(n)]
    out["joints"] = [joints[k] for k in keep] + [{"name": "tool_joint", "type": "fixed", "parent": joints[keep[-1]]["child"], "child": "tool", "xyz": [0.0, 0.0, 0.12],
                                                 "rpy": [0.0, 0.0, 0.0]}]
    out["links"] = [links[k] for k in ["world"] + [joints[k]["child"] for k in keep]] + [
        {"name": "tool", "inertial": {"mass": 0.4, "xyz": [0.0, 0.0, 0.03], "rpy": [0.0, 0.0, 0.0], "inertia": [0.001, 0.0, 0.0, 0.001, 0.0, 0.0008]}}]
    path = os.path.join(str(tmp_path), f"med{n}.kin.json")
    json.dump(out, open(path, "w"))
    return path


def _robots(tmp_path):
    """(tag, kin file, link, nominal configuration)"""
    out = [("tester2", TESTER_REV_KIN, "eff", np.array([0.4, -0.3]))]
    for n in (3, 4, 5, 6):
        out.append((f"med{n}", _med7_cut(tmp_path, n), "tool", np.deg2rad([0, 45, 0, -90, 0, -45, 0])[:n]))
    return out


def _binding_limits(orc, link, qn):
    """Effort limits that bind without making the arm fall: three quarters of the peak torque of the unlimited plan on the joints that carry little gravity load,
    one and a half times the peak on the others (a limit below the holding torque of the nominal configuration leaves the first knots no feasible torque)."""
    n = orc.ndof
    prob = TorqueProblem(orc, link, T=T, dt=0.1, tau_lim=1e3, **W)
    hold = np.abs(rnea_batch(prob.tb, qn, np.zeros(n), np.zeros(n)))
    peak = np.abs(solve_torque_ipm(prob, qn, np.zeros(n), prob.goal_figure_eight(qn, scale=0.5))["tau"]).max(0)
    return np.where(hold < 0.3 * peak, 0.75 * peak, 1.5 * peak)


def _backend(kin, link, lim, **kw):
    robot = RobotModel(urdf_filename=kin, time_derivs=[0, 1, 2])
    return TorqueBackend(robot.kinematic_chain(link), robot.dynamics_tables(), T=T, dt=0.1, tau_lo=-lim, tau_up=lim, **W, **kw)


def test_every_chain_length_equals_the_port_and_is_a_kkt_point_of_the_literal_problem(hip_lib, tmp_path):
    rng = np.random.default_rng(SEED + 40)
    for tag, kin, link, qn in _robots(tmp_path):
        orc = OracleRobot(kin)
        n = orc.ndof
        for lim in (np.full(n, 1e3), _binding_limits(orc, link, qn)):
            prob = TorqueProblem(orc, link, T=T, dt=0.1, tau_lim=lim, **W)
            nlp = TorqueMPCNLP(prob)
            be = _backend(kin, link, lim)
            B = 6
            qc = qn + np.concatenate([np.zeros((1, n)), rng.uniform(-0.1, 0.1, (B - 1, n))])
            goal = np.stack([prob.goal_figure_eight(q, scale=0.5) for q in qc])
            p = np.stack([nlp.pack_p(qc[b], np.zeros(n), goal[b]) for b in range(B)])
            res = be.solve(np.stack([nlp.seed(q) for q in qc]), p)
            assert _lib.status_ok(res.status).all(), (tag, lim, res.status)
            lam = be.multipliers(B)
            assert lam.shape == (B, T, 2 * n) and lam.min() > 0.0
            slack = min(nlp.k(res.x[b], p[b]).min() for b in range(B))
            if lim.max() < 1e3:  # an effort row sits at its bound (slack = mu_b / lam with mu_b <= 1e-8) and carries a multiplier far above the inactive rows' mu_b / s
                assert slack < 1e-4 and lam.max() > 1e3 * np.median(lam), (tag, slack, lam.max(), "the effort rows were meant to bind")
            for b in range(B):
                r = solve_torque_ipm(prob, qc[b], np.zeros(n), goal[b])
                assert r["status"] == 0 and abs(r["f"] - res.f[b]) <= 1e-9 * r["f"] and abs(r["iters"] - res.iters[b]) <= 2, (tag, lim, b, r["f"], res.f[b], r["iters"], res.iters[b])
                x = res.x[b]
                X = x.reshape(4, T, n)
                assert np.abs(X[0] - r["Q"]).max() <= 1e-6 and np.abs(X[3] - r["tau"]).max() <= 1e-4 * max(1.0, np.abs(r["tau"]).max())
                assert np.abs(X[3] - rnea_batch(prob.tb, X[0], X[1], X[2])).max() <= 1e-10  # the torques in x ARE the inverse dynamics of its states
                assert abs(nlp.f(x, p[b]) - res.f[b]) <= 1e-12 * res.f[b]
                assert np.abs(nlp.a(x, p[b])).max() <= 1e-12 and np.abs(nlp.h(x, p[b])).max() <= 1e-10 and nlp.k(x, p[b]).min() > 0.0
                lk = np.concatenate([lam[b][:, :n].reshape(-1), lam[b][:, n:].reshape(-1)])  # k = [vec(TAU) - lo; up - vec(TAU)]
                k = kkt_reference_form(nlp, x, p[b], lam_kg=lk)
                assert k["stationarity"] <= 1e-6 and k["feasibility"] <= 1e-10 and k["complementarity"] <= 1e-8, (tag, lim, b, k)
            # one instance alone = the same instance inside the batch, bit for bit (the family's answers do not depend on the batch)
            alone = be.solve(nlp.seed(qc[2])[None], p[2:3])
            assert np.array_equal(alone.x[0], res.x[2]) and alone.iters[0] == res.iters[2]
            be.close()


def test_dual_number_path_and_velocity_rows_on_a_short_chain(hip_lib, tmp_path):
    """The other instantiations of the evaluation kernel (d tau / dz by dual numbers instead of the closed form: option tq_jac_dual; joint-velocity rows on the
    velocity states) on the 4-joint arm: the dual-number path reaches the closed form's optimum, and velocity limits that bind are respected and carry multipliers."""
    kin = _med7_cut(tmp_path, 4)
    orc = OracleRobot(kin)
    n, link = 4, "tool"
    lim = _binding_limits(orc, link, np.deg2rad([0, 45, 0, -90]))
    prob = TorqueProblem(orc, link, T=T, dt=0.1, tau_lim=lim, **W)
    nlp = TorqueMPCNLP(prob)
    qc = np.deg2rad([0, 45, 0, -90]) + np.array([[0.0] * 4, [0.05, -0.04, 0.03, 0.02]])
    goal = np.stack([prob.goal_figure_eight(q, scale=0.5) for q in qc])
    p = np.stack([nlp.pack_p(qc[b], np.zeros(n), goal[b]) for b in range(2)])
    x0 = np.stack([nlp.seed(q) for q in qc])
    be = _backend(kin, link, lim)
    ref = be.solve(x0, p)
    be.set_option("tq_jac_dual", 1)
    dual = be.solve(x0, p)
    assert _lib.status_ok(ref.status).all() and _lib.status_ok(dual.status).all()
    assert np.all(np.abs(dual.f - ref.f) <= 1e-9 * ref.f) and np.abs(dual.x - ref.x).reshape(2, 4, T, n)[:, 0].max() <= 1e-6
    be.close()
    vmax = 0.8 * np.abs(ref.x.reshape(2, 4, T, n)[:, 1]).max()
    bv = _backend(kin, link, lim, dq_lo=-vmax, dq_up=vmax)
    rv = bv.solve(x0, p)
    assert _lib.status_ok(rv.status).all()
    dq = rv.x.reshape(2, 4, T, n)[:, 1]
    lam = bv.multipliers(2)
    assert np.abs(dq).max() < vmax and np.abs(dq).max() > 0.99 * vmax and lam.shape == (2, T, 4 * n) and lam[:, :, 2 * n :].max() > 1e-3
    assert np.all(rv.f >= ref.f - 1e-12)  # a restriction of the same problem
    bv.close()


def test_closed_loop_on_the_device_equals_the_port_on_a_five_joint_arm(hip_lib, tmp_path):
    """oh_tq_rollout (warm-started receding horizon, point_mass_mpc.py:156-175) on five joints: states and step counts of the port's loop."""
    kin = _med7_cut(tmp_path, 5)
    orc = OracleRobot(kin)
    n, link, ticks = 5, "tool", 4
    lim = _binding_limits(orc, link, np.deg2rad([0, 45, 0, -90, 0]))
    prob = TorqueProblem(orc, link, T=T, dt=0.1, tau_lim=lim, **W)
    be = _backend(kin, link, lim)
    q0 = np.deg2rad([0, 45, 0, -90, 0]) + np.array([[0.0] * 5, [0.05, -0.04, 0.03, 0.02, -0.05]])
    e, Re, _, _ = prob.chain.fk(q0)
    ts = np.arange(ticks + T) * 0.1
    loc = 0.5 * np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros_like(ts)], 1)
    table = np.stack([e[b][None] + loc @ Re[b].T for b in range(2)])
    states, tau0, f, iters, status = be.rollout(np.concatenate([q0, np.zeros((2, n))], 1), table, ticks)
    assert _lib.status_ok(status).all()
    for b in range(2):
        r = rollout_torque_ipm(prob, q0[b], np.zeros(n), table[b], ticks)
        assert np.abs(np.asarray(r["states"]) - states[:, b]).max() <= 1e-7 and np.abs(np.asarray(r["f"]) - f[:, b]).max() <= 1e-8 * np.abs(f[:, b]).max()
        assert np.abs(np.asarray(r["iters"]) - iters[:, b]).max() <= 2
    be.close()


def test_reference_script_flow_on_a_four_joint_arm_through_hipsolver(hip_lib, tmp_path):
    """The builder calls of the reference's torque-MPC script (examples/torque_mpc.py:build_problem) on the 4-joint arm: the lowering recognises the family, the
    answer is the port's, the dictionary has the reference's keys and shapes (solver.py:137-155), and the problem's own functions agree with the literal NLP."""
    import optas_amd as optas
    from examples.torque_mpc import build_problem, figure_eight_goal

    kin = _med7_cut(tmp_path, 4)
    orc = OracleRobot(kin)
    qc = np.deg2rad([0, 45, 0, -90]) + np.array([0.02, -0.03, 0.04, 0.01])
    lim = _binding_limits(orc, "tool", np.deg2rad([0, 45, 0, -90]))
    robot, link, opt = build_problem(T, 0.1, effort=lim, robot=RobotModel(urdf_filename=kin, time_derivs=[0, 1, 2]), link="tool")
    solver = optas.HIPSolver(opt).setup("hip_sqp")
    assert type(solver.backend).__name__ == "TorqueBackend" or "Torque" in type(solver.backend).__name__
    prob = TorqueProblem(orc, "tool", T=T, dt=0.1, tau_lim=lim, **W)
    nlp = TorqueMPCNLP(prob)
    goal = prob.goal_figure_eight(qc, scale=0.5)
    name = robot.get_name()
    pd = {"qc": qc, "dqc": np.zeros(4), "goal": goal.T}
    solver.reset_parameters(pd)
    solver.reset_initial_seed({f"{name}/q/x": np.tile(qc[:, None], (1, T))})
    sol = solver.solve()
    r = solve_torque_ipm(prob, qc, np.zeros(4), goal)
    assert solver.did_solve() and r["status"] == 0 and abs(solver.stats()["f"][0] - r["f"]) <= 1e-9 * r["f"]
    for key in (f"{name}/q", f"{name}/dq", f"{name}/ddq", "tau/y", "tau/y/x"):
        assert sol[key].shape == (4, T)
    assert np.all(np.abs(sol["tau/y"]) <= lim[:, None] + 1e-8)
    x = np.asarray(solver.opt.decision_variables.dict2vec({k: v for k, v in sol.items() if k.endswith("/x")})).reshape(-1)
    p = np.asarray(solver.opt.parameters.dict2vec(pd)).reshape(-1)
    po = nlp.pack_p(qc, np.zeros(4), goal)
    assert np.abs(p - po).max() == 0.0 and abs(opt.f(x, p) - nlp.f(x, po)) <= 1e-10 * nlp.f(x, po)
    for fn in ("k", "a", "h", "v"):
        assert np.abs(getattr(opt, fn)(x, p) - getattr(nlp, fn)(x, po)).max() <= 1e-9, fn


def test_abi_limits(hip_lib):
    lib = _lib.load()
    import ctypes as C

    for ndof, ok in ((1, False), (2, True), (7, True), (8, False)):
        desc = _lib.oh_torque_desc(T=8, ndof=ndof, dt=0.1, w_path=1.0, w_vel=0.0, w_tau=1e-3)
        for i in range(min(ndof, 8)):
            desc.tau_lo[i], desc.tau_up[i] = -1.0, 1.0
        h = C.c_void_p()
        rc = lib.oh_create_torque(C.byref(desc), C.byref(h))
        assert (rc == 0) == ok, (ndof, rc)
        if rc == 0:
            lib.oh_destroy(h)
