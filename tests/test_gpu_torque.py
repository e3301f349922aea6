"""BASELINE configs[4] on the GPU: OH_PROBLEM_TORQUE_MPC (torque MPC, RobotModel.rnea as equality rows, T = 30, batches up to 8192)
through the C ABI / HIPSolver against the oracle.

Round 4: the state machine is a primal-dual interior point (oracle/torque_ipm.py is its numpy port; the augmented-Lagrangian machine of rounds
1-3, oracle/torque.py:solve_torque_lm, stays as an independent second solver of the same problem).

Tolerances: objective 1e-9 relative to the numpy port's optimum, 1e-8 to the augmented-Lagrangian machine's and to scipy L-BFGS-B's (reduced
problem), 1e-7 to scipy trust-constr's (reference wiring, literal layout) and SLSQP's, all from tests/golden/torque_golden.npz; reference-form KKT
on the literal 1680-row v WITH THE RETURNED MULTIPLIERS (lam_i = mu_b / s_i): stationarity <= 1e-6, rows strictly feasible, complementarity
<= 1e-8; linear rows <= 1e-12, dynamics rows <= 1e-10; step counts equal the port's (same state machine, same arithmetic up to rounding)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, MED7_KIN, SEED, oh_debug
from optas_amd import _lib
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
from oracle.problems import TorqueMPCNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.torque import TorqueProblem, rnea_jacobian, solve_torque_lm
from oracle.torque_ipm import solve_torque_ipm

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.torque_mpc import build_problem, figure_eight_goal  # noqa: E402

pytestmark = pytest.mark.gpu
LINK = "lbr_link_ee"
W = dict(w_path=1000.0, w_vel=0.1, w_tau=1e-4)


@pytest.fixture(scope="module")
def ctx():
    med7 = OracleRobot(MED7_KIN)
    robot = RobotModel.builtin("med7")
    return med7, robot, np.load(os.path.join(GOLDEN, "torque_golden.npz"))


def backend(robot, T, lim, **kw):
    """lim > 1e8 marks the golden cases generated with the URDF's own effort limits (100 N m on every med7 joint): never active at the
    optimum, but they do reject early trial points, so both sides must carry them."""
    lim = 100.0 if lim is None or lim > 1e8 else float(lim)
    return TorqueBackend(robot.kinematic_chain(LINK), robot.dynamics_tables(), T=T, dt=0.1, tau_lo=-lim, tau_up=lim, **W, **kw)


def test_golden_instances_objective_solution_steps_and_literal_kkt(hip_lib, ctx):
    med7, robot, g = ctx
    for tag, T in (("t30", 30), ("t30lim", 30), ("t6", 6), ("t6lim", 6)):
        lim = float(g[tag + "_lim"])
        prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=None if lim > 1e8 else lim, **W)
        nlp = TorqueMPCNLP(prob)
        be = backend(robot, T, lim)
        qc, goal = g[tag + "_qc"], g[tag + "_goal"]
        B = len(qc)
        p = np.stack([nlp.pack_p(qc[b], np.zeros(7), goal[b]) for b in range(B)])
        res = be.solve(np.stack([nlp.seed(q) for q in qc]), p)
        assert _lib.status_ok(res.status).all(), (tag, res.status)
        assert np.all(np.abs(res.f - g[tag + "_f"]) <= 1e-9 * g[tag + "_f"]), (tag, res.f - g[tag + "_f"])
        # (the end game is Newton's: the step count does not hover around the tolerance as the Gauss-Newton tail of rounds 1-3 did)
        assert np.all(np.abs(res.iters - g[tag + "_iters"]) <= 2), (tag, res.iters, g[tag + "_iters"])
        assert np.all(np.abs(res.f - g[tag + "_f_al"]) <= 1e-8 * res.f), (tag, res.f - g[tag + "_f_al"])
        # the solution is pinned as tightly as the stopping rule pins it: |grad| <= 1e-6 leaves a component with curvature c free to 1e-6 / c,
        # and the curvature along the wrist accelerations is 2 w_tau M_77^2 ~ 1e-8 .. 1e-6 (joint-space inertia 0.007 .. 0.1 kg m^2)
        dX = np.abs(res.x - g[tag + "_x"]).reshape(B, 4, T, 7).max((0, 2, 3))
        assert dX[0] < 1e-4 and dX[1] < 1e-3 and dX[2] < 0.1 and dX[3] < 1e-2, (tag, dX)
        if lim > 1e8:
            assert np.all(np.abs(res.f - g[tag + "_f_lbfgs"]) <= 1e-8 * res.f)
        if T == 6:
            assert np.all(np.abs(res.f - g[tag + "_f_trust_constr"]) <= 1e-7 * res.f)
        if lim < 1e8:  # effort rows active: scipy SLSQP on the problem reduced to the controls, rows with their exact Jacobian (tools/make_golden.py)
            assert np.all(np.abs(res.f - g[tag + "_f_slsqp"]) <= 1e-7 * res.f)
        lam = be.multipliers(B)
        assert lam.shape == (B, T, 14) and lam.min() > 0.0 and np.abs(lam - g[tag + "_lam"]).max() <= 1e-5 * max(1.0, np.abs(g[tag + "_lam"]).max())
        for b in range(B):
            x = res.x[b]
            assert abs(nlp.f(x, p[b]) - res.f[b]) <= 1e-12 * res.f[b]
            assert np.abs(nlp.a(x, p[b])).max() <= 1e-12 and np.abs(nlp.h(x, p[b])).max() <= 1e-10
            assert nlp.k(x, p[b]).min() > 0.0  # interior
            lk = np.concatenate([lam[b][:, :7].reshape(-1), lam[b][:, 7:].reshape(-1)])  # k = [vec(TAU) - lo; up - vec(TAU)]
            k = kkt_reference_form(nlp, x, p[b], lam_kg=lk)
            assert k["stationarity"] <= 1e-6 and k["feasibility"] <= 1e-10 and k["complementarity"] <= 1e-8, (tag, b, k)
        if lim < 1e8:
            assert lam.max() > 1e-3  # the effort rows are active in these instances
        else:
            assert lam.max() <= 1e-9  # mu_b / s with every row tens of N m from its bound
        be.close()


def test_first_twelve_steps_equal_the_port(hip_lib, ctx):
    """Same state machine, same iterates: stopped after 12 evaluations the GPU and the numpy port hold the same point -- objective 1e-9
    relative, ddq 1e-6.  With and without active effort rows."""
    med7, robot, g = ctx
    for tag in ("t30", "t30lim"):
        lim = float(g[tag + "_lim"])
        prob = TorqueProblem(med7, LINK, T=30, dt=0.1, tau_lim=None if lim > 1e8 else lim, **W)
        nlp = TorqueMPCNLP(prob)
        be = backend(robot, 30, lim, max_iter=12)
        qc, goal = g[tag + "_qc"], g[tag + "_goal"]
        B = len(qc)
        res = be.solve(np.stack([nlp.seed(q) for q in qc]), np.stack([nlp.pack_p(qc[b], np.zeros(7), goal[b]) for b in range(B)]))
        assert (res.status == 1).all() and (res.iters == 12).all()
        for b in range(B):
            r = solve_torque_ipm(prob, qc[b], np.zeros(7), goal[b], max_iter=12)
            assert r["status"] == 1 and r["iters"] == 12
            assert abs(r["f"] - res.f[b]) <= 1e-9 * r["f"], (tag, b, r["f"], res.f[b])
            # (the wrist accelerations are the loosest directions of the problem -- curvature 2 w_tau M_77^2 ~ 1e-8 .. 1e-6 -- and mid-way through the
            #  iteration the last bits of the Riccati elimination show there: the kernel eliminates by Gauss-Jordan steps across the lanes, numpy by
            #  Cholesky; the states and torques the accelerations produce agree far tighter)
            X = res.x[b].reshape(4, 30, 7)
            assert np.abs(X[2] - r["U"]).max() <= 1e-3 * max(1.0, np.abs(r["U"]).max())
            assert np.abs(X[0] - r["Q"]).max() <= 1e-7 and np.abs(X[3] - r["tau"]).max() <= 1e-5
        be.close()


def test_first_evaluation_equals_the_oracle_functions(hip_lib, ctx):
    """max_iter = 0: the solve stops after evaluating the seed -- f and the torques in x are the oracle's f(x0) and rnea(x0)."""
    med7, robot, g = ctx
    T = 30
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, **W)
    nlp = TorqueMPCNLP(prob)
    rng = np.random.default_rng(SEED + 11)
    B = 8
    qc = g["t30_qc"][0][None] + rng.uniform(-0.3, 0.3, (B, 7))
    dqc = rng.uniform(-0.5, 0.5, (B, 7))
    U = rng.uniform(-2.0, 2.0, (B, T, 7))
    goal = np.stack([prob.goal_figure_eight(q) for q in qc])
    x0 = np.stack([nlp.join(*prob.rollout(qc[b], dqc[b], U[b]), U[b], np.zeros((T, 7))) for b in range(B)])
    p = np.stack([nlp.pack_p(qc[b], dqc[b], goal[b]) for b in range(B)])
    be = TorqueBackend(robot.kinematic_chain(LINK), robot.dynamics_tables(), T=T, dt=0.1, max_iter=1, tol=1e-30, **W)
    res = be.solve(x0, p)
    # after one step the returned point is the first trial or the seed; evaluate it with the literal functions
    for b in range(B):
        x = res.x[b]
        assert np.abs(nlp.h(x, p[b])).max() <= 1e-10 and np.abs(nlp.a(x, p[b])).max() <= 1e-12
        assert abs(nlp.f(x, p[b]) - res.f[b]) <= 1e-12 * max(1.0, res.f[b])
    be.close()


def test_closed_form_jacobian_path_equals_the_dual_number_path(hip_lib, ctx, monkeypatch):
    """k_tq_eval3 takes d tau / dz in closed form (rnea_idsva) when the dynamics tables describe a rigid-body chain, else from dual numbers through the
    recursion (OH_TQ_JAC=dual forces that path): same step counts, objectives to 1e-12 relative, solutions to 1e-9."""
    med7, robot, g = ctx
    T = 30
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    nlp = TorqueMPCNLP(prob)
    rng = np.random.default_rng(SEED + 31)
    B = 64
    qc = g["t30_qc"][0][None] + rng.uniform(-0.1, 0.1, (B, 7))
    goal = np.stack([prob.goal_figure_eight(q) for q in qc])
    p = np.stack([nlp.pack_p(qc[b], np.zeros(7), goal[b]) for b in range(B)])
    x0 = np.stack([nlp.seed(q) for q in qc])
    be = backend(robot, T, 58.0)
    ra = be.solve(x0, p)
    oh_debug(monkeypatch, tq_jac="dual")
    rb = be.solve(x0, p)
    oh_debug(monkeypatch, tq_jac=None)
    assert _lib.status_ok(ra.status).all() and _lib.status_ok(rb.status).all()
    assert np.mean(np.asarray(ra.iters) == np.asarray(rb.iters)) >= 0.9  # a ratio test decided by the last bits may differ on an instance or two
    same = np.asarray(ra.iters) == np.asarray(rb.iters)
    assert np.abs(ra.f - rb.f).max() <= 1e-9 * np.abs(ra.f).max()
    assert np.abs(ra.x[same] - rb.x[same]).max() <= 1e-7
    be.close()


def test_tables_that_are_no_rigid_body_chain_take_the_dual_number_path_and_equal_the_port(hip_lib, ctx):
    """The reference adds the angular velocity iRp @ axis (models.py:1821-1823); with a joint-origin rotation that moves the axis this is not the
    axis the joint rotation turns about, the recursion is no rigid-body dynamics any more and the closed form does not apply.  The library detects it
    (R0^T axis != axis) and differentiates the literal recursion; the numpy port does the same by complex step."""
    med7, robot, g = ctx
    T = 6
    from oracle.robot import rpy2r
    from oracle.torque import rnea_jacobian_spatial
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    Rx = rpy2r([0.3, -0.2, 0.1])
    prob.tb.R0[2] = Rx  # joint 3: axis z, origin now rotated about x / y as well
    q, qd, qdd = np.random.default_rng(SEED + 32).uniform(-1, 1, (3, 4, 7))
    assert np.abs(rnea_jacobian_spatial(prob.tb, q, qd, qdd)[1] - rnea_jacobian(prob.tb, q, qd, qdd)).max() > 1e-6
    dyn = robot.dynamics_tables()
    for k in range(9):
        dyn.R0[2][k] = float(Rx.reshape(-1)[k])
    nlp = TorqueMPCNLP(prob)
    qc = g["t6_qc"][0][None] + np.random.default_rng(SEED + 33).uniform(-0.1, 0.1, (2, 7))
    goal = np.stack([prob.goal_figure_eight(q) for q in qc])
    p = np.stack([nlp.pack_p(qc[b], np.zeros(7), goal[b]) for b in range(2)])
    be = TorqueBackend(robot.kinematic_chain(LINK), dyn, T=T, dt=0.1, tau_lo=-58.0, tau_up=58.0, **W)
    res = be.solve(np.stack([nlp.seed(q) for q in qc]), p)
    for b in range(2):
        o = solve_torque_ipm(prob, qc[b], np.zeros(7), goal[b])
        assert res.status[b] == o["status"] and o["status"] in (0, 4)
        assert abs(res.f[b] - o["f"]) <= 1e-9 * o["f"]
        assert abs(int(res.iters[b]) - o["iters"]) <= 1
    be.close()


def test_instance_at_the_arithmetic_floor_ends_at_the_optimum(hip_lib, ctx):
    """Instance 7359 of tools/gpu_tq_ipm_probe.py's batch of 8192: start within 1.5 N m of the gravity torque limit, at the optimum an effort row with
    slack 1.6e-8 whose multiplier mu_b / s is good to ~1e-6 relative -- the reduced gradient cannot be pushed below ~2e-6 in this arithmetic.  One build of
    round 4 reached the optimum at step 90 and then fed the watchdog for 450 steps (NUMERICAL).  Pinned: status converged (by tol or by the acceptable
    level: 25 steps at the floor of the barrier parameter within 10 tol), the objective equals the numpy port's to 1e-9, rows strictly feasible."""
    med7, robot, g = ctx
    T = 30
    qc = np.array([-0.009206280161798391, 0.6187609522084117, -0.030419334043609858, -1.4849330895196382, -0.03916752292410011, -0.6112430533197809,
                   0.04844326754516143])
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    nlp = TorqueMPCNLP(prob)
    goal = prob.goal_figure_eight(qc)
    be = backend(robot, T, 58.0, max_iter=600)
    res = be.solve(nlp.seed(qc)[None], nlp.pack_p(qc, np.zeros(7), goal)[None])
    o = solve_torque_ipm(prob, qc, np.zeros(7), goal, max_iter=600)
    assert res.status[0] == o["status"] and o["status"] in (0, 4)
    assert res.iters[0] <= 150
    assert abs(res.f[0] - o["f"]) <= 1e-9 * o["f"]
    kkt = np.asarray(res.kkt)[0]
    assert kkt[0] <= 1e-5 and kkt[1] == 0.0 and kkt[2] <= 1e-8
    be.close()


def test_reference_script_flow_through_hipsolver(hip_lib, ctx):
    med7, robot_, g = ctx
    import optas_amd as optas

    T, dt = 30, 0.1
    robot, link, opt = build_problem(T, dt, effort=55.0)
    solver = optas.HIPSolver(opt).setup("hip_sqp")
    qc = g["t30lim_qc"][0]
    goal = figure_eight_goal(robot, link, qc, T, dt)
    assert np.abs(goal.T - g["t30lim_goal"][0]).max() < 1e-12
    pd = {"qc": qc, "dqc": np.zeros(7), "goal": goal}
    solver.reset_parameters(pd)
    solver.reset_initial_seed({"med7/q/x": np.tile(qc[:, None], (1, T))})
    sol = solver.solve()
    assert solver.did_solve() and abs(solver.stats()["f"][0] - g["t30lim_f"][0]) <= 1e-9 * g["t30lim_f"][0]
    for key, shape in (("med7/q", (7, T)), ("med7/dq", (7, T)), ("med7/ddq", (7, T)), ("tau/y", (7, T)), ("tau/y/x", (7, T))):
        assert sol[key].shape == shape  # solver.py:137-155
    assert np.abs(sol["tau/y"]).max() <= 55.0 + 1e-8
    # numeric members of the problem IR (FK and RNEA through liboptas_hip) at the solution, in the reference's layout and signs
    x = solver.opt.decision_variables.dict2vec({k: v for k, v in sol.items() if k.endswith("/x")})
    p = solver.opt.parameters.dict2vec(pd)
    prob = TorqueProblem(med7, LINK, T=T, dt=dt, tau_lim=55.0, **W)
    nlp = TorqueMPCNLP(prob)
    po = nlp.pack_p(qc, np.zeros(7), goal.T)
    assert np.abs(np.asarray(p).reshape(-1) - po).max() == 0.0
    x = np.asarray(x).reshape(-1)
    assert abs(opt.f(x, p) - nlp.f(x, po)) <= 1e-10 * nlp.f(x, po)
    for name in ("k", "a", "h", "v"):
        assert np.abs(getattr(opt, name)(x, p) - getattr(nlp, name)(x, po)).max() <= 1e-9, name
    rng = np.random.default_rng(SEED + 12)
    xr = x + rng.normal(0, 0.05, x.shape)
    assert np.abs(opt.h(xr, p) - nlp.h(xr, po)).max() <= 1e-9 and np.abs(opt.a(xr, p) - nlp.a(xr, po)).max() <= 1e-12


def test_batch_of_8192_properties_determinism_and_scalar_equivalence(hip_lib, ctx):
    med7, robot, g = ctx
    T, B = 30, 8192  # BASELINE configs[4]
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    nlp = TorqueMPCNLP(prob)
    rng = np.random.default_rng(SEED)
    qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    qc = qn[None] + rng.uniform(-0.1, 0.1, (B, 7))
    pose, _ = robot._kin(LINK).fk_jac(qc, want_jac=False)
    ts = np.arange(T) * 0.1
    loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])  # (3, T)
    goal = np.empty((B, T, 3))
    from oracle.structured import FoldedChain

    Re = FoldedChain(med7, LINK).fk(qc)[1]
    goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
    assert np.abs(goal[7] - prob.goal_figure_eight(qc[7])).max() < 1e-12
    p = np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1)
    x0 = np.zeros((B, nlp.nx))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    be = backend(robot, T, 58.0, max_iter=600)  # effort rows bind in 70 % of this batch; round 4: p50 27 steps, the slowest instance ~130 (round 3: 39 / 600)
    res = be.solve(x0, p)
    assert _lib.status_ok(res.status).all() and np.median(res.iters) <= 30 and res.iters.max() <= 250
    assert res.kkt[:, 0].max() <= 1e-6 and res.kkt[:, 1].max() == 0.0 and res.kkt[:, 2].max() <= 1e-8
    tau = res.x[:, 3 * 7 * T:]
    assert np.abs(tau).max() < 58.0 and (np.abs(tau).max(1) > 58.0 - 1e-5).any()  # limits hold strictly (interior) and bind somewhere
    # linear rows of every instance (vectorised): q_{t+1} = q_t + dt dq_t, dq_{t+1} = dq_t + dt ddq_t, q_0 = qc, dq_0 = 0
    X = res.x.reshape(B, 4, T, 7)
    assert np.abs(X[:, 0, 1:] - X[:, 0, :-1] - 0.1 * X[:, 1, :-1]).max() <= 1e-12
    assert np.abs(X[:, 1, 1:] - X[:, 1, :-1] - 0.1 * X[:, 2, :-1]).max() <= 1e-12
    assert np.abs(X[:, 0, 0] - qc).max() == 0.0 and np.abs(X[:, 1, 0]).max() == 0.0
    # dynamics rows and objective of a sample with the literal functions
    for b in rng.integers(0, B, 6):
        assert np.abs(nlp.h(res.x[b], p[b])).max() <= 1e-10
        assert abs(nlp.f(res.x[b], p[b]) - res.f[b]) <= 1e-12 * res.f[b]
    # two runs agree bit for bit; one instance alone equals the same instance inside the batch
    res2 = be.solve(x0, p)
    assert np.array_equal(res.x, res2.x) and np.array_equal(res.iters, res2.iters)
    for b in (0, 4097, B - 1):
        r1 = be.solve(x0[b], p[b])
        assert np.array_equal(r1.x[0], res.x[b]) and r1.iters[0] == res.iters[b] and r1.f[0] == res.f[b]
    # three instances against the numpy port run here (port: ~1 s each)
    for b in (1, 2, 3):
        r = solve_torque_ipm(prob, qc[b], np.zeros(7), goal[b])
        assert abs(r["f"] - res.f[b]) <= 1e-9 * r["f"] and abs(r["iters"] - res.iters[b]) <= 2
        al = solve_torque_lm(prob, qc[b], np.zeros(7), goal[b])  # the independent second machine
        assert al["status"] == 0 and abs(al["f"] - res.f[b]) <= 1e-8 * al["f"]
    be.close()


def test_stored_curvature_term_leaves_the_optima_alone(hip_lib, ctx):
    """Round 5 (tq_curv_lag, default 3): the exact-curvature term of a knot is computed at every fourth evaluation of an instance and the stored one
    added in between (k_tq_eval3 / k_tq_curv; numpy: oracle/torque_ipm.py:solve_torque_ipm(curv_lag=)).  The term only shapes the quadratic model:
    pinned here -- with the lag and without it 2048 instances end at the same optima (objective 1e-9 relative, the same status), in the same number
    of steps to within a few, and the port with the same lag takes the GPU's steps."""
    med7, robot, g = ctx
    T, B = 30, 2048
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    nlp = TorqueMPCNLP(prob)
    rng = np.random.default_rng(SEED + 17)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])[None] + rng.uniform(-0.1, 0.1, (B, 7))
    goal = np.stack([prob.goal_figure_eight(q) for q in qc])
    p = np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1)
    x0 = np.zeros((B, nlp.nx))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    out = {}
    for lag in (3, 0, 8):
        be = backend(robot, T, 58.0, max_iter=600)
        be.set_option("tq_curv_lag", lag)
        assert be.get_option("tq_curv_lag") == lag
        r = be.solve(x0, p)
        out[lag] = (np.array(r.status), np.array(r.f), np.array(r.iters), np.array(r.kkt))
        be.close()
    for lag in (3, 8):
        assert _lib.status_ok(out[lag][0]).all() and _lib.status_ok(out[0][0]).all()
        rel = np.abs(out[lag][1] - out[0][1]) / np.abs(out[0][1])
        assert (rel <= 1e-9).mean() >= 0.999 and np.median(rel) <= 1e-12, (lag, rel.max())  # (an instance in a thousand may end in another KKT point)
        assert out[lag][3][:, 0].max() <= 1e-6 and out[lag][3][:, 2].max() <= 1e-8
        assert abs(out[lag][2].mean() - out[0][2].mean()) <= 0.5 and np.median(out[lag][2]) <= np.median(out[0][2]) + 1
    for b in (0, 5):
        for lag in (3, 0):
            ref = solve_torque_ipm(prob, qc[b], np.zeros(7), goal[b], curv_lag=lag)
            assert abs(ref["f"] - out[lag][1][b]) <= 1e-9 * ref["f"] and abs(ref["iters"] - out[lag][2][b]) <= 2, (b, lag, ref["iters"], out[lag][2][b])


def test_batch_solved_in_two_parts_on_two_streams_is_invisible(hip_lib, ctx):
    """Round 5 (solve_split): from tq_split_min instances on a batch is solved in two contiguous parts, each on a stream and a host thread of its
    own (the latency-bound sweep of one part overlaps the evaluation of the other: 8192 instances 56 -> 51 ms).  An instance's path does not depend
    on its batch in this family, so the split must be invisible: x, f, step counts, status and the returned multipliers bit for bit."""
    med7, robot, g = ctx
    T, B = 30, 2048
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    nlp = TorqueMPCNLP(prob)
    rng = np.random.default_rng(SEED + 23)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])[None] + rng.uniform(-0.1, 0.1, (B, 7))
    goal = np.stack([prob.goal_figure_eight(q) for q in qc])
    p = np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1)
    x0 = np.zeros((B, nlp.nx))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    out = {}
    for S in (2, 1):
        be = backend(robot, T, 58.0, max_iter=600)
        be.set_option("streams", S)
        r = be.solve(x0, p)
        out[S] = (np.array(r.x), np.array(r.f), np.array(r.iters), np.array(r.status), np.array(r.kkt), be.multipliers(B))
        be.close()
    for a, b in zip(out[2], out[1]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert _lib.status_ok(out[2][3]).all() and np.abs(out[2][5]).max() > 0


def test_abi_errors(hip_lib, ctx):
    med7, robot, g = ctx
    lib = hip_lib
    d = _lib.oh_torque_desc(T=30, ndof=8, dt=0.1, w_path=1.0, w_vel=0.0, w_tau=1.0)  # (2 .. 7 joints since the end of round 5: tests/test_gpu_torque_chain_lengths.py)
    h = C.c_void_p()
    assert lib.oh_create_torque(C.byref(d), C.byref(h)) == _lib.OH_ERR_INVALID and b"ndof" in lib.oh_last_error()
    d.ndof = 7
    assert lib.oh_create_torque(C.byref(d), C.byref(h)) == _lib.OH_ERR_INVALID and b"tau_lo" in lib.oh_last_error()
    for i in range(7):
        d.tau_lo[i], d.tau_up[i] = -1.0, 1.0
    d.w_tau = 0.0
    assert lib.oh_create_torque(C.byref(d), C.byref(h)) == _lib.OH_ERR_INVALID
    d.w_tau = 1.0
    assert lib.oh_create_torque(C.byref(d), C.byref(h)) == _lib.OH_OK
    x = np.zeros((1, 840))
    p = np.zeros((1, 104))
    assert lib.oh_solve(h, 1, _lib._ptr(x), _lib._ptr(p), None, None, None, None, None) == _lib.OH_ERR_STATE  # constants missing
    chain = robot.kinematic_chain(LINK)
    assert lib.oh_set_constants(h, C.byref(chain)) == _lib.OH_OK
    assert lib.oh_solve(h, 1, _lib._ptr(x), _lib._ptr(p), None, None, None, None, None) == _lib.OH_ERR_STATE and b"oh_set_dynamics" in lib.oh_last_error()
    lib.oh_destroy(h)


def test_host_buffer_solve_in_chunks_on_two_lanes_is_invisible(hip_lib, ctx):
    """Round 6 (solve_pipelined): oh_solve takes a large host batch in chunks on two lanes.  The torque family's answers do not depend on the batch, so the
    chunked call must return the bits of the unchunked one -- x, f, step counts, status and the multipliers (kept per chunk in a device-side cache)."""
    med7, robot, g = ctx
    T, B = 30, 3000
    rng = np.random.default_rng(SEED + 21)
    qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    qc = qn + rng.uniform(-0.1, 0.1, (B, 7))
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    goal = np.stack([prob.goal_figure_eight(q) for q in qc[:8]])
    goal = np.concatenate([goal, np.tile(goal, (B // 8 + 1, 1, 1))])[:B]  # (eight goal paths, recycled: the batch is about plumbing)
    p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
    x0 = np.zeros((B, 4 * 7 * T))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    a = backend(robot, T, 58.0, max_iter=600).set_options(pipe_chunk=1024)  # chunks of 1024, 1024, 952 on lanes 0, 1, 0
    b = backend(robot, T, 58.0, max_iter=600).set_options(pipe=0)
    ra, rb = a.solve(x0, p), b.solve(x0, p)
    assert _lib.status_ok(rb.status).mean() >= 0.99
    assert np.array_equal(ra.x, rb.x) and np.array_equal(ra.f, rb.f) and np.array_equal(ra.iters, rb.iters) and np.array_equal(ra.status, rb.status)
    assert np.array_equal(a.multipliers(B), b.multipliers(B))
    a.close()
    b.close()


def test_parts_of_a_split_solve_follow_dynamics_set_again(hip_lib, ctx):
    """oh_set_dynamics on a handle that has solved in parts reaches the peers (they are dropped and rebuilt): a heavier last link changes every part's answer,
    and the parts equal the same instances on a fresh handle with those dynamics."""
    med7, robot, g = ctx
    T, B = 30, 2048  # split from tq_split_min = 1024 instances on
    rng = np.random.default_rng(SEED + 22)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0]) + rng.uniform(-0.1, 0.1, (B, 7))
    prob = TorqueProblem(med7, LINK, T=T, dt=0.1, tau_lim=58.0, **W)
    goal = np.tile(prob.goal_figure_eight(qc[0])[None], (B, 1, 1))
    p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
    x0 = np.zeros((B, 4 * 7 * T))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    a = backend(robot, T, 58.0, max_iter=600)
    r1 = a.solve(x0, p)
    dyn = robot.dynamics_tables()
    dyn2 = type(dyn).from_buffer_copy(dyn)
    dyn2.mass[dyn2.n - 1] *= 1.5
    _lib.check(_lib.load().oh_set_dynamics(a.handle if hasattr(a, "handle") else a._h, C.byref(dyn2)), "oh_set_dynamics")
    r2 = a.solve(x0, p)
    fresh = TorqueBackend(robot.kinematic_chain(LINK), dyn2, T=T, dt=0.1, tau_lo=-58.0, tau_up=58.0, max_iter=600, **W)
    r3 = fresh.solve(x0, p)
    assert not np.array_equal(r2.f, r1.f) and np.array_equal(r2.x, r3.x) and np.array_equal(r2.f, r3.f) and np.array_equal(r2.iters, r3.iters)
    assert (np.abs(r2.f[B // 2 :] - r1.f[B // 2 :]) > 1e-9).all()  # the second part (the peer's) saw the new dynamics too
    a.close()
    fresh.close()
