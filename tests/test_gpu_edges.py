"""Edge cases of the figure-eight family through the C ABI: shortest and longest horizons (T = 3 has a single free knot,
T = 128 (OH_MAX_T until round 6; 256 since) exceeds the one-wave-per-instance tail kernel and runs on the batched kernels only), non-finite inputs
(reported per instance, never a hang), call-order and descriptor errors (int codes + oh_last_error, no exceptions across the ABI)."""
import ctypes as C

import numpy as np
import pytest

from conftest import KUKA_KIN, SEED
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
from oracle.robot import OracleRobot
from oracle.structured import StructuredFigureEight, solve_structured_lm

pytestmark = pytest.mark.gpu
LINK = "end_effector_ball"
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])


@pytest.mark.parametrize("T", [3, 4, 7, 128])
def test_horizon_extremes_match_port(hip_lib, T):
    prob = StructuredFigureEight(OracleRobot(KUKA_KIN), LINK, T=T)
    robot = RobotModel(urdf_filename=KUKA_KIN)
    be = FigureEightBackend(robot.kinematic_chain(LINK), T, prob.dt, prob.local_path, max_iter=300, tol=1e-6)
    rng = np.random.default_rng(SEED + T)
    B = 3
    qc = QC0 + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.1, 0.1, (B - 1, 7))])
    x0 = np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], axis=1)
    r = be.solve(x0, qc)
    assert r.x.shape == (B, 7 * T + 7 * (T - 1))
    for b in range(B):
        s = solve_structured_lm(prob, qc[b], max_iter=300, tol=1e-6)
        assert r.status[b] == s["status"] == 0 and abs(int(r.iters[b]) - s["iters"]) <= 1
        assert abs(r.f[b] - s["f"]) <= 1e-9 * max(1.0, abs(s["f"]))
        Q = r.x[b, : 7 * T].reshape(T, 7)
        assert np.array_equal(Q[0], qc[b]) and np.array_equal(Q[1], qc[b])  # q_0 = q_1 = qc (fixed rows)
    be.close()


def test_non_finite_inputs_are_reported_per_instance(hip_lib):
    prob = StructuredFigureEight(OracleRobot(KUKA_KIN), LINK, T=50)
    robot = RobotModel(urdf_filename=KUKA_KIN)
    be = FigureEightBackend(robot.kinematic_chain(LINK), 50, prob.dt, prob.local_path, max_iter=50, tol=1e-6)
    B = 70
    qc = np.tile(QC0, (B, 1))
    qc[3, 2] = np.nan
    qc[66, 0] = np.inf
    x0 = np.concatenate([np.tile(qc, (1, 50)), np.zeros((B, 343))], axis=1)
    r = be.solve(x0, qc)
    bad = np.zeros(B, bool)
    bad[[3, 66]] = True
    assert (r.status[bad] != 0).all() and (r.status[~bad] == 0).all()  # neighbours in the same wavefront are unaffected
    assert np.isfinite(r.f[~bad]).all() and np.ptp(r.f[~bad]) == 0.0
    be.close()


def test_non_finite_inputs_other_families(hip_lib):
    from optas_amd.backend import IKBackend, PointMassBackend

    robot = RobotModel(urdf_filename=KUKA_KIN)
    ik = IKBackend(robot.kinematic_chain(LINK), robot.lower_actuated_joint_limits, robot.upper_actuated_joint_limits)
    qn = np.tile(np.deg2rad([0, 45, 0, -90, 0, -45, 0]), (5, 1))
    p = np.concatenate([qn, np.tile([0.5, 0.2, 0.6], (5, 1))], axis=1)
    p[1, 8] = np.nan
    p[4, 0] = np.inf
    r = ik.solve(qn, p)
    assert list(r.status != 0) == [False, True, False, False, True]
    pm = PointMassBackend()
    P = np.zeros((3, 84))
    P[:, 4:44:2] = P[:, 5:44:2] = np.linspace(0.0, 1.0, 20)  # goal ramp, obstacle at the origin far behind
    P[:, 0:2] = -1.0
    P[:, 44:84] = -5.0
    P[1, 10] = np.nan
    r = pm.solve(np.zeros((3, 80)), P)
    assert r.status[0] == 0 and r.status[2] == 0 and r.status[1] != 0


def test_abi_error_codes(hip_lib):
    lib = hip_lib
    lib.oh_last_error.restype = C.c_char_p
    lp = np.zeros((50, 3))
    h = C.c_void_p()

    def desc(**kw):
        d = dict(kind=_lib.OH_PROBLEM_FIGURE_EIGHT, T=50, ndof=7, dt=0.2, w_path=1000.0, w_vel=0.01,
                 local_path=lp.ctypes.data_as(C.POINTER(C.c_double)), lock_orientation=1, fix_dq0=1, path_in_frame=1, max_iter=10, tol=1e-6,
                 tol_feas=1e-9, hessian=0, mu0=0.0)
        d.update(kw)
        return _lib.oh_problem_desc(**d)

    for bad in (dict(ndof=3), dict(ndof=9), dict(T=2), dict(T=_lib.OH_MAX_T + 1), dict(dt=0.0), dict(hessian=7), dict(kind=42)):  # (ndof 4 ... 8 with orientation rows since round 5)
        d = desc(**bad)
        assert lib.oh_create(C.byref(d), C.byref(h)) == 1 and lib.oh_last_error()  # OH_ERR_INVALID
    d = desc()
    assert lib.oh_create(C.byref(d), C.byref(h)) == 0
    x = np.zeros((1, 693))
    p = np.zeros((1, 7))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.oh_solve(h, 1, vp(x), vp(p), vp(x), None, None, None, None) == 3  # OH_ERR_STATE: constants not set
    assert b"oh_set_constants" in lib.oh_last_error()
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain("lwr_arm_3_link")  # 3 of 7 joints: not a solver chain
    assert lib.oh_set_constants(h, C.byref(chain)) == 0
    assert lib.oh_solve(h, 1, vp(x), vp(p), vp(x), None, None, None, None) == 1
    assert lib.oh_solve(h, 0, vp(x), vp(p), vp(x), None, None, None, None) == 1
    g = _lib.oh_guards()
    g.limits = 1
    g.n_links = g.n_obstacles = 1
    assert lib.oh_set_guards(h, C.byref(g)) == 1  # sphere rows are not lowered for the orientation-locked family (joint limits are)
    lib.oh_destroy(h)


def test_other_robot_med7_matches_port(hip_lib):
    """The reference script's default robot is the KUKA LBR Med7 (figure_eight_plan.py:26-28): different joint origins/axes
    exercise the general-axis / non-identity pre-rotation paths of the kinematics walk."""
    from conftest import MED7_KIN

    prob = StructuredFigureEight(OracleRobot(MED7_KIN), "lbr_link_ee", T=50)
    robot = RobotModel(urdf_filename=MED7_KIN)
    be = FigureEightBackend(robot.kinematic_chain("lbr_link_ee"), 50, prob.dt, prob.local_path, max_iter=300, tol=1e-6)
    rng = np.random.default_rng(SEED + 50)
    B = 4
    qc = QC0 + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.1, 0.1, (B - 1, 7))])
    x0 = np.concatenate([np.tile(qc, (1, 50)), np.zeros((B, 343))], axis=1)
    r = be.solve(x0, qc)
    for b in range(B):
        s = solve_structured_lm(prob, qc[b], max_iter=300, tol=1e-6)
        assert r.status[b] == s["status"] == 0 and abs(int(r.iters[b]) - s["iters"]) <= 1
        assert abs(r.f[b] - s["f"]) <= 1e-9 * abs(s["f"])
    be.close()


def test_parameterised_lead_joint_six_optimised_joints(hip_lib):
    """example/figure_eight_plan_6dof.py: RobotModel(param_joints=["lwr_arm_0_joint"]) -- six optimised joints behind a parameterised
    one (the N = 6 kernels, a per-knot lead frame).  Same state machine as the numpy port over the optimised joints; the solution is
    checked through the mirrored Optimization's own f, a, h (forward kinematics of the full 7-joint chain)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from examples.figure_eight_plan_6dof import plan, setup_solver
    from oracle.structured import lead_problem

    kuka, solver = setup_solver()
    o = solver.opt
    assert (o.nx, o.np, o.na, o.nh, o.nv) == (594, 106, 306, 200, 1012)
    qc = QC0.copy()
    qc[0] = 0.2
    sol, interp = plan(kuka, solver, qc)
    assert solver.did_solve()
    assert sol["kuka/q"].shape == (7, 50) and np.allclose(sol["kuka/q"][0], 0.2) and sol["kuka/q/x"].shape == (6, 50)  # solver.py:137-155
    prob = lead_problem(OracleRobot(KUKA_KIN), LINK, 0, np.full(50, 0.2), 0.2, T=50)
    s = solve_structured_lm(prob, qc[1:], max_iter=300, tol=1e-6)
    assert s["status"] == 0 and abs(solver.number_of_iterations() - s["iters"]) <= 1
    assert abs(solver.stats()["f"][0] - s["f"]) < 1e-8 * s["f"] and np.abs(np.asarray(sol["kuka/q/x"]).T - s["Q"]).max() < 1e-4
    x = o.decision_variables.dict2vec(sol)
    Q0 = np.diag(qc) @ np.ones((7, 50))
    p = o.parameters.dict2vec({"qc": qc, "kuka/q/p": Q0[[0]]})
    assert abs(o.f(x, p) - solver.stats()["f"][0]) < 1e-9 and np.abs(o.a(x, p)).max() < 1e-12 and np.abs(o.h(x, p)).max() < 1e-9
    # a moving parameterised joint (the parameter is a trajectory, 1 x T) and a batch
    th = 0.2 + 0.15 * np.sin(np.linspace(0.0, np.pi, 50))
    th[:2] = 0.2
    solver.reset_parameters({"qc": qc, "kuka/q/p": th.reshape(1, -1)})
    sol2 = solver.solve()
    s2 = solve_structured_lm(lead_problem(OracleRobot(KUKA_KIN), LINK, 0, th, 0.2, T=50), qc[1:], max_iter=300, tol=1e-6)
    assert solver.did_solve() and abs(solver.stats()["f"][0] - s2["f"]) < 1e-8 * s2["f"] and np.allclose(sol2["kuka/q"][0], th)
