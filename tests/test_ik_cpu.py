"""BASELINE config 1 (example/example.py) on the CPU: the augmented-Lagrangian port (oracle/ik_al.py, the state machine
the HIP kernel runs) against the committed golden optima, which come from scipy SLSQP in the reference's wiring
(solver.py:652-679) wherever both agree; lowering of the builder problem to OH_PROBLEM_IK."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, KUKA_KIN
from oracle.ik_al import position_curvature, solve_ik_al, _pos_jac_frames
from oracle.problems import IKExampleNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import FoldedChain

LINK = "end_effector_ball"


@pytest.fixture(scope="module")
def setup():
    kuka = OracleRobot(KUKA_KIN)
    return IKExampleNLP(kuka, LINK), FoldedChain(kuka, LINK), np.load(os.path.join(GOLDEN, "ik_golden.npz"))


def test_position_curvature_matches_finite_differences(setup):
    ik, ch, _ = setup
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, 7)
    y = rng.normal(size=3)
    _, Jp, om = _pos_jac_frames(ch, q)
    C = position_curvature(ch, Jp, om, y)
    eps = 1e-6
    Cfd = np.zeros((7, 7))
    for j in range(7):
        d = np.zeros(7)
        d[j] = eps
        Cfd[:, j] = (_pos_jac_frames(ch, q + d)[1].T @ y - _pos_jac_frames(ch, q - d)[1].T @ y) / (2 * eps)
    assert np.abs(C - Cfd).max() < 1e-8 and np.abs(C - C.T).max() == 0.0


def test_port_reproduces_golden_optima(setup):
    ik, ch, g = setup
    assert np.array_equal(g["lo"], ik.lo) and np.array_equal(g["up"], ik.up)
    assert abs(g["f"][0] - 0.29579887518) < 1e-9  # SURVEY App. D known answer of the script's instance
    assert g["agree"].all() and (g["nactive"] > 0).sum() >= 6  # SLSQP and the port found the same optimum everywhere
    for p, x0, x, f, na in zip(g["p"], g["x0"], g["x"], g["f"], g["nactive"]):
        r = solve_ik_al(ch, x0, p[:7], p[7:], ik.lo, ik.up, tol=1e-6, tol_feas=1e-9, max_iter=200)
        assert r["status"] == 0 and r["iterations"] <= 80
        assert abs(r["f"] - f) < 1e-7 and np.abs(r["x"] - x).max() < 1e-6
        k = kkt_reference_form(ik, r["x"], p, active_tol=1e-7)
        assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-9 and k["complementarity"] < 1e-7
        # multipliers in the reference's v >= 0 form
        lam = np.concatenate([r["z_lo"], r["z_up"], np.maximum(r["lam_h"], 0), np.maximum(-r["lam_h"], 0)])
        assert lam.min() >= 0.0
        res = ik.df(r["x"], p) - ik.dv(r["x"], p).T @ lam
        assert np.abs(res).max() < 1e-6
        assert int((r["z_lo"] > 0).sum() + (r["z_up"] > 0).sum()) == na


def test_builder_problem_lowers_to_ik_family():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from examples.example import setup_solver
    import optas_amd
    from optas_amd.lowering import IkSpec, lower
    from optas_amd.optimization import QuadraticCostNonlinearConstraints

    robot, opt = setup_solver(build_only=True)
    assert isinstance(opt, QuadraticCostNonlinearConstraints)
    assert (opt.nx, opt.np, opt.nk, opt.nh, opt.nv) == (7, 10, 14, 3, 20)
    kind, spec = lower(opt)
    assert kind == optas_amd._lib.OH_PROBLEM_IK and isinstance(spec, IkSpec)
    assert spec.link == LINK and spec.w_nominal == 1.0 and (spec.qn_name, spec.pg_name) == ("q_nominal", "p_goal")
    kuka = OracleRobot(KUKA_KIN)
    assert np.array_equal(spec.lo, kuka.lower_actuated_joint_limits) and np.array_equal(spec.up, kuka.upper_actuated_joint_limits)
    assert list(opt.parameters.offsets().items())[-2:] == [("q_nominal", 0), ("p_goal", 7)]
