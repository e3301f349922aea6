"""scipy trust-constr wired the way the reference's ScipyMinimizeSolver wires it (optas/solver.py:680-712: k and a as LinearConstraints
with their own bounds, g and h as NonlinearConstraints -- the split interface, as opposed to the stacked v >= 0 that SLSQP gets) on the
literal NLPs of configs 1, 3 and 4: a third solver, after SLSQP and the ports, that must find the committed optima.  (Config 2: no scipy
method converges on the literal rank-deficient quaternion rows; config 5: tests/golden/torque_golden.npz holds the
trust-constr optimum, 6-8 minutes per instance.)  CPU only."""
import os

import numpy as np

from conftest import GOLDEN, KUKA_KIN
from oracle.problems import DualArmNLP, IKExampleNLP, PointMassMPCNLP
from oracle.robot import OracleRobot
from oracle.solvers import scipy_minimize
from oracle.structured import FoldedChain, solve_free_lm


def test_config_1_ik_trust_constr_finds_the_golden_optimum(golden_nlp):
    ik = IKExampleNLP(OracleRobot(KUKA_KIN), "end_effector_ball")
    p = golden_nlp["ik_p"]
    r = scipy_minimize(ik, p[:7], p, method="trust-constr", tol=1e-10, options={"maxiter": 3000})  # seed = q_nominal
    assert abs(r.fun - float(golden_nlp["ik_f"])) <= 1e-9 and np.abs(r.x - golden_nlp["ik_x"]).max() <= 1e-6
    assert r.constr_violation <= 1e-9


def test_config_3_point_mass_trust_constr_finds_the_golden_optimum():
    pm = np.load(os.path.join(GOLDEN, "pm_golden.npz"))
    nlp = PointMassMPCNLP()
    r = scipy_minimize(nlp, np.zeros(80), pm["p"][1], method="trust-constr", tol=1e-10, options={"maxiter": 3000})
    # trust-constr is an interior method: it stops a barrier parameter away from the active obstacle row
    assert abs(r.fun - pm["f"][1]) <= 1e-6 * pm["f"][1] and np.abs(r.x - pm["x"][1]).max() <= 5e-3 and r.constr_violation <= 1e-9


def test_config_4_dual_arm_trust_constr_agrees_with_the_port_on_a_short_horizon():
    T = 5
    rl, rr = OracleRobot(KUKA_KIN, name="kukal"), OracleRobot(KUKA_KIN, name="kukar")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    nlp = DualArmNLP(rl, rr, T=T, Tmax=10.0 * (T - 1) / 49.0)
    qc = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    p = np.concatenate([qc, qc])
    x0 = np.concatenate([np.concatenate([np.tile(qc, T), np.zeros(7 * (T - 1))]) for _ in range(2)])
    r = scipy_minimize(nlp, x0, p, method="trust-constr", tol=1e-10, options={"maxiter": 2000})
    off = nlp.offsets
    f_port = sum(solve_free_lm(FoldedChain(rob, "end_effector_ball"), T, nlp.dt, off[arm].T, qc, Q0=np.tile(qc, (T, 1)), tol=1e-10)["f"]
                 for arm, rob in (("l", rl), ("r", rr)))
    assert abs(r.fun - f_port) <= 1e-8 * max(1.0, f_port) and r.constr_violation <= 1e-10
