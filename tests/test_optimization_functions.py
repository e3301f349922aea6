"""The function members of the mirrored Optimization classes (f, k, a, g, h, v, M, c, A, b, P, q; optimization.py:192-306)
against the oracle's literal restatement of the same problems: two independent implementations of the builder's layout
and sign conventions (rhs - lhs storage, v = [k; g; a; -a; h; -h], column-major blocks) must produce the same numbers.
The robot problems need forward kinematics, which the product evaluates through liboptas_hip: those are GPU tests."""
import os
import sys

import numpy as np
import pytest

from conftest import KUKA_KIN, SEED
from oracle.problems import FigureEightNLP, GuardedDualArmNLP, IKExampleNLP, PointMassMPCNLP
from oracle.robot import OracleRobot

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def test_point_mass_functions_match_oracle():
    from examples.point_mass_mpc import Controller, obstacle_and_goal

    o = Controller(build_only=True).optimization
    nlp = PointMassMPCNLP()
    rng = np.random.default_rng(SEED)
    x = rng.normal(size=o.nx)
    curr = np.array([-0.45, -0.35])
    obs, goal = obstacle_and_goal(2.0, curr)
    p = o.parameters.dict2vec({"curr": curr, "dcurr": [0.6, 0.6], "goal": goal, "obs": obs})
    assert abs(o.f(x, p) - nlp.f(x, p)) < 1e-12 and np.abs(o.v(x, p) - nlp.v(x, p)).max() < 1e-13
    assert (o.v(x, p).shape[0], o.lbv.shape[0], o.ubv[0]) == (264, 264, 1e10)  # optimization.py:301-303
    M, c, A, b = o.M(p), o.c(p), o.A(p), o.b(p)
    assert np.abs(M @ x + c - o.k(x, p)).max() < 1e-13 and np.abs(M - nlp.dk(x, p)).max() == 0.0  # k = Mx + c (:245-248)
    assert np.abs(A @ x + b - o.a(x, p)).max() < 1e-13 and np.abs(A - nlp.da(x, p)).max() == 0.0  # a = Ax + b (:257-260)
    P, q = o.P(p), o.q(p)
    assert np.abs(2.0 * P - nlp.ddf(x, p)).max() < 1e-12 and np.abs(q - nlp.df(np.zeros(o.nx), p)).max() < 1e-12  # P = ddf/2, q = df(0)
    assert abs(x @ P @ x + q @ x + o.f(np.zeros(o.nx), p) - o.f(x, p)) < 1e-11


def test_quadratic_task_problem_known_answer():
    """A two-variable quadratic in the reference's test style (tests/test_optimization.py:138-143): f = x^T P x + q^T x."""
    import optas_amd
    from optas_amd.builder import OptimizationBuilder
    from optas_amd.expr import sumsqr
    from optas_amd.optimization import QuadraticCostLinearConstraints

    task = optas_amd.TaskModel("t", 2, time_derivs=[0], dlim={0: [-3.0, 4.0]})
    b = OptimizationBuilder(1, tasks=task)
    y = b.get_model_state("t", 0)
    goal = b.add_parameter("goal", 2)
    b.add_cost_term("c", 3.0 * sumsqr(y - goal))
    b.enforce_model_limits("t")
    o = b.build()
    assert isinstance(o, QuadraticCostLinearConstraints) and (o.nx, o.np, o.nk, o.nv) == (2, 2, 4, 4)
    p = np.array([1.0, -2.0])
    assert np.allclose(o.P(p), 3.0 * np.eye(2)) and np.allclose(o.q(p), [-6.0, 12.0])
    assert np.allclose(o.M(p), np.vstack([np.eye(2), -np.eye(2)])) and np.allclose(o.c(p), [3.0, 3.0, 4.0, 4.0])  # x - lo; up - x
    assert np.allclose(o.v(np.array([0.5, 0.25]), p), [3.5, 3.25, 3.5, 3.75])


@pytest.mark.gpu
def test_robot_problem_functions_match_oracle(hip_lib):
    from examples.dual_arm import N_OBSTACLES, SPHERE_LINKS, obstacle_parameters
    from examples.dual_arm import setup_solver as dual_arm
    from examples.example import setup_solver as ik_example
    from examples.figure_eight_plan import setup_solver as figure_eight

    rng = np.random.default_rng(SEED)
    kuka = OracleRobot(KUKA_KIN)
    # config 2
    _, o = figure_eight(build_only=True)
    nlp = FigureEightNLP(kuka, "end_effector_ball", T=50)
    x, p = rng.uniform(-1, 1, o.nx), rng.uniform(-1, 1, 7)
    assert abs(o.f(x, p) - nlp.f(x, p)) < 1e-9 * abs(nlp.f(x, p)) and np.abs(o.v(x, p) - nlp.v(x, p)).max() < 1e-12
    assert np.abs(o.A(p) - nlp.da(x, p)).max() == 0.0
    # config 1
    _, o = ik_example(build_only=True)
    nlp = IKExampleNLP(kuka, "end_effector_ball")
    x, p = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 10)
    assert abs(o.f(x, p) - nlp.f(x, p)) < 1e-13 and np.abs(o.v(x, p) - nlp.v(x, p)).max() < 1e-13
    assert np.allclose(o.P(p), np.eye(7)) and np.allclose(o.q(p), -2.0 * p[:7])
    # config 4 synthetic
    T = 6
    _, o = dual_arm(T=T, build_only=True, limits=True, collision=True)
    rl = OracleRobot(KUKA_KIN, name="kukal")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr = OracleRobot(KUKA_KIN, name="kukar")
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    nlp = GuardedDualArmNLP(rl, rr, SPHERE_LINKS, N_OBSTACLES, T=T)
    x = rng.uniform(-1, 1, o.nx)
    p = o.parameters.dict2vec({"qcl": rng.uniform(-1, 1, 7), "qcr": rng.uniform(-1, 1, 7), **obstacle_parameters()})
    assert abs(o.f(x, p) - nlp.f(x, p)) < 1e-12 * abs(nlp.f(x, p)) and np.abs(o.v(x, p) - nlp.v(x, p)).max() < 1e-12


def test_derivative_members_of_a_task_problem():
    """df / dk / dv / ddf of a problem without kinematics (no GPU): optimization.py:8-24, 198, 304-306."""
    import optas_amd
    from optas_amd.builder import OptimizationBuilder
    from optas_amd.expr import sumsqr

    task = optas_amd.TaskModel("t", 2, time_derivs=[0], dlim={0: [-3.0, 4.0]})
    b = OptimizationBuilder(3, tasks=task)
    Y = b.get_model_states("t", 0)
    goal = b.add_parameter("goal", 2, 3)
    b.add_cost_term("c", 3.0 * sumsqr(Y - goal))
    b.enforce_model_limits("t")
    o = b.build()
    rng = np.random.default_rng(SEED)
    x, p = rng.normal(size=o.nx), rng.normal(size=o.np)
    assert o.df(x, p).shape == (1, 6) and np.allclose(o.df(x, p)[0], 6.0 * (x - p))
    assert np.allclose(o.dk(x, p), np.vstack([np.eye(6), -np.eye(6)])) and np.allclose(o.dv(x, p), o.dk(x, p))
    assert np.allclose(o.ddf(x, p), 6.0 * np.eye(6), atol=1e-6)


@pytest.mark.gpu
def test_derivative_members_match_the_oracle_derivatives(hip_lib):
    """df, da, dh, dk, dv of the robot problems (FK Jacobians through oh_fk_jac, inverse-dynamics Jacobian through oh_rnea_jac) against the
    oracle's analytic / complex-step derivatives of the literal NLPs; ddf against a difference of the oracle's df."""
    from conftest import MED7_KIN
    from examples.example import setup_solver as ik_example
    from examples.figure_eight_plan import setup_solver as figure_eight
    from examples.torque_mpc import build_problem
    from oracle.problems import TorqueMPCNLP
    from oracle.torque import TorqueProblem

    rng = np.random.default_rng(SEED + 5)
    kuka = OracleRobot(KUKA_KIN)
    # config 2, short horizon
    T = 6
    _, o = figure_eight(T=T, Tmax=10.0 * (T - 1) / 49.0, build_only=True)
    nlp = FigureEightNLP(kuka, "end_effector_ball", T=T, Tmax=10.0 * (T - 1) / 49.0)
    x, p = rng.uniform(-1, 1, o.nx), rng.uniform(-1, 1, 7)
    assert np.abs(o.df(x, p)[0] - nlp.df(x, p)).max() < 1e-9 * max(1.0, np.abs(nlp.df(x, p)).max())
    assert np.abs(o.da(x, p) - nlp.da(x, p)).max() == 0.0 and np.abs(o.dh(x, p) - nlp.dh(x, p)).max() < 1e-12
    assert np.abs(o.dv(x, p) - nlp.dv(x, p)).max() < 1e-12 and o.dv(x, p).shape == (o.nv, o.nx)
    # config 1: Hessian of the quadratic cost, second derivatives of the position rows against differences of the oracle's dh
    _, o = ik_example(build_only=True)
    nlp = IKExampleNLP(kuka, "end_effector_ball")
    x, p = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 10)
    assert np.abs(o.df(x, p)[0] - nlp.df(x, p)).max() < 1e-13 and np.abs(o.dv(x, p) - nlp.dv(x, p)).max() < 1e-12
    assert np.abs(o.ddf(x, p) - 2.0 * np.eye(7)).max() < 1e-6
    H = o.ddh(x, p)
    assert H.shape == (3, 7, 7)
    e = np.zeros(7)
    e[2] = 1e-6
    assert np.abs(H[:, :, 2] - (nlp.dh(x + e, p) - nlp.dh(x - e, p)) / 2e-6).max() < 1e-5
    # config 5: the dynamics rows
    T = 4
    _, _, o = build_problem(T=T, effort=60.0)
    prob = TorqueProblem(OracleRobot(MED7_KIN), "lbr_link_ee", T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=60.0)
    nlp = TorqueMPCNLP(prob)
    x, p = rng.uniform(-1, 1, o.nx), rng.uniform(-1, 1, o.np)
    assert np.abs(o.dh(x, p) - nlp.dh(x, p)).max() < 1e-9 and np.abs(o.df(x, p)[0] - nlp.df(x, p)).max() < 1e-9 * max(1.0, np.abs(nlp.df(x, p)).max())
    assert np.abs(o.da(x, p) - nlp.da(x, p)).max() == 0.0 and np.abs(o.dk(x, p) - nlp.dk(x, p)).max() == 0.0
