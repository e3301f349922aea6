"""example/torque_control_example.py (:19-104) through HIPSolver: the squares-of-affine rows are lowered as bands to the dense-QP family; the
answers are compared with the exact minimiser of the literal problem (oracle/problems.py:TorqueControlNLP + band_qp_exact: active-set
enumeration, no iterative tolerance) and graded by the reference-form KKT residuals of the literal rows 'eps - d^2 >= 0'."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle():
    import optas_amd
    from oracle.problems import TorqueControlNLP
    from oracle.robot import OracleRobot

    robot = OracleRobot(os.path.join(os.path.dirname(optas_amd.__file__), "robots", "med7.kin.json"))
    return TorqueControlNLP(robot)


def _instances(nlp, count, seed=0):
    rng = np.random.default_rng(seed)
    q0 = np.deg2rad([0, 30, 0, -90, 0, 60, 0])
    out = []
    for _ in range(count):
        qc = q0 + rng.uniform(-0.3, 0.3, 7)
        pc = np.asarray(nlp.robot.get_global_link_position(nlp.link, qc)).reshape(3)
        quat = rng.normal(size=4)
        quat /= np.linalg.norm(quat)
        pg = np.concatenate([pc + rng.uniform(-0.003, 0.003, 3), quat if rng.random() < 0.5 else [0.0, 1.0, 0.0, 0.0]])
        out.append(np.concatenate([qc, pg]))
    return np.array(out)


def _exact(nlp, p):
    from oracle.problems import band_qp_exact

    A, b, _, _ = nlp.pieces(p)
    z = np.zeros(nlp.nx)
    return band_qp_exact(nlp.ddf(z, p), nlp.df(z, p), A, b, np.sqrt(nlp.bounds))


def test_problem_class_and_numeric_members_match_the_literal_problem():
    from examples.torque_control_example import TrackingController
    from optas_amd.optimization import QuadraticCostNonlinearConstraints

    ctrl = TrackingController(1.0 / 500.0, build_only=True)
    opt, nlp = ctrl.optimization, _oracle()
    assert isinstance(opt, QuadraticCostNonlinearConstraints)
    assert (opt.nx, opt.np, opt.nk, opt.ng, opt.na, opt.nh) == (7, 14, 0, 3, 0, 0)
    rng = np.random.default_rng(1)
    for p in _instances(nlp, 3, seed=2):
        x = rng.normal(size=7)
        assert abs(opt.f(x, p) - nlp.f(x, p)) <= 1e-10 * abs(nlp.f(x, p))
        assert np.abs(np.asarray(opt.g(x, p)).reshape(-1) - nlp.g(x, p)).max() <= 1e-12
        assert np.abs(np.asarray(opt.df(x, p)).reshape(-1) - nlp.df(x, p)).max() <= 1e-9 * np.abs(nlp.df(x, p)).max()
        assert np.abs(np.asarray(opt.dg(x, p)) - nlp.dg(x, p)).max() <= 1e-12
        assert np.abs(np.asarray(opt.ddf(x, p)) - nlp.ddf(x, p)).max() <= 1e-9 * np.abs(nlp.ddf(x, p)).max()
        assert np.abs(np.asarray(opt.v(x, p)).reshape(-1) - nlp.v(x, p)).max() <= 1e-12


def test_single_solve_equals_the_exact_minimiser_of_the_literal_problem():
    from examples.torque_control_example import TrackingController
    from optas_amd.lowering import QpSpec
    from oracle.solvers import kkt_reference_form

    ctrl = TrackingController(1.0 / 500.0)
    assert isinstance(ctrl.solver._spec, QpSpec) and ctrl.solver._spec.bands == ("eff_x", "eff_y", "eff_z")
    nlp = _oracle()
    for p in _instances(nlp, 4, seed=3):
        dq = ctrl.compute_target_velocity(p[:7], p[7:])
        assert ctrl.solver.did_solve()
        x, fval, state, _ = _exact(nlp, p)
        assert np.abs(dq - x).max() <= 1e-6 * max(1.0, np.abs(x).max()), (dq, x)
        assert abs(nlp.f(dq, p) - nlp.f(x, p)) <= 1e-9 * abs(nlp.f(x, p))
        assert nlp.g(dq, p).min() >= -1e-12  # the literal rows eps - d^2: 1e-12 of a bound of 1e-8
        kkt = kkt_reference_form(nlp, dq, p, active_tol=1e-11)
        assert kkt["stationarity"] <= 1e-6 and kkt["feasibility"] <= 1e-12 and kkt["complementarity"] <= 1e-6, kkt


def test_batch_of_ticks_in_one_launch():
    from examples.torque_control_example import TrackingController

    ctrl = TrackingController(1.0 / 500.0)
    nlp = _oracle()
    P = _instances(nlp, 256, seed=4)
    res = ctrl.solver.solve_batch_arrays(np.zeros((len(P), 7)), P)
    X = np.asarray(res.x).reshape(len(P), 7)
    assert np.all(np.asarray(res.status) == 0)
    worst = 0.0
    active = 0
    for i in range(0, len(P), 8):
        x, _, state, _ = _exact(nlp, P[i])
        worst = max(worst, np.abs(X[i] - x).max() / max(1.0, np.abs(x).max()))
        active += sum(1 for s in state if s)
    assert worst <= 1e-6, worst
    assert active > 0  # the bands bind on this sample: they are what moves the arm (the speed penalty outweighs the tracking weight)


def test_closed_loop_follows_the_goal():
    from examples.torque_control_example import main

    assert main(ticks=10) == 0
