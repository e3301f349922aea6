"""Host pieces behind example/torque_control_example.py that need no GPU: the structural expression node (transpose, skew, horzcat), the
reference's Quaternion.getrotm restated, ``@`` with a scalar operand, and the lowering of rows ``c - e*e >= 0`` to bands of a dense QP.
CPU: values, first and second derivatives of the nodes against numpy / differences, the rewritten problem's M, c, the oracle's literal NLP
(oracle/problems.py:TorqueControlNLP) against differences, its exact active-set solver against SLSQP and the interior-point oracle."""
import os

import numpy as np
import pytest

import optas_amd
from conftest import SEED
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import Gather, horzcat, sumsqr, transpose, vertcat
from optas_amd.lowering import LoweringError, QpSpec, lower, match_qp
from optas_amd.optimization import QuadraticCostLinearConstraints, QuadraticCostNonlinearConstraints
from optas_amd.spatialmath import I3, Quaternion, skew


def _builder():
    b = OptimizationBuilder(1)
    x = b.add_decision_variables("x", 3)
    y = b.add_decision_variables("y", 2)
    a = b.add_parameter("a", 3)
    return b, x, y, a


def test_structural_nodes_values_and_derivatives():
    b, x, y, a = _builder()
    S = skew(x)  # 3 x 3, linear in x
    assert isinstance(S, Gather) and S.shape == (3, 3) and S.degree() == 1
    H = horzcat(x, a, 2.0 * x)  # 3 x 3
    assert H.shape == (3, 3)
    Rt = transpose(vertcat(x, y))  # 1 x 5
    assert Rt.shape == (1, 5) and x.T.shape == (1, 3)
    b.add_cost_term("c1", sumsqr(S @ a))  # ||x x a||^2
    b.add_cost_term("c2", (x.T @ (H @ x)))  # x'(H x), cubic
    b.add_cost_term("c3", sumsqr(Rt @ np.arange(1.0, 6.0).reshape(5, 1)))
    b.add_cost_term("c4", sumsqr((skew(y[0]) @ y) - np.array([[1.0], [2.0]])))
    opt = b.build()
    rng = np.random.default_rng(SEED)
    xv, p = rng.normal(size=5), rng.normal(size=3)

    def f_np(v):
        X, Y = v[:3], v[3:]
        Hm = np.column_stack([X, p, 2 * X])
        s2 = np.array([[0.0, -Y[0]], [Y[0], 0.0]])
        return (np.sum(np.cross(X, p) ** 2) + X @ (Hm @ X) + (np.arange(1.0, 6.0) @ v) ** 2 + np.sum((s2 @ Y - np.array([1.0, 2.0])) ** 2))

    assert abs(opt.f(xv, p) - f_np(xv)) <= 1e-12 * max(1.0, abs(f_np(xv)))
    h = 1e-5
    g_fd = np.array([(f_np(xv + h * e) - f_np(xv - h * e)) / (2 * h) for e in np.eye(5)])
    assert np.abs(np.asarray(opt.df(xv, p)).reshape(-1) - g_fd).max() <= 1e-7 * max(1.0, np.abs(g_fd).max())
    H_fd = np.array([(np.asarray(opt.df(xv + h * e, p)).reshape(-1) - np.asarray(opt.df(xv - h * e, p)).reshape(-1)) / (2 * h) for e in np.eye(5)])
    assert np.abs(np.asarray(opt.ddf(xv, p)) - 0.5 * (H_fd + H_fd.T)).max() <= 1e-6 * max(1.0, np.abs(H_fd).max())
    assert np.array_equal(np.asarray(skew(np.array([1.0, 2.0, 3.0]))), np.array([[0.0, -3.0, 2.0], [3.0, 0.0, -1.0], [-2.0, 1.0, 0.0]]))


def test_structural_nodes_on_the_tape():
    from optas_amd.tape import compile_problem
    from oracle import tape_ref

    b, x, y, a = _builder()
    b.add_cost_term("c", sumsqr(skew(x) @ a) + sumsqr(horzcat(x, a).T @ x))
    b.add_leq_inequality_constraint("r", (x.T @ a) * (x.T @ a), 4.0)
    opt = b.build()
    tape = compile_problem(opt)
    rng = np.random.default_rng(SEED + 1)
    xv, p = rng.normal(size=5), rng.normal(size=3)
    v = tape_ref.forward(tape, xv, p)
    f = v[tape.out_cost]
    assert abs(f - opt.f(xv, p)) <= 1e-12 * max(1.0, abs(f))
    assert abs(v[tape.out_rows[0]] - (4.0 - (xv[:3] @ p) ** 2)) <= 1e-12


def test_the_example_compiles_to_a_tape_equal_to_the_literal_problem():
    """The whole of example/torque_control_example.py:44-95 -- J(qc), p(qc), R(qc) of the med7 chain, the goal matrix out of the quaternion
    entries, transposes and the skew -- as one scalar tape, for the problem as written and for the banded QP the kernel is handed."""
    from examples.torque_control_example import TrackingController
    from optas_amd.tape import compile_problem
    from oracle import tape_ref

    ctrl = TrackingController(1.0 / 500.0, build_only=True)
    nlp = _control_nlp()
    rng = np.random.default_rng(SEED + 5)
    tape = compile_problem(ctrl.optimization)
    assert (tape.nx, tape.np_, tape.n_ineq, tape.n_eq) == (7, 14, 3, 0) and len(tape.op) < 1000
    _, spec = lower(ctrl.optimization)
    band = compile_problem(spec.problem)
    assert (band.n_ineq, band.n_eq) == (6, 0)
    for _ in range(3):
        x = rng.normal(size=7)
        quat = rng.normal(size=4)
        p = np.concatenate([rng.uniform(-1.5, 1.5, 7), rng.uniform(-0.5, 0.5, 3), quat / np.linalg.norm(quat)])
        v = tape_ref.forward(tape, x, p)
        assert abs(v[tape.out_cost] - nlp.f(x, p)) <= 1e-11 * abs(nlp.f(x, p))
        assert np.abs(v[tape.out_rows] - nlp.g(x, p)).max() <= 1e-14
        assert np.abs(tape_ref.reverse(tape, v, {tape.out_cost: 1.0}) - nlp.df(x, p)).max() <= 1e-10 * np.abs(nlp.df(x, p)).max()
        A, b, _, _ = nlp.pieces(p)
        w = tape_ref.forward(band, x, p)
        d, half = A @ x - b, np.sqrt(nlp.bounds)
        assert np.abs(w[band.out_rows] - np.stack([d + half, half - d], axis=1).reshape(-1)).max() <= 1e-14


def test_matmul_with_a_scalar_operand_is_a_scaling():
    b, x, y, a = _builder()
    e = x.T @ 0.5 @ a  # casadi.mtimes: a scalar operand multiplies elementwise (torque_control_example.py:69 'Rc.T @ dt @ dp[:3]')
    b.add_cost_term("c", e * e)
    opt = b.build()
    xv, p = np.array([1.0, 2.0, 3.0, 0.0, 0.0]), np.array([0.5, -1.0, 2.0])
    assert abs(opt.f(xv, p) - (0.5 * xv[:3] @ p) ** 2) < 1e-14


def test_getrotm_follows_the_reference_formula():
    from oracle.problems import TorqueControlNLP

    rng = np.random.default_rng(SEED + 2)
    for _ in range(5):
        q = rng.normal(size=4)
        assert np.allclose(Quaternion(*q).getrotm(), TorqueControlNLP.getrotm(q), rtol=0, atol=1e-15)  # same terms, summed in another order
    # where the reference's entries coincide with the textbook matrix: rotations about y and the identity
    assert np.allclose(Quaternion(0.0, 1.0, 0.0, 0.0).getrotm(), np.diag([-1.0, 1.0, -1.0]))
    assert np.allclose(Quaternion(0.0, 0.0, 0.0, 1.0).getrotm(), np.eye(3))
    # symbolic components: entries of a parameter block
    b = OptimizationBuilder(1)
    x = b.add_decision_variables("x", 3)
    pg = b.add_parameter("pg", 4)
    R = Quaternion(pg[0], pg[1], pg[2], pg[3]).getrotm()
    assert R.shape == (3, 3) and R.degree() == 0
    b.add_cost_term("c", sumsqr(R @ x - np.ones((3, 1))))
    opt = b.build()
    q, xv = rng.normal(size=4), rng.normal(size=3)
    assert abs(opt.f(xv, q) - np.sum((TorqueControlNLP.getrotm(q) @ xv - 1.0) ** 2)) <= 1e-12


def test_band_rows_are_lowered_to_a_dense_qp():
    b, x, y, a = _builder()
    e = a.T @ x + 2.0 * y[0] - 0.25  # affine in (x, y)
    b.add_cost_term("c", sumsqr(x - a) + 3.0 * sumsqr(y))
    b.add_leq_inequality_constraint("band", e * e, 1e-4)
    d = x - y[1]
    b.add_leq_inequality_constraint("vec", d[0] * d[0], 4e-6)
    b.add_leq_inequality_constraint("lin", x[2], 5.0)
    opt = b.build()
    assert isinstance(opt, QuadraticCostNonlinearConstraints) and (opt.nk, opt.ng) == (1, 2)
    kind, spec = lower(opt)
    assert kind == optas_amd._lib.OH_PROBLEM_QP and isinstance(spec, QpSpec) and spec.bands == ("band", "vec")
    qp = spec.problem
    assert isinstance(qp, QuadraticCostLinearConstraints) and (qp.nx, qp.nk, qp.na) == (5, 5, 0) and (spec.n, spec.m, spec.me) == (5, 5, 0)
    p = np.array([0.3, -0.2, 0.5])
    M, c = np.asarray(qp.M(p)), np.asarray(qp.c(p)).reshape(-1)
    row = np.concatenate([p, [2.0, 0.0]])
    expect_M = np.array([[0, 0, -1.0, 0, 0], row, -row, [1.0, 0, 0, 0, -1.0], [-1.0, 0, 0, 0, 1.0]])
    expect_c = np.array([5.0, -0.25 + 1e-2, 0.25 + 1e-2, 2e-3, 2e-3])
    assert np.allclose(M, expect_M, atol=1e-14) and np.allclose(c, expect_c, atol=1e-14)
    rng = np.random.default_rng(SEED + 3)
    for _ in range(20):  # the same feasible set
        xv = rng.normal(size=5) * 0.3
        in_band = np.all(np.asarray(opt.g(xv, p)).reshape(-1) >= 0) and np.all(np.asarray(opt.k(xv, p)).reshape(-1) >= 0)
        assert in_band == bool(np.all(M @ xv + c >= 0))
        assert qp.f(xv, p) == opt.f(xv, p)


def test_other_nonlinear_rows_are_not_taken_for_bands():
    b, x, y, a = _builder()
    b.add_cost_term("c", sumsqr(x) + sumsqr(y))
    b.add_leq_inequality_constraint("r", x[0] * x[1], 1.0)  # a product of two different expressions
    with pytest.raises(LoweringError):
        match_qp(b.build())
    b2, x2, y2, a2 = _builder()
    b2.add_cost_term("c", sumsqr(x2) + sumsqr(y2))
    b2.add_geq_inequality_constraint("r", x2[0] * x2[0], 1.0)  # x^2 >= 1: not a band (the feasible set is not convex)
    with pytest.raises(LoweringError):
        match_qp(b2.build())


def _control_nlp():
    from oracle.problems import TorqueControlNLP
    from oracle.robot import OracleRobot

    return TorqueControlNLP(OracleRobot(os.path.join(os.path.dirname(optas_amd.__file__), "robots", "med7.kin.json")))


def test_literal_control_nlp_derivatives_and_exact_solver():
    from oracle.ipm_reference_form import attach_hessian, solve_ipm
    from oracle.problems import band_qp_exact
    from oracle.solvers import kkt_reference_form, scipy_minimize

    nlp = _control_nlp()
    rng = np.random.default_rng(SEED + 4)
    qc = np.deg2rad([0, 30, 0, -90, 0, 60, 0])
    pc = np.asarray(nlp.robot.get_global_link_position(nlp.link, qc)).reshape(3)
    p = np.concatenate([qc, pc + [0.002, -0.001, 0.0015], [0.0, 1.0, 0.0, 0.0]])
    x = rng.normal(size=7) * 0.1
    h = 1e-2  # f and g are quadratic: central differences are exact up to rounding
    g_fd = np.array([(nlp.f(x + h * e, p) - nlp.f(x - h * e, p)) / (2 * h) for e in np.eye(7)])
    assert np.abs(g_fd - nlp.df(x, p)).max() <= 1e-9 * np.abs(g_fd).max()
    J_fd = np.array([(nlp.g(x + h * e, p) - nlp.g(x - h * e, p)) / (2 * h) for e in np.eye(7)]).T
    assert np.abs(J_fd - nlp.dg(x, p)).max() <= 1e-9 * np.abs(J_fd).max()
    H_fd = np.array([(nlp.df(x + h * e, p) - nlp.df(x - h * e, p)) / (2 * h) for e in np.eye(7)])
    assert np.abs(H_fd - nlp.ddf(x, p)).max() <= 1e-9 * np.abs(H_fd).max()
    lam = np.array([1.0, -2.0, 0.5])
    Hg_fd = np.array([(nlp.dg(x + h * e, p).T @ lam - nlp.dg(x - h * e, p).T @ lam) / (2 * h) for e in np.eye(7)])
    assert np.abs(Hg_fd - nlp.ddg_dot(x, p, lam)).max() <= 1e-9 * np.abs(Hg_fd).max()
    # the reference's cost does not depend on the current orientation: diffR = Rg_ee' R = Rg' (I + dt skew(w)) (torque_control_example.py:77-80)
    A, b, C, c0 = nlp.pieces(p)
    assert np.allclose(c0.reshape(3, 3).T, nlp.getrotm(p[10:]).T - np.eye(3), atol=1e-12)
    z = np.zeros(7)
    xs, fs, state, nu = band_qp_exact(nlp.ddf(z, p), nlp.df(z, p), A, b, np.sqrt(nlp.bounds))
    assert state == (1, 1, 1) and abs(0.5 * xs @ nlp.ddf(z, p) @ xs + nlp.df(z, p) @ xs + nlp.f(z, p) - nlp.f(xs, p)) <= 1e-9
    k = kkt_reference_form(nlp, xs, p, active_tol=1e-13)
    assert k["stationarity"] <= 1e-10 and k["feasibility"] <= 1e-20 and (k["lam"] >= 0).all()
    s = scipy_minimize(nlp, z, p, method="SLSQP", tol=1e-14)
    assert np.abs(s["x"] - xs).max() <= 1e-6
    # the interior-point oracle relaxes every bound by 1e-8 (IPOPT's bound_relax_factor), which doubles the 1e-8 bands of this problem:
    # it lands on the minimiser of the relaxed bands, 2 % away -- the rows' scale, not a different basin
    r = solve_ipm(attach_hessian(nlp), z, p)
    assert r["status"] == "optimal" and np.abs(r["x"] - xs).max() <= 0.05 and np.all(nlp.g(r["x"], p) >= -1.01e-8)
    xr, _, _, _ = band_qp_exact(nlp.ddf(z, p), nlp.df(z, p), A, b, np.sqrt(nlp.bounds + 1e-8))
    assert np.abs(r["x"] - xr).max() <= 1e-4


def test_vector_rows_with_per_entry_bounds_and_the_square_node():
    b = OptimizationBuilder(1)
    x = b.add_decision_variables("x", 3)
    a = b.add_parameter("a", 3)
    d = x - a
    b.add_cost_term("c", sumsqr(x))
    b.add_leq_inequality_constraint("v", d * d, np.array([1e-2, 4e-2, 9e-2]))  # three rows, one bound each
    b.add_leq_inequality_constraint("w", d**2, 0.25)  # the Square node, one bound for all rows
    _, spec = lower(b.build())
    qp, p = spec.problem, np.array([1.0, 2.0, 3.0])
    assert spec.bands == ("v", "w") and qp.nk == 12
    I = np.eye(3)
    assert np.allclose(np.asarray(qp.M(p)), np.vstack([I, -I, I, -I]), atol=1e-15)
    half = np.array([0.1, 0.2, 0.3])
    assert np.allclose(np.asarray(qp.c(p)).reshape(-1), np.concatenate([-p + half, p + half, -p + 0.5, p + 0.5]), atol=1e-15)


def test_symbolic_quaternion_refuses_what_is_not_lowered():
    b = OptimizationBuilder(1)
    b.add_decision_variables("x", 1)
    pg = b.add_parameter("pg", 4)
    q = Quaternion(pg[0], pg[1], pg[2], pg[3])
    assert len(q.split()) == 4
    for call in (q.getquat, q.sumsqr, q.inv, q.getrpy, lambda: q * Quaternion(0.0, 0.0, 0.0, 1.0)):
        with pytest.raises(NotImplementedError):
            call()
