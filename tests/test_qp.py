"""Dense QP family and the reference's only numeric solver test (tests/test_solver.py:22-54): the Booth
function, (a, b) = (2, 7) -> (x, y) = (1, 3), with and without the huge bound rows of its `constraint=True` variant.
CPU: numpy port (oracle/qp_ipm.py) vs scipy and the known answer, builder/lowering.  GPU: the same through HIPSolver."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import minimize

import optas_amd
from conftest import SEED
from optas_amd.builder import OptimizationBuilder
from optas_amd.lowering import QpSpec, lower
from optas_amd.optimization import QuadraticCostLinearConstraints, QuadraticCostUnconstrained
from oracle.qp_ipm import solve_qp_ipm


def booth_builder(constraint=False):
    """tests/test_solver.py:22-38, the same builder calls."""
    builder = OptimizationBuilder(1)
    x = builder.add_decision_variables("x")
    y = builder.add_decision_variables("y")
    a = builder.add_parameter("a")
    b = builder.add_parameter("b")
    f = (x + a * y - b) ** 2 + (2.0 * x + y - 5.0) ** 2
    builder.add_cost_term("f", f)
    if constraint:
        builder.add_bound_inequality_constraint("bnd", -1e9, x, 1e9)
    return builder


def random_qp(rng, n, m, me):
    G = rng.normal(size=(n, n))
    P = 0.5 * (G @ G.T) + 0.1 * np.eye(n)
    q = rng.normal(size=n)
    xf = rng.normal(size=n)  # a strictly feasible point exists
    M = rng.normal(size=(m, n))
    c = -M @ xf + rng.uniform(0.1, 1.0, m)
    A = rng.normal(size=(me, n))
    b = -A @ xf
    return P, q, M, c, A, b


def test_booth_builder_classes_and_matrices():
    o = booth_builder().build()
    assert isinstance(o, QuadraticCostUnconstrained) and (o.nx, o.np, o.nv) == (2, 2, 0)
    p = np.array([2.0, 7.0])
    # f = (x + 2y - 7)^2 + (2x + y - 5)^2 = x^T P x + q^T x + 74 with P = [[5, 4], [4, 5]], q = (-34, -38)
    assert np.allclose(o.P(p), [[5.0, 4.0], [4.0, 5.0]]) and np.allclose(o.q(p), [-34.0, -38.0]) and abs(o.f(np.zeros(2), p) - 74.0) < 1e-12
    assert abs(o.f(np.array([1.0, 3.0]), p)) < 1e-12
    kind, spec = lower(o)
    assert kind == optas_amd._lib.OH_PROBLEM_QP and isinstance(spec, QpSpec) and (spec.n, spec.m, spec.me) == (2, 0, 0)
    oc = booth_builder(constraint=True).build()
    assert isinstance(oc, QuadraticCostLinearConstraints) and (oc.nk, oc.nv) == (2, 2)
    assert np.allclose(oc.M(p), [[1.0, 0.0], [-1.0, 0.0]]) and np.allclose(oc.c(p), [1e9, 1e9])


def test_port_booth_known_answer_and_random_qps():
    for constraint in (False, True):
        o = booth_builder(constraint).build()
        p = np.array([2.0, 7.0])
        M, c = (o.M(p), o.c(p)) if o.nk else (np.zeros((0, 2)), np.zeros(0))
        r = solve_qp_ipm(o.P(p), o.q(p), M, c, np.zeros((0, 2)), np.zeros(0), x0=np.zeros(2))
        assert r["status"] == 0 and np.isclose(r["x"], [1.0, 3.0]).all()  # np.isclose defaults, as the reference asserts
    rng = np.random.default_rng(SEED)
    for n, m, me in ((3, 0, 0), (5, 8, 0), (6, 10, 2), (12, 30, 4), (7, 0, 3)):
        P, q, M, c, A, b = random_qp(rng, n, m, me)
        r = solve_qp_ipm(P, q, M, c, A, b)
        assert r["status"] == 0 and r["iters"] <= 40
        cons = []
        if m:
            cons.append({"type": "ineq", "fun": lambda x: M @ x + c, "jac": lambda x: M})
        if me:
            cons.append({"type": "eq", "fun": lambda x: A @ x + b, "jac": lambda x: A})
        s = minimize(lambda x: x @ P @ x + q @ x, np.zeros(n), jac=lambda x: 2 * P @ x + q, method="SLSQP", constraints=cons, tol=1e-13,
                     options={"maxiter": 500})
        assert s.success and abs(s.fun - r["f"]) < 1e-7 and np.abs(s.x - r["x"]).max() < 1e-5
        assert (r["lam"] >= 0).all() and np.abs(2 * P @ r["x"] + q - M.T @ r["lam"] - A.T @ r["nu"]).max() < 1e-8


@pytest.mark.gpu
def test_reference_solver_test_through_hipsolver(hip_lib):
    """tests/test_solver.py:42-54 with HIPSolver in place of CasADiSolver / ScipyMinimizeSolver / OSQPSolver / CVXOPTSolver."""
    from optas_amd.solver import HIPSolver

    for constraint in (False, True):
        solver = HIPSolver(booth_builder(constraint).build()).setup("hip_sqp")
        solver.reset_initial_seed({"x": 0, "y": 0})
        solver.reset_parameters({"a": 2.0, "b": 7.0})
        result = solver.solve()
        assert np.isclose(np.asarray(result["x"]).flatten(), 1.0).all() and np.isclose(np.asarray(result["y"]).flatten(), 3.0).all()
        assert solver.did_solve() and abs(solver.stats()["f"][0]) < 1e-9
    # batch: every (a, b) has the closed-form minimiser of the 2 x 2 linear system
    solver = HIPSolver(booth_builder().build()).setup("hip_sqp")
    rng = np.random.default_rng(SEED)
    ab = rng.uniform(0.5, 8.0, (64, 2))
    ab[:, 0] = np.where(np.abs(ab[:, 0] - 0.5) < 0.05, 1.0, ab[:, 0])  # a = 1/2 makes the two rows parallel
    solver.reset_parameters_batch({"a": ab[:, 0], "b": ab[:, 1]})
    sols = solver.solve_batch()
    for (a, b), sol in zip(ab, sols):
        xy = np.linalg.solve(np.array([[1.0, a], [2.0, 1.0]]), np.array([b, 5.0]))
        assert abs(np.asarray(sol["x"]).item() - xy[0]) < 1e-7 and abs(np.asarray(sol["y"]).item() - xy[1]) < 1e-7


@pytest.mark.gpu
def test_dense_qp_kernel_matches_port(hip_lib):
    from optas_amd.backend import QPBackend

    rng = np.random.default_rng(SEED + 1)
    for n, m, me in ((3, 0, 0), (6, 10, 2), (12, 30, 4), (32, 64, 8)):
        be = QPBackend(n, m, me)
        qps = [random_qp(rng, n, m, me) for _ in range(70)]
        r = be.solve(np.zeros((70, n)), np.stack([QPBackend.pack(*qp) for qp in qps]))
        lam, nu = be.multipliers(70)
        assert (r.status == 0).all()
        for i in (0, 17, 69):
            P, q, M, c, A, b = qps[i]
            s = solve_qp_ipm(P, q, M, c, A, b)
            assert int(r.iters[i]) == s["iters"] and abs(r.f[i] - s["f"]) < 1e-9 * max(1.0, abs(s["f"])) and np.abs(r.x[i] - s["x"]).max() < 1e-8
            assert np.abs(lam[i] - s["lam"]).max(initial=0.0) < 1e-6 and np.abs(nu[i] - s["nu"]).max(initial=0.0) < 1e-6
            assert (M @ r.x[i] + c >= -1e-9).all() and (np.abs(A @ r.x[i] + b) <= 1e-9).all()
        # a few instances take the wavefront-per-instance kernel (lanes over rows / matrix entries): same iteration, sums associated differently
        rows = np.stack([QPBackend.pack(*qp) for qp in qps[:40]])
        rw = be.solve(np.zeros((40, n)), rows)
        lam_w, nu_w = be.multipliers(40)
        assert (rw.status == 0).all() and (rw.iters == r.iters[:40]).all()
        assert np.abs(rw.x - r.x[:40]).max() < 1e-9 and np.abs(rw.f - r.f[:40]).max() < 1e-9 * max(1.0, np.abs(r.f).max())
        assert np.abs(lam_w - lam[:40]).max(initial=0.0) < 1e-7 and np.abs(nu_w - nu[:40]).max(initial=0.0) < 1e-7
        be.close()


def test_differential_ik_is_a_quadratic_program():
    """Builder classification of the velocity-IK problem (example/experiment1.py's type): quadratic cost, linear rows only."""
    from examples.differential_ik import DifferentialIK

    ik = DifferentialIK(build_only=True)
    o = ik.optimization
    assert isinstance(o, QuadraticCostLinearConstraints) and (o.nx, o.np, o.nk, o.na, o.nv) == (7, 7, 16, 0, 16)
    assert list(o.decision_variables.keys()) == ["kuka/dq/x"]
    kind, spec = lower(o)
    assert kind == optas_amd._lib.OH_PROBLEM_QP and (spec.n, spec.m, spec.me) == (7, 16, 0)


@pytest.mark.gpu
def test_differential_ik_through_the_qp_family(hip_lib):
    """The QP's data comes from forward kinematics / the geometric Jacobian at qc (evaluated on the GPU through oh_fk_jac, memoised per
    parameter vector); the solution is checked against scipy SLSQP on the same data and against the task it encodes."""
    from conftest import KUKA_KIN
    from examples.differential_ik import DifferentialIK
    from oracle.robot import OracleRobot

    ik = DifferentialIK(height_band=(0.0, 2.0))
    o = ik.optimization
    kuka = OracleRobot(KUKA_KIN)
    qc = np.deg2rad([0, 30, 0, -90, 0, 60, 0])
    P, q, M, c = o.P(qc), o.q(qc), o.M(qc), o.c(qc)
    J = kuka.get_global_link_geometric_jacobian("end_effector_ball", qc)
    vg = np.array([0.1, 0, 0, 0, 0, 0])
    assert np.abs(P - (np.eye(7) + 1000.0 * J.T @ J)).max() < 1e-9 and np.abs(q + 2000.0 * J.T @ vg).max() < 1e-9  # f = x^T P x + q^T x + const
    assert np.abs(M[:7] - 0.1 * np.eye(7)).max() < 1e-15 and np.abs(M[14] - 0.1 * J[2]).max() < 1e-12
    dq, qn = ik.step(qc)
    assert ik.solver.did_solve()
    s = minimize(lambda x: x @ P @ x + q @ x, np.zeros(7), jac=lambda x: 2 * P @ x + q, method="SLSQP",
                 constraints=[{"type": "ineq", "fun": lambda x: M @ x + c, "jac": lambda x: M}], tol=1e-14, options={"maxiter": 300})
    assert s.success and np.abs(dq - s.x).max() < 1e-6
    assert np.abs(J @ dq - vg).max() < 2e-3 and (M @ dq + c >= -1e-9).all()  # the end-effector moves along +x at 0.1 m/s, nothing else moves
    # an active height band: the step may not lift the end-effector above its current height
    z = kuka.get_global_link_position("end_effector_ball", qc)[2]
    ik2 = DifferentialIK(planar_direction=(0.0, 0.0), height_band=(0.0, z - 0.002))
    dq2, _ = ik2.step(qc)
    assert ik2.solver.did_solve() and abs(z + 0.1 * (J @ dq2)[2] - (z - 0.002)) < 1e-8  # pushed down exactly onto the band


def test_planar_idk_is_a_quadratic_program_with_equalities():
    from examples.planar_idk import setup_solver

    robot, o = setup_solver(build_only=True)
    assert isinstance(o, QuadraticCostLinearConstraints) and (robot.ndof, o.nx, o.np, o.nk, o.na, o.nv) == (3, 3, 3, 8, 2, 12)
    kind, spec = lower(o)
    assert kind == optas_amd._lib.OH_PROBLEM_QP and (spec.n, spec.m, spec.me) == (3, 8, 2)


@pytest.mark.gpu
def test_planar_idk_known_answer(hip_lib):
    """example/planar_idk.py: with the bounds inactive the minimum-norm solution of J_xy dq = dx is pinv(J_xy) dx -- the comparison the
    reference script prints (:58)."""
    from examples.planar_idk import setup_solver

    robot, solver = setup_solver()
    q_t = np.array([2.39, -2.55, -0.46])
    solver.reset_initial_seed({"planar_3dof/dq/x": [0.0, 0.0, 0.0]} if robot.get_name() == "planar_3dof" else {f"{robot.get_name()}/dq/x": [0.0, 0.0, 0.0]})
    solver.reset_parameters({"q": q_t})
    sol = solver.solve()
    assert solver.did_solve()
    dq = np.asarray(sol[f"{robot.get_name()}/dq"]).reshape(-1)
    Jxy = np.asarray(robot.get_global_link_linear_jacobian("end", q_t))[0:2]
    expect = np.linalg.pinv(Jxy) @ np.array([0.01, 0.0])
    o = solver.opt
    assert np.abs(o.a(dq, q_t)).max() < 1e-9 and o.k(dq, q_t).min() > -1e-9  # the equality holds, the rows are feasible
    if np.abs(expect).max() < 0.1 and o.k(expect, q_t).min() > 1e-6:  # bounds inactive at the pinv solution: it is the answer
        assert np.abs(dq - expect).max() < 1e-7
    else:  # otherwise at least no worse than any feasible scaling of it
        assert dq @ dq >= expect @ expect - 1e-9


@pytest.mark.gpu
def test_qp_data_assembled_on_the_device_equals_the_host_route(hip_lib):
    """oh_qp_set_tape: the kernel reads [P | q | M | c | A | b] off the problem's instruction tape with the probes the host route applies to the
    numeric members (optimization.py:219-260 are cs.Functions of p in the reference; solver.py:453-467 evaluates them before a solve).
    Same rows to rounding, same solutions; p of a solve is the problem's parameter vector."""
    from examples.differential_ik import DifferentialIK
    from optas_amd.backend import QPBackend
    from optas_amd.solver import HIPSolver
    from optas_amd.tape import compile_problem

    ik = DifferentialIK(height_band=(0.02, 1.18), build_only=True)  # the upper band is active for part of the batch
    o = ik.optimization
    rng = np.random.default_rng(SEED + 5)
    B = 96
    qc = np.deg2rad([0, 30, 0, -90, 0, 60, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
    dev = HIPSolver(o).setup("hip_sqp")
    host = HIPSolver(o).setup("hip_sqp", {"device_assembly": False})
    assert dev.backend.be.tape is not None and host.backend.be.tape is None
    x0 = np.zeros((B, o.nx))
    rd, rh = dev.solve_batch_arrays(x0, qc), host.solve_batch_arrays(x0, qc)
    assert (rd.status == 0).all() and (rh.status == 0).all()
    assert np.abs(rd.x - rh.x).max() < 1e-9 and np.abs(rd.f - rh.f).max() < 1e-9 * np.abs(rh.f).max()
    lam_d, lam_h = dev.backend.be.multipliers(B)[0], host.backend.be.multipliers(B)[0]
    assert np.abs(lam_d - lam_h).max() < 1e-6 * max(1.0, np.abs(lam_h).max()) and (lam_h > 1e-6).any()
    # the reference-form objective of the returned point, constant term included
    for b in (0, 17, 95):
        assert abs(rd.f[b] - o.f(rd.x[b], qc[b])) < 1e-9 * max(1.0, abs(rd.f[b]))
    # ABI: sizes must match the handle; only QP handles take a tape
    tape = compile_problem(o)
    wrong = QPBackend(o.nx, o.nk + 1, 0)
    with pytest.raises(optas_amd._lib.OptasHipError, match="differ from the handle"):
        wrong.set_tape(tape)
    wrong.close()
    from optas_amd.backend import PointMassBackend, TapeBackend

    pm = PointMassBackend()
    td = TapeBackend.descriptor(tape)
    assert hip_lib.oh_qp_set_tape(pm._h, C.byref(td)) == optas_amd._lib.OH_ERR_STATE
    assert hip_lib.oh_qp_set_tape(None, C.byref(td)) == optas_amd._lib.OH_ERR_INVALID
    pm.close()
    dev.backend.close()
    host.backend.close()


@pytest.mark.gpu
def test_parameter_free_qp_with_device_assembly(hip_lib):
    """A QuadraticCost problem without parameters: the tape has no parameter loads, the ABI still takes one (unused) column."""
    from optas_amd.solver import HIPSolver

    builder = OptimizationBuilder(1)
    x = builder.add_decision_variables("x")
    y = builder.add_decision_variables("y")
    builder.add_cost_term("f", (x - 1.0) ** 2 + (y - 2.0) ** 2)
    builder.add_leq_inequality_constraint("sum", x + y, 2.0)
    opt = builder.build()
    assert opt.np == 0
    solver = HIPSolver(opt).setup("hip_sqp")
    assert solver.backend.be.tape is not None
    sol = solver.solve()
    assert solver.did_solve()
    assert abs(np.asarray(sol["x"]).item() - 0.5) < 1e-7 and abs(np.asarray(sol["y"]).item() - 1.5) < 1e-7  # projection of (1, 2) onto x + y = 2
    assert abs(solver.stats()["f"][0] - 0.5) < 1e-8
