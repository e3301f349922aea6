"""CasADi SX tape -> kernel tape, and the literal ``optas.solver.Solver`` subclass.

Where the reference itself is installed (casadi + optas), a maintainer does not need the ``optas_amd`` front-end at all: the problem's
``cs.Function`` members ``f, k, g, a, h`` (optimization.py:95-160, built in builder.py:885-1040) are SX virtual-machine programs, and
this module walks their instructions (``n_instructions / instruction_id / instruction_input / instruction_output /
instruction_constant``: the public introspection API of ``casadi.Function``) into the same instruction tape the GPU evaluates
(``optas_amd.tape.Tape`` -> ``oh_create_tape``).  ``make_solver_class(optas.solver)`` then returns a subclass of the reference's own
``Solver`` whose ``_solve`` runs ``oh_solve`` -- used exactly like ``CasADiSolver`` / ``ScipyMinimizeSolver`` (solver.py:317-398,558-742).

casadi is not importable in the build image, so nothing here imports it: the caller passes the module (``cs=casadi``), and the tests
drive the walker with a stand-in object that implements the five introspection methods over a hand-written instruction list.
Nothing is evaluated on the CPU: the walk only re-encodes instructions.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .tape import MAX_TAPE, Tape, TapeBuilder, UnsupportedInstruction  # noqa: F401


# (defined next to the builder that raises it too: optas_amd/tape.py)


# casadi opcode names -> how the tape builder expresses them (casadi/core/calculus.hpp enumerates the names; the integer values are read
# from the module at run time because they have shifted between casadi releases)
_UNARY = {
    "OP_ASSIGN": lambda tb, a: a,
    "OP_NEG": lambda tb, a: tb.neg(a),
    "OP_SIN": lambda tb, a: tb.sin(a),
    "OP_COS": lambda tb, a: tb.cos(a),
    "OP_SQRT": lambda tb, a: tb.sqrt(a),
    "OP_SQ": lambda tb, a: tb.sqr(a),
    "OP_TWICE": lambda tb, a: tb.add(a, a),
    "OP_INV": lambda tb, a: tb.div(tb.const(1.0), a),
    "OP_TAN": lambda tb, a: tb.tan(a),
    # round 5: user costs and constraints written with `from casadi import *` (optas/__init__.py:2)
    "OP_EXP": lambda tb, a: tb.exp(a),
    "OP_LOG": lambda tb, a: tb.log(a),
    "OP_ACOS": lambda tb, a: tb.acos(a),
    "OP_ATAN": lambda tb, a: tb.atan(a),
    "OP_TANH": lambda tb, a: tb.tanh(a),
    "OP_SINH": lambda tb, a: tb.sinh(a),
    "OP_COSH": lambda tb, a: tb.cosh(a),
    "OP_ASINH": lambda tb, a: tb.asinh(a),
    "OP_ACOSH": lambda tb, a: tb.acosh(a),
    "OP_ATANH": lambda tb, a: tb.atanh(a),
    "OP_LOG1P": lambda tb, a: tb.log1p(a),
    "OP_EXPM1": lambda tb, a: tb.expm1(a),
    "OP_SIGN": lambda tb, a: tb.sign(a),
    "OP_ASIN": lambda tb, a: tb.asin(a),  # Quaternion.getrpy (spatialmath.py:384-404) -> get_global_link_rpy and the analytical Jacobians
    "OP_FABS": lambda tb, a: tb.fabs(a),
    "OP_NOT": lambda tb, a: tb.lnot(a),
}
_BINARY = {
    "OP_ADD": lambda tb, a, b: tb.add(a, b),
    "OP_SUB": lambda tb, a, b: tb.sub(a, b),
    "OP_MUL": lambda tb, a, b: tb.mul(a, b),
    "OP_DIV": lambda tb, a, b: tb.div(a, b),
    "OP_ATAN2": lambda tb, a, b: tb.atan2(a, b),
    "OP_FMIN": lambda tb, a, b: tb.fmin(a, b),  # optas.clip (__init__.py:29-41)
    "OP_FMAX": lambda tb, a, b: tb.fmax(a, b),
    "OP_LT": lambda tb, a, b: tb.lt(a, b),
    "OP_LE": lambda tb, a, b: tb.le(a, b),
    "OP_EQ": lambda tb, a, b: tb.eq(a, b),
    "OP_NE": lambda tb, a, b: tb.ne(a, b),
    "OP_AND": lambda tb, a, b: tb.land(a, b),
    "OP_OR": lambda tb, a, b: tb.lor(a, b),
    "OP_IF_ELSE_ZERO": lambda tb, a, b: tb.ifz(a, b),  # cs.if_else(c, x, y) = if_else_zero(c, x) + if_else_zero(!c, y) in an SX graph
}


def _pow(tb: TapeBuilder, a: int, b: int) -> int:
    """OP_POW / OP_CONSTPOW: TapeBuilder.pow (small integer and half exponents exactly, anything else as exp(y log x), x > 0)."""
    return tb.pow(a, b)


def opcode_table(cs) -> Dict[int, tuple]:
    """{integer opcode of this casadi build: (kind, handler)}."""
    table = {}
    for name, fn in _UNARY.items():
        if hasattr(cs, name):
            table[int(getattr(cs, name))] = ("unary", fn)
    for name, fn in _BINARY.items():
        if hasattr(cs, name):
            table[int(getattr(cs, name))] = ("binary", fn)
    for name in ("OP_POW", "OP_CONSTPOW"):
        if hasattr(cs, name):
            table[int(getattr(cs, name))] = ("binary", _pow)
    for name, kind in (("OP_CONST", "const"), ("OP_INPUT", "input"), ("OP_OUTPUT", "output")):
        table[int(getattr(cs, name))] = (kind, None)
    return table


def walk(tb: TapeBuilder, fn, args: Sequence[Sequence[int]], cs, table: Optional[dict] = None) -> List[List[int]]:
    """Re-encode one SX ``Function`` into ``tb``.  ``args[i][j]`` is the register holding the j-th nonzero of input i (the inputs of the
    reference's functions are the dense column vectors x and p, optimization.py:12-24, so j is the element index).  Returns, per output,
    the registers of its dense column-major entries (structural zeros become the constant 0)."""
    table = table or opcode_table(cs)
    if hasattr(fn, "is_a") and not fn.is_a("SXFunction"):
        raise UnsupportedInstruction("only SX functions expose a scalar instruction list (expand() an MX function first)")
    work: Dict[int, int] = {}
    outs: List[Dict[int, int]] = [dict() for _ in range(fn.n_out())]
    for k in range(fn.n_instructions()):
        op = int(fn.instruction_id(k))
        kind, handler = table.get(op, (None, None))
        o, i = list(fn.instruction_output(k)), list(fn.instruction_input(k))
        if kind == "const":
            work[o[0]] = tb.const(float(fn.instruction_constant(k)))
        elif kind == "input":
            work[o[0]] = int(args[i[0]][i[1]])
        elif kind == "output":
            outs[o[0]][o[1]] = work[i[0]]
        elif kind == "unary":
            work[o[0]] = handler(tb, work[i[0]])
        elif kind == "binary":
            work[o[0]] = handler(tb, work[i[0]], work[i[1]])
        else:
            raise UnsupportedInstruction(f"casadi instruction {op} (instruction {k} of {fn.name()}) has no tape counterpart")
    zero = None
    dense: List[List[int]] = []
    for j, nz in enumerate(outs):
        sp = fn.sparsity_out(j)
        m, n = int(sp.size1()), int(sp.size2())
        rows, cols = list(sp.row()), list(sp.get_col())
        regs = [None] * (m * n)
        for e, r in nz.items():
            regs[rows[e] + cols[e] * m] = r
        for e in range(m * n):
            if regs[e] is None:
                zero = tb.const(0.0) if zero is None else zero
                regs[e] = zero
        dense.append(regs)
    return dense


def tape_from_functions(cs, nx: int, np_: int, f, ineq: Sequence = (), eq: Sequence = ()) -> Tape:
    """One tape for the cost ``f(x, p)`` and the rows of the ``ineq`` functions (>= 0) then the ``eq`` functions (= 0)."""
    tb = TapeBuilder()
    args = [[tb.x(k) for k in range(nx)], [tb.p(k) for k in range(np_)]]
    table = opcode_table(cs)
    cost = walk(tb, f, args, cs, table)[0]
    if len(cost) != 1:
        raise UnsupportedInstruction("the cost function must be scalar")
    rows_i = [r for g in ineq if g is not None for r in walk(tb, g, args, cs, table)[0]]
    rows_e = [r for h in eq if h is not None for r in walk(tb, h, args, cs, table)[0]]
    if len(tb.op) > MAX_TAPE:
        raise UnsupportedInstruction(f"tape of {len(tb.op)} instructions exceeds {MAX_TAPE}")
    return Tape(np.asarray(tb.op, dtype=np.int32), np.asarray(tb.a, dtype=np.int32), np.asarray(tb.b, dtype=np.int32), np.asarray(tb.c, dtype=np.float64),
                int(cost[0]), np.asarray(rows_i + rows_e, dtype=np.int32), len(rows_i), len(rows_e), int(nx), int(np_))


def tape_from_optimization(opt, cs) -> Tape:
    """The reference's ``Optimization`` object (any of its seven classes): f, then k and g (>= 0), then a and h (= 0) -- the rows of
    ``v`` (optimization.py:27-51) without the mirrored equality rows."""
    if opt.has_discrete_variables():
        raise UnsupportedInstruction("discrete variables are not supported")
    return tape_from_functions(cs, opt.nx, opt.np, opt.f, ineq=(opt.k, opt.g), eq=(opt.a, opt.h))


def make_solver_class(solver_module, cs):
    """``HIPSolver(optas.solver.Solver)``: the literal drop-in.  ``solver_module`` is the imported ``optas.solver``; ``cs`` is casadi.

        HIPSolver = make_solver_class(optas.solver, casadi)
        solver = HIPSolver(builder.build()).setup("hip_sqp", {"tol": 1e-6})
        solver.reset_initial_seed({...}); solver.reset_parameters({...}); solution = solver.solve()
    """
    from . import _lib
    from .backend import tape_backend, tape_default_max_iter

    class HIPSolver(solver_module.Solver):
        def setup(self, solver_name: str = "hip_sqp", solver_options: Optional[dict] = None):
            """Structured family first (optas_amd.probe_lowering: labels and shapes of the problem's containers, its own numeric functions
            probed and then verified -- the headline figure-eight kernels behind a real ``Optimization``), generic tape family otherwise.
            ``solver_options["family"]`` = "figure_eight" | "torque_mpc" | "ik" | "point_mass" | "multi_arm" | "qp" | "tape" forces one route; ``"link"`` may name
            the tracked link.  The QuadraticCost* classes without nonlinear rows go to the dense-QP family."""
            if solver_name != "hip_sqp":
                raise ValueError(f"unknown solver '{solver_name}' (this interface provides 'hip_sqp')")
            o = dict(solver_options or {})
            family = o.pop("family", None)
            link = o.pop("link", None)
            self._family = None
            if family not in ("tape", "qp"):
                from .backend import IKBackend, MultiArmBackend, PointMassBackend, TorqueBackend
                from .lowering import LoweringError
                from . import probe_lowering as pl
                from .solver import figure_eight_backend

                probes = pl.PROBES
                if family is not None and family not in probes:
                    raise ValueError(f"unknown family '{family}' ({', '.join(probes)}, tape)")
                try:
                    fam, spec = (family, probes[family](self.opt, link=link)) if family else pl.probe(self.opt, link=link)
                    if fam == "figure_eight":
                        hess = {"gauss_newton": 0, "exact": 1, "hybrid": 2}[o.pop("hessian", "hybrid")]
                        self._backend = figure_eight_backend(spec, o, hess)
                        if spec.lead is not None:  # one parameterised joint ahead of the chain: the reference's parameter vector re-packed per solve
                            from .solver import _LeadAdapter

                            self._backend = _LeadAdapter(self.opt, spec, self._backend)
                    elif fam == "torque_mpc":
                        self._backend = TorqueBackend(spec.robot.solver_chain(spec.link), spec.robot.dynamics_tables(), T=spec.T, dt=spec.dt, w_path=spec.w_path,
                                                      w_vel=spec.w_vel, w_tau=spec.w_tau, tau_lo=spec.tau_lo, tau_up=spec.tau_up, dq_lo=getattr(spec, "dq_lo", None), dq_up=getattr(spec, "dq_up", None),
                                                      max_iter=int(o.pop("max_iter", 300)), tol=float(o.pop("tol", 1e-6)), tol_compl=float(o.pop("tol_compl", 1e-8)),
                                                      mu_barrier0=float(o.pop("mu_barrier0", 0.0)), mu0=float(o.pop("mu0", 0.0)))
                    elif fam == "point_mass":
                        pl = spec.planner  # example/point_mass_planner.py: final-knot tracking, velocity cost, final velocity fixed, constant obstacle
                        self._backend = PointMassBackend(spec.T, spec.dt, spec.w_acc, spec.ylim, spec.vlim, spec.safe, max_iter=int(o.pop("max_iter", 200 if pl else 100)),
                                                         tol=float(o.pop("tol", 1e-8)), track_final_only=pl is not None, w_vel=pl["w_vel"] if pl else 0.0,
                                                         fix_final_velocity=pl is not None)
                        if pl is not None:
                            from .solver import _PlannerAdapter

                            self._backend = _PlannerAdapter(spec, self._backend)
                    elif fam == "multi_arm":
                        self._backend = MultiArmBackend(spec, self.opt, max_iter=int(o.pop("max_iter", 200)), tol=float(o.pop("tol", 1e-6)))
                    else:
                        self._backend = IKBackend(spec.robot.kinematic_chain(spec.link), spec.lo, spec.up, w_nominal=spec.w_nominal,
                                                  max_iter=int(o.pop("max_iter", 200)), tol=float(o.pop("tol", 1e-6)), tol_feas=float(o.pop("tol_feas", 1e-9)))
                    self._family, self._spec = fam, spec
                except LoweringError:
                    if family is not None:
                        raise
            if self._family is None and family in (None, "qp"):
                # the QuadraticCost* classes without nonlinear rows (what the reference hands to OSQP / CVXOPT / qpOASES, solver.py:421-584):
                # the dense-QP family, with P, q, M, c, A, b read off the walked tape on the device (oh_qp_set_tape)
                from .backend import QPBackend

                is_qp = type(self.opt).__name__ in ("QuadraticCostUnconstrained", "QuadraticCostLinearConstraints")
                nk = int(getattr(self.opt, "nk", 0) or 0)
                na = int(getattr(self.opt, "na", 0) or 0)
                if is_qp and self.opt.nx <= 32 and nk <= 256 and na <= min(32, self.opt.nx) and not self.opt.has_discrete_variables():
                    self._tape = tape_from_functions(cs, self.opt.nx, self.opt.np, self.opt.f, ineq=(getattr(self.opt, "k", None),), eq=(getattr(self.opt, "a", None),))
                    self._backend = QPBackend(self.opt.nx, nk, na, max_iter=int(o.pop("max_iter", 100)), tol=float(o.pop("tol", 1e-9)), tape=self._tape)
                    self._family = "qp"
                elif type(self.opt).__name__ == "QuadraticCostNonlinearConstraints" and not int(getattr(self.opt, "nh", 0) or 0) and self.opt.nx <= 32 \
                        and not self.opt.has_discrete_variables():
                    # quadratic cost whose only curved rows are squares of affine expressions under a constant (example/torque_control_example.py:93-95,
                    # handed to sqpmethod by the reference): bands, i.e. the same QP with two linear rows each (tape.band_rewrite)
                    from .tape import band_rewrite

                    banded = band_rewrite(tape_from_optimization(self.opt, cs))
                    if banded is not None and banded.n_ineq <= 256 and banded.n_eq <= min(32, self.opt.nx):
                        self._tape = banded
                        self._backend = QPBackend(self.opt.nx, banded.n_ineq, banded.n_eq, max_iter=int(o.pop("max_iter", 100)), tol=float(o.pop("tol", 1e-9)), tape=banded)
                        self._family = "qp"
                if self._family is None and family == "qp":
                    raise ValueError("family 'qp' needs a QuadraticCostUnconstrained / QuadraticCostLinearConstraints problem (or one whose nonlinear rows are "
                                     "squares of affine expressions under a constant) with nx <= 32, nk <= 256, na <= 32")
            if self._family is None:
                self._tape = tape_from_optimization(self.opt, cs)
                self._backend = tape_backend(self._tape, eliminate=bool(o.pop("eliminate", True)), max_iter=int(o.pop("max_iter", tape_default_max_iter(self._tape.nx))),
                                             tol=float(o.pop("tol", 1e-6)), tol_feas=float(o.pop("tol_feas", 1e-9)), rho0=(float(o.pop("rho0")) if "rho0" in o else None), jit=bool(o.pop("jit", True)), metric=o.pop("metric", None))
                self._family = "tape"
            if o:
                raise ValueError(f"unknown solver options {sorted(o)}")
            self._stats = None
            return self

        def _solve(self):
            x0 = np.asarray(self.x0, dtype=np.float64).reshape(1, -1)
            p = np.asarray(self.p, dtype=np.float64).reshape(1, -1)
            r = self._backend.solve(x0, p)
            self._stats = {"status": int(r.status[0]), "success": bool(_lib.status_ok(r.status[0])), "iter_count": int(r.iters[0]), "f": float(r.f[0]),
                           "kkt": [float(v) for v in r.kkt[0]], "family": self._family,
                           "solve_ms": self._backend.solve_ms() if hasattr(self._backend, "solve_ms") else self._backend.timing()["solve_ms"]}
            return cs.DM(r.x[0])

        def stats(self):
            return self._stats

        def number_of_iterations(self) -> int:
            return self._stats["iter_count"]

        def did_solve(self) -> bool:
            return self._stats["success"]

    return HIPSolver
