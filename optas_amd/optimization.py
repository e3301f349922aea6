"""Problem IR handed to a Solver: the reference's seven ``Optimization`` classes
(optas/optimization.py:54-568) with the same attribute names for sizes, bounds, containers and functions.

The solve itself happens inside the lowered HIP kernels; the function members ``f, k, a, g, h, v`` and -- where the reference
defines them -- ``M, c, A, b`` (``k = Mx + c``, ``a = Ax + b``, optimization.py:225-260) and ``P, q``
(``f = x^T P x + q^T x``, optimization.py:219-223) are numeric callables over the expression trees (optas_amd.evaluate;
link functions inside them run through liboptas_hip), used for diagnostics and for checking the builder's layout and sign
conventions against the oracle.  The derivative members the reference derives with CasADi (optimization.py:8-24) are here too:
``df, dk, da, dg, dh, dv`` are exact (forward propagation through the trees with the geometric Jacobian of oh_fk_jac and the
dual-number Jacobian of oh_rnea_jac, optas_amd.evaluate.jacobian); ``ddf, ddg, ddh, ddv`` are exact as well (optas_amd.evaluate.weighted_hessian:
second-order kinematics from the same geometric Jacobian), except rows through inverse dynamics, which are central differences of their
exact first derivatives (step 1e-6, symmetrised) -- the kernels carry their own second-order terms and never call them.
``v = [k; g; a; -a; h; -h]``, ``nv = nk + ng + 2 na + 2 nh``, bounds ``0 <= v <= 1e10`` (optimization.py:27-51,292-306).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .sx_container import SXContainer


class Optimization:
    inf = 1.0e10  # optimization.py:58

    def __init__(self, decision_variables: SXContainer, parameters: SXContainer, cost_terms: SXContainer):
        self.models: Optional[List] = None
        self.decision_variables = decision_variables
        self.parameters = parameters
        self.cost_terms = cost_terms
        self.lin_eq_constraints = SXContainer()
        self.lin_ineq_constraints = SXContainer()
        self.eq_constraints = SXContainer()
        self.ineq_constraints = SXContainer()
        self.nk = self.na = self.ng = self.nh = self.nv = 0
        self.nx = decision_variables.numel()
        self.np = parameters.numel()

    def set_models(self, models) -> None:
        self.models = models

    # ---- numeric function members (cs.Function objects in the reference) --------------------------------------------
    def _vec(self, container, x, p) -> np.ndarray:
        """container.vec() (sx_container.py:83-89) evaluated at (x, p): items in order, each column-major."""
        from .evaluate import evaluate

        x, p = np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(p, dtype=np.float64).reshape(-1)
        assert x.shape[0] == self.nx and p.shape[0] == self.np, f"expected x ({self.nx}) and p ({self.np})"
        parts = []
        for term in container.values():
            m, n = term.shape
            parts.append(np.broadcast_to(np.asarray(evaluate(term, self, x, p), dtype=np.float64), (m, n)).T.reshape(-1))
        return np.concatenate(parts) if parts else np.zeros(0)

    def f(self, x, p) -> float:
        """Sum of the cost terms (optimization.py:192-195)."""
        return float(np.sum(self._vec(self.cost_terms, x, p)))

    def k(self, x, p) -> np.ndarray:
        return self._vec(self.lin_ineq_constraints, x, p)

    def a(self, x, p) -> np.ndarray:
        return self._vec(self.lin_eq_constraints, x, p)

    def g(self, x, p) -> np.ndarray:
        return self._vec(self.ineq_constraints, x, p)

    def h(self, x, p) -> np.ndarray:
        return self._vec(self.eq_constraints, x, p)

    def v(self, x, p) -> np.ndarray:
        """vertcon (optimization.py:27-51): [k; g; a; -a; h; -h] >= 0."""
        a, h = self.a(x, p), self.h(x, p)
        return np.concatenate([self.k(x, p), self.g(x, p), a, -a, h, -h])

    # ---- derivative members (optimization.py:8-24, 198, 276-306) ------------------------------------------------------------------
    def _jac(self, container, x, p) -> np.ndarray:
        from .evaluate import jacobian

        x, p = np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(p, dtype=np.float64).reshape(-1)
        parts = [jacobian(term, self, x, p)[1] for term in container.values()]
        return np.concatenate(parts, axis=0) if parts else np.zeros((0, self.nx))

    def df(self, x, p) -> np.ndarray:
        """Gradient of f as a 1 x nx row (casadi.jacobian(f, x))."""
        return np.sum(self._jac(self.cost_terms, x, p), axis=0, keepdims=True)

    def dk(self, x, p) -> np.ndarray:
        return self._jac(self.lin_ineq_constraints, x, p)

    def da(self, x, p) -> np.ndarray:
        return self._jac(self.lin_eq_constraints, x, p)

    def dg(self, x, p) -> np.ndarray:
        return self._jac(self.ineq_constraints, x, p)

    def dh(self, x, p) -> np.ndarray:
        return self._jac(self.eq_constraints, x, p)

    def dv(self, x, p) -> np.ndarray:
        da, dh = self.da(x, p), self.dh(x, p)
        return np.concatenate([self.dk(x, p), self.dg(x, p), da, -da, dh, -dh], axis=0)

    def _second(self, first, x, p, step=1e-6) -> np.ndarray:
        """d/dx of a first-derivative member by central differences: (rows, nx, nx), symmetrised in the last two axes."""
        x = np.asarray(x, dtype=np.float64).reshape(-1).copy()
        J0 = first(x, p)
        H = np.zeros(J0.shape + (self.nx,))
        for i in range(self.nx):
            xi = x[i]
            x[i] = xi + step
            Jp = first(x, p)
            x[i] = xi - step
            Jm = first(x, p)
            x[i] = xi
            H[..., i] = (Jp - Jm) / (2.0 * step)
        return 0.5 * (H + np.swapaxes(H, -1, -2))

    def _hess_rows(self, container, x, p) -> np.ndarray:
        """(rows, nx, nx): exact Hessian of every entry of the container's terms (evaluate.weighted_hessian: the weights travel down the
        expression trees, second-order kinematics from the geometric Jacobian) -- what casadi.jacobian(casadi.jacobian(.)) gives the
        reference (optimization.py:8-24).  NotImplementedError where a node has no rule (inverse-dynamics rows)."""
        from .evaluate import weighted_hessian

        x, p = np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(p, dtype=np.float64).reshape(-1)
        out = []
        for term in container.values():
            m, n = term.shape
            for c in range(n):  # column-major, like vec()
                for r in range(m):
                    W = np.zeros((m, n))
                    W[r, c] = 1.0
                    out.append(weighted_hessian(term, self, x, p, W))
        return np.array(out).reshape(len(out), self.nx, self.nx)

    def ddf(self, x, p) -> np.ndarray:
        """Hessian of f, exact (round 3; central differences of df only where a cost term has a node without a second-derivative rule)."""
        from .evaluate import weighted_hessian

        xx, pp = np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(p, dtype=np.float64).reshape(-1)
        try:
            return sum((weighted_hessian(term, self, xx, pp, np.ones((1, 1))) for term in self.cost_terms.values()), np.zeros((self.nx, self.nx)))
        except NotImplementedError:
            return self._second(self.df, x, p)[0]

    def ddg(self, x, p) -> np.ndarray:
        try:
            return self._hess_rows(self.ineq_constraints, x, p)
        except NotImplementedError:
            return self._second(self.dg, x, p)

    def ddh(self, x, p) -> np.ndarray:
        try:
            return self._hess_rows(self.eq_constraints, x, p)
        except NotImplementedError:
            return self._second(self.dh, x, p)

    def ddv(self, x, p) -> np.ndarray:
        """Hessians of the rows of v = [k; g; a; -a; h; -h]: zero for the linear blocks."""
        ddg, ddh = self.ddg(x, p), self.ddh(x, p)
        zk, za = np.zeros((self.nk, self.nx, self.nx)), np.zeros((self.na, self.nx, self.nx))
        return np.concatenate([zk, ddg, za, za, ddh, -ddh], axis=0)

    def _affine(self, fun, p):
        """(matrix, offset) of an affine map x -> fun(x, p): offset = fun(0), column i = fun(e_i) - offset (exact)."""
        z = np.zeros(self.nx)
        off = fun(z, p)
        Mx = np.zeros((off.shape[0], self.nx))
        for i in range(self.nx):
            z[i] = 1.0
            Mx[:, i] = fun(z, p) - off
            z[i] = 0.0
        return Mx, off

    def M(self, p) -> np.ndarray:
        return self._affine(self.k, p)[0]

    def c(self, p) -> np.ndarray:
        return self.k(np.zeros(self.nx), p)

    def A(self, p) -> np.ndarray:
        return self._affine(self.a, p)[0]

    def b(self, p) -> np.ndarray:
        return self.a(np.zeros(self.nx), p)

    @property
    def lbk(self):
        return np.zeros(self.nk)

    @property
    def ubk(self):
        return self.inf * np.ones(self.nk)

    @property
    def lba(self):
        return np.zeros(self.na)

    @property
    def uba(self):
        return np.zeros(self.na)

    @property
    def lbg(self):
        return np.zeros(self.ng)

    @property
    def ubg(self):
        return self.inf * np.ones(self.ng)

    @property
    def lbh(self):
        return np.zeros(self.nh)

    @property
    def ubh(self):
        return np.zeros(self.nh)

    @property
    def lbv(self):
        return np.zeros(self.nv)

    @property
    def ubv(self):
        return self.inf * np.ones(self.nv)

    def specify_linear_constraints(self, lin_ineq_constraints, lin_eq_constraints) -> None:
        self.lin_ineq_constraints = lin_ineq_constraints
        self.lin_eq_constraints = lin_eq_constraints
        self.nk = lin_ineq_constraints.numel()
        self.na = lin_eq_constraints.numel()

    def specify_nonlinear_constraints(self, ineq_constraints, eq_constraints) -> None:
        self.ineq_constraints = ineq_constraints
        self.eq_constraints = eq_constraints
        self.ng = ineq_constraints.numel()
        self.nh = eq_constraints.numel()

    def specify_v(self) -> None:
        self.nv = self.nk + self.ng + 2 * self.na + 2 * self.nh

    def has_discrete_variables(self) -> bool:
        return self.decision_variables.has_discrete_variables()


class _QuadraticCost:
    """specify_quadratic_cost (optimization.py:219-223): P = 1/2 ddf, q = df(0); exact differences of the quadratic f."""

    def q(self, p) -> np.ndarray:
        e = np.zeros(self.nx)
        out = np.zeros(self.nx)
        for i in range(self.nx):
            e[i] = 1.0
            fp = self.f(e, p)
            e[i] = -1.0
            fm = self.f(e, p)
            e[i] = 0.0
            out[i] = 0.5 * (fp - fm)
        return out

    def P(self, p) -> np.ndarray:
        n = self.nx
        z = np.zeros(n)
        f0 = self.f(z, p)
        f1 = np.zeros(n)
        for i in range(n):
            z[i] = 1.0
            f1[i] = self.f(z, p)
            z[i] = 0.0
        q = self.q(p)
        Pm = np.zeros((n, n))
        for i in range(n):
            Pm[i, i] = f1[i] - f0 - q[i]
            for j in range(i):
                z[i] = z[j] = 1.0
                Pm[i, j] = Pm[j, i] = 0.5 * (self.f(z, p) - f1[i] - f1[j] + f0)
                z[i] = z[j] = 0.0
        return Pm


class QuadraticCostUnconstrained(_QuadraticCost, Optimization):
    pass


class QuadraticCostLinearConstraints(_QuadraticCost, Optimization):
    def __init__(self, decision_variables, parameters, cost_terms, lin_eq_constraints, lin_ineq_constraints):
        super().__init__(decision_variables, parameters, cost_terms)
        self.specify_linear_constraints(lin_ineq_constraints, lin_eq_constraints)
        self.specify_v()


class QuadraticCostNonlinearConstraints(_QuadraticCost, Optimization):
    def __init__(self, decision_variables, parameters, cost_terms, lin_eq_constraints, lin_ineq_constraints, eq_constraints, ineq_constraints):
        super().__init__(decision_variables, parameters, cost_terms)
        self.specify_linear_constraints(lin_ineq_constraints, lin_eq_constraints)
        self.specify_nonlinear_constraints(ineq_constraints, eq_constraints)
        self.specify_v()


class NonlinearCostUnconstrained(Optimization):
    pass


class NonlinearCostLinearConstraints(Optimization):
    def __init__(self, decision_variables, parameters, cost_terms, lin_eq_constraints, lin_ineq_constraints):
        super().__init__(decision_variables, parameters, cost_terms)
        self.specify_linear_constraints(lin_ineq_constraints, lin_eq_constraints)
        self.specify_v()


class NonlinearCostNonlinearConstraints(Optimization):
    def __init__(self, decision_variables, parameters, cost_terms, lin_eq_constraints, lin_ineq_constraints, eq_constraints, ineq_constraints):
        super().__init__(decision_variables, parameters, cost_terms)
        self.specify_linear_constraints(lin_ineq_constraints, lin_eq_constraints)
        self.specify_nonlinear_constraints(ineq_constraints, eq_constraints)
        self.specify_v()


class MixedIntegerNonlinearCostNonlinearConstrained(NonlinearCostNonlinearConstraints):
    pass
