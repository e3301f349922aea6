"""Problem IR handed to a Solver: the reference's seven ``Optimization`` classes
(optas/optimization.py:54-568) with the same attribute names for sizes and containers.  The
``cs.Function`` members (f, df, ddf, k, a, g, h, v, dv, ...) do not exist here -- evaluation happens
inside the lowered HIP kernels -- but the row counts follow the reference exactly:
``nv = nk + ng + 2*na + 2*nh`` for ``v = [k; g; a; -a; h; -h]`` (:27-51,292-306).
"""
from __future__ import annotations

from typing import List, Optional

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


class QuadraticCostUnconstrained(Optimization):
    pass


class QuadraticCostLinearConstraints(Optimization):
    def __init__(self, decision_variables, parameters, cost_terms, lin_eq_constraints, lin_ineq_constraints):
        super().__init__(decision_variables, parameters, cost_terms)
        self.specify_linear_constraints(lin_ineq_constraints, lin_eq_constraints)
        self.specify_v()


class QuadraticCostNonlinearConstraints(Optimization):
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
