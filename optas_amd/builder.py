"""``OptimizationBuilder`` with the reference's method names, argument meaning, block naming and
constraint routing (optas/builder.py), recording ``optas_amd.expr`` nodes instead of CasADi SX.

Layout rules that the solver relies on (and the tests pin against the reference's own expectations,
tests/test_builder.py:25-37,258-419):
  * decision blocks ``"{name}/{d*'d'}{symbol}/x"`` of shape dim x (T-d) (or x T with derivs_align),
    created model by model, derivative by derivative (builder.py:90-99); robot parameter blocks
    ``".../p"`` likewise (possibly 0 rows);
  * every constraint is stored as ``rhs - lhs`` (builder.py:313,354); ``add_bound`` makes ``_l``/``_r``
    blocks (:334-335); linear ones go to the ``lin_*`` containers (:314-317,357-360);
  * ``build()`` picks the Optimization subclass like builder.py:545-635.
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np

from .expr import Const, Expr, ParamRef, RobotStates, Scale, StateRef, Sub, SumSqr, VarRef, as_expr
from .models import Model, RobotModel, TaskModel
from .optimization import (
    MixedIntegerNonlinearCostNonlinearConstrained,
    NonlinearCostLinearConstraints,
    NonlinearCostNonlinearConstraints,
    NonlinearCostUnconstrained,
    Optimization,
    QuadraticCostLinearConstraints,
    QuadraticCostNonlinearConstraints,
    QuadraticCostUnconstrained,
)
from .sx_container import SXContainer


class IntegrationResidual(Expr):
    """x_t + dt_t * xd_t - x_{t+1} for t = 0..n-1 (builder.py:429-438), dim x n."""

    def __init__(self, x: StateRef, xd: StateRef, dt: np.ndarray, n: int):
        self.x, self.xd, self.dt, self.n = x, xd, np.asarray(dt, dtype=np.float64).reshape(-1), n
        self.shape = (x.m, n)

    def degree(self):
        return 1


class OptimizationBuilder:
    """Same constructor arguments, block names and layout as the reference's builder (optas/builder.py:14-99); the bookkeeping is this
    package's own: a name -> model table and one routine that lays a model's blocks down."""

    def __init__(self, T: int, robots: List[RobotModel] = [], tasks: List[TaskModel] = [], derivs_align: bool = False):
        self.T = int(T)
        self.derivs_align = bool(derivs_align)
        models = (robots if isinstance(robots, list) else [robots]) + (tasks if isinstance(tasks, list) else [tasks])
        self._by_name = {}
        for model in models:
            assert model.get_name() not in self._by_name, f"two models are called '{model.get_name()}': names identify the blocks of x and p"
            self._by_name[model.get_name()] = model
        self._models = models
        assert self.T >= 1, "the horizon needs at least one knot"
        if not self.derivs_align:
            # derivative d of a model lives on T - d knots: the highest one must keep at least one
            deepest = max((d for model in models for d in model.time_derivs), default=0)
            assert self.T > deepest, f"a horizon of {self.T} knots leaves none for the time derivative of order {deepest}"
        self._decision_variables, self._parameters, self._cost_terms = SXContainer(), SXContainer(), SXContainer()
        self._lin_eq_constraints, self._lin_ineq_constraints = SXContainer(), SXContainer()
        self._eq_constraints, self._ineq_constraints = SXContainer(), SXContainer()
        for model in models:
            self._lay_down(model)

    def _lay_down(self, model: Model) -> None:
        """x (and, for robots, p) blocks of one model, derivative by derivative: "{name}/{d x 'd'}{symbol}/x" with dim rows and T - d columns (T with
        derivs_align); a robot's parameterised joints get the matching ".../p" block, empty when every joint is optimised."""
        robot = isinstance(model, RobotModel)
        for order in model.time_derivs:
            knots = self.T if self.derivs_align else self.T - order
            label = model.state_optimized_name(order)
            self._decision_variables[label] = StateRef(label, model.get_name(), order, model.num_opt_joints if robot else model.dim, knots)
            if robot:
                self.add_parameter(model.state_parameter_name(order), model.num_param_joints, knots)
            elif model.is_discrete:
                self._decision_variables.variable_is_discrete(label)

    # ---- models ------------------------------------------------------------------------------------------
    def get_model_names(self) -> List[str]:
        return list(self._by_name)

    def get_model_index(self, name: str) -> int:
        return self.get_model_names().index(name)

    def get_model(self, name: str) -> Model:
        if name not in self._by_name:
            raise ValueError(f"'{name}' is not in list")  # what list.index raises in the reference (builder.py:107-118)
        return self._by_name[name]

    def _block(self, container: SXContainer, name: str, order: int, label_of):
        model = self.get_model(name)
        assert order in model.time_derivs, f"'{name}' carries the time derivatives {list(model.time_derivs)}, not order {order}"
        return container[label_of(model)(order)]

    def get_model_states(self, name: str, time_deriv: int = 0) -> StateRef:
        return self._block(self._decision_variables, name, time_deriv, lambda m: m.state_optimized_name)

    def get_model_state(self, name: str, t: int, time_deriv: int = 0) -> StateRef:
        return self.get_model_states(name, time_deriv)[:, t]

    def get_model_parameters(self, name: str, time_deriv: int = 0) -> ParamRef:
        return self._block(self._parameters, name, time_deriv, lambda m: m.state_parameter_name)

    def get_model_parameter(self, name: str, t: int, time_deriv: int = 0):
        """builder.py:165-176: column t of the parameter block of a model's (parameterised joints') states."""
        return self.get_model_parameters(name, time_deriv)[:, t]

    # ---- variables / parameters / terms ------------------------------------------------------------------
    def get_robot_states_and_parameters(self, name: str, time_deriv: int = 0):
        """builder.py:178-204: dim x n array with the optimised rows from the decision variables and the parameterised rows from
        the parameters "{name}/{d}q/p"."""
        model = self.get_model(name)
        assert isinstance(model, RobotModel), "this method only applies to robot models"
        states = self.get_model_states(name, time_deriv=time_deriv)
        if model.num_param_joints == 0:
            return states
        return RobotStates(states, self.get_model_parameters(name, time_deriv=time_deriv), tuple(model.optimized_joint_indexes),
                           tuple(model.parameter_joint_indexes))

    def add_decision_variables(self, name: str, m: int = 1, n: int = 1, is_discrete: bool = False) -> VarRef:
        x = VarRef(name, m, n)
        self._decision_variables[name] = x
        if is_discrete:
            self._decision_variables.variable_is_discrete(name)
        return x

    def add_parameter(self, name: str, m: int = 1, n: int = 1) -> ParamRef:
        p = ParamRef(name, m, n)
        self._parameters[name] = p
        return p

    def add_cost_term(self, name: str, cost_term) -> None:
        cost_term = as_expr(cost_term)
        m, n = cost_term.shape
        assert m == 1 and n == 1, "cost term must be scalar"
        self._cost_terms[name] = cost_term

    def _is_linear_in_x(self, y: Expr) -> bool:
        return y.degree() <= 1

    def is_cost_quadratic(self) -> bool:
        return all(c.degree() <= 2 for c in self._cost_terms.values())

    def add_geq_inequality_constraint(self, name: str, lhs, rhs=None) -> None:
        lhs = as_expr(lhs)
        if rhs is None:
            rhs = Const(np.zeros(lhs.shape))
        self.add_leq_inequality_constraint(name, rhs, lhs)

    def add_leq_inequality_constraint(self, name: str, lhs, rhs=None) -> None:
        lhs = as_expr(lhs)
        rhs = Const(np.zeros(lhs.shape)) if rhs is None else as_expr(rhs)
        diff = Sub(rhs, lhs)  # diff >= 0
        if self._is_linear_in_x(diff):
            self._lin_ineq_constraints[name] = diff
        else:
            self._ineq_constraints[name] = diff

    def add_bound_inequality_constraint(self, name: str, lhs, mid, rhs) -> None:
        self.add_leq_inequality_constraint(name + "_l", lhs, mid)
        self.add_leq_inequality_constraint(name + "_r", mid, rhs)

    def add_equality_constraint(self, name: str, lhs, rhs=None, reduce_constraint: bool = False) -> None:
        lhs = as_expr(lhs)
        rhs = Const(np.zeros(lhs.shape)) if rhs is None else as_expr(rhs)
        diff = Sub(rhs, lhs)  # diff == 0
        if reduce_constraint:
            diff = SumSqr(diff)
        if self._is_linear_in_x(diff):
            self._lin_eq_constraints[name] = diff
        else:
            self._eq_constraints[name] = diff

    # ---- common constraints ----------------------------------------------------------------------------
    def sphere_collision_avoidance_constraints(self, name: str, obstacle_names, link_names=None, base_link=None, *, link_radii_prefix: str = "") -> None:
        """builder.py:366-417: for every knot t, link l and obstacle j the row ||p_l(q_t) - o_j||^2 - (r_l + r_j)^2 >= 0;
        creates the parameters "{link}_radii", "{obs}_position" (3) and "{obs}_radii" in that order.  ``link_radii_prefix``
        is an additive extension: the reference names the link-radius parameters after the bare link name, so two robots
        built from the same URDF collide on them (KeyError, sx_container.py:50-51); a prefix lifts that."""
        model = self.get_model(name)
        assert isinstance(model, RobotModel), "this method only applies to robot models"
        if base_link is not None and base_link != model.get_root_link():
            raise NotImplementedError("sphere constraints are lowered in the root frame only")
        Q = self.get_model_states(name)
        n = Q.shape[1]
        if link_names is None:
            link_names = model.link_names
        assert len(link_names), "at least one link should be named"
        links = {}
        for ln in link_names:
            links[ln] = self.add_parameter(link_radii_prefix + ln + "_radii")
        assert len(obstacle_names), "at least one obstacle should be named"
        obstacles = {}
        for on in obstacle_names:
            obstacles[on] = (self.add_parameter(on + "_position", 3), self.add_parameter(on + "_radii"))
        for t in range(n):
            q = Q[:, t]
            for ln, linkrad in links.items():
                p = model.get_global_link_position(ln, q)
                for on, (obs, obsrad) in obstacles.items():
                    dist2 = SumSqr(Sub(p, obs))
                    bnd2 = (linkrad + obsrad) ** 2
                    self.add_leq_inequality_constraint(f"sphere_col_avoid_{t}_{ln}_{on}", bnd2, dist2)

    def integrate_model_states(self, name: str, time_deriv: int, dt) -> None:
        n = self.T - (1 if self.derivs_align else time_deriv)
        dt = np.asarray(dt, dtype=np.float64).reshape(-1)
        if dt.shape[0] == 1:
            dt = dt[0] * np.ones(n)
        assert dt.shape[0] == n, f"The array for dt has an incorrect length, expected {n}, got {dt.shape[0]}"
        xd = self.get_model_states(name, time_deriv)
        x = self.get_model_states(name, time_deriv - 1)
        cname = f"__integrate_model_states_{name}_{time_deriv}__"
        self.add_equality_constraint(cname, IntegrationResidual(x, xd, dt, n))

    def enforce_model_limits(self, name: str, time_deriv: int = 0, lo=None, up=None, safe_frac=1.0) -> None:
        assert 0.0 < safe_frac <= 1.0, f"Given safe_frac '{safe_frac}' must be in range (0, 1]."
        x = self.get_model_states(name, time_deriv)
        xlo, xup = lo, up
        if (xlo is None) or (xup is None):
            mlo, mup = self.get_model(name).get_limits(time_deriv)
            xlo = mlo if xlo is None else xlo
            xup = mup if xup is None else xup
        xlo, xup = np.asarray(xlo, dtype=np.float64).reshape(-1), np.asarray(xup, dtype=np.float64).reshape(-1)
        if safe_frac < 1.0:
            mid, diff = 0.5 * (xlo + xup), xup - xlo
            xlo, xup = mid - 0.5 * safe_frac * diff, mid + 0.5 * safe_frac * diff
        n = f"__{name}_model_limit_{time_deriv}__"
        self.add_bound_inequality_constraint(n, Const(xlo), x, Const(xup))

    def initial_configuration(self, name: str, init=None, time_deriv: int = 0) -> None:
        x0 = self.get_model_state(name, 0, time_deriv=time_deriv)
        self.add_equality_constraint(f"__{name}_initial_configuration_{time_deriv}__", lhs=x0, rhs=init)

    def fix_configuration(self, name: str, config=None, time_deriv: int = 0, t: int = 0) -> None:
        x0 = self.get_model_state(name, t, time_deriv=time_deriv)
        self.add_equality_constraint(f"__{name}_fix_configuration_{time_deriv}_{t}__", lhs=x0, rhs=config)

    # ---- build -------------------------------------------------------------------------------------------
    def build(self) -> Optimization:
        nlin = self._lin_ineq_constraints.numel() + self._lin_eq_constraints.numel()
        nnlin = self._ineq_constraints.numel() + self._eq_constraints.numel()
        dv, pa, ct = self._decision_variables, self._parameters, self._cost_terms
        le, li, eq, iq = self._lin_eq_constraints, self._lin_ineq_constraints, self._eq_constraints, self._ineq_constraints
        if dv.has_discrete_variables():
            opt = MixedIntegerNonlinearCostNonlinearConstrained(dv, pa, ct, le, li, eq, iq)
        elif self.is_cost_quadratic():
            if nnlin > 0:
                opt = QuadraticCostNonlinearConstraints(dv, pa, ct, le, li, eq, iq)
            elif nlin > 0:
                opt = QuadraticCostLinearConstraints(dv, pa, ct, le, li)
            else:
                opt = QuadraticCostUnconstrained(dv, pa, ct)
        else:
            if nnlin > 0:
                opt = NonlinearCostNonlinearConstraints(dv, pa, ct, le, li, eq, iq)
            elif nlin > 0:
                opt = NonlinearCostLinearConstraints(dv, pa, ct, le, li)
            else:
                opt = NonlinearCostUnconstrained(dv, pa, ct)
        opt.set_models(self._models)
        return opt
