"""Structural lowering: recognise the task-term pattern of an ``Optimization`` and bind it to the
hand-written HIP kernel family that evaluates it.

The reference hands IPOPT generic CasADi tapes (optas/solver.py:346-363); the north star replaces that
with kernels written per problem *family*.  Families lowered so far:

  figure-eight  (example/figure_eight_plan.py:16-113)  -> OH_PROBLEM_FIGURE_EIGHT

Anything that does not match raises ``NotImplementedError`` naming the first term that failed to
match -- there is no generic/CPU evaluation path to fall back to.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .builder import IntegrationResidual
from .expr import Const, LinkFunction, ParamRef, PathInFrame, Scale, StateRef, Sub, SumSqr
from .models import RobotModel
from .optimization import Optimization


class LoweringError(NotImplementedError):
    pass


@dataclass
class FigureEightSpec:
    robot: RobotModel
    link: str
    T: int
    dt: float
    w_path: float
    w_vel: float
    local_path: np.ndarray  # (T, 3)
    qc_name: str
    q_name: str
    dq_name: str


def _unscale(e):
    w = 1.0
    while isinstance(e, Scale):
        w *= e.w
        e = e.a
    return w, e


def _is_zero_const(e) -> bool:
    return isinstance(e, Const) and not np.any(e.value)


def match_figure_eight(opt: Optimization) -> FigureEightSpec:
    def no(msg):
        raise LoweringError(f"figure-eight lowering: {msg}")

    robots = [m for m in (opt.models or []) if isinstance(m, RobotModel)]
    if len(opt.models or []) != 1 or len(robots) != 1:
        no("expected exactly one RobotModel and no task models")
    robot = robots[0]
    if list(robot.time_derivs) != [0, 1] or robot.num_param_joints != 0:
        no("robot must have time_derivs=[0, 1] and no parameterised joints")
    name = robot.get_name()
    q_name, dq_name = robot.state_optimized_name(0), robot.state_optimized_name(1)
    if list(opt.decision_variables.keys()) != [q_name, dq_name]:
        no(f"decision variables must be exactly [{q_name}, {dq_name}]")
    Q: StateRef = opt.decision_variables[q_name]
    dQ: StateRef = opt.decision_variables[dq_name]
    T = Q.n
    if dQ.n != T - 1:
        no("derivs_align=True is not lowered")
    if opt.nk or opt.ng:
        no("inequality rows are not lowered for this family yet")

    # linear equalities: fix q_0 = qc, fix dq_0 = 0, Euler integration
    qc: Optional[ParamRef] = None
    dt = None
    seen = set()
    for label, diff in opt.lin_eq_constraints.items():
        if not isinstance(diff, Sub):
            no(f"linear equality '{label}' not recognised")
        rhs, lhs = diff.a, diff.b
        if isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 0 and isinstance(rhs, ParamRef):
            qc = rhs
            seen.add("fix_q")
        elif isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 1 and _is_zero_const(rhs):
            seen.add("fix_dq")
        elif isinstance(lhs, IntegrationResidual) and _is_zero_const(rhs) and lhs.xd.time_deriv == 1:
            if not np.allclose(lhs.dt, lhs.dt[0], rtol=0, atol=0):
                no("non-uniform dt is not lowered")
            dt = float(lhs.dt[0])
            seen.add("integr")
        else:
            no(f"linear equality '{label}' not recognised")
    if seen != {"fix_q", "fix_dq", "integr"}:
        no(f"need fix_configuration(q, qc), fix_configuration(dq) and integrate_model_states; found {sorted(seen)}")
    if qc.shape != (robot.ndof, 1):
        no("qc must be an ndof-vector parameter")
    params = [k for k, v in opt.parameters.items() if v.numel() > 0]
    if params != [qc.name]:
        no(f"the only non-empty parameter must be '{qc.name}', found {params}")

    # nonlinear equality: quat(Q) == quat(qc)
    if len(opt.eq_constraints) != 1:
        no("expected exactly one nonlinear equality (end-effector quaternion lock)")
    (label, diff), = opt.eq_constraints.items()
    ok = (
        isinstance(diff, Sub)
        and isinstance(diff.a, LinkFunction)
        and isinstance(diff.b, LinkFunction)
        and diff.a.what == diff.b.what == "quaternion"
        and diff.a.q is qc
        and diff.b.q is Q
        and diff.a.link == diff.b.link
        and diff.a.robot is robot
        and diff.b.robot is robot
    )
    if not ok:
        no(f"equality '{label}' is not quat(link, Q) == quat(link, qc)")
    link = diff.a.link

    # costs
    w_path = w_vel = None
    local = None
    for label, term in opt.cost_terms.items():
        w, e = _unscale(term)
        if not isinstance(e, SumSqr):
            no(f"cost '{label}' is not a weighted sumsqr")
        inner = e.a
        if inner is dQ:
            w_vel = w
        elif isinstance(inner, Sub) and isinstance(inner.a, PathInFrame) and isinstance(inner.b, LinkFunction):
            pth, pos = inner.a, inner.b
            good = (
                pos.what == "position" and pos.q is Q and pos.link == link and pos.robot is robot
                and isinstance(pth.origin, LinkFunction) and pth.origin.what == "position" and pth.origin.q is qc and pth.origin.link == link
                and isinstance(pth.rotation, LinkFunction) and pth.rotation.what == "rotation" and pth.rotation.q is qc and pth.rotation.link == link
                and pth.local.shape == (3, T)
            )
            if not good:
                no(f"cost '{label}' is not sumsqr(path_in_frame(p(qc), R(qc), local) - p(Q))")
            w_path, local = w, pth.local
        else:
            no(f"cost '{label}' not recognised")
    if w_path is None or w_vel is None:
        no("need both the path-tracking and the joint-velocity cost terms")
    return FigureEightSpec(robot, link, T, dt, w_path, w_vel, np.ascontiguousarray(local.T), qc.name, q_name, dq_name)


def lower(opt: Optimization):
    """Return (kind, spec).  Raises LoweringError if no kernel family matches."""
    errors = []
    for kind, fn in ((_lib.OH_PROBLEM_FIGURE_EIGHT, match_figure_eight),):
        try:
            return kind, fn(opt)
        except LoweringError as e:
            errors.append(str(e))
    raise LoweringError("no HIP kernel family matches this optimization problem: " + "; ".join(errors))
