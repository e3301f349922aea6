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
from .expr import Add, Const, LinkFunction, Mul, ParamCol, ParamRef, PathInFrame, RneaFunction, RobotStates, Rows, Scale, Square, StateCols, StateRef, Sub, SumSqr
from .models import RobotModel, TaskModel
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
    lo: Optional[np.ndarray] = None  # joint limits (enforce_model_limits), None: no such rows
    up: Optional[np.ndarray] = None
    spheres: Optional["GuardSpec"] = None  # sphere clearances (sphere_collision_avoidance_constraints)
    vlo: Optional[np.ndarray] = None  # joint-velocity limits (enforce_model_limits(time_deriv=1)), None: no such rows
    vup: Optional[np.ndarray] = None
    lead: Optional[dict] = None  # one parameterised joint ahead of the chain (param_joints): {"par", "opt", "qp", "dqp"}


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
    if list(robot.time_derivs) != [0, 1] or robot.num_param_joints > 1:
        no("robot must have time_derivs=[0, 1] and at most one parameterised joint")
    name = robot.get_name()
    q_name, dq_name = robot.state_optimized_name(0), robot.state_optimized_name(1)
    if list(opt.decision_variables.keys()) != [q_name, dq_name]:
        no(f"decision variables must be exactly [{q_name}, {dq_name}]")
    Q: StateRef = opt.decision_variables[q_name]
    dQ: StateRef = opt.decision_variables[dq_name]
    T = Q.n
    if dQ.n != T - 1:
        no("derivs_align=True is not lowered")
    lead = None
    if robot.num_param_joints == 1:
        lead = {"par": robot.parameter_joint_indexes[0], "opt": list(robot.optimized_joint_indexes), "qp": robot.state_parameter_name(0),
                "dqp": robot.state_parameter_name(1)}

    def is_full(e, X):
        """e is the robot's full trajectory of block X: X itself, or get_robot_states_and_parameters over X."""
        if lead is None:
            return e is X
        return isinstance(e, RobotStates) and e.states is X and list(e.opt_idx) == lead["opt"] and list(e.par_idx) == [lead["par"]]

    sph = {}
    for label, d in opt.ineq_constraints.items():
        ok = (isinstance(d, Sub) and isinstance(d.a, SumSqr) and isinstance(d.a.a, Sub) and isinstance(d.a.a.a, LinkFunction)
              and d.a.a.a.what == "position" and isinstance(d.a.a.b, ParamRef) and d.a.a.b.shape == (3, 1)
              and isinstance(d.b, Square) and isinstance(d.b.a, Add) and isinstance(d.b.a.a, ParamRef) and isinstance(d.b.a.b, ParamRef)
              and isinstance(d.a.a.a.q, StateRef) and d.a.a.a.q.t is not None and d.a.a.a.q.var_name == q_name and d.a.a.a.robot is robot)
        if not ok:
            no(f"inequality '{label}' is not a sphere clearance ||p_link(q_t) - o||^2 >= (r_link + r_o)^2")
        sph[(d.a.a.a.q.t, d.a.a.a.link, d.a.a.b.name)] = (d.b.a.a.name, d.b.a.b.name)
    spheres = None
    if sph:
        links, obst = [], []
        for (t_, ln, on) in sph:
            if ln not in links:
                links.append(ln)
            if on not in obst:
                obst.append(on)
        if len(sph) != T * len(links) * len(obst):
            no("sphere rows must cover every (knot, link, obstacle) combination")
        lrad = {ln: sph[(0, ln, obst[0])][0] for ln in links}
        orad = {on: sph[(0, links[0], on)][1] for on in obst}
        for (t_, ln, on), (lr, orr) in sph.items():
            if lr != lrad[ln] or orr != orad[on]:
                no("inconsistent radius parameters in the sphere rows")
        spheres = GuardSpec(None, None, links, [lrad[ln] for ln in links], [(on, orad[on]) for on in obst])
    lo = up = vlo = vup = None
    n = robot.ndof
    for label, d in opt.lin_ineq_constraints.items():
        if isinstance(d, Sub) and d.a is Q and isinstance(d.b, Const) and d.b.value.shape == (n, 1):
            lo = d.b.value[:, 0] if lo is None else np.maximum(lo, d.b.value[:, 0])
        elif isinstance(d, Sub) and d.b is Q and isinstance(d.a, Const) and d.a.value.shape == (n, 1):
            up = d.a.value[:, 0] if up is None else np.minimum(up, d.a.value[:, 0])
        elif isinstance(d, Sub) and d.a is dQ and isinstance(d.b, Const) and d.b.value.shape == (n, 1):
            vlo = d.b.value[:, 0] if vlo is None else np.maximum(vlo, d.b.value[:, 0])
        elif isinstance(d, Sub) and d.b is dQ and isinstance(d.a, Const) and d.a.value.shape == (n, 1):
            vup = d.a.value[:, 0] if vup is None else np.minimum(vup, d.a.value[:, 0])
        else:
            no(f"linear inequality '{label}' is not a joint-position or joint-velocity bound over the whole trajectory")
    if (lo is None) != (up is None) or (vlo is None) != (vup is None):
        no("limits need both the lower and the upper row block")

    # linear equalities: fix q_0 = qc, fix dq_0 = 0, Euler integration
    qc: Optional[ParamRef] = None
    dt = None
    seen = set()
    for label, diff in opt.lin_eq_constraints.items():
        if not isinstance(diff, Sub):
            no(f"linear equality '{label}' not recognised")
        rhs, lhs = diff.a, diff.b
        if isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 0 and isinstance(rhs, ParamRef) and lead is None:
            qc = rhs
            seen.add("fix_q")
        elif (isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 0 and lead is not None and isinstance(rhs, Rows)
              and isinstance(rhs.a, ParamRef) and list(rhs.idx) == lead["opt"]):
            qc = rhs.a  # initial_configuration(name, robot.extract_optimized_dimensions(qc)), figure_eight_plan_6dof.py:46-49
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
    expect = [qc.name]
    if lead is not None:
        expect = [lead["qp"], lead["dqp"], qc.name]
        if spheres is not None or lo is not None or vlo is not None:
            no("inequality rows together with a parameterised joint are not lowered")
    if spheres is not None:
        expect += list(spheres.link_radii) + [x for ob in spheres.obstacles for x in ob]
    if params != expect:
        no(f"the non-empty parameters must be {expect} in this order (the kernel family reads p = [qc; link radii; obstacles]), found {params}")

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
        and is_full(diff.b.q, Q)
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
        if inner is dQ or is_full(inner, dQ):
            if w_vel is not None:
                no(f"cost '{label}': a second joint-velocity term (weights are not summed or overwritten silently)")
            w_vel = w
        elif isinstance(inner, Sub) and isinstance(inner.a, PathInFrame) and isinstance(inner.b, LinkFunction):
            if w_path is not None:
                no(f"cost '{label}': a second path-tracking term (weights are not summed or overwritten silently)")
            pth, pos = inner.a, inner.b
            good = (
                pos.what == "position" and is_full(pos.q, Q) and pos.link == link and pos.robot is robot
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
    return FigureEightSpec(robot, link, T, dt, w_path, w_vel, np.ascontiguousarray(local.T), qc.name, q_name, dq_name, lo, up, spheres, vlo, vup, lead)


@dataclass
class PointMassSpec:
    T: int
    dt: float
    w_acc: float
    ylim: float
    vlim: float
    safe: float
    names: tuple  # (curr, dcurr, goal, obs) parameter labels
    y_name: str
    dy_name: str
    planner: Optional[dict] = None  # point_mass_planner.py variant: {"w_vel", "obstacle" (2,), "init", "goal"}


def match_point_mass_planner(opt: Optimization) -> PointMassSpec:
    """example/point_mass_planner.py:17-55 (Planner): same plant and rows as the MPC tick, but the tracking cost sits on the last knot
    only, the velocity is penalised, the initial velocity is fixed to zero, the final velocity is an equality row and the obstacle
    is a constant."""

    def no(msg):
        raise LoweringError(f"point-mass planner lowering: {msg}")

    tasks = [m for m in (opt.models or []) if isinstance(m, TaskModel)]
    if len(opt.models or []) != 1 or len(tasks) != 1:
        no("expected exactly one TaskModel")
    tm = tasks[0]
    if tm.dim != 2 or list(tm.time_derivs) != [0, 1]:
        no("task model must be planar (dim 2) with time_derivs=[0, 1]")
    y_name, dy_name = tm.state_optimized_name(0), tm.state_optimized_name(1)
    if list(opt.decision_variables.keys()) != [y_name, dy_name]:
        no("decision variables must be exactly the position and velocity trajectories")
    Y, dY = opt.decision_variables[y_name], opt.decision_variables[dy_name]
    T = Y.n
    if dY.n != T or opt.nh:
        no("needs derivs_align=True and no nonlinear equalities")
    lim = {}
    for label, diff in opt.lin_ineq_constraints.items():
        hi, lo = (diff.a, diff.b) if isinstance(diff, Sub) else (None, None)
        if isinstance(hi, StateRef) and hi.t is None and isinstance(lo, Const) and lo.value.size == 1:
            lim[(hi.time_deriv, "l")] = float(lo.value.reshape(-1)[0])
        elif isinstance(lo, StateRef) and lo.t is None and isinstance(hi, Const) and hi.value.size == 1:
            lim[(lo.time_deriv, "r")] = float(hi.value.reshape(-1)[0])
        else:
            no(f"linear inequality '{label}' is not a scalar box limit on a whole trajectory")
    if set(lim) != {(0, "l"), (0, "r"), (1, "l"), (1, "r")} or lim[(0, "l")] != -lim[(0, "r")] or lim[(1, "l")] != -lim[(1, "r")]:
        no("need symmetric enforce_model_limits for time_deriv 0 and 1")
    init = dt = None
    seen = set()
    for label, diff in opt.lin_eq_constraints.items():
        rhs, lhs = diff.a, diff.b
        if isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 0 and isinstance(rhs, ParamRef):
            init = rhs
        elif isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 1 and _is_zero_const(rhs):
            seen.add("dy0")
        elif isinstance(lhs, StateRef) and lhs.t == T - 1 and lhs.time_deriv == 1 and _is_zero_const(rhs):
            seen.add("dyT")
        elif isinstance(lhs, IntegrationResidual) and _is_zero_const(rhs) and lhs.xd.time_deriv == 1 and np.all(lhs.dt == lhs.dt[0]):
            dt = float(lhs.dt[0])
        else:
            no(f"linear equality '{label}' not recognised")
    if init is None or dt is None or seen != {"dy0", "dyT"}:
        no("need fix_configuration(init), zero initial and final velocity, integrate_model_states")
    if len(opt.ineq_constraints) != T:
        no(f"expected {T} obstacle rows")
    obstacle = safe_sq = None
    for i, (label, diff) in enumerate(opt.ineq_constraints.items()):
        ok = (isinstance(diff, Sub) and isinstance(diff.a, SumSqr) and isinstance(diff.b, Const) and diff.b.value.size == 1
              and isinstance(diff.a.a, Sub) and isinstance(diff.a.a.a, Const) and diff.a.a.a.value.shape == (2, 1)
              and isinstance(diff.a.a.b, StateRef) and diff.a.a.b.t == i and diff.a.a.b.time_deriv == 0)
        if not ok:
            no(f"inequality '{label}' is not ||obstacle - y_{i}||^2 >= const with a constant obstacle")
        o_i, r_i = diff.a.a.a.value[:, 0], float(diff.b.value.reshape(-1)[0])
        if obstacle is None:
            obstacle, safe_sq = o_i, r_i
        elif not np.array_equal(o_i, obstacle) or r_i != safe_sq:
            no("all obstacle rows must use the same obstacle and radius")
    goal = w_vel = w_acc = None
    for label, term in opt.cost_terms.items():
        w, e = _unscale(term)
        if not isinstance(e, SumSqr):
            no(f"cost '{label}' is not a weighted sumsqr")
        inner = e.a
        if (isinstance(inner, Sub) and isinstance(inner.a, ParamRef) and isinstance(inner.b, StateRef) and inner.b.t == T - 1
                and inner.b.time_deriv == 0 and w == 1.0):
            goal = inner.a
        elif inner is dY:
            w_vel = w
        else:
            s2, d = _unscale(inner)
            good = (isinstance(d, Sub) and isinstance(d.a, StateCols) and isinstance(d.b, StateCols) and d.a.state is dY and d.b.state is dY
                    and (d.a.lo, d.a.hi, d.b.lo, d.b.hi) == (1, T, 0, T - 1) and abs(s2 * dt - 1.0) < 1e-12)
            if not good:
                no(f"cost '{label}' not recognised")
            w_acc = w
    if goal is None or w_vel is None or w_acc is None:
        no("need the final-state, velocity and acceleration cost terms")
    if list(opt.parameters.keys()) != [init.name, goal.name] or (init.shape, goal.shape) != ((2, 1), (2, 1)):
        no("parameters must be init(2), goal(2) in this order")
    return PointMassSpec(T, dt, w_acc, lim[(0, "r")], lim[(1, "r")], float(np.sqrt(safe_sq)), (init.name, goal.name), y_name, dy_name,
                         planner={"w_vel": float(w_vel), "obstacle": np.asarray(obstacle, dtype=np.float64), "init": init.name, "goal": goal.name})


def match_point_mass(opt: Optimization) -> PointMassSpec:
    """example/point_mass_mpc.py:88-154 (Controller)."""

    def no(msg):
        raise LoweringError(f"point-mass MPC lowering: {msg}")

    tasks = [m for m in (opt.models or []) if isinstance(m, TaskModel)]
    if len(opt.models or []) != 1 or len(tasks) != 1:
        no("expected exactly one TaskModel")
    tm = tasks[0]
    if tm.dim != 2 or list(tm.time_derivs) != [0, 1]:
        no("task model must be planar (dim 2) with time_derivs=[0, 1]")
    y_name, dy_name = tm.state_optimized_name(0), tm.state_optimized_name(1)
    if list(opt.decision_variables.keys()) != [y_name, dy_name]:
        no("decision variables must be exactly the position and velocity trajectories")
    Y, dY = opt.decision_variables[y_name], opt.decision_variables[dy_name]
    T = Y.n
    if dY.n != T:
        no("needs derivs_align=True")
    if opt.nh:
        no("nonlinear equalities are not part of this family")
    # box limits
    lim = {}
    for label, diff in opt.lin_ineq_constraints.items():
        if not (isinstance(diff, Sub) and diff.numel() == 2 * T):
            no(f"linear inequality '{label}' not recognised")
        hi, lo = diff.a, diff.b  # stored as rhs - lhs
        if isinstance(hi, StateRef) and isinstance(lo, Const) and lo.value.size == 1:
            lim[(hi.time_deriv, "l")] = float(lo.value.reshape(-1)[0])
        elif isinstance(lo, StateRef) and isinstance(hi, Const) and hi.value.size == 1:
            lim[(lo.time_deriv, "r")] = float(hi.value.reshape(-1)[0])
        else:
            no(f"linear inequality '{label}' is not a scalar box limit on a whole trajectory")
    if set(lim) != {(0, "l"), (0, "r"), (1, "l"), (1, "r")}:
        no("need enforce_model_limits for time_deriv 0 and 1")
    if lim[(0, "l")] != -lim[(0, "r")] or lim[(1, "l")] != -lim[(1, "r")]:
        no("box limits must be symmetric")
    # equalities
    curr = dcurr = None
    dt = None
    for label, diff in opt.lin_eq_constraints.items():
        rhs, lhs = diff.a, diff.b
        if isinstance(lhs, StateRef) and lhs.t == 0 and isinstance(rhs, ParamRef):
            if lhs.time_deriv == 0:
                curr = rhs
            else:
                dcurr = rhs
        elif isinstance(lhs, IntegrationResidual) and _is_zero_const(rhs) and lhs.xd.time_deriv == 1:
            if not np.all(lhs.dt == lhs.dt[0]):
                no("non-uniform dt is not lowered")
            dt = float(lhs.dt[0])
        else:
            no(f"linear equality '{label}' not recognised")
    if curr is None or dcurr is None or dt is None:
        no("need fix_configuration for position and velocity and integrate_model_states")
    # obstacle rows
    if len(opt.ineq_constraints) != T:
        no(f"expected {T} obstacle rows, found {len(opt.ineq_constraints)}")
    obs = None
    safe_sq = None
    for i, (label, diff) in enumerate(opt.ineq_constraints.items()):
        ok = (
            isinstance(diff, Sub) and isinstance(diff.a, SumSqr) and isinstance(diff.b, Const) and diff.b.value.size == 1
            and isinstance(diff.a.a, Sub) and isinstance(diff.a.a.a, ParamCol) and isinstance(diff.a.a.b, StateRef)
            and diff.a.a.a.col == i and diff.a.a.b.t == i and diff.a.a.b.time_deriv == 0
        )
        if not ok:
            no(f"inequality '{label}' is not ||obs[:, {i}] - y_{i}||^2 >= const")
        if obs is None:
            obs, safe_sq = diff.a.a.a.param, float(diff.b.value.reshape(-1)[0])
        elif diff.a.a.a.param is not obs or float(diff.b.value.reshape(-1)[0]) != safe_sq:
            no("all obstacle rows must use the same obstacle parameter and radius")
    # costs
    goal = None
    w_acc = None
    for label, term in opt.cost_terms.items():
        w, e = _unscale(term)
        if not isinstance(e, SumSqr):
            no(f"cost '{label}' is not a weighted sumsqr")
        inner = e.a
        if isinstance(inner, Sub) and isinstance(inner.a, ParamRef) and inner.b is Y and w == 1.0:
            goal = inner.a
        else:
            s2, d = _unscale(inner)
            good = (
                isinstance(d, Sub) and isinstance(d.a, StateCols) and isinstance(d.b, StateCols) and d.a.state is dY and d.b.state is dY
                and (d.a.lo, d.a.hi, d.b.lo, d.b.hi) == (1, T, 0, T - 1) and abs(s2 * dt - 1.0) < 1e-12
            )
            if not good:
                no(f"cost '{label}' not recognised")
            w_acc = w
    if goal is None or w_acc is None:
        no("need the path-tracking and the acceleration cost terms")
    names = (curr.name, dcurr.name, goal.name, obs.name)
    if list(opt.parameters.keys()) != list(names):
        no(f"parameters must be created in the order {names}")
    if (curr.shape, dcurr.shape, goal.shape, obs.shape) != ((2, 1), (2, 1), (2, T), (2, T)):
        no("parameter shapes must be curr(2), dcurr(2), goal(2xT), obs(2xT)")
    return PointMassSpec(T, dt, w_acc, lim[(0, "r")], lim[(1, "r")], float(np.sqrt(safe_sq)), names, y_name, dy_name)


@dataclass
class ArmSpec:
    robot: RobotModel
    link: str
    w_path: float
    w_vel: float
    offsets: np.ndarray  # (T, 3): path_t = p(qc) + offsets[t]
    qc_name: str
    q_name: str
    dq_name: str
    guards: Optional["GuardSpec"] = None


@dataclass
class GuardSpec:
    """Inequality rows of one arm: joint limits (enforce_model_limits) and sphere clearances
    (sphere_collision_avoidance_constraints); parameter labels in the order the kernel family expects them."""

    lo: Optional[np.ndarray]
    up: Optional[np.ndarray]
    links: list          # sphere link names
    link_radii: list     # parameter labels, one per link
    obstacles: list      # (position label, radius label) per obstacle
    vlo: Optional[np.ndarray] = None  # joint-velocity limits (enforce_model_limits(name, time_deriv=1)), None: no such rows
    vup: Optional[np.ndarray] = None


@dataclass
class MultiArmSpec:
    """Separable sum of position-only end-effector tracking problems, one per robot (example/dual_arm.py:17-129)."""

    T: int
    dt: float
    arms: list


def match_multi_arm(opt: Optimization) -> MultiArmSpec:
    def no(msg):
        raise LoweringError(f"multi-arm lowering: {msg}")

    robots = [m for m in (opt.models or []) if isinstance(m, RobotModel)]
    if not robots or len(robots) != len(opt.models or []):
        no("expected robot models only")
    if opt.nh:
        no("nonlinear equalities are not lowered for this family")
    for r in robots:
        if list(r.time_derivs) != [0, 1] or r.num_param_joints != 0:
            no("every robot must have time_derivs=[0, 1] and no parameterised joints")
    names = []
    for r in robots:
        names += [r.state_optimized_name(0), r.state_optimized_name(1)]
    if list(opt.decision_variables.keys()) != names:
        no("decision variables must be exactly the q/dq blocks of the robots")
    T = opt.decision_variables[names[0]].n
    arms = {}
    for r in robots:
        Q, dQ = opt.decision_variables[r.state_optimized_name(0)], opt.decision_variables[r.state_optimized_name(1)]
        if Q.n != T or dQ.n != T - 1:
            no("all robots must share T and use derivs_align=False")
        arms[r.get_name()] = {"robot": r, "Q": Q, "dQ": dQ}
    dt = None
    for label, diff in opt.lin_eq_constraints.items():
        if not isinstance(diff, Sub):
            no(f"linear equality '{label}' not recognised")
        rhs, lhs = diff.a, diff.b
        if isinstance(lhs, StateRef) and lhs.t == 0 and lhs.time_deriv == 0 and isinstance(rhs, ParamRef) and lhs.model_name in arms:
            arms[lhs.model_name]["qc"] = rhs
        elif isinstance(lhs, IntegrationResidual) and _is_zero_const(rhs) and lhs.xd.time_deriv == 1 and lhs.x.model_name in arms:
            if not np.all(lhs.dt == lhs.dt[0]) or (dt is not None and float(lhs.dt[0]) != dt):
                no("non-uniform dt is not lowered")
            dt = float(lhs.dt[0])
            arms[lhs.x.model_name]["integr"] = True
        else:
            no(f"linear equality '{label}' not recognised (dq_0 must be free in this family)")
    for label, term in opt.cost_terms.items():
        w, e = _unscale(term)
        if not isinstance(e, SumSqr):
            no(f"cost '{label}' is not a weighted sumsqr")
        inner = e.a
        hit = False
        for a in arms.values():
            if inner is a["dQ"]:
                if "w_vel" in a:
                    no(f"cost '{label}': a second joint-velocity term for one arm (weights are not summed or overwritten silently)")
                a["w_vel"], hit = w, True
            elif isinstance(inner, Sub):
                for pos, pth, sgn in ((inner.a, inner.b, 1), (inner.b, inner.a, 1)):
                    if (isinstance(pos, LinkFunction) and pos.what == "position" and pos.q is a["Q"] and pos.robot is a["robot"]
                            and isinstance(pth, Add) and isinstance(pth.a, LinkFunction) and pth.a.what == "position" and pth.a.link == pos.link
                            and pth.a.robot is a["robot"] and isinstance(pth.a.q, ParamRef) and isinstance(pth.b, Const) and pth.b.value.shape == (3, T)):
                        if "w_path" in a and not hit:
                            no(f"cost '{label}': a second path-tracking term for one arm (weights are not summed or overwritten silently)")
                        a["w_path"], a["link"], a["offsets"], a["qc_path"], hit = w, pos.link, np.ascontiguousarray(pth.b.value.T), pth.a.q, True
        if not hit:
            no(f"cost '{label}' not recognised")
    # inequality rows: joint limits (linear) and sphere clearances (nonlinear), per arm
    for label, d in opt.lin_ineq_constraints.items():
        hit = False
        for a in arms.values():
            n = a["robot"].ndof
            if isinstance(d, Sub) and d.a is a["Q"] and isinstance(d.b, Const) and d.b.value.shape == (n, 1):
                a["lo"], hit = np.maximum(a.get("lo", np.full(n, -np.inf)), d.b.value[:, 0]), True
            elif isinstance(d, Sub) and d.b is a["Q"] and isinstance(d.a, Const) and d.a.value.shape == (n, 1):
                a["up"], hit = np.minimum(a.get("up", np.full(n, np.inf)), d.a.value[:, 0]), True
            elif isinstance(d, Sub) and d.a is a["dQ"] and isinstance(d.b, Const) and d.b.value.shape == (n, 1):
                a["vlo"], hit = np.maximum(a.get("vlo", np.full(n, -np.inf)), d.b.value[:, 0]), True
            elif isinstance(d, Sub) and d.b is a["dQ"] and isinstance(d.a, Const) and d.a.value.shape == (n, 1):
                a["vup"], hit = np.minimum(a.get("vup", np.full(n, np.inf)), d.a.value[:, 0]), True
        if not hit:
            no(f"linear inequality '{label}' is not a joint-position or joint-velocity bound over the whole trajectory")
    for label, d in opt.ineq_constraints.items():
        ok = (isinstance(d, Sub) and isinstance(d.a, SumSqr) and isinstance(d.a.a, Sub) and isinstance(d.a.a.a, LinkFunction)
              and d.a.a.a.what == "position" and isinstance(d.a.a.b, ParamRef) and d.a.a.b.shape == (3, 1)
              and isinstance(d.b, Square) and isinstance(d.b.a, Add) and isinstance(d.b.a.a, ParamRef) and isinstance(d.b.a.b, ParamRef)
              and isinstance(d.a.a.a.q, StateRef) and d.a.a.a.q.t is not None and d.a.a.a.q.model_name in arms)
        if not ok:
            no(f"inequality '{label}' is not a sphere clearance ||p_link(q_t) - o||^2 >= (r_link + r_o)^2")
        pos = d.a.a.a
        a = arms[pos.q.model_name]
        if pos.robot is not a["robot"] or pos.q.var_name != a["Q"].var_name:
            no(f"inequality '{label}' mixes models")
        a.setdefault("spheres", {})[(pos.q.t, pos.link, d.a.a.b.name)] = (d.b.a.a.name, d.b.a.b.name)
    for name, a in arms.items():
        if ("lo" in a) != ("up" in a) or ("vlo" in a) != ("vup" in a):
            no(f"robot '{name}': limits need both the lower and the upper row block")
        g = None
        sph = a.get("spheres")
        if sph:
            links, obst = [], []
            for (t, ln, on) in sph:
                if ln not in links:
                    links.append(ln)
                if on not in obst:
                    obst.append(on)
            if len(sph) != T * len(links) * len(obst):
                no(f"robot '{name}': sphere rows must cover every (knot, link, obstacle) combination")
            lrad = {ln: sph[(0, ln, obst[0])][0] for ln in links}
            orad = {on: sph[(0, links[0], on)][1] for on in obst}
            for (t, ln, on), (lr, orr) in sph.items():
                if lr != lrad[ln] or orr != orad[on]:
                    no(f"robot '{name}': inconsistent radius parameters in the sphere rows")
            g = GuardSpec(a.get("lo"), a.get("up"), links, [lrad[ln] for ln in links], [(on, orad[on]) for on in obst], a.get("vlo"), a.get("vup"))
        elif "lo" in a or "vlo" in a:
            g = GuardSpec(a.get("lo"), a.get("up"), [], [], [], a.get("vlo"), a.get("vup"))
        a["guards"] = g
    out = []
    for name, a in arms.items():
        need = {"qc", "integr", "w_vel", "w_path"}
        if not need <= set(a):
            no(f"robot '{name}' is missing {sorted(need - set(a))}")
        if a["qc_path"] is not a["qc"]:
            no(f"robot '{name}': the path must start from the fixed initial configuration parameter")
        out.append(ArmSpec(a["robot"], a["link"], a["w_path"], a["w_vel"], a["offsets"], a["qc"].name, a["Q"].var_name, a["dQ"].var_name, a["guards"]))
    params = [k for k, v in opt.parameters.items() if v.numel() > 0]
    known = set(a.qc_name for a in out)
    for a in out:
        if a.guards is not None:
            known |= set(a.guards.link_radii) | set(x for ob in a.guards.obstacles for x in ob)
    if set(params) != known:
        no("parameters other than the initial configurations and the sphere radii / obstacle positions are present")
    return MultiArmSpec(T, dt, out)


@dataclass
class IkSpec:
    robot: RobotModel
    link: str
    w_nominal: float
    lo: np.ndarray
    up: np.ndarray
    qn_name: str
    pg_name: str
    q_name: str


def match_ik(opt: Optimization) -> IkSpec:
    """example/example.py:13-60: min w||q - q_nominal||^2 s.t. p_link(q) = p_goal, lo <= q <= up (T = 1)."""

    def no(msg):
        raise LoweringError(f"inverse-kinematics lowering: {msg}")

    robots = [m for m in (opt.models or []) if isinstance(m, RobotModel)]
    if len(opt.models or []) != 1 or len(robots) != 1:
        no("expected exactly one RobotModel and no task models")
    robot = robots[0]
    if list(robot.time_derivs) != [0] or robot.num_param_joints != 0:
        no("robot must have time_derivs=[0] and no parameterised joints")
    q_name = robot.state_optimized_name(0)
    if list(opt.decision_variables.keys()) != [q_name]:
        no(f"decision variables must be exactly [{q_name}]")
    Q: StateRef = opt.decision_variables[q_name]
    if Q.n != 1:
        no("T must be 1")
    n = robot.ndof
    if opt.ng or opt.na:
        no("nonlinear inequality / linear equality rows are not part of this family")

    def is_q(e):
        return isinstance(e, StateRef) and e.var_name == q_name and e.time_deriv == 0 and (e.t in (None, 0))

    # nonlinear equality: p_link(q) == p_goal
    if len(opt.eq_constraints) != 1:
        no("expected exactly one nonlinear equality (link position goal)")
    (label, diff), = opt.eq_constraints.items()
    if not (isinstance(diff, Sub) and isinstance(diff.a, ParamRef) and isinstance(diff.b, LinkFunction) and diff.b.what == "position"
            and is_q(diff.b.q) and diff.b.robot is robot and diff.a.shape == (3, 1)):
        no(f"equality '{label}' is not p(link, q) == p_goal with a 3-vector parameter")
    pg, link = diff.a, diff.b.link

    # cost: w * sumsqr(q - q_nominal)
    if len(opt.cost_terms) != 1:
        no("expected exactly one cost term")
    (label, term), = opt.cost_terms.items()
    w, e = _unscale(term)
    if not (isinstance(e, SumSqr) and isinstance(e.a, Sub) and is_q(e.a.a) and isinstance(e.a.b, ParamRef) and e.a.b.shape == (n, 1)):
        no(f"cost '{label}' is not w * sumsqr(q - q_nominal)")
    qn = e.a.b
    if not w > 0:
        no("cost weight must be positive")

    # linear inequalities: joint limits (optional)
    lo, up = np.full(n, -1e9), np.full(n, 1e9)
    for label, d in opt.lin_ineq_constraints.items():
        if isinstance(d, Sub) and is_q(d.a) and isinstance(d.b, Const) and d.b.value.shape == (n, 1):
            lo = np.maximum(lo, d.b.value[:, 0])
        elif isinstance(d, Sub) and is_q(d.b) and isinstance(d.a, Const) and d.a.value.shape == (n, 1):
            up = np.minimum(up, d.a.value[:, 0])
        else:
            no(f"linear inequality '{label}' is not a joint bound")
    params = [k for k, v in opt.parameters.items() if v.numel() > 0]
    if params != [qn.name, pg.name]:
        no(f"non-empty parameters must be ['{qn.name}', '{pg.name}'] in this order, found {params}")
    return IkSpec(robot, link, float(w), lo, up, qn.name, pg.name, q_name)


@dataclass
class QpSpec:
    """Generic small dense QP: the matrices are read off the Optimization's numeric members per instance."""

    n: int
    m: int
    me: int
    problem: Optional[Optimization] = None  # the linearly constrained problem the kernel is handed when the user's rows were rewritten (band rows)
    bands: tuple = ()  # names of the rewritten rows


def _same_node(a, b) -> bool:
    """a and b denote the same expression: one node, or equal row / block selections of one node (``d[0] * d[0]`` indexes twice)."""
    from .expr import Block

    if a is b:
        return True
    if isinstance(a, Rows) and isinstance(b, Rows):
        return tuple(a.idx) == tuple(b.idx) and _same_node(a.a, b.a)
    if isinstance(a, Block) and isinstance(b, Block):
        return tuple(a.ridx) == tuple(b.ridx) and tuple(a.cidx) == tuple(b.cidx) and _same_node(a.a, b.a)
    return False


def _band_rows(opt: Optimization):
    """Rows ``c - e * e >= 0`` with ``e`` affine in x and ``c`` a non-negative constant (``add_leq_inequality_constraint(name, d * d, 1e-8)``,
    example/torque_control_example.py:93-95) describe the band ``-sqrt(c) <= e <= sqrt(c)``: the same feasible set, so a quadratic cost over
    them is a QP with the same minimisers.  Returns {name: (e, sqrt c)}; raises LoweringError for any other nonlinear row."""
    out = {}
    for name, term in opt.ineq_constraints.items():
        sq = term.b if isinstance(term, Sub) else None
        inner = None
        if isinstance(sq, Square):
            inner = sq.a
        elif isinstance(sq, Mul) and _same_node(sq.a, sq.b):
            inner = sq.a
        if inner is None or not isinstance(term.a, Const) or inner.degree() > 1:
            raise LoweringError(f"dense-QP lowering: nonlinear inequality '{name}' is not of the form c - e*e with e linear in x")
        if np.any(term.a.value < 0.0):
            raise LoweringError(f"dense-QP lowering: '{name}' bounds a square by a negative number (empty feasible set)")
        out[name] = (inner, np.broadcast_to(np.sqrt(term.a.value), term.shape).copy())
    return out


def match_qp(opt: Optimization) -> QpSpec:
    """QuadraticCostUnconstrained / QuadraticCostLinearConstraints (optimization.py:312-388) with small dense data: what the
    reference's OSQP / CVXOPT / qpOASES back-ends take (solver.py:421-584).  QuadraticCostNonlinearConstraints (optimization.py:391-460)
    whose only nonlinear rows are squares of affine expressions under a constant bound (`_band_rows`) are handed over as the equivalent
    linearly constrained QP.  Last resort: the dedicated families come first."""
    from .optimization import QuadraticCostLinearConstraints, QuadraticCostNonlinearConstraints, QuadraticCostUnconstrained
    from .sx_container import SXContainer

    problem, bands = None, ()
    if isinstance(opt, QuadraticCostNonlinearConstraints):
        if opt.nh:
            raise LoweringError("dense-QP lowering: nonlinear equality rows")
        rows = _band_rows(opt)
        lin = SXContainer()
        for name, term in opt.lin_ineq_constraints.items():
            lin[name] = term
        for name, (e, half) in rows.items():
            lin[name + "__band_l"] = Add(e, Const(half))  # e + sqrt c >= 0
            lin[name + "__band_r"] = Sub(Const(half), e)  # sqrt c - e >= 0
        problem = QuadraticCostLinearConstraints(opt.decision_variables, opt.parameters, opt.cost_terms, opt.lin_eq_constraints, lin)
        problem.models = opt.models
        bands = tuple(rows)
        opt = problem
    if not isinstance(opt, (QuadraticCostUnconstrained, QuadraticCostLinearConstraints)):
        raise LoweringError("dense-QP lowering: the problem is not of a QuadraticCost{Unconstrained, LinearConstraints} class")
    if opt.has_discrete_variables():
        raise LoweringError("dense-QP lowering: discrete variables are not supported")
    n, m, me = opt.nx, opt.nk, opt.na
    if not (1 <= n <= 32 and m <= 256 and me <= min(32, n)):
        raise LoweringError(f"dense-QP lowering: sizes nx={n}, nk={m}, na={me} exceed the dense kernel's limits (32, 256, 32)")
    return QpSpec(n, m, me, problem, bands)


@dataclass
class TapeSpec:
    tape: object


def match_tape(opt: Optimization) -> TapeSpec:
    """Last resort: any problem whose expression trees compile to a scalar tape (optas_amd/tape.py), evaluated on the GPU by generated code;
    dense BFGS up to 48 variables, limited-memory BFGS beyond (trajectory-sized problems: hundreds of variables, tens of thousands of tape
    evaluations per solve -- a fallback that solves, not a fast path)."""
    from .tape import compile_problem

    if opt.has_discrete_variables():
        raise LoweringError("tape lowering: discrete variables are not supported")
    if not 1 <= opt.nx <= 4096:
        raise LoweringError(f"tape lowering: nx={opt.nx} exceeds the generic family's limit of 4096 decision variables (OH_TAPE_MAX_N)")
    try:
        tape = compile_problem(opt)
    except NotImplementedError as e:
        raise LoweringError(f"tape lowering: {e}") from None
    return TapeSpec(tape)


OH_KIND_MULTI_ARM = 101  # host-side composition of OH_PROBLEM_FIGURE_EIGHT handles with lock_orientation = 0


@dataclass
class TorqueSpec:
    """BASELINE configs[4]: torque MPC with RobotModel.rnea as equality rows (OH_PROBLEM_TORQUE_MPC)."""

    robot: RobotModel
    link: str
    T: int
    dt: float
    w_path: float
    w_vel: float
    w_tau: float
    tau_lo: np.ndarray
    tau_up: np.ndarray
    dq_lo: Optional[np.ndarray] = None  # joint-velocity limits on the velocity states (enforce_model_limits(name, time_deriv=1)), None: no such rows
    dq_up: Optional[np.ndarray] = None


def match_torque_mpc(opt: Optimization) -> TorqueSpec:
    def no(msg):
        raise LoweringError(f"torque-MPC lowering: {msg}")

    models = list(opt.models or [])
    robots = [m for m in models if isinstance(m, RobotModel)]
    tasks = [m for m in models if isinstance(m, TaskModel)]
    if len(robots) != 1 or len(tasks) != 1 or len(models) != 2:
        no("expected one RobotModel and one TaskModel (the joint torques)")
    robot, task = robots[0], tasks[0]
    n = robot.ndof
    if list(robot.time_derivs) != [0, 1, 2] or robot.num_param_joints != 0 or list(task.time_derivs) != [0] or task.dim != n:
        no("robot must have time_derivs=[0, 1, 2] and no parameterised joints; the task model must be ndof-dimensional with time_derivs=[0]")
    names = [robot.state_optimized_name(d) for d in (0, 1, 2)] + [task.state_optimized_name(0)]
    if list(opt.decision_variables.keys()) != names:
        no(f"decision variables must be exactly {names}")
    Q, dQ, ddQ, TAU = (opt.decision_variables[k] for k in names)
    T = Q.n
    if not (dQ.n == ddQ.n == TAU.n == T):
        no("derivs_align=True is required (every block has T columns)")
    # linear equalities
    qc = dqc = None
    dts = {}
    for label, diff in opt.lin_eq_constraints.items():
        if not isinstance(diff, Sub):
            no(f"linear equality '{label}' not recognised")
        rhs, lhs = diff.a, diff.b
        if isinstance(lhs, StateRef) and lhs.t == 0 and lhs.model_name == robot.get_name() and lhs.time_deriv in (0, 1) and isinstance(rhs, ParamRef) \
                and rhs.shape == (n, 1):
            if lhs.time_deriv == 0 and qc is None:
                qc = rhs
            elif lhs.time_deriv == 1 and dqc is None:
                dqc = rhs
            else:
                no(f"linear equality '{label}' fixes a configuration twice")
        elif isinstance(lhs, IntegrationResidual) and _is_zero_const(rhs) and lhs.x.model_name == robot.get_name() and lhs.xd.time_deriv in (1, 2) \
                and lhs.xd.time_deriv not in dts:
            if not np.allclose(lhs.dt, lhs.dt[0], rtol=0, atol=0):
                no("non-uniform dt is not lowered")
            dts[lhs.xd.time_deriv] = float(lhs.dt[0])
        else:
            no(f"linear equality '{label}' not recognised")
    if qc is None or dqc is None or sorted(dts) != [1, 2] or dts[1] != dts[2]:
        no("need fix_configuration(q, qc), fix_configuration(dq, dqc, time_deriv=1) and integrate_model_states for time_deriv 1 and 2 with one dt")
    # dynamics rows h = TAU - rnea(Q, dQ, ddQ)
    if len(opt.eq_constraints) != 1:
        no("expected exactly one nonlinear equality (the inverse dynamics)")
    (label, diff), = opt.eq_constraints.items()
    if not (isinstance(diff, Sub) and diff.a is TAU and isinstance(diff.b, RneaFunction) and diff.b.robot is robot and diff.b.q is Q and diff.b.qd is dQ
            and diff.b.qdd is ddQ):
        no(f"equality '{label}' is not add_equality_constraint(lhs=robot.rnea(Q, dQ, ddQ), rhs=TAU)")
    if len(opt.ineq_constraints):
        no("nonlinear inequalities are not lowered")
    # effort limits
    lo = up = vlo = vup = None
    for label, d in opt.lin_ineq_constraints.items():
        if isinstance(d, Sub) and d.a is TAU and isinstance(d.b, Const) and d.b.value.shape == (n, 1) and lo is None:
            lo = d.b.value[:, 0]
        elif isinstance(d, Sub) and d.b is TAU and isinstance(d.a, Const) and d.a.value.shape == (n, 1) and up is None:
            up = d.a.value[:, 0]
        elif isinstance(d, Sub) and d.a is dQ and isinstance(d.b, Const) and d.b.value.shape == (n, 1) and vlo is None:
            vlo = d.b.value[:, 0]
        elif isinstance(d, Sub) and d.b is dQ and isinstance(d.a, Const) and d.a.value.shape == (n, 1) and vup is None:
            vup = d.a.value[:, 0]
        else:
            no(f"linear inequality '{label}' is not an effort or joint-velocity bound over the whole trajectory (or a second one of its kind)")
    if (lo is None) != (up is None) or (vlo is None) != (vup is None):
        no("limits need both the lower and the upper row block")
    if lo is None:
        lo, up = -1e9 * np.ones(n), 1e9 * np.ones(n)
    # costs
    w_path = w_vel = w_tau = None
    link = goal = None
    for label, term in opt.cost_terms.items():
        w, e = _unscale(term)
        if not isinstance(e, SumSqr):
            no(f"cost '{label}' is not a weighted sumsqr")
        inner = e.a
        if inner is dQ and w_vel is None:
            w_vel = w
        elif inner is TAU and w_tau is None:
            w_tau = w
        elif (isinstance(inner, Sub) and w_path is None and isinstance(inner.a, LinkFunction) and inner.a.what == "position" and inner.a.q is Q
              and inner.a.robot is robot and isinstance(inner.b, ParamRef) and inner.b.shape == (3, T)):
            w_path, link, goal = w, inner.a.link, inner.b
        else:
            no(f"cost '{label}' not recognised (or a second term of its kind: weights are not summed silently)")
    if w_path is None or w_tau is None:
        no("need the tracking term sumsqr(p_link(Q) - goal) and the effort term sumsqr(TAU)")
    params = [k for k, v in opt.parameters.items() if v.numel() > 0]
    if params != [qc.name, dqc.name, goal.name]:
        no(f"the non-empty parameters must be [{qc.name}, {dqc.name}, {goal.name}] in this order (the kernel family reads p = [qc; dqc; vec(goal)]), found {params}")
    return TorqueSpec(robot, link, T, dts[1], float(w_path), float(w_vel or 0.0), float(w_tau), np.asarray(lo, float), np.asarray(up, float),
                      None if vlo is None else np.asarray(vlo, float), None if vup is None else np.asarray(vup, float))


def lower(opt: Optimization):
    """Return (kind, spec).  Raises LoweringError if no kernel family matches."""
    errors = []
    for kind, fn in ((_lib.OH_PROBLEM_FIGURE_EIGHT, match_figure_eight), (_lib.OH_PROBLEM_TORQUE_MPC, match_torque_mpc), (_lib.OH_PROBLEM_POINT_MASS_MPC, match_point_mass), (_lib.OH_PROBLEM_POINT_MASS_MPC, match_point_mass_planner),
                     (OH_KIND_MULTI_ARM, match_multi_arm),
                     (_lib.OH_PROBLEM_IK, match_ik), (_lib.OH_PROBLEM_QP, match_qp), (_lib.OH_PROBLEM_TAPE, match_tape)):
        try:
            return kind, fn(opt)
        except LoweringError as e:
            errors.append(str(e))
    raise LoweringError("no HIP kernel family matches this optimization problem: " + "; ".join(errors))
