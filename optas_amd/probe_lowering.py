"""Lowering of a *reference* ``optas.optimization.Optimization`` (CasADi inside) to the structured kernel families -- without
reading a single SX node.

The mirror builder of this repo records expression trees, so ``optas_amd.lowering`` can pattern-match them.  A real optas problem holds
opaque ``casadi.SX`` graphs and ``casadi.Function`` members (optimization.py:60-306).  What both worlds share is the *interface*:

  * the containers ``decision_variables / parameters / lin_eq_constraints / eq_constraints / ...`` with the builder's labels and shapes
    (``"{name}/q/x"``, ``"__{name}_fix_configuration_0_0__"``, ``"__integrate_model_states_{name}_1__"``; builder.py:90-99, 469, 539),
  * ``models`` (``RobotModel`` with its parsed URDF, models.py:286-321),
  * callable numeric members ``f(x, p), a(x, p), h(x, p), k(x, p), g(x, p)``.

``probe_figure_eight`` uses exactly that: (1) labels and shapes decide whether the problem *can* be the figure-eight family of
example/figure_eight_plan.py:16-113; (2) the numbers the kernels need (dt, weights, the path in the end-effector frame, the tracked link)
are read off the problem's own functions by evaluating them at probe points -- ``a`` is affine and ``f`` is a separable sum of squares, so
a handful of evaluations per knot identifies them exactly; (3) the resulting family model is then **verified** against ``f``, ``a`` and
``h`` at random points and a second parameter vector.  Any mismatch raises ``LoweringError``: a problem is either proven (to 1e-9) to be
the family or refused, never approximated.  Kinematics for the probes come from ``oh_fk_jac`` (the GPU is needed here as everywhere).

Used by ``optas_amd.casadi_tape.make_solver_class`` (the literal ``optas.solver.Solver`` subclass): structured family first, generic
tape family as the fallback.  Tested against the mirror's ``Optimization`` objects behind an adapter that hides everything but the
reference interface (tests/test_probe_lowering.py), including a URDF object shaped like urdf_parser_py's.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .lowering import FigureEightSpec, LoweringError
from .models import RobotModel
from .urdf import Inertial, Joint, Limit, Link, RobotDescription


def _vec(fn, x, p) -> np.ndarray:
    """Value of a numeric member (casadi.Function returning DM, or a numpy callable) as a flat float64 vector."""
    if fn is None:
        return np.zeros(0)
    v = fn(x, p)
    if hasattr(v, "toarray"):
        v = v.toarray()
    elif hasattr(v, "full"):
        v = v.full()
    return np.asarray(v, dtype=np.float64).reshape(-1)


def _shape(item):
    s = item.shape
    return int(s[0]), int(s[1])


def description_from_urdf(urdf) -> RobotDescription:
    """``RobotDescription`` from what ``RobotModel.urdf`` holds in the reference (a urdf_parser_py ``Robot``: joints with
    ``origin.xyz/rpy``, ``axis``, ``limit``; links with ``inertial``) -- or from this repo's own description, returned as is."""
    if isinstance(urdf, RobotDescription):
        return urdf
    links = []
    for l in urdf.links:
        ine = getattr(l, "inertial", None)
        inertial = None
        if ine is not None:
            org = getattr(ine, "origin", None)
            I = ine.inertia
            inertial = Inertial(mass=float(ine.mass), xyz=[float(v) for v in (org.xyz if org is not None and org.xyz is not None else (0, 0, 0))],
                                rpy=[float(v) for v in (org.rpy if org is not None and org.rpy is not None else (0, 0, 0))],
                                inertia=[float(getattr(I, k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")])
        links.append(Link(name=l.name, inertial=inertial))
    joints = []
    for j in urdf.joints:
        org = getattr(j, "origin", None)
        lim = getattr(j, "limit", None)
        joints.append(Joint(
            name=j.name, type=j.type, parent=j.parent, child=j.child,
            xyz=None if org is None else [float(v) for v in (org.xyz if org.xyz is not None else (0, 0, 0))],
            rpy=None if org is None else [float(v) for v in (org.rpy if org.rpy is not None else (0, 0, 0))],
            axis=None if getattr(j, "axis", None) is None else [float(v) for v in j.axis],
            limit=None if lim is None else Limit(lower=float(lim.lower or 0.0), upper=float(lim.upper or 0.0), velocity=float(lim.velocity or 0.0),
                                                 effort=float(lim.effort or 0.0))))
    return RobotDescription(name=urdf.name, links=links, joints=joints)


def _mirror_robot(model) -> RobotModel:
    """A RobotModel of this repo with the same kinematic tree, name and derivative orders as the reference's model object."""
    if isinstance(model, RobotModel):
        return model
    return RobotModel.from_description(description_from_urdf(model.urdf), name=model.get_name(), time_derivs=list(model.time_derivs),
                                       param_joints=list(getattr(model, "param_joints", []) or []))


def probe_figure_eight(opt, rng_seed: int = 12345, link: Optional[str] = None) -> FigureEightSpec:
    """See the module docstring.  ``link`` may name the tracked link; by default every link of the robot is tried against ``h``."""

    def no(msg):
        raise LoweringError(f"figure-eight probing: {msg}")

    models = list(opt.models or [])
    if len(models) != 1 or not hasattr(models[0], "urdf"):
        no("expected exactly one robot model")
    m = models[0]
    name = m.get_name()
    if list(m.time_derivs) == [0, 1] and len(getattr(m, "param_joints", []) or []) == 1:
        return _probe_figure_eight_lead(opt, m, rng_seed, link, no)  # example/figure_eight_plan_6dof.py: one joint ahead of the chain is a parameter
    if list(m.time_derivs) != [0, 1] or len(getattr(m, "param_joints", []) or []) != 0:
        no("robot must have time_derivs=[0, 1] and at most one parameterised joint")
    q_name, dq_name = f"{name}/q/x", f"{name}/dq/x"
    if list(opt.decision_variables.keys()) != [q_name, dq_name]:
        no(f"decision variables must be exactly [{q_name}, {dq_name}], found {list(opt.decision_variables.keys())}")
    n, T = _shape(opt.decision_variables[q_name])
    if _shape(opt.decision_variables[dq_name]) != (n, T - 1):
        no("the velocity block must be ndof x (T - 1) (derivs_align=False)")
    params = [(k, _shape(v)) for k, v in opt.parameters.items() if _shape(v)[0] * _shape(v)[1] > 0]
    if not params or params[0][1] != (n, 1):
        no(f"expected the initial configuration ({n}, 1) as the first non-empty parameter, found {params}")
    extras = [k for k, _ in params[1:]]  # link radii, obstacle positions / radii of sphere rows (builder.py:366-417); attributed and verified below
    if extras and not len(opt.ineq_constraints):
        no(f"parameters beyond the initial configuration without sphere rows: {extras}")
    poff, o_ = {}, 0
    for k_, v_ in opt.parameters.items():
        m_, n_ = _shape(v_)
        poff[k_] = (o_, m_ * n_)
        o_ += m_ * n_
    p_harmless = np.zeros(o_)  # radii 0.05, obstacles far away: the rows stay inactive wherever the probes go
    for k_ in extras:
        a0, l0 = poff[k_]
        p_harmless[a0 : a0 + l0] = 0.05 if l0 == 1 else 5.0 + np.arange(l0)
    qc_slice = slice(poff[params[0][0]][0], poff[params[0][0]][0] + n)

    def pfull(qc_):
        pv = p_harmless.copy()
        pv[qc_slice] = np.asarray(qc_, dtype=float).reshape(-1)
        return pv

    real = opt
    if extras:
        class _QcOnly:  # the figure-eight probes below speak p = qc; this hands the problem's functions the full parameter vector
            def __getattr__(self, name):
                attr = getattr(real, name)
                if name in ("f", "a", "h", "k", "g", "v"):
                    return lambda x, p, _fn=attr: _fn(x, pfull(p))
                return attr

        opt = _QcOnly()
    want = {f"__{name}_fix_configuration_0_0__": (n, 1), f"__{name}_fix_configuration_1_0__": (n, 1), f"__integrate_model_states_{name}_1__": (n, T - 1)}
    got = {k: _shape(v) for k, v in opt.lin_eq_constraints.items()}
    if got != want:
        no(f"linear equalities must be {want} (fix_configuration of q and dq at t = 0, integrate_model_states), found {got}")
    # inequality rows (round 4): enforce_model_limits(name) / (name, time_deriv=1) (builder.py:471-509) and sphere_collision_avoidance_constraints
    # (builder.py:366-417): found by label, attributed, read off k / g and verified against them by the routine the multi-arm family uses
    guarded = len(opt.lin_ineq_constraints) > 0 or len(opt.ineq_constraints) > 0
    eq = [(k, _shape(v)) for k, v in opt.eq_constraints.items()]
    if len(eq) != 1 or eq[0][1] != (4, T):
        no(f"expected one nonlinear equality of shape (4, {T}) (the end-effector quaternion lock), found {eq}")
    nx, np_ = n * T + n * (T - 1), int(opt.np)
    if int(opt.nx) != nx or np_ != n + sum(poff[k_][1] for k_ in extras):
        no("unexpected nx / np")
    robot = _mirror_robot(m)
    if robot.ndof != n:
        no("the robot's ndof does not match the decision variables")
    rng = np.random.default_rng(rng_seed)
    nq = n * T
    lo, up = robot.lower_actuated_joint_limits, robot.upper_actuated_joint_limits
    mid, half = 0.5 * (lo + up), 0.3 * np.minimum(up - lo, 4.0)
    qc = mid + rng.uniform(-1, 1, n) * half

    def xvec(Q, dQ):  # (T, n), (T-1, n) -> vec order (blocks column-major: x[n t + j])
        return np.concatenate([Q.reshape(-1), dQ.reshape(-1)])

    Qc = np.tile(qc, (T, 1))
    Z = np.zeros((T - 1, n))
    x0 = xvec(Qc, Z)
    # ---- linear rows: a = [qc - q_0; 0 - dq_0; -(q_t + dt dq_t - q_{t+1})]: dt from one probe, then the whole block is checked
    a0 = _vec(opt.a, x0, qc)
    if a0.shape != (2 * n + n * (T - 1),) or np.abs(a0).max() > 1e-12:
        no("a(x, p) does not vanish at q_t = qc, dq = 0")
    d = np.zeros_like(x0)
    d[nq + n] = 1.0  # dq_1[0]
    dt = -float(_vec(opt.a, x0 + d, qc)[2 * n + n])
    if not (dt > 0):
        no("could not read a positive dt off the integration rows")
    Qr, dQr, pr = rng.normal(size=(T, n)), rng.normal(size=(T - 1, n)), rng.normal(size=n)
    a_model = np.concatenate([pr - Qr[0], -dQr[0], -(Qr[:-1] + dt * dQr - Qr[1:]).reshape(-1)])
    if np.abs(_vec(opt.a, xvec(Qr, dQr), pr) - a_model).max() > 1e-9:
        no("the linear equalities are not [qc - q_0; -dq_0; Euler integration with a uniform dt]")
    # ---- nonlinear equality h = quat(link, qc) - quat(link, q_t): find the link
    Qh = qc[None] + rng.uniform(-0.3, 0.3, (T, n))
    h_val = _vec(opt.h, xvec(Qh, Z), qc).reshape(T, 4)
    cands = [link] if link is not None else [l for l in robot.link_names if l != robot.get_root_link()]
    found = None
    for cand in cands:
        try:
            chain_ok = len(robot.urdf.get_chain(robot.get_root_link(), cand)) > 0
        except ValueError:
            chain_ok = False
        if not chain_ok:
            continue
        quat_c = np.asarray(robot.get_global_link_quaternion(cand, qc)).reshape(4)
        quat = np.asarray(robot.get_global_link_quaternion(cand, Qh.T)).reshape(4, T).T
        if np.abs(h_val - (quat_c[None] - quat)).max() <= 1e-9:
            found = cand  # several links may share the orientation (fixed joints): the tracking cost below decides
            p_c = np.asarray(robot.get_global_link_position(cand, qc)).reshape(3)
            R_c = np.asarray(robot.get_global_link_rotation(cand, qc))
            spec = _probe_costs(opt, robot, cand, n, T, dt, qc, p_c, R_c, xvec, rng)
            if spec is not None:
                lo_ = up_ = vlo_ = vup_ = sph_ = None
                if guarded:
                    from types import SimpleNamespace

                    from .lowering import GuardSpec

                    arm = SimpleNamespace(guards=None)
                    _probe_arm_guards(real, [m], [robot], [arm], T, lambda Qs, Zs: xvec(Qs[0], np.reshape(Zs[0], (T - 1, n))), [Qc], [Z],
                                      lambda qs: pfull(qs[0]), [qc], poff, extras, rng, no)
                    gs = arm.guards
                    if gs is not None:
                        lo_, up_, vlo_, vup_ = gs.lo, gs.up, gs.vlo, gs.vup
                        if gs.links:
                            sph_ = GuardSpec(None, None, gs.links, gs.link_radii, gs.obstacles)
                return FigureEightSpec(robot, cand, T, dt, spec[0], spec[1], spec[2], params[0][0], q_name, dq_name, lo=lo_, up=up_, spheres=sph_, vlo=vlo_, vup=vup_)
    if found is None:
        no("h(x, p) is not quat(link, qc) - quat(link, q_t) for any link of the robot")
    no(f"the orientation rows match link '{found}' but the cost is not w_path sumsqr(path_in_frame - p(link, Q)) + w_vel sumsqr(dQ)")


def _probe_figure_eight_lead(opt, m, rng_seed, link, no) -> FigureEightSpec:
    """The figure-eight plan with one parameterised joint (RobotModel(param_joints=[...]), example/figure_eight_plan_6dof.py:17-128): decision variables
    are the optimised joints' positions / velocities, the parameterised joint's trajectory and its velocity arrive as parameters `{name}/q/p`,
    `{name}/dq/p` next to `qc` (all joints).  Labels and shapes decide whether the problem can be the family; the linear rows are checked against
    [qc[opt] - q_0; -dq_0; Euler] on the optimised joints; h and f are probed on the FULL joint trajectory (states and parameter joint merged) with
    the routines of the plain family, and verified."""
    name = m.get_name()
    robot = _mirror_robot(m)
    n = robot.ndof
    opt_idx, par = list(robot.optimized_joint_indexes), int(robot.parameter_joint_indexes[0])
    no_ = len(opt_idx)
    q_name, dq_name, qp_name, dqp_name = f"{name}/q/x", f"{name}/dq/x", f"{name}/q/p", f"{name}/dq/p"
    if list(opt.decision_variables.keys()) != [q_name, dq_name]:
        no(f"decision variables must be exactly [{q_name}, {dq_name}], found {list(opt.decision_variables.keys())}")
    n_x, T = _shape(opt.decision_variables[q_name])
    if n_x != no_ or _shape(opt.decision_variables[dq_name]) != (no_, T - 1):
        no("the state blocks must be (optimised joints) x T and x (T - 1)")
    params = [(k, _shape(v)) for k, v in opt.parameters.items() if _shape(v)[0] * _shape(v)[1] > 0]
    if len(params) != 3 or params[0] != (qp_name, (1, T)) or params[1] != (dqp_name, (1, T - 1)) or params[2][1] != (n, 1):
        no(f"parameters must be [{qp_name} (1, T), {dqp_name} (1, T-1), initial configuration ({n}, 1)], found {params}")
    qc_name = params[2][0]
    want = {f"__{name}_initial_configuration_0__": (no_, 1), f"__{name}_initial_configuration_1__": (no_, 1), f"__integrate_model_states_{name}_1__": (no_, T - 1)}
    got = {k: _shape(v) for k, v in opt.lin_eq_constraints.items()}
    if got != want:
        no(f"linear equalities must be {want}, found {got}")
    if len(opt.lin_ineq_constraints) or len(opt.ineq_constraints):
        no("inequality rows are not lowered for the lead-joint variant")
    eq = [(k, _shape(v)) for k, v in opt.eq_constraints.items()]
    if len(eq) != 1 or eq[0][1] != (4, T):
        no(f"expected one nonlinear equality of shape (4, {T}) (the end-effector quaternion lock), found {eq}")
    if int(opt.nx) != no_ * T + no_ * (T - 1) or int(opt.np) != T + (T - 1) + n:
        no("unexpected nx / np")
    rng = np.random.default_rng(rng_seed)

    def split(Q, dQ, qc_):  # full trajectories (T, n), (T-1, n) -> the problem's own (x, p)
        x = np.concatenate([Q[:, opt_idx].reshape(-1), dQ[:, opt_idx].reshape(-1)])
        return x, np.concatenate([Q[:, par], dQ[:, par], np.asarray(qc_, dtype=float).reshape(-1)])

    lo, up = robot.lower_actuated_joint_limits, robot.upper_actuated_joint_limits
    lo, up = (np.asarray(lo), np.asarray(up))
    if lo.size != n:  # limits of the optimised joints only: the probe ranges need no more than a box
        lo, up = np.full(n, -2.0), np.full(n, 2.0)
    qc = 0.5 * (lo + up) + rng.uniform(-1, 1, n) * 0.3 * np.minimum(up - lo, 4.0)
    Qc, Z = np.tile(qc, (T, 1)), np.zeros((T - 1, n))
    # ---- linear rows on the optimised joints
    a0 = _vec(opt.a, *split(Qc, Z, qc))
    if a0.shape != (2 * no_ + no_ * (T - 1),) or np.abs(a0).max() > 1e-12:
        no("a(x, p) does not vanish at q_t = qc, dq = 0")
    d = Z.copy()
    d[1, opt_idx[0]] = 1.0
    dt = -float(_vec(opt.a, *split(Qc, d, qc))[2 * no_ + no_])
    if not (dt > 0):
        no("could not read a positive dt off the integration rows")
    Qr, dQr, pr = rng.normal(size=(T, n)), rng.normal(size=(T - 1, n)), rng.normal(size=n)
    a_model = np.concatenate([pr[opt_idx] - Qr[0, opt_idx], -dQr[0, opt_idx], -(Qr[:-1, opt_idx] + dt * dQr[:, opt_idx] - Qr[1:, opt_idx]).reshape(-1)])
    if np.abs(_vec(opt.a, *split(Qr, dQr, pr)) - a_model).max() > 1e-9:
        no("the linear equalities are not [qc[opt] - q_0; -dq_0; Euler integration with a uniform dt] on the optimised joints")

    class _Full:  # h and f as functions of the full trajectory ((Q, dQ), qc): what the plain family's probes speak
        @staticmethod
        def f(xq, qc_):
            return opt.f(*split(xq[0], xq[1], qc_))

        @staticmethod
        def h(xq, qc_):
            return opt.h(*split(xq[0], xq[1], qc_))

    xfull = lambda Q, dQ: (Q, dQ)  # noqa: E731
    Qh = qc[None] + rng.uniform(-0.3, 0.3, (T, n))
    h_val = _vec(_Full.h, xfull(Qh, Z), qc).reshape(T, 4)
    cands = [link] if link is not None else [l for l in robot.link_names if l != robot.get_root_link()]
    found = None
    for cand in cands:
        try:
            chain_ok = len(robot.urdf.get_chain(robot.get_root_link(), cand)) > 0
        except ValueError:
            chain_ok = False
        if not chain_ok:
            continue
        quat_c = np.asarray(robot.get_global_link_quaternion(cand, qc)).reshape(4)
        quat = np.asarray(robot.get_global_link_quaternion(cand, Qh.T)).reshape(4, T).T
        if np.abs(h_val - (quat_c[None] - quat)).max() <= 1e-9 or np.abs(h_val - (quat - quat_c[None])).max() <= 1e-9:
            found = cand
            p_c = np.asarray(robot.get_global_link_position(cand, qc)).reshape(3)
            R_c = np.asarray(robot.get_global_link_rotation(cand, qc))
            spec = _probe_costs(_Full, robot, cand, n, T, dt, qc, p_c, R_c, xfull, rng)
            if spec is not None:
                lead = {"par": par, "opt": opt_idx, "qp": qp_name, "dqp": dqp_name}
                return FigureEightSpec(robot, cand, T, dt, spec[0], spec[1], spec[2], qc_name, q_name, dq_name, lead=lead)
    if found is None:
        no("h(x, p) is not the quaternion lock of any link of the robot on the merged joint trajectory")
    no(f"the orientation rows match link '{found}' but the cost is not w_path sumsqr(path_in_frame - p(link, Q)) + w_vel sumsqr(dQ)")


def _probe_costs(opt, robot, link, n, T, dt, qc, p_c, R_c, xvec, rng):
    """(w_path, w_vel, local_path (T, 3)) if f is the family's cost for this link, else None."""
    Qc = np.tile(qc, (T, 1))
    Z = np.zeros((T - 1, n))
    f0 = float(_vec(opt.f, xvec(Qc, Z), qc)[0])
    # velocity term: separable quadratic in dQ
    d = Z.copy()
    d[3 % (T - 1), 1 % n] = 0.7
    w_vel = (float(_vec(opt.f, xvec(Qc, d), qc)[0]) - f0) / 0.49
    if not (w_vel >= 0):
        return None
    # tracking term, one knot at a time: f(q_t = q') - f0 = w (|p'|^2 - |p_c|^2) - 2 (w path_t) . (p' - p_c): linear in (w, w path_t)
    K = 6
    probes = qc[None] + rng.uniform(-0.4, 0.4, (K, n))
    P = np.asarray(robot.get_global_link_position(link, probes.T)).reshape(3, K).T
    A = np.concatenate([(np.sum(P * P, 1) - p_c @ p_c)[:, None], -2.0 * (P - p_c[None])], 1)  # (K, 4)
    ws, paths = [], []
    for t in range(T):
        rhs = np.empty(K)
        for k in range(K):
            Q = Qc.copy()
            Q[t] = probes[k]
            rhs[k] = float(_vec(opt.f, xvec(Q, Z), qc)[0]) - f0
        sol, res, rank, _ = np.linalg.lstsq(A, rhs, rcond=None)
        if rank < 4 or np.abs(A @ sol - rhs).max() > 1e-8 * max(1.0, np.abs(rhs).max()):
            return None
        ws.append(sol[0])
        paths.append(sol[1:] / sol[0] if sol[0] != 0 else np.zeros(3))
    ws = np.array(ws)
    if not (ws.min() > 0) or np.abs(ws - ws[0]).max() > 1e-7 * ws[0]:
        return None
    w_path = float(np.median(ws))
    local = (np.array(paths) - p_c[None]) @ R_c  # R_c^T (path_t - p_c), rows
    # ---- verification at a random point and a second parameter vector: the whole cost, as the kernels will evaluate it
    for trial in range(2):
        qc2 = qc + rng.uniform(-0.2, 0.2, n)
        Q = qc2[None] + rng.uniform(-0.3, 0.3, (T, n))
        dQ = rng.normal(size=(T - 1, n))
        pc2 = np.asarray(robot.get_global_link_position(link, qc2)).reshape(3)
        Rc2 = np.asarray(robot.get_global_link_rotation(link, qc2))
        pos = np.asarray(robot.get_global_link_position(link, Q.T)).reshape(3, T).T
        path = pc2[None] + local @ Rc2.T
        f_model = w_path * np.sum((path - pos) ** 2) + w_vel * np.sum(dQ * dQ)
        f_ref = float(_vec(opt.f, xvec(Q, dQ), qc2)[0])
        if abs(f_model - f_ref) > 1e-9 * max(1.0, abs(f_ref)):
            return None
    return w_path, w_vel, np.ascontiguousarray(local)


# ------------------------------------------------------------------------------------------------------------------------------------
# The other families a BASELINE config uses, from the same interface: labels / shapes decide whether a problem *can* be the family,
# its own numeric members are probed for the numbers, and the recovered model is verified against them before anything is returned.
# ------------------------------------------------------------------------------------------------------------------------------------
def _nonempty_params(opt):
    return [(k, _shape(v)) for k, v in opt.parameters.items() if _shape(v)[0] * _shape(v)[1] > 0]


def _links_to_try(robot, link):
    out = []
    for cand in ([link] if link is not None else [l for l in robot.link_names if l != robot.get_root_link()]):
        try:
            if len(robot.urdf.get_chain(robot.get_root_link(), cand)) > 0:
                out.append(cand)
        except ValueError:
            pass
    return out


def _probe_bounds(opt, what, x_of, n, blocks, xr, pr, no):
    """Joint-bound style rows k(x, p) = [z - lo; up - z] (z = x_of(x), one block of n x cols rows per container item, any order): (lo, up) or
    None for a side that has no block.  ``blocks`` = [(label, cols)].  Read at z = 0 and one unit step, then verified at the random point."""
    lo = up = None
    k0 = _vec(opt.k, x_of(np.zeros_like(xr)), pr)
    k1 = _vec(opt.k, x_of(np.ones_like(xr)), pr)
    off = 0
    model = []
    for label, cols in blocks:
        m = n * cols
        c, s = k0[off : off + m].reshape(cols, n), (k1 - k0)[off : off + m].reshape(cols, n)
        if np.abs(c - c[0][None]).max() > 1e-12 * max(1.0, np.abs(c).max()):
            no(f"{what} '{label}': the bound is not the same at every knot")
        if np.abs(s - 1.0).max() <= 1e-12 and lo is None:
            lo = -c[0]
            model.append((xr - lo[None]).reshape(-1))
        elif np.abs(s + 1.0).max() <= 1e-12 and up is None:
            up = c[0]
            model.append((up[None] - xr).reshape(-1))
        else:
            no(f"{what} '{label}' is not a bound block z - lo >= 0 or up - z >= 0 (or a second one of its kind)")
        off += m
    if off != k0.size:
        no(f"{what}: k(x, p) has {k0.size} rows, the containers account for {off}")
    if model and np.abs(_vec(opt.k, x_of(xr), pr) - np.concatenate(model)).max() > 1e-9:
        no(f"{what}: k(x, p) is not the bound rows read off it")
    return lo, up


def probe_ik(opt, rng_seed: int = 12345, link: Optional[str] = None):
    """example/example.py:13-60 behind the reference interface: min w ||q - q_nominal||^2 s.t. p(link, q) = p_goal, lo <= q <= up."""
    from .lowering import IkSpec

    def no(msg):
        raise LoweringError(f"inverse-kinematics probing: {msg}")

    models = list(opt.models or [])
    if len(models) != 1 or not hasattr(models[0], "urdf"):
        no("expected exactly one robot model")
    m = models[0]
    name = m.get_name()
    if list(m.time_derivs) != [0] or len(getattr(m, "param_joints", []) or []) != 0:
        no("robot must have time_derivs=[0] and no parameterised joints")
    q_name = f"{name}/q/x"
    if list(opt.decision_variables.keys()) != [q_name]:
        no(f"decision variables must be exactly [{q_name}]")
    n, T = _shape(opt.decision_variables[q_name])
    if T != 1:
        no("T must be 1")
    params = _nonempty_params(opt)
    if [s for _, s in params] != [(n, 1), (3, 1)]:
        no(f"non-empty parameters must be (q_nominal ({n}, 1), p_goal (3, 1)) in this order, found {params}")
    if len(opt.lin_eq_constraints) or len(opt.ineq_constraints):
        no("linear equality / nonlinear inequality rows are not part of this family")
    eq = [(k, _shape(v)) for k, v in opt.eq_constraints.items()]
    if len(eq) != 1 or eq[0][1] != (3, 1):
        no(f"expected one nonlinear equality of shape (3, 1) (the position goal), found {eq}")
    if any(_shape(v) != (n, 1) for v in opt.lin_ineq_constraints.values()):
        no("linear inequalities must be joint-bound blocks of shape (ndof, 1)")
    robot = _mirror_robot(m)
    if robot.ndof != n:
        no("the robot's ndof does not match the decision variables")
    rng = np.random.default_rng(rng_seed)
    lo_j, up_j = robot.lower_actuated_joint_limits, robot.upper_actuated_joint_limits
    mid, half = 0.5 * (lo_j + up_j), 0.3 * np.minimum(up_j - lo_j, 4.0)
    q, qn, pg = mid + rng.uniform(-1, 1, n) * half, mid + rng.uniform(-1, 1, n) * half, rng.normal(size=3)
    p = np.concatenate([qn, pg])
    lo, up = _probe_bounds(opt, "linear inequality", lambda z: z.reshape(-1), n, [(k, 1) for k in opt.lin_ineq_constraints.keys()], q[None], p, no)
    lo = np.full(n, -1e9) if lo is None else lo
    up = np.full(n, 1e9) if up is None else up
    # cost: w ||q - qn||^2
    if abs(float(_vec(opt.f, qn, p)[0])) > 1e-12:
        no("f does not vanish at q = q_nominal")
    d = np.zeros(n)
    d[0] = 0.5
    w = float(_vec(opt.f, qn + d, p)[0]) / 0.25
    if not (w > 0) or abs(float(_vec(opt.f, q, p)[0]) - w * float(np.sum((q - qn) ** 2))) > 1e-9 * max(1.0, w):
        no("the cost is not w * sumsqr(q - q_nominal)")
    # h = p_goal - p(link, q)
    h = _vec(opt.h, q, p)
    for cand in _links_to_try(robot, link):
        if np.abs(h - (pg - np.asarray(robot.get_global_link_position(cand, q)).reshape(3))).max() <= 1e-9:
            q2, pg2 = mid + rng.uniform(-1, 1, n) * half, rng.normal(size=3)
            h2 = _vec(opt.h, q2, np.concatenate([qn, pg2]))
            if np.abs(h2 - (pg2 - np.asarray(robot.get_global_link_position(cand, q2)).reshape(3))).max() <= 1e-9:
                return IkSpec(robot, cand, w, np.asarray(lo, float), np.asarray(up, float), params[0][0], params[1][0], q_name)
    no("h(x, p) is not p_goal - p(link, q) for any link of the robot")


def probe_torque_mpc(opt, rng_seed: int = 12345, link: Optional[str] = None):
    """BASELINE configs[4] behind the reference interface: x = [Q; dQ; ddQ; TAU] (derivs_align), two Euler integrations, q_0 / dq_0 fixed,
    h = TAU - rnea(Q, dQ, ddQ), effort bounds, f = w_path sumsqr(p(link, Q) - goal) + w_vel sumsqr(dQ) + w_tau sumsqr(TAU)."""
    from .lowering import TorqueSpec

    def no(msg):
        raise LoweringError(f"torque-MPC probing: {msg}")

    models = list(opt.models or [])
    robots = [m for m in models if hasattr(m, "urdf")]
    tasks = [m for m in models if not hasattr(m, "urdf")]
    if len(robots) != 1 or len(tasks) != 1 or len(models) != 2:
        no("expected one robot model and one task model (the joint torques)")
    m, task = robots[0], tasks[0]
    name = m.get_name()
    if list(m.time_derivs) != [0, 1, 2] or len(getattr(m, "param_joints", []) or []) != 0 or list(task.time_derivs) != [0]:
        no("robot must have time_derivs=[0, 1, 2] and no parameterised joints; the task model time_derivs=[0]")
    names = [f"{name}/q/x", f"{name}/dq/x", f"{name}/ddq/x", task.state_optimized_name(0)]
    if list(opt.decision_variables.keys()) != names:
        no(f"decision variables must be exactly {names}, found {list(opt.decision_variables.keys())}")
    n, T = _shape(opt.decision_variables[names[0]])
    if any(_shape(opt.decision_variables[k]) != (n, T) for k in names):
        no("every block must be ndof x T (derivs_align=True)")
    params = _nonempty_params(opt)
    if [s for _, s in params] != [(n, 1), (n, 1), (3, T)]:
        no(f"non-empty parameters must be (qc ({n}, 1), dqc ({n}, 1), goal (3, {T})) in this order, found {params}")
    lin = [(k, _shape(v)) for k, v in opt.lin_eq_constraints.items()]
    kinds = {f"__{name}_fix_configuration_0_0__": ("fix", 0), f"__{name}_fix_configuration_1_0__": ("fix", 1),
             f"__integrate_model_states_{name}_1__": ("int", 1), f"__integrate_model_states_{name}_2__": ("int", 2)}
    if sorted(k for k, _ in lin) != sorted(kinds) or any(s != ((n, 1) if kinds[k][0] == "fix" else (n, T - 1)) for k, s in lin):
        no(f"linear equalities must be fix_configuration of q and dq at t = 0 and integrate_model_states for time_deriv 1 and 2, found {lin}")
    if len(opt.ineq_constraints):
        no("nonlinear inequalities are not lowered")
    eq = [(k, _shape(v)) for k, v in opt.eq_constraints.items()]
    if len(eq) != 1 or eq[0][1] != (n, T):
        no(f"expected one nonlinear equality of shape ({n}, {T}) (the inverse dynamics), found {eq}")
    if any(_shape(v) != (n, T) for v in opt.lin_ineq_constraints.values()):
        no("linear inequalities must be effort-bound blocks of shape (ndof, T)")
    robot = _mirror_robot(m)
    if robot.ndof != n:
        no("the robot's ndof does not match the decision variables")
    rng = np.random.default_rng(rng_seed)
    nT = n * T

    def xvec(Q, dQ, ddQ, TAU):  # (T, n) each -> vec order
        return np.concatenate([Q.reshape(-1), dQ.reshape(-1), ddQ.reshape(-1), TAU.reshape(-1)])

    lo_j, up_j = robot.lower_actuated_joint_limits, robot.upper_actuated_joint_limits
    mid, half = 0.5 * (lo_j + up_j), 0.3 * np.minimum(up_j - lo_j, 4.0)
    qc = mid + rng.uniform(-1, 1, n) * half
    Z = np.zeros((T, n))
    Qc = np.tile(qc, (T, 1))
    p0 = np.concatenate([qc, np.zeros(n), np.zeros(3 * T)])
    x0 = xvec(Qc, Z, Z, Z)
    # ---- linear rows: dt from one probe, then the whole block in the containers' order
    a0 = _vec(opt.a, x0, p0)
    if a0.size != 2 * n + 2 * n * (T - 1) or np.abs(a0).max() > 1e-12:
        no("a(x, p) does not vanish at q_t = qc, dq = ddq = 0")
    off, where = 0, {}
    for k, s in lin:
        where[kinds[k]] = off
        off += s[0] * s[1]
    d = np.zeros_like(x0)
    d[nT + 0] = 1.0  # dq_0[0] enters the first integration row with -dt (read at q = 0, qc = 0: nothing to cancel, dt comes out exactly)
    dt = -float(_vec(opt.a, d, np.zeros_like(p0))[where[("int", 1)]])
    if not (dt > 0):
        no("could not read a positive dt off the integration rows")
    Qr, dQr, ddQr, TAUr = (rng.normal(size=(T, n)) for _ in range(4))
    pr = np.concatenate([rng.normal(size=n), rng.normal(size=n), rng.normal(size=3 * T)])
    rows = {("fix", 0): pr[:n] - Qr[0], ("fix", 1): pr[n : 2 * n] - dQr[0], ("int", 1): -(Qr[:-1] + dt * dQr[:-1] - Qr[1:]).reshape(-1),
            ("int", 2): -(dQr[:-1] + dt * ddQr[:-1] - dQr[1:]).reshape(-1)}
    a_model = np.concatenate([rows[kinds[k]] for k, _ in lin])
    xr = xvec(Qr, dQr, ddQr, TAUr)
    if np.abs(_vec(opt.a, xr, pr) - a_model).max() > 1e-9:
        no("the linear equalities are not [qc - q_0; dqc - dq_0; Euler integration of q and dq with one uniform dt]")
    # ---- effort bounds
    lo, up = _probe_bounds(opt, "linear inequality", lambda z: xvec(Qr, dQr, ddQr, z), n, [(k, T) for k in opt.lin_ineq_constraints.keys()], TAUr, pr, no)
    if (lo is None) != (up is None):
        no("effort limits need both the lower and the upper row block")
    if lo is None:
        lo, up = -1e9 * np.ones(n), 1e9 * np.ones(n)
    # ---- dynamics rows h = TAU - rnea(Q, dQ, ddQ)
    Qh = qc[None] + rng.uniform(-0.3, 0.3, (T, n))
    dQh, ddQh = rng.uniform(-1, 1, (T, n)), rng.uniform(-2, 2, (T, n))
    tau = np.asarray(robot.rnea(Qh.T, dQh.T, ddQh.T)).reshape(n, T).T
    h = _vec(opt.h, xvec(Qh, dQh, ddQh, TAUr), pr).reshape(T, n)
    if np.abs(h - (TAUr - tau)).max() > 1e-9 * max(1.0, np.abs(tau).max()):
        no("h(x, p) is not TAU - rnea(Q, dQ, ddQ)")
    # ---- cost: at q_t = qc, dq = ddq = tau = 0 and goal_t = g:  f = w_path T |p_c - g|^2  ->  w_path and p_c from five goals
    def f_goal(g):
        return float(_vec(opt.f, x0, np.concatenate([qc, np.zeros(n), np.tile(g, T)]))[0])

    f0, e = f_goal(np.zeros(3)), np.eye(3)
    fe = [f_goal(e[i]) for i in range(3)]
    w_path = (f_goal(2.0 * e[0]) - 2.0 * fe[0] + f0) / (2.0 * T)
    if not (w_path > 0):
        no("could not read a positive tracking weight off f")
    p_c = np.array([0.5 * (1.0 - (fe[i] - f0) / (w_path * T)) for i in range(3)])

    fb = f_goal(p_c)  # the tracking term vanishes here: the small weights are read without cancellation against it

    def bump(block, t, j, v):
        dd = np.zeros_like(x0)
        dd[block * nT + t * n + j] = v
        return float(_vec(opt.f, x0 + dd, np.concatenate([qc, np.zeros(n), np.tile(p_c, T)]))[0]) - fb

    w_vel, w_acc, w_tau = bump(1, 3 % T, 1 % n, 0.5) / 0.25, bump(2, 2 % T, 2 % n, 0.5) / 0.25, bump(3, 4 % T, 3 % n, 0.5) / 0.25
    if abs(w_acc) > 1e-12 * max(1.0, w_path) or not (w_vel >= -1e-12 and w_tau > 0):
        no("the cost must be w_path sumsqr(p(link, Q) - goal) + w_vel sumsqr(dQ) + w_tau sumsqr(TAU) (no acceleration term, w_tau > 0)")
    w_vel = max(w_vel, 0.0)
    for cand in _links_to_try(robot, link):
        if np.abs(np.asarray(robot.get_global_link_position(cand, qc)).reshape(3) - p_c).max() > 1e-8:
            continue
        ok = True
        for _ in range(2):  # the whole cost at random points, as the kernels will evaluate it
            Qv = qc[None] + rng.uniform(-0.3, 0.3, (T, n))
            dQv, ddQv, TAUv, G = rng.normal(size=(T, n)), rng.normal(size=(T, n)), rng.normal(size=(T, n)), rng.normal(size=(T, 3))
            pos = np.asarray(robot.get_global_link_position(cand, Qv.T)).reshape(3, T).T
            f_model = w_path * np.sum((pos - G) ** 2) + w_vel * np.sum(dQv**2) + w_tau * np.sum(TAUv**2)
            f_ref = float(_vec(opt.f, xvec(Qv, dQv, ddQv, TAUv), np.concatenate([qc, np.zeros(n), G.reshape(-1)]))[0])
            ok = ok and abs(f_model - f_ref) <= 1e-9 * max(1.0, abs(f_ref))
        if ok:
            return TorqueSpec(robot, cand, T, dt, float(w_path), float(w_vel), float(w_tau), np.asarray(lo, float), np.asarray(up, float))
    no("the tracking term is not sumsqr(p(link, Q) - goal) for any link of the robot")


def probe(opt, link: Optional[str] = None):
    """(family name, spec) of the first structured family the problem is proven to be; LoweringError (with every family's reason) if none."""
    reasons = []
    for family, fn in PROBES.items():
        try:
            return family, fn(opt, link=link)
        except LoweringError as e:
            reasons.append(str(e))
    raise LoweringError("no structured family matches: " + " | ".join(reasons))


def probe_point_mass(opt, rng_seed: int = 12345, link: Optional[str] = None):
    """example/point_mass_mpc.py:88-154 behind the reference interface: a planar point mass, x = [Y; dY] (derivs_align), y_0 / dy_0 fixed,
    Euler integration, symmetric box limits on Y and dY, one obstacle row ||obs_t - y_t||^2 >= safe^2 per knot,
    f = sumsqr(goal - Y) + w_acc sumsqr((dY[:, 1:] - dY[:, :-1]) / dt).  No kinematics: every number is read off the problem's own members."""
    from .lowering import PointMassSpec

    def no(msg):
        raise LoweringError(f"point-mass MPC probing: {msg}")

    models = list(opt.models or [])
    if len(models) != 1 or hasattr(models[0], "urdf"):
        no("expected exactly one task model")
    tm = models[0]
    if int(tm.dim) != 2 or list(tm.time_derivs) != [0, 1]:
        no("task model must be planar (dim 2) with time_derivs=[0, 1]")
    name = tm.get_name()
    y_name, dy_name = tm.state_optimized_name(0), tm.state_optimized_name(1)
    if list(opt.decision_variables.keys()) != [y_name, dy_name]:
        no("decision variables must be exactly the position and velocity trajectories")
    _, T = _shape(opt.decision_variables[y_name])
    if _shape(opt.decision_variables[y_name]) != (2, T) or _shape(opt.decision_variables[dy_name]) != (2, T):
        no("needs derivs_align=True (both blocks 2 x T)")
    if len(opt.eq_constraints):
        no("nonlinear equalities are not part of this family")
    params = [(k, _shape(v)) for k, v in opt.parameters.items()]
    planner = [s for _, s in params] == [(2, 1), (2, 1)]  # example/point_mass_planner.py:27-28: init, goal
    if not planner and [s for _, s in params] != [(2, 1), (2, 1), (2, T), (2, T)]:
        no(f"parameters must be curr (2), dcurr (2), goal (2 x T), obs (2 x T) in this order (or init (2), goal (2) for the planner), found {params}")
    kinds = {f"__{name}_fix_configuration_0_0__": "fix0", f"__{name}_fix_configuration_1_0__": "fix1", f"__integrate_model_states_{name}_1__": "int"}
    lin = [(k, _shape(v)) for k, v in opt.lin_eq_constraints.items()]
    extra = [k for k, _ in lin if k not in kinds]
    if planner and len(extra) == 1:
        kinds[extra[0]] = "final"  # example/point_mass_planner.py:41-43: the final velocity as a user-labelled equality row
    if sorted(k for k, _ in lin) != sorted(kinds) or any(s != ((2, T - 1) if kinds[k] == "int" else (2, 1)) for k, s in lin):
        no(f"linear equalities must be fix_configuration of y and dy at t = 0 and integrate_model_states{' and one final-velocity row' if planner else ''}, found {lin}")
    if [_shape(v) for v in opt.lin_ineq_constraints.values()] != [(2, T)] * 4:
        no("need enforce_model_limits for time_deriv 0 and 1 (four blocks of shape (2, T))")
    if [_shape(v) for v in opt.ineq_constraints.values()] != [(1, 1)] * T:
        no(f"expected {T} scalar obstacle rows")
    rng = np.random.default_rng(rng_seed)
    m = 2 * T

    def xvec(Y, dY):  # (T, 2) each
        return np.concatenate([Y.reshape(-1), dY.reshape(-1)])

    def pvec(c, dc, G, O):  # the planner's vector is [init; goal]: dc and O do not exist there, G is one point
        return np.concatenate([c, G.reshape(-1)[:2]]) if planner else np.concatenate([c, dc, G.reshape(-1), O.reshape(-1)])

    Zt = np.zeros((T, 2))
    z_p = pvec(np.zeros(2), np.zeros(2), Zt, Zt)
    d = np.zeros(2 * m)
    d[m] = 1.0
    off, where = 0, {}
    for k, s in lin:
        where[kinds[k]] = off
        off += s[0] * s[1]
    dt = -float(_vec(opt.a, d, z_p)[where["int"]])
    if not (dt > 0):
        no("could not read a positive dt off the integration rows")
    Yr, dYr, Gr, Or = (rng.normal(size=(T, 2)) for _ in range(4))
    cr, dcr = rng.normal(size=2), rng.normal(size=2)
    if planner:
        Gr = np.tile(Gr[:1], (T, 1))  # one goal point
    xr, pr = xvec(Yr, dYr), pvec(cr, dcr, Gr, Or)
    rows = {"fix0": cr - Yr[0], "fix1": (0.0 if planner else dcr) - dYr[0], "int": -(Yr[:-1] + dt * dYr[:-1] - Yr[1:]).reshape(-1)}
    a_r = _vec(opt.a, xr, pr)
    if planner:
        sgn = a_r[where["final"]] / dYr[-1, 0]
        if abs(abs(sgn) - 1.0) > 1e-12:
            no("the extra equality row is not the final velocity")
        rows["final"] = np.sign(sgn) * dYr[-1]
    if np.abs(a_r - np.concatenate([rows[kinds[k]] for k, _ in lin])).max() > 1e-9:
        no("the linear equalities are not [curr - y_0; dcurr - dy_0; Euler integration with a uniform dt]" if not planner else
           "the linear equalities are not [init - y_0; -dy_0; Euler integration with a uniform dt; dy_{T-1}]")
    # ---- box limits: every block has slope +-1 on exactly one of Y, dY and one constant
    k0 = _vec(opt.k, np.zeros(2 * m), pr)
    sY, sD = _vec(opt.k, xvec(np.ones((T, 2)), Zt), pr) - k0, _vec(opt.k, xvec(Zt, np.ones((T, 2))), pr) - k0
    if k0.size != 4 * m:
        no("k(x, p) must have 4 x 2 T rows")
    lim, model = {}, []
    for b in range(4):
        sl = slice(b * m, (b + 1) * m)
        c = k0[sl]
        if np.abs(c - c[0]).max() > 0:
            no("box limits must be one scalar per block")
        for which, s_this, s_other, Z in ((0, sY[sl], sD[sl], Yr), (1, sD[sl], sY[sl], dYr)):
            if np.abs(s_other).max() == 0 and np.abs(np.abs(s_this) - 1.0).max() == 0 and np.abs(s_this - s_this[0]).max() == 0:
                side = "l" if s_this[0] > 0 else "r"  # z - lo >= 0  /  up - z >= 0
                if (which, side) in lim:
                    no("two limit blocks of the same kind")
                lim[(which, side)] = -c[0] if side == "l" else c[0]
                model.append(s_this[0] * Z.reshape(-1) + c[0])
    if set(lim) != {(0, "l"), (0, "r"), (1, "l"), (1, "r")}:
        no("need a lower and an upper box limit on both the positions and the velocities")
    if lim[(0, "l")] != -lim[(0, "r")] or lim[(1, "l")] != -lim[(1, "r")]:
        no("box limits must be symmetric")
    if np.abs(_vec(opt.k, xr, pr) - np.concatenate(model)).max() > 1e-12:
        no("k(x, p) is not the box rows read off it")
    if planner:
        return _probe_planner_rows(opt, T, dt, lim, params, y_name, dy_name, xvec, Yr, dYr, xr, pr, rng, no)
    # ---- obstacle rows g_t = |obs_t - y_t|^2 - safe^2
    g_at = _vec(opt.g, xvec(Or, dYr), pr)
    if g_at.size != T or np.abs(g_at - g_at[0]).max() > 0 or not (g_at[0] < 0):
        no("the inequality rows are not ||obs_t - y_t||^2 >= safe^2 with one radius")
    safe_sq = -float(g_at[0])
    if np.abs(_vec(opt.g, xr, pr) - (np.sum((Or - Yr) ** 2, 1) - safe_sq)).max() > 1e-12 * max(1.0, safe_sq):
        no("the inequality rows are not ||obs_t - y_t||^2 >= safe^2")
    # ---- cost
    if abs(float(_vec(opt.f, xvec(Gr, Zt), pr)[0])) > 1e-12:
        no("f does not vanish at Y = goal, dY = 0")
    dd = Zt.copy()
    dd[T // 2, 0] = 1.0  # an interior velocity entry appears in two differences
    w_acc = float(_vec(opt.f, xvec(Gr, dd), pr)[0]) * dt * dt / 2.0
    f_model = np.sum((Gr - Yr) ** 2) + w_acc * np.sum(((dYr[1:] - dYr[:-1]) / dt) ** 2)
    if not (w_acc > 0) or abs(float(_vec(opt.f, xr, pr)[0]) - f_model) > 1e-9 * max(1.0, abs(f_model)):
        no("the cost is not sumsqr(goal - Y) + w_acc sumsqr((dY[:, 1:] - dY[:, :-1]) / dt)")
    return PointMassSpec(T, dt, float(w_acc), float(lim[(0, "r")]), float(lim[(1, "r")]), float(np.sqrt(safe_sq)), tuple(k for k, _ in params), y_name, dy_name)


def _probe_planner_rows(opt, T, dt, lim, params, y_name, dy_name, xvec, Yr, dYr, xr, pr, rng, no):
    """example/point_mass_planner.py:45-66: the obstacle is a constant (read off g as the stationary point of every row: g_t is a quadratic
    in y_t alone with Hessian 2 I), the tracking cost sits on the last knot, the velocities are penalised."""
    from .lowering import PointMassSpec

    Zt = np.zeros((T, 2))
    g0 = _vec(opt.g, xvec(Zt, dYr), pr)
    ex, ey = Zt.copy(), Zt.copy()
    ex[:, 0], ey[:, 1] = 1.0, 1.0
    gx, gy = _vec(opt.g, xvec(ex, dYr), pr), _vec(opt.g, xvec(ey, dYr), pr)
    if g0.size != T:
        no(f"expected {T} obstacle rows")
    # g(y) = |o - y|^2 - s: g(e_i) - g(0) = 1 - 2 o_i
    ox, oy = (1.0 - (gx - g0)) / 2.0, (1.0 - (gy - g0)) / 2.0
    if np.abs(ox - ox[0]).max() > 1e-12 or np.abs(oy - oy[0]).max() > 1e-12:
        no("the obstacle rows do not share one constant obstacle")
    obstacle = np.array([float(ox[0]), float(oy[0])])
    safe_sq = float(np.sum(obstacle * obstacle) - g0[0])
    if not (safe_sq > 0) or np.abs(g0 - g0[0]).max() > 1e-12:
        no("the obstacle rows do not share one radius")
    p2 = pr + rng.normal(size=pr.size)
    if np.abs(_vec(opt.g, xr, p2) - (np.sum((obstacle[None, :] - Yr) ** 2, 1) - safe_sq)).max() > 1e-12 * max(1.0, safe_sq):
        no("the inequality rows are not ||obstacle - y_t||^2 >= safe^2 with a constant obstacle")
    # ---- cost: sumsqr(goal - y_{T-1}) + w_vel sumsqr(dY) + w_acc sumsqr((dY[:, 1:] - dY[:, :-1]) / dt)
    goal = pr[2:4]
    Yg = Yr.copy()
    Yg[-1] = goal
    if abs(float(_vec(opt.f, xvec(Yg, Zt), pr)[0])) > 1e-12:
        no("f does not vanish at y_{T-1} = goal, dY = 0")
    d_all = np.ones((T, 2))  # constant velocities: no acceleration
    w_vel = float(_vec(opt.f, xvec(Yg, d_all), pr)[0]) / (2.0 * T)
    dd = Zt.copy()
    dd[T // 2, 0] = 1.0  # an interior velocity entry: once in the velocity term, in two differences
    w_acc = (float(_vec(opt.f, xvec(Yg, dd), pr)[0]) - w_vel) * dt * dt / 2.0
    f_model = np.sum((goal - Yr[-1]) ** 2) + w_vel * np.sum(dYr * dYr) + w_acc * np.sum(((dYr[1:] - dYr[:-1]) / dt) ** 2)
    if not (w_acc > 0 and w_vel >= 0) or abs(float(_vec(opt.f, xr, pr)[0]) - f_model) > 1e-9 * max(1.0, abs(f_model)):
        no("the cost is not sumsqr(goal - y_{T-1}) + w_vel sumsqr(dY) + w_acc sumsqr((dY[:, 1:] - dY[:, :-1]) / dt)")
    names = tuple(k for k, _ in params)
    return PointMassSpec(T, dt, float(w_acc), float(lim[(0, "r")]), float(lim[(1, "r")]), float(np.sqrt(safe_sq)), names, y_name, dy_name,
                         planner={"w_vel": float(w_vel), "obstacle": obstacle, "init": names[0], "goal": names[1]})


def probe_multi_arm(opt, rng_seed: int = 12345, link: Optional[str] = None):
    """example/dual_arm.py:17-129 as shipped, behind the reference interface: one position-tracking problem per robot (q_0 fixed to a
    parameter, dq_0 free, Euler integration, path_t = p(link, qc) + offset_t), summed.  The arms are probed one at a time with the others held
    at their fixed configuration.  Round 3: the inequality rows of BASELINE configs[3] as stated -- enforce_model_limits (builder.py:471-509,
    blocks "__{name}_model_limit_0___l/_r") and sphere_collision_avoidance_constraints (builder.py:366-417, rows
    "sphere_col_avoid_{t}_{link}_{obstacle}", parameters "{link}_radii", "{obstacle}_position", "{obstacle}_radii") -- are recognised from their
    labels, attributed to an arm numerically, read off k and g, and verified against them (_probe_arm_guards)."""
    from .lowering import ArmSpec, MultiArmSpec

    def no(msg):
        raise LoweringError(f"multi-arm probing: {msg}")

    models = list(opt.models or [])
    if len(models) < 2 or not all(hasattr(m, "urdf") for m in models):
        no("expected two or more robot models and nothing else")
    if len(opt.eq_constraints):
        no("nonlinear equality rows are not part of this family")
    names, T = [], None
    for m in models:
        if list(m.time_derivs) != [0, 1] or len(getattr(m, "param_joints", []) or []) != 0:
            no("every robot must have time_derivs=[0, 1] and no parameterised joints")
        names += [f"{m.get_name()}/q/x", f"{m.get_name()}/dq/x"]
    if list(opt.decision_variables.keys()) != names:
        no(f"decision variables must be exactly {names}")
    robots = [_mirror_robot(m) for m in models]
    shapes = [_shape(opt.decision_variables[k]) for k in names]
    T = shapes[0][1]
    for r, sq, sd in zip(robots, shapes[0::2], shapes[1::2]):
        if sq != (r.ndof, T) or sd != (r.ndof, T - 1):
            no("all robots must share T and use derivs_align=False")
    params_all = _nonempty_params(opt)
    guarded = len(opt.ineq_constraints) > 0 or len(opt.lin_ineq_constraints) > 0
    params = params_all[: len(models)]  # the initial configurations come first (dual_arm.py:31-32 before any sphere parameter, builder.py:391-405)
    if (not guarded and len(params_all) != len(models)) or [s for _, s in params] != [(r.ndof, 1) for r in robots]:
        no(f"expected one initial-configuration parameter per robot, in the robots' order, found {params_all}")
    want = {}
    for m, r in zip(models, robots):
        want[f"__{m.get_name()}_fix_configuration_0_0__"] = (r.ndof, 1)
        want[f"__integrate_model_states_{m.get_name()}_1__"] = (r.ndof, T - 1)
    lin = [(k, _shape(v)) for k, v in opt.lin_eq_constraints.items()]
    if dict(lin) != want:
        no(f"linear equalities must be {want} (dq_0 is free in this family), found {dict(lin)}")
    rng = np.random.default_rng(rng_seed)
    ns = [r.ndof for r in robots]
    qcs = []
    for r in robots:
        lo, up = r.lower_actuated_joint_limits, r.upper_actuated_joint_limits
        qcs.append(0.5 * (lo + up) + rng.uniform(-1, 1, r.ndof) * 0.3 * np.minimum(up - lo, 4.0))

    def xvec(Qs, dQs):
        return np.concatenate([np.concatenate([Q.reshape(-1), dQ.reshape(-1)]) for Q, dQ in zip(Qs, dQs)])

    Qc = [np.tile(q, (T, 1)) for q in qcs]
    Zs = [np.zeros((T - 1, n)) for n in ns]
    poff, o = {}, 0
    for k, v in opt.parameters.items():
        m_, n_ = _shape(v)
        poff[k] = (o, m_ * n_)
        o += m_ * n_
    np_total = o
    # parameter vector with the initial configurations in place and every other parameter at a harmless value (radii 0.05, obstacles far away)
    pc = np.zeros(np_total)
    for (k, _), q in zip(params, qcs):
        pc[poff[k][0] : poff[k][0] + q.size] = q
    for k, _ in params_all[len(models):]:
        a0, l0 = poff[k]
        pc[a0 : a0 + l0] = 0.05 if l0 == 1 else 5.0 + np.arange(l0)
    qc_slices = [slice(poff[k][0], poff[k][0] + q.size) for (k, _), q in zip(params, qcs)]

    def pvec(qlist):
        pv = pc.copy()
        for sl, q in zip(qc_slices, qlist):
            pv[sl] = q
        return pv

    x0 = xvec(Qc, Zs)
    # ---- linear rows
    off, where = 0, {}
    for k, s in lin:
        where[k] = off
        off += s[0] * s[1]
    d = np.zeros_like(x0)
    d[ns[0] * T] = 1.0  # dq_0[0] of the first robot
    dt = -float(_vec(opt.a, d, np.zeros_like(pc))[where[f"__integrate_model_states_{models[0].get_name()}_1__"]])
    if not (dt > 0):
        no("could not read a positive dt off the integration rows")
    Qr, dQr = [rng.normal(size=(T, n)) for n in ns], [rng.normal(size=(T - 1, n)) for n in ns]
    pr = pvec([rng.normal(size=n) for n in ns])
    rows = {}
    for m, n, Q, dQ, sl in zip(models, ns, Qr, dQr, qc_slices):
        rows[f"__{m.get_name()}_fix_configuration_0_0__"] = pr[sl] - Q[0]
        rows[f"__integrate_model_states_{m.get_name()}_1__"] = -(Q[:-1] + dt * dQ - Q[1:]).reshape(-1)
    if np.abs(_vec(opt.a, xvec(Qr, dQr), pr) - np.concatenate([rows[k] for k, _ in lin])).max() > 1e-9:
        no("the linear equalities are not [qc - q_0; Euler integration with one uniform dt] per robot")
    # ---- costs, one arm at a time
    f0 = float(_vec(opt.f, x0, pc)[0])
    arms = []
    for i, (m, r, n) in enumerate(zip(models, robots, ns)):
        def with_arm(Q, dQ):
            Qs, dQs = list(Qc), list(Zs)
            Qs[i], dQs[i] = Q, dQ
            return float(_vec(opt.f, xvec(Qs, dQs), pc)[0])

        dd = Zs[i].copy()
        dd[3 % (T - 1), 1 % n] = 0.7
        w_vel = (with_arm(Qc[i], dd) - f0) / 0.49
        if not (w_vel >= 0):
            no(f"robot '{m.get_name()}': no joint-velocity term")
        K = 6
        probes = qcs[i][None] + rng.uniform(-0.4, 0.4, (K, n))
        found = None
        for cand in _links_to_try(r, link):
            p_c = np.asarray(r.get_global_link_position(cand, qcs[i])).reshape(3)
            P = np.asarray(r.get_global_link_position(cand, probes.T)).reshape(3, K).T
            A = np.concatenate([(np.sum(P * P, 1) - p_c @ p_c)[:, None], -2.0 * (P - p_c[None])], 1)
            ws, paths, ok = [], [], True
            for t in [1] + list(range(1, T)):  # q_0 is fixed: its tracking error is a constant of the problem.  Knot 1 first, on its own:
                # a wrong candidate link is dropped after K evaluations instead of K (T - 1)
                if len(ws) == 1 and t == 1:
                    ws, paths = [], []
                rhs = np.empty(K)
                for k in range(K):
                    Q = Qc[i].copy()
                    Q[t] = probes[k]
                    rhs[k] = with_arm(Q, Zs[i]) - f0
                sol, _, rank, _ = np.linalg.lstsq(A, rhs, rcond=None)
                if rank < 4 or np.abs(A @ sol - rhs).max() > 1e-8 * max(1.0, np.abs(rhs).max()) or not (sol[0] > 0):
                    ok = False
                    break
                ws.append(sol[0])
                paths.append(sol[1:] / sol[0])
            if not ok or np.abs(np.array(ws) - ws[0]).max() > 1e-7 * ws[0]:
                continue
            w_path = float(np.median(ws))
            offsets = np.concatenate([np.zeros((1, 3)), np.array(paths) - p_c[None]])
            # knot 0: f0 contains w |p_c - (p_c + offset_0)|^2 of every arm; it is pinned by the verification below together with the rest
            found = (cand, w_path, offsets)
            break
        if found is None:
            no(f"robot '{m.get_name()}': the cost is not w_path sumsqr(p(link, Q) - (p(link, qc) + offsets)) + w_vel sumsqr(dQ) for any link")
        arms.append(ArmSpec(r, found[0], found[1], float(w_vel), np.ascontiguousarray(found[2]), params[i][0], names[2 * i], names[2 * i + 1], None))
    # offset of knot 0 (the fixed knot): read off f at the fixed configuration, one arm at a time is not possible -- it is a constant of the
    # whole problem.  With every other term known, sum_i w_i |offset_0,i|^2 = f0 - (known part); only its total matters to the optimiser
    # (a constant), so it is attributed to the first arm's x component for the reported objective and then verified.
    known = sum(a.w_path * np.sum(a.offsets[1:] ** 2) for a in arms)
    rest = f0 - known
    if rest < -1e-9 * max(1.0, abs(f0)):
        no("the cost at the fixed configuration is smaller than the tracking terms read off it")
    arms[0].offsets[0, 0] = np.sqrt(max(rest, 0.0) / arms[0].w_path)
    for _ in range(2):  # the whole cost at random points and other fixed configurations, as the kernels will evaluate it
        q2 = [q + rng.uniform(-0.2, 0.2, q.size) for q in qcs]
        Qv = [q[None] + rng.uniform(-0.3, 0.3, (T, q.size)) for q in q2]
        for Q, q in zip(Qv, q2):
            Q[0] = q  # the kernels eliminate the fixed knot
        dQv = [rng.normal(size=(T - 1, n)) for n in ns]
        f_model = 0.0
        for a, Q, dQ, q in zip(arms, Qv, dQv, q2):
            pos = np.asarray(a.robot.get_global_link_position(a.link, Q.T)).reshape(3, T).T
            p_c = np.asarray(a.robot.get_global_link_position(a.link, q)).reshape(3)
            f_model += a.w_path * np.sum((pos - (p_c[None] + a.offsets)) ** 2) + a.w_vel * np.sum(dQ * dQ)
        f_ref = float(_vec(opt.f, xvec(Qv, dQv), pvec(q2))[0])
        if abs(f_model - f_ref) > 1e-9 * max(1.0, abs(f_ref)):
            no("the summed cost does not match the per-arm models read off it")
    if guarded:
        _probe_arm_guards(opt, models, robots, arms, T, xvec, Qc, Zs, pvec, qcs, poff, [k for k, _ in params_all[len(models):]], rng, no)
    return MultiArmSpec(T, dt, arms)


def _probe_arm_guards(opt, models, robots, arms, T, xvec, Qc, Zs, pvec, qcs, poff, extra_params, rng, no):
    """Joint-limit blocks and sphere-clearance rows of a multi-arm problem, from labels and numbers only.  Fills arms[i].guards."""
    from .lowering import GuardSpec

    ns = [r.ndof for r in robots]
    x0, p0 = xvec(Qc, Zs), pvec(qcs)
    used_params = set()
    # ---- joint limits: "__{name}_model_limit_0___l" = x - lo, "..._r" = up - x, each vec of an n x T block (builder.py:334-335, 471-509)
    koff, o = {}, 0
    for k, v in opt.lin_ineq_constraints.items():
        m_, n_ = _shape(v)
        koff[k] = (o, m_, n_)
        o += m_ * n_
    lims = [None] * len(models)
    expected = set()
    for i, m in enumerate(models):
        lab = f"__{m.get_name()}_model_limit_0__"
        if lab + "_l" in koff or lab + "_r" in koff:
            if not (lab + "_l" in koff and lab + "_r" in koff) or koff[lab + "_l"][1:] != (ns[i], T) or koff[lab + "_r"][1:] != (ns[i], T):
                no(f"robot '{m.get_name()}': joint limits need both blocks, n x T each")
            expected |= {lab + "_l", lab + "_r"}
            k0 = _vec(opt.k, np.zeros_like(x0), p0)
            lo = -k0[koff[lab + "_l"][0] : koff[lab + "_l"][0] + ns[i] * T].reshape(T, ns[i])
            up = k0[koff[lab + "_r"][0] : koff[lab + "_r"][0] + ns[i] * T].reshape(T, ns[i])
            if np.abs(lo - lo[0]).max() > 0 or np.abs(up - up[0]).max() > 0 or not (lo[0] < up[0]).all():
                no(f"robot '{m.get_name()}': the limit rows are not one bound pair per joint over the whole trajectory")
            lims[i] = (lo[0].copy(), up[0].copy())
    # ---- joint-velocity limits: "__{name}_model_limit_1___l" / "_r", each vec of an n x (T-1) block over the dq states
    vlims = [None] * len(models)
    for i, m in enumerate(models):
        lab = f"__{m.get_name()}_model_limit_1__"
        if lab + "_l" in koff or lab + "_r" in koff:
            if not (lab + "_l" in koff and lab + "_r" in koff) or koff[lab + "_l"][1:] != (ns[i], T - 1) or koff[lab + "_r"][1:] != (ns[i], T - 1):
                no(f"robot '{m.get_name()}': joint-velocity limits need both blocks, n x (T-1) each")
            expected |= {lab + "_l", lab + "_r"}
            k0 = _vec(opt.k, np.zeros_like(x0), p0)
            m_ = ns[i] * (T - 1)
            vlo = -k0[koff[lab + "_l"][0] : koff[lab + "_l"][0] + m_].reshape(T - 1, ns[i])
            vup = k0[koff[lab + "_r"][0] : koff[lab + "_r"][0] + m_].reshape(T - 1, ns[i])
            if np.abs(vlo - vlo[0]).max() > 0 or np.abs(vup - vup[0]).max() > 0 or not (vlo[0] < vup[0]).all():
                no(f"robot '{m.get_name()}': the velocity-limit rows are not one bound pair per joint over the whole trajectory")
            vlims[i] = (vlo[0].copy(), vup[0].copy())
    if set(koff) != expected:
        no(f"linear inequality blocks other than the robots' joint and joint-velocity limits: {sorted(set(koff) - expected)}")
    if expected:
        Qr = [rng.normal(size=(T, n)) for n in ns]
        kr = _vec(opt.k, xvec(Qr, Zs), pvec([rng.normal(size=n) for n in ns]))
        for i, m in enumerate(models):
            if lims[i] is None:
                continue
            lab = f"__{m.get_name()}_model_limit_0__"
            a0, b0 = koff[lab + "_l"][0], koff[lab + "_r"][0]
            if (np.abs(kr[a0 : a0 + ns[i] * T].reshape(T, ns[i]) - (Qr[i] - lims[i][0][None])).max() > 1e-9
                    or np.abs(kr[b0 : b0 + ns[i] * T].reshape(T, ns[i]) - (lims[i][1][None] - Qr[i])).max() > 1e-9):
                no(f"robot '{m.get_name()}': the limit rows are not [Q - lo; up - Q]")
        Zr = [rng.normal(size=np.shape(z)) for z in Zs]
        kv = _vec(opt.k, xvec(Qr, Zr), pvec([rng.normal(size=n) for n in ns]))
        for i, m in enumerate(models):
            if vlims[i] is None:
                continue
            lab = f"__{m.get_name()}_model_limit_1__"
            a0, b0, m_ = koff[lab + "_l"][0], koff[lab + "_r"][0], ns[i] * (T - 1)
            if (np.abs(kv[a0 : a0 + m_].reshape(T - 1, ns[i]) - (np.reshape(Zr[i], (T - 1, ns[i])) - vlims[i][0][None])).max() > 1e-9
                    or np.abs(kv[b0 : b0 + m_].reshape(T - 1, ns[i]) - (vlims[i][1][None] - np.reshape(Zr[i], (T - 1, ns[i])))).max() > 1e-9):
                no(f"robot '{m.get_name()}': the velocity-limit rows are not [dQ - lo; up - dQ]")
    # ---- sphere clearances: "sphere_col_avoid_{t}_{link}_{obstacle}", one scalar row each (builder.py:407-415)
    glabels = list(opt.ineq_constraints.keys())
    sph = [dict() for _ in models]
    if glabels:
        g0 = _vec(opt.g, x0, p0)
        if g0.size != len(glabels):
            no("the nonlinear inequality rows are not scalar rows")
        owner = np.full(len(glabels), -1)
        for i in range(len(models)):  # which arm does a row belong to?  Move that arm alone and see which rows answer
            Qs = list(Qc)
            Qs[i] = Qc[i] + rng.uniform(-0.3, 0.3, Qc[i].shape)
            moved = np.abs(_vec(opt.g, xvec(Qs, Zs), p0) - g0) > 1e-12
            if (owner[moved] >= 0).any():
                no("a sphere row depends on two robots")
            owner[moved] = i
        if (owner < 0).any():
            no("an inequality row depends on no robot's configuration")
        for r_, lab in enumerate(glabels):
            i = int(owner[r_])
            pre = "sphere_col_avoid_"
            if not lab.startswith(pre):
                no(f"inequality '{lab}' is not a sphere-clearance row")
            t_s, _, rest = lab[len(pre):].partition("_")
            hit = None
            for ln in sorted(robots[i].link_names, key=len, reverse=True):
                if rest.startswith(ln + "_"):
                    hit = (ln, rest[len(ln) + 1:])
                    break
            if hit is None or not t_s.isdigit():
                no(f"inequality '{lab}': no link of robot '{models[i].get_name()}' in the label")
            sph[i][(int(t_s), hit[0], hit[1])] = r_
    for i, (m, r) in enumerate(zip(models, robots)):
        links, obst = [], []
        for (t, ln, on) in sph[i]:
            if ln not in links:
                links.append(ln)
            if on not in obst:
                obst.append(on)
        if sph[i] and len(sph[i]) != T * len(links) * len(obst):
            no(f"robot '{m.get_name()}': sphere rows must cover every (knot, link, obstacle) combination")
        lrad_names = []
        for ln in links:  # "{link}_radii", possibly behind a prefix (this repo's additive extension for two robots of one URDF): the parameter that moves this arm's rows
            cands = [k for k in extra_params if k.endswith(ln + "_radii")]
            pick = None
            for k in cands:
                pv = p0.copy()
                pv[poff[k][0]] += 0.01
                ch = np.abs(_vec(opt.g, x0, pv) - g0) > 1e-12
                mine = np.array([sph[i][(t, ln, on)] for t in range(T) for on in obst])
                if ch[mine].all() and ch.sum() == mine.size:
                    pick = k
            if pick is None:
                no(f"robot '{m.get_name()}': no radius parameter for link '{ln}'")
            lrad_names.append(pick)
        obs_names = []
        for on in obst:
            if on + "_position" not in poff or on + "_radii" not in poff or poff[on + "_position"][1] != 3:
                no(f"obstacle '{on}': parameters '{on}_position' (3) and '{on}_radii' expected")
            obs_names.append((on + "_position", on + "_radii"))
        used_params |= set(lrad_names) | set(x for ob in obs_names for x in ob)
        if sph[i]:
            # verification at random configurations and parameters: g = ||p_link(q_t) - o||^2 - (r_link + r_o)^2, computed with this package's kinematics
            for _ in range(2):
                Qs = list(Qc)
                Qs[i] = qcs[i][None] + rng.uniform(-0.5, 0.5, (T, ns[i]))
                pv = p0.copy()
                for k in lrad_names + [b for _, b in obs_names]:
                    pv[poff[k][0]] = rng.uniform(0.02, 0.2)
                for a_, _ in obs_names:
                    pv[poff[a_][0] : poff[a_][0] + 3] = rng.uniform(-0.8, 0.8, 3)
                gv = _vec(opt.g, xvec(Qs, Zs), pv)
                for ln, lk in zip(links, lrad_names):
                    P = np.asarray(r.get_global_link_position(ln, Qs[i].T)).reshape(3, T).T
                    for (pk, rk) in obs_names:
                        on = pk[: -len("_position")]
                        want = np.sum((P - pv[poff[pk][0] : poff[pk][0] + 3][None]) ** 2, 1) - (pv[poff[lk][0]] + pv[poff[rk][0]]) ** 2
                        got = np.array([gv[sph[i][(t, ln, on)]] for t in range(T)])
                        if np.abs(want - got).max() > 1e-9:
                            no(f"robot '{m.get_name()}': rows of link '{ln}' / obstacle '{on}' are not ||p_link(q_t) - o||^2 - (r_link + r_o)^2")
        if lims[i] is not None or sph[i] or vlims[i] is not None:
            lo, up = lims[i] if lims[i] is not None else (None, None)
            vlo, vup = vlims[i] if vlims[i] is not None else (None, None)
            arms[i].guards = GuardSpec(lo, up, links, lrad_names, obs_names, vlo, vup)
    if set(extra_params) != used_params:
        no(f"parameters that belong to no recognised row: {sorted(set(extra_params) - used_params)}")


PROBES = {"figure_eight": probe_figure_eight, "torque_mpc": probe_torque_mpc, "ik": probe_ik, "point_mass": probe_point_mass, "multi_arm": probe_multi_arm}
