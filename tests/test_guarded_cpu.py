"""Synthetic config 4 (8(d) C4) on the CPU: builder counts of example/dual_arm.py + enforce_model_limits +
sphere_collision_avoidance_constraints, lowering to the guarded position-tracking family, and the augmented-Lagrangian port
(oracle/guarded.py, the state machine the HIP kernels run) against the golden optima that scipy SLSQP (reference wiring:
inequality rows passed as g >= 0, solver.py:672-679) and the port agree on; reference-form KKT on the literal layout."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, KUKA_KIN
from oracle.guarded import Guards, guard_values, solve_free_al
from oracle.problems import GuardedDualArmNLP, dual_arm_offsets
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import FoldedChain

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.dual_arm import N_OBSTACLES, SPHERE_LINKS, obstacle_parameters, setup_solver  # noqa: E402

QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])


def _robots():
    rl = OracleRobot(KUKA_KIN, name="kukal")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr = OracleRobot(KUKA_KIN, name="kukar")
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    return rl, rr


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "guard_golden.npz"))


def test_link_attachments_match_literal_forward_kinematics():
    rl, _ = _robots()
    ch = FoldedChain(rl, "end_effector_ball")
    rng = np.random.default_rng(0)
    Q = rng.uniform(-1, 1, (3, 7))
    C, J = ch.link_positions(Q, SPHERE_LINKS)
    for i, q in enumerate(Q):
        for li, ln in enumerate(SPHERE_LINKS):
            assert np.abs(C[i, li] - rl.get_global_link_position(ln, q)).max() < 1e-15
            assert np.abs(J[i, li] - rl.get_global_link_linear_jacobian(ln, q)).max() < 1e-15


def test_builder_counts_and_lowering():
    from optas_amd.lowering import OH_KIND_MULTI_ARM, LoweringError, lower
    from optas_amd.optimization import NonlinearCostNonlinearConstraints

    (kl, kr), o = setup_solver(T=100, build_only=True, limits=True, collision=True)
    assert isinstance(o, NonlinearCostNonlinearConstraints)
    # SURVEY 8(a) H4 synthetic: nx = 2786, na = 1400, ng = 4800 (+ nk = 2 * 2 * 7 * 100 limit rows)
    assert (o.nx, o.np, o.nk, o.na, o.ng, o.nh) == (2786, 70, 2800, 1400, 4800, 0) and o.nv == 2800 + 4800 + 2 * 1400
    labels = list(o.ineq_constraints.keys())
    assert labels[0] == "sphere_col_avoid_0_end_effector_ball_kukal_obs0" and labels[1] == "sphere_col_avoid_0_end_effector_ball_kukal_obs1"
    assert list(o.parameters.keys())[4:12] == ["qcl", "qcr", "kukal_end_effector_ball_radii", "kukal_lwr_arm_7_link_radii", "kukal_lwr_arm_5_link_radii",
                                               "kukal_lwr_arm_6_link_radii", "kukal_obs0_position", "kukal_obs0_radii"]  # builder.py:391-405 order
    kind, spec = lower(o)
    assert kind == OH_KIND_MULTI_ARM and all(a.guards is not None for a in spec.arms)
    g = spec.arms[1].guards
    assert g.links == SPHERE_LINKS and len(g.obstacles) == N_OBSTACLES and g.obstacles[0] == ("kukar_obs0_position", "kukar_obs0_radii")
    rl, _ = _robots()
    assert np.array_equal(g.lo, rl.lower_actuated_joint_limits) and np.array_equal(g.up, rl.upper_actuated_joint_limits)
    # the reference's naming collides for two robots built from the same URDF (sx_container.py:50-51): kept
    from optas_amd.builder import OptimizationBuilder
    import optas_amd

    r1 = optas_amd.RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="a")
    r2 = optas_amd.RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="b")
    b = OptimizationBuilder(T=3, robots=[r1, r2])
    b.sphere_collision_avoidance_constraints("a", ["o1"], link_names=["lwr_arm_5_link"])
    with pytest.raises(KeyError):
        b.sphere_collision_avoidance_constraints("b", ["o2"], link_names=["lwr_arm_5_link"])
    # sphere rows on a subset of knots are outside the structured family: the problem (154 variables) goes to the generic tape family -- until round 3
    # that family stopped at 32 variables and this raised LoweringError
    (kl, kr), o2 = setup_solver(T=6, build_only=True, collision=True)
    del o2.ineq_constraints["sphere_col_avoid_3_lwr_arm_5_link_kukal_obs2"]
    from optas_amd import _lib

    kind2, spec2 = lower(o2)
    assert kind2 == _lib.OH_PROBLEM_TAPE and spec2.tape.nx == o2.nx


def test_literal_layout_matches_builder_and_derivatives():
    rl, rr = _robots()
    T = 8
    nlp = GuardedDualArmNLP(rl, rr, SPHERE_LINKS, N_OBSTACLES, T=T)
    (kl, kr), o = setup_solver(T=T, build_only=True, limits=True, collision=True)
    assert (nlp.nx, nlp.np_, nlp.nk, nlp.na, nlp.ng, nlp.nv) == (o.nx, o.np, o.nk, o.na, o.ng, o.nv)
    p = o.parameters.dict2vec({"qcl": QC, "qcr": QC + 0.02, **obstacle_parameters()})
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, nlp.nx)
    Jg, Jk, h = nlp.dg(x, p), nlp.dk(x, p), 1e-6
    for i in rng.choice(nlp.nx, 8, replace=False):
        d = np.zeros(nlp.nx)
        d[i] = h
        assert np.abs((nlp.g(x + d, p) - nlp.g(x - d, p)) / (2 * h) - Jg[:, i]).max() < 1e-8
        assert np.abs((nlp.k(x + d, p) - nlp.k(x - d, p)) / (2 * h) - Jk[:, i]).max() < 1e-8
    v = nlp.v(x, p)
    assert v.shape == (nlp.nv,) and np.array_equal(v[: nlp.nk], nlp.k(x, p)) and np.array_equal(v[nlp.nk : nlp.nk + nlp.ng], nlp.g(x, p))


@pytest.mark.parametrize("tag,T,arm", [("T20l", 20, "l"), ("T20r", 20, "r"), ("T50l", 50, "l")])
def test_port_reproduces_golden(golden, tag, T, arm):
    rl, rr = _robots()
    rob = rl if arm == "l" else rr
    ch = FoldedChain(rob, "end_effector_ball")
    qc = golden[tag + "_qc"]
    G = Guards(lo=rob.lower_actuated_joint_limits, up=rob.upper_actuated_joint_limits, links=SPHERE_LINKS, link_radii=np.full(4, 0.15),
               obs_pos=golden["obs"], obs_radii=np.full(6, 0.1))
    s = solve_free_al(ch, T, 10.0 / (T - 1), dual_arm_offsets(T)[arm].T, qc, G, Q0=np.tile(qc, (T, 1)), rho0=10.0, exact=False)
    assert s["status"] == 0 and s["iters"] <= 80
    # tol 1e-6 on the gradient leaves ~1e-5 rad of play along the weakly curved velocity-regularised directions
    assert abs(s["f"] - float(golden[tag + "_f"])) < 1e-8 and np.abs(s["Q"] - golden[tag + "_Q"]).max() < 5e-5
    assert np.abs(s["Q"] - golden[tag + "_Q_slsqp"]).max() < 5e-5  # the reference-wired SLSQP optimum
    gv, _ = guard_values(ch, s["Q"], G)
    assert gv[1:].min() > -1e-9 and (s["lam"] >= 0).all() and np.abs(s["lam"] * gv)[1:].max() < 1e-8
    assert ((s["lam"] > 0) == (golden[tag + "_lam"] > 0)).mean() > 0.999  # same active set


def test_reference_form_kkt_of_the_port_solution(golden):
    rl, rr = _robots()
    T = 20
    nlp = GuardedDualArmNLP(rl, rr, SPHERE_LINKS, N_OBSTACLES, T=T)
    (kl, kr), o = setup_solver(T=T, build_only=True, limits=True, collision=True)
    p = o.parameters.dict2vec({"qcl": golden["T20l_qc"], "qcr": golden["T20r_qc"], **obstacle_parameters()})
    xs = []
    for tag in ("T20l", "T20r"):
        Q = golden[tag + "_Q"]
        xs += [Q.reshape(-1), (np.diff(Q, axis=0) / nlp.dt).reshape(-1)]
    x = np.concatenate(xs)
    assert abs(nlp.f(x, p) - float(golden["T20l_f"]) - float(golden["T20r_f"])) < 1e-12
    k = kkt_reference_form(nlp, x, p, active_tol=1e-6)
    assert k["stationarity"] < 1e-7 and k["feasibility"] < 1e-10 and k["complementarity"] < 1e-9


def test_figure_eight_lowering_with_limits_and_spheres():
    from examples.figure_eight_plan import setup_solver as figure_eight
    from optas_amd import _lib
    from optas_amd.lowering import LoweringError, lower
    from oracle.problems import GuardedFigureEightNLP

    rl, _ = _robots()
    kuka, o = figure_eight(build_only=True, limits=True, obstacles=["obs0", "obs1"], sphere_links=SPHERE_LINKS)
    assert (o.nk, o.ng, o.np, o.nv) == (700, 400, 7 + 4 + 8, 700 + 400 + 2 * 357 + 2 * 200)
    kind, spec = lower(o)
    assert kind == _lib.OH_PROBLEM_FIGURE_EIGHT and spec.spheres.links == SPHERE_LINKS and len(spec.spheres.obstacles) == 2
    kuka_o = OracleRobot(KUKA_KIN)
    assert np.array_equal(spec.lo, kuka_o.lower_actuated_joint_limits) and np.array_equal(spec.up, kuka_o.upper_actuated_joint_limits)
    # the mirrored builder's rows equal the literal restatement's on a random point (numpy only: no link function is evaluated
    # on the GPU here because k, a are linear and g is checked in the GPU suite)
    nlp = GuardedFigureEightNLP(kuka_o, "end_effector_ball", SPHERE_LINKS, 2, lo=spec.lo, up=spec.up, T=50)
    assert (nlp.nx, nlp.np_, nlp.nk, nlp.na, nlp.ng, nlp.nh, nlp.nv) == (o.nx, o.np, o.nk, o.na, o.ng, o.nh, o.nv)
    rng = np.random.default_rng(2)
    x, p = rng.uniform(-1, 1, o.nx), rng.uniform(-1, 1, o.np)
    assert np.abs(o.k(x, p) - nlp.k(x, p)).max() < 1e-14 and np.abs(o.a(x, p) - nlp.a(x, p)).max() < 1e-14
    # a parameter created between qc and the sphere parameters breaks the p layout the kernels read: refused
    from optas_amd.builder import OptimizationBuilder  # noqa: F401

    kuka2, o2 = figure_eight(build_only=True, obstacles=["obs0"], sphere_links=SPHERE_LINKS[:1])
    keys = list(o2.parameters.keys())
    assert keys[-4:] == ["qc", "end_effector_ball_radii", "obs0_position", "obs0_radii"]
    # velocity limits alone (no cost, no path): none of the structured families; the generic tape family takes the 63 variables (round 3)
    import optas_amd

    r = optas_amd.RobotModel.builtin("kuka_lwr", time_derivs=[0, 1])
    b = OptimizationBuilder(T=5, robots=[r])
    b.enforce_model_limits(r.get_name(), time_deriv=1)
    assert lower(b.build())[0] == _lib.OH_PROBLEM_TAPE


def test_parameterised_joint_problem_builds_and_lowers():
    """example/figure_eight_plan_6dof.py on the CPU: builder counts with param_joints, lowering to the lead-joint chain."""
    from examples.figure_eight_plan_6dof import setup_solver
    from optas_amd import _lib
    from optas_amd.lowering import lower

    kuka, o = setup_solver(build_only=True)
    assert (kuka.ndof, kuka.num_opt_joints, kuka.num_param_joints) == (7, 6, 1)
    assert (o.nx, o.np, o.na, o.nh, o.nv) == (6 * 50 + 6 * 49, 50 + 49 + 7, 6 + 6 + 6 * 49, 200, 2 * 306 + 2 * 200)
    assert list(o.parameters.keys()) == ["kuka/q/p", "kuka/dq/p", "qc"] and o.parameters["kuka/q/p"].shape == (1, 50)
    kind, spec = lower(o)
    assert kind == _lib.OH_PROBLEM_FIGURE_EIGHT and spec.lead == {"par": 0, "opt": [1, 2, 3, 4, 5, 6], "qp": "kuka/q/p", "dqp": "kuka/dq/p"}
    ch = kuka.solver_chain("end_effector_ball")
    full = kuka.kinematic_chain("end_effector_ball")
    assert (ch.ndof, ch.n_chain, ch.has_lead) == (6, 6, 1) and list(ch.qidx[:6]) == [0, 1, 2, 3, 4, 5]
    assert list(ch.lead_axis) == list(full.axis[0]) and list(ch.lead_p0) == list(full.p0[0]) and list(ch.R0[0]) == list(full.R0[1])
    # linear rows of the mirrored problem: q_x[:, 0] = qc[1:], dq_x[:, 0] = 0, Euler integration over the optimised block only
    rng = np.random.default_rng(4)
    x, p = rng.normal(size=o.nx), rng.normal(size=o.np)
    a = o.a(x, p)
    qc = p[99:106]
    assert np.allclose(a[:6], qc[1:] - x[:6]) and np.allclose(a[6:12], -x[300:306])


def test_velocity_limits_port_against_literal_kkt_and_slsqp():
    """enforce_model_limits(time_deriv=1) on the orientation-locked family (oracle/structured.py:vel_terms, the state machine k_couple_vel runs):
    the port's optimum satisfies the KKT conditions of the literal problem with its 2 n (T-1) extra k rows, the limits bind, and scipy SLSQP in
    the reference's wiring on a short horizon (rank-3 orientation rows: the xyz part only, where SLSQP converges) cannot find a better point."""
    from conftest import KUKA_KIN
    from oracle.problems import LimitedFigureEightNLP
    from oracle.robot import OracleRobot
    from oracle.solvers import kkt_reference_form
    from oracle.structured import StructuredFigureEight, solve_structured_lm

    orc = OracleRobot(KUKA_KIN)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    T = 50
    prob = StructuredFigureEight(orc, "end_effector_ball", T=T)
    free = solve_structured_lm(prob, qc, tol=1e-7)
    vfree = np.abs(np.diff(free["Q"], axis=0) / prob.dt).max(0)
    vl = np.asarray(orc.velocity_actuated_joint_limits)
    assert vfree[0] > vl[0]  # the shipped script's optimum violates the LWR velocity limit
    s = solve_structured_lm(prob, qc, tol=1e-7, vlimits=(-vl, vl), max_iter=600)
    assert s["status"] == 0 and s["f"] > free["f"] and s["iters"] < 60
    dQ = np.diff(s["Q"], axis=0) / prob.dt
    assert np.all(np.abs(dQ).max(0) <= vl + 1e-9) and abs(np.abs(dQ[:, 0]).max() - vl[0]) < 1e-9 and s["lam_v"].max() > 0
    nlp = LimitedFigureEightNLP(orc, "end_effector_ball", vlo=-vl, vup=vl, T=T)
    assert nlp.nk == 2 * 7 * (T - 1) and nlp.nv == 1114 + 686
    x = nlp.join(s["Q"].T, dQ.T)
    k = kkt_reference_form(nlp, x, qc, active_tol=1e-7)
    assert k["stationarity"] <= 1e-8 and k["feasibility"] <= 1e-8 and k["complementarity"] <= 1e-6


def test_dual_arm_with_velocity_limits_port_lowering_and_literal_kkt():
    """Round 3 (verdict Missing 3): enforce_model_limits(name, time_deriv=1) on the position-tracking family.  The mirror builder's k rows equal
    the literal restatement's, the lowering recognises them (GuardSpec.vlo / vup), and the numpy port's optimum satisfies the reference-form KKT
    conditions on the literal layout with the velocity rows binding."""
    from optas_amd.lowering import MultiArmSpec, lower

    T, vmax = 20, 0.08
    vl = np.full(7, vmax)
    (kl, kr), o = setup_solver(T=T, build_only=True, velocity_limits=(-vl, vl))
    assert o.nk == 4 * 7 * (T - 1) and o.ng == 0
    kind, spec = lower(o)
    assert isinstance(spec, MultiArmSpec) and all(np.array_equal(a.guards.vlo, -vl) and np.array_equal(a.guards.vup, vl) and a.guards.lo is None for a in spec.arms)
    rl, rr = _robots()
    nlp = GuardedDualArmNLP(rl, rr, [], 0, T=T, limits=False, vlimits=(-vl, vl))
    assert (nlp.nx, nlp.nk, nlp.na, nlp.ng) == (o.nx, o.nk, o.na, 0)
    rng = np.random.default_rng(5)
    xr, p = rng.uniform(-1, 1, o.nx), np.concatenate([QC, QC + 0.03])
    assert np.abs(o.k(xr, p) - nlp.k(xr, p)).max() < 1e-14 and np.array_equal(o.dk(xr, p), nlp.dk(xr, p))
    xs, f = [], 0.0
    for rob, arm, qc in ((rl, "l", p[:7]), (rr, "r", p[7:])):
        ch = FoldedChain(rob, "end_effector_ball")
        free = solve_free_al(ch, T, 10.0 / (T - 1), dual_arm_offsets(T)[arm].T, qc, Guards(), Q0=np.tile(qc, (T, 1)), rho0=10.0, exact=False)
        assert np.abs(np.diff(free["Q"], axis=0) / nlp.dt).max() > vmax  # the limit cuts into the unconstrained optimum
        s = solve_free_al(ch, T, 10.0 / (T - 1), dual_arm_offsets(T)[arm].T, qc, Guards(), Q0=np.tile(qc, (T, 1)), rho0=10.0, exact=False,
                          vlimits=(-vl, vl), max_iter=600, tol=1e-7)
        assert s["status"] == 0 and s["f"] > free["f"] and (s["lam_v"] > 0).sum() >= 3
        dQ = np.diff(s["Q"], axis=0) / nlp.dt
        assert np.abs(dQ).max() <= vmax + 1e-8
        xs += [s["Q"].reshape(-1), dQ.reshape(-1)]
        f += s["f"]
    x = np.concatenate(xs)
    assert abs(nlp.f(x, p) - f) < 1e-12 and np.abs(nlp.a(x, p)).max() < 1e-13
    k = kkt_reference_form(nlp, x, p, active_tol=1e-6)
    assert k["stationarity"] < 1e-5 and k["feasibility"] < 1e-8 and k["complementarity"] < 1e-6, k
