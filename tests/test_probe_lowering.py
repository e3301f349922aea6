"""The drop-in route for a REAL optas problem (CasADi graphs inside): optas_amd.probe_lowering recognises the figure-eight family from the
labels / shapes of the problem's containers and the values of its own numeric functions, verifies the model it read off, and the literal
``optas.solver.Solver`` subclass (optas_amd.casadi_tape.make_solver_class) then runs the structured kernels.  casadi is not installed
here, so the problem object is this repo's mirror ``Optimization`` behind an adapter that exposes nothing but the reference's interface
(containers of shaped items, ``models`` with a urdf_parser_py-shaped URDF object, numeric callables): the expression trees the mirror's
own pattern matcher uses are hidden, exactly as they would be with SX graphs."""
import os
import sys
import types

import numpy as np
import pytest

from conftest import SEED

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

pytestmark = pytest.mark.gpu


class _Shaped:
    def __init__(self, shape):
        self.shape = shape


class _Container(dict):
    """Ordered label -> item with a shape (what SXContainer is to the code under test)."""


def _hide(container):
    out = _Container()
    for k, v in container.items():
        out[k] = _Shaped(tuple(v.shape))
    return out


def _urdf_parser_py_like(desc):
    """The same kinematic tree as objects shaped like urdf_parser_py's (joint.origin.xyz / .rpy, joint.limit.lower, ...)."""
    ns = types.SimpleNamespace

    def joint(j):
        return ns(name=j.name, type=j.type, parent=j.parent, child=j.child, origin=None if j.xyz is None else ns(xyz=list(j.xyz), rpy=list(j.rpy)),
                  axis=None if j.axis is None else list(j.axis),
                  limit=None if j.limit is None else ns(lower=j.limit.lower, upper=j.limit.upper, velocity=j.limit.velocity, effort=j.limit.effort))

    def link(l):
        if l.inertial is None:
            return ns(name=l.name, inertial=None)
        ixx, ixy, ixz, iyy, iyz, izz = l.inertial.inertia
        return ns(name=l.name, inertial=ns(mass=l.inertial.mass, origin=ns(xyz=list(l.inertial.xyz), rpy=list(l.inertial.rpy)),
                                           inertia=ns(ixx=ixx, ixy=ixy, ixz=ixz, iyy=iyy, iyz=iyz, izz=izz)))

    return ns(name=desc.name, joints=[joint(j) for j in desc.joints], links=[link(l) for l in desc.links])


class ReferenceLikeOptimization:
    """Only what optas.optimization.Optimization offers to a Solver (optimization.py:60-306)."""

    def __init__(self, opt):
        self._o = opt
        self.nx, self.np, self.nk, self.na, self.ng, self.nh, self.nv = opt.nx, opt.np, opt.nk, opt.na, opt.ng, opt.nh, opt.nv
        self.decision_variables = opt.decision_variables  # dict2vec / vec2dict are needed by Solver.solve; the items only expose .shape below
        self.parameters = opt.parameters
        self.lin_eq_constraints, self.lin_ineq_constraints = _hide(opt.lin_eq_constraints), _hide(opt.lin_ineq_constraints)
        self.eq_constraints, self.ineq_constraints = _hide(opt.eq_constraints), _hide(opt.ineq_constraints)
        self.cost_terms = _hide(opt.cost_terms)
        self.models = []
        for m in opt.models:
            ns = types.SimpleNamespace(get_name=m.get_name, time_derivs=list(m.time_derivs), dim=m.dim, state_name=m.state_name,
                                       state_optimized_name=m.state_optimized_name, state_parameter_name=m.state_parameter_name)
            if hasattr(m, "urdf"):  # a RobotModel; a TaskModel has no URDF (models.py:189-214)
                pj = list(getattr(m, "param_joints", []) or [])  # (models.py: RobotModel(param_joints=...))
                ns.param_joints, ns.urdf, ns.ndof, ns.num_param_joints = pj, _urdf_parser_py_like(m.urdf), m.ndof, len(pj)
            self.models.append(ns)
        self.calls = 0

    def _count(self, fn):
        def g(x, p):
            self.calls += 1
            return fn(np.asarray(x, float).reshape(-1), np.asarray(p, float).reshape(-1))

        return g

    def __getattr__(self, name):
        if name in ("f", "a", "h", "k", "g", "v"):
            return self._count(getattr(self._o, name))
        raise AttributeError(name)

    def has_discrete_variables(self):
        return False


def test_probing_recovers_the_family_from_the_reference_interface(hip_lib, golden_nlp):
    from examples.figure_eight_plan import setup_solver
    from optas_amd.lowering import FigureEightSpec, match_figure_eight
    from optas_amd.probe_lowering import probe_figure_eight

    kuka, solver = setup_solver(build_only=True)
    want = match_figure_eight(solver)  # build_only returns the Optimization: the tree-matching route, for comparison
    ref = ReferenceLikeOptimization(solver)
    spec = probe_figure_eight(ref)
    assert isinstance(spec, FigureEightSpec)
    assert (spec.link, spec.T) == ("end_effector_ball", 50) and abs(spec.dt - want.dt) < 1e-14
    assert abs(spec.w_path - 1000.0) < 1e-6 and abs(spec.w_vel - 0.01) < 1e-9
    assert np.abs(spec.local_path - want.local_path).max() < 1e-9
    assert ref.calls < 500  # a few evaluations per knot, not a search
    # constants folded from the urdf_parser_py-shaped object equal the ones folded from the mirror's own description
    assert bytes(spec.robot.solver_chain(spec.link)) == bytes(kuka.solver_chain("end_effector_ball"))


def test_probing_refuses_what_is_not_the_family(hip_lib):
    from examples.figure_eight_plan import setup_solver
    from optas_amd.lowering import LoweringError
    from optas_amd.probe_lowering import probe_figure_eight

    kuka, opt = setup_solver(build_only=True)
    ref = ReferenceLikeOptimization(opt)
    f_true = opt.f
    # same labels and shapes, a different cost (extra quartic term): the verification step must catch it
    opt.f = lambda x, p: f_true(x, p) + 1e-3 * float(np.sum(np.asarray(x)[:7] ** 4))
    with pytest.raises(LoweringError, match="cost"):
        probe_figure_eight(ref)
    opt.f = f_true
    # a missing row block is refused on labels alone
    del ref.lin_eq_constraints["__kuka_fix_configuration_1_0__"]
    with pytest.raises(LoweringError, match="linear equalities"):
        probe_figure_eight(ref)


def test_literal_solver_subclass_runs_the_structured_kernels(hip_lib, golden_nlp):
    """make_solver_class over a stand-in for optas.solver (the mirror's Solver ABC has the reference's methods) and a stand-in for casadi
    (only ``DM``): the subclass must pick the figure-eight family by probing and reproduce the golden optimum."""
    from examples.figure_eight_plan import setup_solver
    from optas_amd import solver as mirror_solver
    from optas_amd.casadi_tape import make_solver_class

    cs = types.SimpleNamespace(DM=lambda a: np.asarray(a, dtype=np.float64).reshape(-1, 1))
    HIPSolver = make_solver_class(mirror_solver, cs)
    kuka, opt = setup_solver(build_only=True)
    ref = ReferenceLikeOptimization(opt)
    s = HIPSolver(ref).setup("hip_sqp", {"tol": 1e-7})
    qc = golden_nlp["fig8_qc"]
    s.reset_parameters({"qc": qc})
    s.reset_initial_seed({"kuka/q/x": np.tile(qc[:, None], (1, 50))})
    sol = s.solve()
    st = s.stats()
    assert st["family"] == "figure_eight" and s.did_solve()
    assert abs(st["f"] - float(golden_nlp["fig8_f"])) <= 1e-8 * float(golden_nlp["fig8_f"])  # dense SQP on the literal layout (independent)
    assert sol["kuka/q"].shape == (7, 50) and sol["kuka/dq"].shape == (7, 49)
    with pytest.raises(ValueError):
        HIPSolver(ref).setup("ipopt")


def _standins():
    from optas_amd import solver as mirror_solver
    from optas_amd.casadi_tape import make_solver_class

    cs = types.SimpleNamespace(DM=lambda a: np.asarray(a, dtype=np.float64).reshape(-1, 1))
    return make_solver_class(mirror_solver, cs)


def test_torque_mpc_is_recognised_and_solved_through_the_reference_interface(hip_lib):
    """BASELINE configs[4] (RNEA equality rows) behind the reference interface: two models (robot + task), four variable blocks, the
    dynamics rows verified against oh_rnea; the literal Solver subclass lands on the torque kernels and reproduces the tree-matched route."""
    from examples.torque_mpc import build_problem, figure_eight_goal
    from optas_amd.lowering import LoweringError, TorqueSpec, match_torque_mpc
    from optas_amd.probe_lowering import probe, probe_torque_mpc
    from optas_amd.solver import HIPSolver as MirrorHIPSolver

    T, dt = 12, 0.1
    robot, link, opt = build_problem(T, dt, effort=60.0)
    want = match_torque_mpc(opt)
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "torque_mpc" and isinstance(spec, TorqueSpec) and (spec.link, spec.T) == (link, T)
    assert abs(spec.dt - dt) < 1e-14 and abs(spec.w_path - want.w_path) < 1e-6 and abs(spec.w_vel - want.w_vel) < 1e-12 and abs(spec.w_tau - want.w_tau) < 1e-14
    assert np.abs(spec.tau_lo + 60.0).max() < 1e-12 and np.abs(spec.tau_up - 60.0).max() < 1e-12
    assert ref.calls < 80
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    inputs = {"qc": qc, "dqc": np.zeros(7), "goal": figure_eight_goal(robot, link, qc, T, dt)}
    seed = {f"{robot.get_name()}/q/x": np.tile(qc[:, None], (1, T))}
    s = _standins()(ref).setup("hip_sqp", {"tol": 1e-6})
    s.reset_parameters(inputs)
    s.reset_initial_seed(seed)
    sol = s.solve()
    assert s.stats()["family"] == "torque_mpc" and s.did_solve()
    m = MirrorHIPSolver(opt).setup("hip_sqp", {"tol": 1e-6})
    m.reset_parameters(inputs)
    m.reset_initial_seed(seed)
    sol_m = m.solve()
    assert abs(s.stats()["f"] - float(m.stats()["f"][0])) <= 1e-12 * abs(s.stats()["f"])
    assert np.abs(np.asarray(sol["tau/y"]) - np.asarray(sol_m["tau/y"])).max() <= 1e-9
    # a different dynamics row (gravity-free torque, say) keeps every label and shape: the verification must refuse it
    h_true = opt.h
    opt.h = lambda x, p: h_true(x, p) + 1e-3
    with pytest.raises(LoweringError, match="rnea"):
        probe_torque_mpc(ref)
    opt.h = h_true


def test_ik_is_recognised_and_solved_through_the_reference_interface(hip_lib, golden_nlp):
    from examples.example import script_inputs, setup_solver
    from optas_amd.lowering import IkSpec, match_ik
    from optas_amd.probe_lowering import probe

    robot, opt = setup_solver(build_only=True)
    want = match_ik(opt)
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "ik" and isinstance(spec, IkSpec) and spec.link == want.link and abs(spec.w_nominal - want.w_nominal) < 1e-12
    assert np.abs(spec.lo - want.lo).max() < 1e-12 and np.abs(spec.up - want.up).max() < 1e-12
    q_nominal, p_goal = script_inputs(robot)
    s = _standins()(ref).setup("hip_sqp")
    s.reset_parameters({"q_nominal": q_nominal, "p_goal": p_goal})
    s.reset_initial_seed({robot.get_name() + "/q": q_nominal})  # not a variable label: zero seed, as in the reference script
    sol = s.solve()
    assert s.stats()["family"] == "ik" and s.did_solve()
    assert abs(s.stats()["f"] - 0.29579887518) < 1e-8  # the SLSQP-wired known answer of example.py (tests/test_gpu_ik.py)


def test_point_mass_mpc_is_recognised_and_solved_through_the_reference_interface(hip_lib):
    """BASELINE configs[2] has no robot at all: a task model, box limits, one obstacle row per knot.  Every number is read off the problem's
    own members; the known answer of the reference-wired SLSQP run comes back through the literal Solver subclass."""
    from examples.point_mass_mpc import Controller, obstacle_and_goal
    from optas_amd.lowering import LoweringError, PointMassSpec, match_point_mass
    from optas_amd.probe_lowering import probe, probe_point_mass

    opt = Controller(build_only=True).optimization
    want = match_point_mass(opt)
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "point_mass" and isinstance(spec, PointMassSpec)
    assert (spec.T, spec.names, spec.y_name, spec.dy_name) == (want.T, want.names, want.y_name, want.dy_name)
    for a, b in ((spec.dt, want.dt), (spec.w_acc, want.w_acc), (spec.ylim, want.ylim), (spec.vlim, want.vlim), (spec.safe, want.safe)):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(b))
    assert ref.calls < 20
    s = _standins()(ref).setup("hip_sqp", {"tol": 1e-9})
    curr, dcurr = np.array([-0.45, -0.35]), np.array([0.6, 0.6])
    obs, goal = obstacle_and_goal(2.0, curr)
    s.reset_parameters({"curr": curr, "dcurr": dcurr, "goal": goal, "obs": obs})
    sol = s.solve()
    assert s.stats()["family"] == "point_mass" and s.did_solve() and abs(s.stats()["f"] - 0.1759064919) < 1e-7
    assert np.asarray(sol["point_mass/y"]).shape == (2, 20)
    g_true = opt.g
    opt.g = lambda x, p: g_true(x, p) * 1.0001  # not a squared distance any more
    with pytest.raises(LoweringError, match="inequality rows"):
        probe_point_mass(ref)
    opt.g = g_true


def test_point_mass_planner_variant_is_recognised_by_probing_and_solved(hip_lib):
    """example/point_mass_planner.py behind the reference interface: parameters init, goal only; the obstacle is a constant inside g, the
    tracking cost sits on the last knot, the velocities are penalised and the final velocity is a user-labelled equality row.  Every number
    is read off the problem's callables and agrees with the tree matcher of the mirror builder; the literal subclass reaches the optimum of
    the mirror route."""
    from examples.point_mass_planner import Planner
    from optas_amd.lowering import PointMassSpec, match_point_mass_planner
    from optas_amd.probe_lowering import probe

    pl = Planner(solver_options={"tol": 1e-9})
    opt = pl.solver.opt
    want = match_point_mass_planner(opt)
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "point_mass" and isinstance(spec, PointMassSpec) and spec.planner is not None
    assert (spec.T, spec.names, spec.y_name, spec.dy_name) == (want.T, want.names, want.y_name, want.dy_name)
    for a, b in ((spec.dt, want.dt), (spec.w_acc, want.w_acc), (spec.ylim, want.ylim), (spec.vlim, want.vlim), (spec.safe, want.safe),
                 (spec.planner["w_vel"], want.planner["w_vel"])):
        assert abs(a - b) <= 1e-10 * max(1.0, abs(b))
    assert np.abs(spec.planner["obstacle"] - want.planner["obstacle"]).max() < 1e-12
    assert ref.calls < 25
    init, goal = [-1.2, -0.4], [1.0, 0.7]
    _, _, sol_m = pl.plan(init, goal)
    s = _standins()(ref).setup("hip_sqp", {"tol": 1e-9})
    s.reset_parameters({"init": init, "goal": goal})
    sol = s.solve()
    assert s.stats()["family"] == "point_mass" and s.did_solve()
    assert abs(s.stats()["f"] - pl.solver.stats()["f"][0]) < 1e-12
    assert np.abs(np.asarray(sol["point_mass/y"]) - np.asarray(sol_m["point_mass/y"])).max() < 1e-12


def test_dual_arm_is_recognised_and_solved_through_the_reference_interface(hip_lib):
    """BASELINE configs[3] as shipped (example/dual_arm.py: two robots on base frames, separable): probed one arm at a time."""
    from examples.dual_arm import setup_solver
    from optas_amd.lowering import MultiArmSpec, match_multi_arm
    from optas_amd.probe_lowering import probe

    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    (kl, kr), opt = setup_solver(build_only=True)
    want = match_multi_arm(opt)
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "multi_arm" and isinstance(spec, MultiArmSpec) and (spec.T, len(spec.arms)) == (want.T, 2) and abs(spec.dt - want.dt) < 1e-14
    for a, b in zip(spec.arms, want.arms):
        assert (a.link, a.qc_name, a.q_name, a.dq_name) == (b.link, b.qc_name, b.q_name, b.dq_name)
        assert abs(a.w_path - b.w_path) < 1e-7 * b.w_path and abs(a.w_vel - b.w_vel) < 1e-9
        assert np.abs(a.offsets[1:] - b.offsets[1:]).max() < 1e-8
        # base frames come out of the URDF-shaped object (add_base_frame edits the tree, models.py:323-361)
        assert bytes(a.robot.kinematic_chain(a.link)) == bytes(b.robot.kinematic_chain(b.link))
    s = _standins()(ref).setup("hip_sqp", {"tol": 1e-8, "max_iter": 300})
    s.reset_parameters({"qcl": QC, "qcr": QC})
    sol = s.solve()
    assert s.stats()["family"] == "multi_arm" and s.did_solve()
    assert abs(s.stats()["f"] - 0.00480191855905) <= 1e-8  # the known answer of tests/test_dual_arm.py (oracle, reference wiring)
    assert np.asarray(sol["kukal/q"]).shape == (7, 50)


def test_guarded_dual_arm_is_recognised_and_solved_through_the_reference_interface(hip_lib):
    """BASELINE configs[3] AS STATED (dual-arm + joint limits + sphere-collision inequalities) behind the reference interface (round-2 verdict,
    Missing 2): limit blocks and sphere rows are recognised from their labels, attributed to an arm and read off k and g numerically, verified,
    and the guarded kernels run.  Known answer: the SLSQP-wired optimum of tests/golden/guard_golden.npz (reference wiring, T = 20)."""
    from conftest import GOLDEN
    from examples.dual_arm import N_OBSTACLES, SPHERE_LINKS, obstacle_parameters, setup_solver
    from optas_amd.lowering import LoweringError, MultiArmSpec, match_multi_arm
    from optas_amd.probe_lowering import probe, probe_multi_arm

    golden = np.load(os.path.join(GOLDEN, "guard_golden.npz"))
    T = 20
    (kl, kr), opt = setup_solver(T=T, build_only=True, limits=True, collision=True)
    want = match_multi_arm(opt)  # the tree matcher of the mirror builder: what the probing route has to reproduce without seeing a tree
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "multi_arm" and isinstance(spec, MultiArmSpec) and len(spec.arms) == 2
    for a, b in zip(spec.arms, want.arms):
        assert a.guards is not None and b.guards is not None
        assert a.guards.links == b.guards.links == SPHERE_LINKS and a.guards.link_radii == b.guards.link_radii and a.guards.obstacles == b.guards.obstacles
        assert len(a.guards.obstacles) == N_OBSTACLES and np.array_equal(a.guards.lo, b.guards.lo) and np.array_equal(a.guards.up, b.guards.up)
        assert abs(a.w_path - b.w_path) < 1e-7 * b.w_path and np.abs(a.offsets[1:] - b.offsets[1:]).max() < 1e-8
    s = _standins()(ref).setup("hip_sqp", {"max_iter": 400})
    pd = {"qcl": golden["T20l_qc"], "qcr": golden["T20r_qc"], **obstacle_parameters()}
    s.reset_parameters(pd)
    s.reset_initial_seed({"kukal/q/x": np.tile(pd["qcl"].reshape(-1, 1), (1, T)), "kukar/q/x": np.tile(pd["qcr"].reshape(-1, 1), (1, T))})
    sol = s.solve()
    assert s.stats()["family"] == "multi_arm" and s.did_solve()
    assert abs(s.stats()["f"] - (float(golden["T20l_f"]) + float(golden["T20r_f"]))) < 1e-8  # scipy SLSQP in the reference's wiring (tools/make_golden.py)
    assert np.abs(np.asarray(sol["kukal/q"]).T - golden["T20l_Q"]).max() < 5e-5 and np.abs(np.asarray(sol["kukar/q"]).T - golden["T20r_Q"]).max() < 5e-5
    # a problem whose rows only look like sphere rows is refused, not approximated
    g_true = opt.g
    opt.g = lambda x, p: g_true(x, p) + 1e-4 * np.sin(np.asarray(x)[3])
    with pytest.raises(LoweringError):
        probe_multi_arm(ref)
    opt.g = g_true
    k_true = opt.k
    opt.k = lambda x, p: k_true(x, p) * 1.001
    with pytest.raises(LoweringError, match="limit"):
        probe_multi_arm(ref)
    opt.k = k_true


@pytest.mark.gpu
def test_velocity_limited_dual_arm_through_the_reference_interface(hip_lib):
    """enforce_model_limits(name, time_deriv=1) on the position-tracking family (round 3): the "__{name}_model_limit_1___l/_r" blocks are recognised
    from their labels, read off k numerically and verified; the velocity-row kernels run and the limit binds."""
    from examples.dual_arm import setup_solver
    from optas_amd.lowering import LoweringError, MultiArmSpec, match_multi_arm
    from optas_amd.probe_lowering import probe, probe_multi_arm

    T, vmax = 20, 0.08
    vl = np.full(7, vmax)
    (kl, kr), opt = setup_solver(T=T, build_only=True, velocity_limits=(-vl, vl))
    want = match_multi_arm(opt)
    ref = ReferenceLikeOptimization(opt)
    fam, spec = probe(ref)
    assert fam == "multi_arm" and isinstance(spec, MultiArmSpec)
    for a, b in zip(spec.arms, want.arms):
        assert a.guards.lo is None and not a.guards.links and np.array_equal(a.guards.vlo, b.guards.vlo) and np.array_equal(a.guards.vup, vl)
    s = _standins()(ref).setup("hip_sqp", {"max_iter": 600})
    qc = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    s.reset_parameters({"qcl": qc, "qcr": qc})
    s.reset_initial_seed({"kukal/q/x": np.tile(qc.reshape(-1, 1), (1, T)), "kukar/q/x": np.tile(qc.reshape(-1, 1), (1, T))})
    sol = s.solve()
    assert s.stats()["family"] == "multi_arm" and s.did_solve()
    dq = np.asarray(sol["kukal/dq"])
    assert np.abs(dq).max() <= vmax + 1e-8 and np.abs(dq).max() >= vmax - 1e-6
    k_true = opt.k
    opt.k = lambda x, p: k_true(x, p) + 1e-3 * np.asarray(x)[0]
    with pytest.raises(LoweringError):
        probe_multi_arm(ref)
    opt.k = k_true


def test_limit_rows_of_the_figure_eight_are_recognised_by_probing(hip_lib):
    """Round 4 (verdict r03 Missing 4, first part): enforce_model_limits(name) and (name, time_deriv=1) on config 2 reach the kernels from the
    reference interface too -- blocks found by label and shape, bounds read off k(0, p), k verified at a random point -- and give the spec the tree
    matcher gives.  A block whose rows are not [x - lo; up - x] is refused."""
    from examples.figure_eight_plan import setup_solver
    from optas_amd.lowering import LoweringError, match_figure_eight
    from optas_amd.probe_lowering import probe_figure_eight

    vl = np.full(7, 1.2)
    kuka, opt = setup_solver(build_only=True, limits=True, velocity_limits=(-vl, vl))
    want = match_figure_eight(opt)
    ref = ReferenceLikeOptimization(opt)
    spec = probe_figure_eight(ref)
    assert np.abs(spec.lo - want.lo).max() < 1e-12 and np.abs(spec.up - want.up).max() < 1e-12
    assert np.abs(spec.vlo + vl).max() < 1e-12 and np.abs(spec.vup - vl).max() < 1e-12
    assert spec.spheres is None and abs(spec.w_path - 1000.0) < 1e-6
    kuka, opt = setup_solver(build_only=True, velocity_limits=True)
    spec = probe_figure_eight(ReferenceLikeOptimization(opt))
    assert spec.lo is None and spec.vlo is not None and np.abs(spec.vup - match_figure_eight(opt).vup).max() < 1e-12
    k_true = opt.k
    opt.k = lambda x, p: np.asarray(k_true(x, p)) * 1.0 + 1e-3 * np.sin(np.asarray(x)[:1])  # same labels, rows no longer affine with slope one
    with pytest.raises(LoweringError, match="limit rows"):
        probe_figure_eight(ReferenceLikeOptimization(opt))


def test_velocity_limited_figure_eight_through_the_reference_interface(hip_lib, golden_nlp):
    """... and the literal subclass then solves it with the structured kernels: same optimum as the mirror route (HIPSolver on the tree-matched problem),
    velocity rows satisfied."""
    from examples.figure_eight_plan import setup_solver

    HIPSolver = _standins()
    qc = golden_nlp["fig8_qc"]
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"tol": 1e-7, "max_iter": 600})
    solver.reset_parameters({"qc": qc})
    solver.reset_initial_seed({"kuka/q/x": np.tile(qc[:, None], (1, 50))})
    want = solver.solve()
    f_want = solver.stats()["f"]
    kuka, opt = setup_solver(build_only=True, velocity_limits=True)
    s = HIPSolver(ReferenceLikeOptimization(opt)).setup("hip_sqp", {"tol": 1e-7, "max_iter": 600})
    s.reset_parameters({"qc": qc})
    s.reset_initial_seed({"kuka/q/x": np.tile(qc[:, None], (1, 50))})
    sol = s.solve()
    st = s.stats()
    assert st["family"] == "figure_eight" and s.did_solve()
    assert abs(st["f"] - f_want) <= 1e-9 * f_want
    vmax = kuka.velocity_actuated_joint_limits if hasattr(kuka, "velocity_actuated_joint_limits") else None
    dq = np.asarray(sol["kuka/dq"])
    assert np.abs(dq - np.asarray(want["kuka/dq"])).max() <= 1e-7
    if vmax is not None:
        assert (np.abs(dq) <= np.asarray(vmax)[:, None] + 1e-9).all()


def test_sphere_rows_of_the_figure_eight_are_recognised_by_probing_and_solved(hip_lib):
    """Round 4 (verdict r03 Missing 4): sphere_collision_avoidance_constraints on config 2 from the reference interface -- rows found by label, radius /
    obstacle parameters attributed numerically, g verified against ||p_link(q_t) - o||^2 - (r_l + r_o)^2 -- gives the tree matcher's spec, and the
    literal subclass solves it on the structured kernels to the mirror route's optimum (obstacle beside the path: rows active)."""
    from examples.dual_arm import SPHERE_LINKS
    from examples.figure_eight_plan import setup_solver
    from optas_amd.lowering import match_figure_eight
    from optas_amd.probe_lowering import probe_figure_eight

    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    obs = np.array([-0.848, 0.12, 0.461])
    pd = {"qc": qc, "obs0_position": obs, "obs0_radii": 0.05, **{ln + "_radii": 0.05 for ln in SPHERE_LINKS}}
    kuka, opt = setup_solver(build_only=True, obstacles=["obs0"], sphere_links=SPHERE_LINKS, limits=True)
    want = match_figure_eight(opt)
    spec = probe_figure_eight(ReferenceLikeOptimization(opt))
    assert spec.spheres is not None and list(spec.spheres.links) == list(want.spheres.links)
    assert list(spec.spheres.link_radii) == list(want.spheres.link_radii) and [tuple(o) for o in spec.spheres.obstacles] == [tuple(o) for o in want.spheres.obstacles]
    assert np.abs(spec.lo - want.lo).max() < 1e-12 and spec.vlo is None
    kuka, solver = setup_solver(obstacles=["obs0"], sphere_links=SPHERE_LINKS, limits=True, solver_options={"max_iter": 400})
    solver.reset_parameters(pd)
    solver.reset_initial_seed({"kuka/q/x": np.tile(qc.reshape(-1, 1), (1, 50))})
    solver.solve()
    f_want = solver.stats()["f"][0]
    HIPSolver = _standins()
    s = HIPSolver(ReferenceLikeOptimization(opt)).setup("hip_sqp", {"max_iter": 400})
    s.reset_parameters(pd)
    s.reset_initial_seed({"kuka/q/x": np.tile(qc.reshape(-1, 1), (1, 50))})
    s.solve()
    st = s.stats()
    assert st["family"] == "figure_eight" and s.did_solve()
    assert abs(np.ravel(st["f"])[0] - f_want) <= 1e-9 * f_want and f_want > 8.9


def test_lead_joint_variant_is_recognised_by_probing_and_solved(hip_lib):
    """Round 4 (verdict r03 Missing 4): example/figure_eight_plan_6dof.py -- RobotModel(param_joints=[first joint]) -- from the reference interface: six
    optimised joints in x, the parameterised joint's trajectory in p; probing merges them, finds the link, weights and path, and the literal subclass
    reaches the optimum of the mirror route."""
    from examples.figure_eight_plan_6dof import plan, setup_solver
    from optas_amd.lowering import match_figure_eight
    from optas_amd.probe_lowering import probe_figure_eight

    kuka, opt = setup_solver(build_only=True)
    want = match_figure_eight(opt)
    spec = probe_figure_eight(ReferenceLikeOptimization(opt))
    assert spec.lead == want.lead and spec.link == want.link and (spec.qc_name, spec.q_name, spec.dq_name) == (want.qc_name, want.q_name, want.dq_name)
    assert abs(spec.w_path - want.w_path) < 1e-6 and abs(spec.w_vel - want.w_vel) < 1e-9 and abs(spec.dt - want.dt) < 1e-14
    assert np.abs(spec.local_path - want.local_path).max() < 1e-9
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    kuka, solver = setup_solver()
    plan(kuka, solver, qc)
    f_want = np.ravel(solver.stats()["f"])[0]
    HIPSolver = _standins()
    s = HIPSolver(ReferenceLikeOptimization(opt)).setup("hip_sqp", {})
    Q0 = np.diag(qc) @ np.ones((7, 50))
    s.reset_initial_seed({"kuka/q/x": kuka.extract_optimized_dimensions(Q0)})
    s.reset_parameters({"qc": qc, "kuka/q/p": kuka.extract_parameter_dimensions(Q0)})
    s.solve()
    st = s.stats()
    assert st["family"] == "figure_eight" and s.did_solve()
    assert abs(np.ravel(st["f"])[0] - f_want) <= 1e-9 * f_want
