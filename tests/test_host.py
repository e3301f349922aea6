"""Host-side logic of optas_amd (no GPU): URDF loading, container layout, builder routing/counts
(the reference's own expectations, tests/test_builder.py:25-37,258-419, tests/test_sx_container.py),
lowering, chain folding."""
import ctypes as C
import os

import numpy as np
import pytest

import optas_amd
from conftest import KUKA_KIN, TESTER_KIN
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import ParamRef, StateRef, path_in_frame, sumsqr
from optas_amd.lowering import LoweringError, lower
from optas_amd.models import RobotModel, TaskModel
from optas_amd.optimization import (
    NonlinearCostNonlinearConstraints,
    QuadraticCostLinearConstraints,
    QuadraticCostNonlinearConstraints,
    QuadraticCostUnconstrained,
)
from optas_amd.sx_container import SXContainer
from optas_amd.urdf import RobotDescription
from oracle.robot import OracleRobot
from oracle.structured import FoldedChain

URDF = """<?xml version="1.0" ?>
<robot name="mini">
<link name="base"/><link name="a"><inertial><origin xyz="0.1 0 0" rpy="0 0 0"/><mass value="2.0"/>
<inertia ixx="1" ixy="0" ixz="0" iyy="2" iyz="0" izz="3"/></inertial></link><link name="b"/><link name="tip"/>
<joint name="j0" type="continuous"><origin xyz="0 0 0.5" rpy="0 0 1.0"/><axis xyz="0 0 2"/><parent link="base"/><child link="a"/></joint>
<transmission><joint name="j0"/></transmission>
<joint name="j1" type="prismatic"><origin xyz="1 0 0"/><axis xyz="1 0 0"/><limit lower="-0.5" upper="0.5" velocity="2" effort="10"/><parent link="a"/><child link="b"/></joint>
<joint name="jt" type="fixed"><parent link="b"/><child link="tip"/></joint>
</robot>"""


def test_urdf_parser():
    r = RobotDescription.from_xml_string(URDF)
    assert r.name == "mini" and [j.name for j in r.joints] == ["j0", "j1", "jt"]  # nested <joint/> stubs ignored
    assert r.get_root() == "base" and r.get_chain("base", "tip") == ["j0", "j1", "jt"]
    assert r.get_chain("base", "tip", links=True, joints=False) == ["base", "a", "b", "tip"]
    assert r.joint_map["jt"].xyz is None and r.joint_map["jt"].axis is None and r.joint_map["j0"].limit is None
    assert r.link_map["a"].inertial.mass == 2.0
    r2 = RobotDescription.from_dict(r.to_dict())
    assert r2.to_dict() == r.to_dict()
    m = RobotModel(urdf_string=URDF)
    assert m.ndof == 2 and m.actuated_joint_names == ["j0", "j1"]
    assert np.allclose(m.lower_actuated_joint_limits, [-1e9, -0.5]) and np.allclose(m.velocity_actuated_joint_limits, [1e9, 2])
    assert np.allclose(m.get_joint_axis(m.urdf.joint_map["j0"]), [0, 0, 1])  # normalised
    assert np.allclose(m.get_joint_axis(m.urdf.joint_map["jt"]), [1, 0, 0])  # default
    with pytest.raises(AssertionError):
        RobotModel()


def test_robot_model_tester_known_answers():
    m = RobotModel(urdf_filename=TESTER_KIN)
    assert m.joint_names == ["joint0", "joint1", "joint2", "eff_joint"]
    assert m.link_names == ["world", "link1", "link2", "link3", "eff"]
    assert m.optimized_joint_indexes == [0, 1, 2] and m.parameter_joint_indexes == []
    assert np.allclose(m.lower_optimized_joint_limits, [-1e9, -1, 0]) and np.allclose(m.upper_optimized_joint_limits, [1e9, 1, 1])
    mp = RobotModel(urdf_filename=TESTER_KIN, param_joints=["joint0"])
    assert mp.optimized_joint_indexes == [1, 2] and mp.parameter_joint_indexes == [0]
    assert mp.optimized_joint_names == ["joint1", "joint2"] and np.allclose(mp.lower_optimized_joint_limits, [-1, 0])
    assert m.state_name(0) == "test_robot/q" and m.state_optimized_name(0) == "test_robot/q/x" and m.state_parameter_name(0) == "test_robot/q/p"
    with pytest.raises(AssertionError):
        m.state_name(1)


def test_sx_container_layout():
    c = SXContainer()
    c["a"] = StateRef("a", "m", 0, 2, 3)
    c["b"] = ParamRef("b", 1, 2)
    with pytest.raises(KeyError):
        c["a"] = ParamRef("a")
    assert c.numel() == 8
    d = {"a": np.array([[1.0, 2, 3], [4, 5, 6]])}
    v = c.dict2vec(d)
    assert np.allclose(v, [1, 4, 2, 5, 3, 6, 0, 0])  # column-major, missing key zero-filled
    back = c.vec2dict(v)
    assert np.allclose(back["a"], d["a"]) and np.allclose(back["b"], 0) and back["b"].shape == (1, 2)
    assert c.discrete() == [False] * 8
    c.variable_is_discrete("b")
    assert c.has_discrete_variables() and c.discrete()[-2:] == [True, True]


def test_builder_counts_and_classes():
    T = 10
    robot = RobotModel(urdf_filename=TESTER_KIN, time_derivs=[0, 1])
    name = robot.get_name()
    b = OptimizationBuilder(T, robots=[robot])
    assert list(b._decision_variables.keys()) == [f"{name}/q/x", f"{name}/dq/x"]
    assert b._decision_variables[f"{name}/q/x"].shape == (3, T) and b._decision_variables[f"{name}/dq/x"].shape == (3, T - 1)
    b.integrate_model_states(name, 1, 0.1)
    assert b._lin_eq_constraints.numel() == 3 * (T - 1)
    b.enforce_model_limits(name)
    assert b._lin_ineq_constraints.numel() == 2 * 3 * T  # "_l" and "_r" blocks
    assert list(b._lin_ineq_constraints.keys()) == [f"__{name}_model_limit_0___l", f"__{name}_model_limit_0___r"]
    b.fix_configuration(name)
    assert b._lin_eq_constraints.numel() == 3 * (T - 1) + 3
    b.add_cost_term("c", sumsqr(b.get_model_states(name, 1)))
    opt = b.build()
    assert isinstance(opt, QuadraticCostLinearConstraints)
    assert opt.nv == opt.nk + 2 * opt.na and opt.nx == 3 * T + 3 * (T - 1)
    with pytest.raises(AssertionError):
        OptimizationBuilder(1, robots=[robot])  # T too low for time_derivs [0,1]
    with pytest.raises(AssertionError):
        b.add_cost_term("bad", b.get_model_states(name))  # not scalar
    b2 = OptimizationBuilder(1, tasks=[TaskModel("t", 2)])
    b2.add_cost_term("c", sumsqr(b2.get_model_state("t", 0)))
    assert isinstance(b2.build(), QuadraticCostUnconstrained)


def test_ik_example_builder_matches_reference_counts():
    robot = RobotModel(urdf_filename=KUKA_KIN)
    name = robot.get_name()
    b = OptimizationBuilder(1, robots=robot)
    qn = b.add_parameter("q_nominal", robot.ndof)
    pg = b.add_parameter("p_goal", 3)
    q = b.get_model_state(name, 0)
    b.add_equality_constraint("end_goal", robot.get_global_link_position("end_effector_ball", q), pg)
    b.add_cost_term("nominal", sumsqr(q - qn))
    b.enforce_model_limits(name)
    opt = b.build()
    assert isinstance(opt, QuadraticCostNonlinearConstraints)
    assert (opt.nx, opt.np, opt.nk, opt.nh, opt.nv) == (7, 10, 14, 3, 20)
    kind, _ = lower(opt)
    assert kind == optas_amd._lib.OH_PROBLEM_IK
    b.add_cost_term("extra", sumsqr(q))
    kind, spec = lower(b.build())  # a second cost term is outside the hand-written IK family: the generic tape family takes it
    assert kind == optas_amd._lib.OH_PROBLEM_TAPE and spec.tape.nx == 7 and (spec.tape.n_ineq, spec.tape.n_eq) == (14, 3)
    big = OptimizationBuilder(T=9, robots=robot)  # 63 variables with a coupling cost: no structured family; the generic one takes it with its
    qs = big.get_model_states(name)               # limited-memory solver (refused until round 3: the dense BFGS stopped at 32 variables)
    big.add_cost_term("coupled", sumsqr(robot.get_global_link_position("end_effector_ball", qs[:, 0]) - robot.get_global_link_position("end_effector_ball", qs[:, 8])))
    kind, spec = lower(big.build())
    assert kind == optas_amd._lib.OH_PROBLEM_TAPE and spec.tape.nx == 63
    huge = OptimizationBuilder(T=600, robots=robot)  # 4200 variables: beyond OH_TAPE_MAX_N, refused loudly, never approximated
    qh = huge.get_model_states(name)
    huge.add_cost_term("coupled", sumsqr(robot.get_global_link_position("end_effector_ball", qh[:, 0]) - robot.get_global_link_position("end_effector_ball", qh[:, 8])))
    with pytest.raises(LoweringError):
        lower(huge.build())


def test_figure_eight_builder_and_lowering():
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from examples.figure_eight_plan import setup_solver

    kuka, opt = setup_solver(build_only=True)
    assert isinstance(opt, NonlinearCostNonlinearConstraints)
    assert (opt.nx, opt.np, opt.nk, opt.na, opt.ng, opt.nh, opt.nv) == (693, 7, 0, 357, 0, 200, 1114)
    kind, spec = lower(opt)
    assert kind == optas_amd._lib.OH_PROBLEM_FIGURE_EIGHT and spec.link == "end_effector_ball" and spec.T == 50
    assert np.isclose(spec.dt, 10 / 49) and spec.w_path == 1000.0 and spec.w_vel == 0.01 and spec.local_path.shape == (50, 3)
    off = opt.decision_variables.offsets()
    assert off == {"kuka/q/x": 0, "kuka/dq/x": 350}
    assert opt.parameters.offsets()["qc"] == 0


def test_chain_folding_matches_oracle():
    robot = RobotModel(urdf_filename=KUKA_KIN)
    ch = robot.kinematic_chain("end_effector_ball")
    fc = FoldedChain(OracleRobot(KUKA_KIN), "end_effector_ball")
    assert ch.ndof == 7 and ch.n_chain == 7 and list(ch.qidx)[:7] == list(range(7)) and list(ch.jtype)[:7] == [0] * 7
    for k in range(7):
        assert np.allclose(np.array(ch.R0[k]).reshape(3, 3), fc.R0[k], atol=1e-15)
        assert np.allclose(ch.p0[k], fc.p0[k], atol=1e-15) and np.allclose(ch.axis[k], fc.axis[k], atol=1e-15)
    assert np.allclose(np.array(ch.R_tool).reshape(3, 3), fc.R_tool, atol=1e-15) and np.allclose(ch.p_tool, [0, 0, 0.2323], atol=1e-15)
    assert list(ch.axcode)[:7] == [3, 2, 3, -2, 3, 2, 3] and list(ch.r0ident)[:7] == [0, 1, 1, 1, 1, 1, 0]
    mid = robot.kinematic_chain("lwr_arm_4_link")
    assert mid.n_chain == 4 and mid.ndof == 7
    assert C.sizeof(optas_amd._lib.oh_chain) == 2952  # the block that is broadcast over RCCL (< 3 KB)
    t = RobotModel(urdf_filename=TESTER_KIN).kinematic_chain("eff")
    assert list(t.jtype)[:3] == [0, 0, 1] and np.allclose(t.p_tool, [0, 0, 0.5])
    with pytest.raises(AssertionError):
        robot.kinematic_chain("no_such_link")


def test_batched_container_conversions_match_the_scalar_ones():
    from optas_amd.expr import ParamRef
    from optas_amd.sx_container import SXContainer

    c = SXContainer()
    c["a"] = ParamRef("a", 3, 4)
    c["b"] = ParamRef("b", 2, 1)
    c["s"] = ParamRef("s", 1, 1)
    rng = np.random.default_rng(0)
    B = 5
    d = {"a": rng.normal(size=(B, 3, 4)), "b": rng.normal(size=(B, 2)), "s": rng.normal(size=B)}
    V = c.dict2vec_batch(d, B)
    for i in range(B):
        assert np.array_equal(V[i], c.dict2vec({"a": d["a"][i], "b": d["b"][i], "s": d["s"][i]}))
    back = c.vec2dict_batch(V)
    assert np.array_equal(back["a"], d["a"]) and back["b"].shape == (B, 2, 1) and np.array_equal(back["s"].reshape(-1), d["s"])
    assert np.array_equal(c.dict2vec_batch({"b": d["b"]}, B)[:, :12], np.zeros((B, 12)))  # missing labels are zero-filled
