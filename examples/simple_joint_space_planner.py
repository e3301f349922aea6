"""The reference's example/simple_joint_space_planner.py (lines 15-73) written against optas_amd: a T = 20 joint-space plan of the KUKA med7
from the current configuration to an end-effector pose, with the elbow and the end effector kept above the table, a nominal-posture cost and
velocity / acceleration costs (derivs_align: 7 x 20 configurations + 7 x 20 velocities = 280 decision variables; 154 equality rows, 40
inequality rows).  It matches none of the structured kernel families (final-pose rows, height rows of two links, an acceleration cost across
neighbouring velocity columns), so HIPSolver compiles it to one instruction tape and the generic family solves it on the GPU
(OH_PROBLEM_TAPE; with more than 48 variables its inner solver is the limited-memory BFGS)."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr
from optas_amd.solver import HIPSolver

EE_LINK, ELBOW_LINK = "lbr_link_ee", "lbr_link_3"


def setup_solver(T=20, duration=4.0, build_only=False, solver_options=None):
    dt = duration / float(T - 1)
    robot = optas_amd.RobotModel.builtin("med7", time_derivs=[0, 1])
    name = robot.get_name()
    builder = OptimizationBuilder(T=T, robots=[robot], derivs_align=True)
    qn = builder.add_parameter("nominal_joint_state", robot.ndof)
    qc = builder.add_parameter("current_joint_state", robot.ndof)
    pg = builder.add_parameter("position_goal", 3)
    og = builder.add_parameter("orientation_goal", 4)
    builder.fix_configuration(name, config=qc)  # initial configuration
    qF = builder.get_model_state(name, -1)
    builder.add_equality_constraint("final_position", robot.get_global_link_position(EE_LINK, qF), pg)
    builder.add_equality_constraint("final_orientation", robot.get_global_link_quaternion(EE_LINK, qF), og)
    builder.integrate_model_states(name, time_deriv=1, dt=dt)
    zpad = 0.05
    for t in range(T):
        q = builder.get_model_state(name, t)
        builder.add_cost_term(f"nominal_{t}", 0.1 * sumsqr(q - qn))
        builder.add_geq_inequality_constraint(f"eff_safe_{t}", robot.get_global_link_position(EE_LINK, q)[2] + zpad)
        builder.add_geq_inequality_constraint(f"elbow_safe_{t}", robot.get_global_link_position(ELBOW_LINK, q)[2] + zpad)
    dQ = builder.get_model_states(name, time_deriv=1)
    builder.add_cost_term("minimize_velocity", 0.1 * sumsqr(dQ))
    ddQ = (dQ[:, 1:] - dQ[:, :-1]) * (1.0 / dt)
    builder.add_cost_term("minimize_acceleration", 10.0 * sumsqr(ddQ))
    builder.fix_configuration(name, t=-1, time_deriv=1)  # final velocity is zero
    optimization = builder.build()
    if build_only:
        return robot, optimization
    return robot, HIPSolver(optimization).setup("hip_sqp", solver_options)


def main():
    robot, solver = setup_solver()
    name = robot.get_name()
    q0 = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    qg = np.deg2rad([20, 55, -10, -70, 10, -40, 15])
    solver.reset_parameters({"nominal_joint_state": q0, "current_joint_state": q0, "position_goal": robot.get_global_link_position(EE_LINK, qg),
                             "orientation_goal": robot.get_global_link_quaternion(EE_LINK, qg)})
    solver.reset_initial_seed({f"{name}/q/x": np.tile(q0.reshape(-1, 1), (1, 20))})
    sol = solver.solve()
    print("did_solve", solver.did_solve(), "evaluations", solver.number_of_iterations())
    print(np.rad2deg(np.asarray(sol[f"{name}/q"])[:, -1]))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
