"""Config 1 of BASELINE.json written against optas_amd: nearest-to-nominal inverse kinematics for the KUKA LWR with a position goal
for `end_effector_ball` and the URDF joint limits (what the reference's example/example.py:13-60 sets up), solved by
HIPSolver("hip_sqp") instead of CasADiSolver("ipopt").  Visualisation is out of scope."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr
from optas_amd.solver import HIPSolver

END_EFFECTOR = "end_effector_ball"
NOMINAL_DEG = (0, 45, 0, -90, 0, -45, 0)
GOAL_OFFSET = (0.0, 0.3, -0.2)


def ik_problem(robot):
    """T = 1 problem: parameters q_nominal (ndof) and p_goal (3); one equality block, one cost term, the model limits."""
    b = OptimizationBuilder(1, robots=robot)
    nominal = b.add_parameter("q_nominal", robot.ndof)
    goal = b.add_parameter("p_goal", 3)
    state = b.get_model_state(robot.get_name(), 0)
    b.add_equality_constraint("end_goal", robot.get_global_link_position(END_EFFECTOR, state), goal)
    b.add_cost_term("nominal", sumsqr(state - nominal))
    b.enforce_model_limits(robot.get_name())
    return b.build()


def setup_solver(robot_name="kuka_lwr", solver_options=None, build_only=False):
    robot = optas_amd.RobotModel.builtin(robot_name)
    problem = ik_problem(robot)
    if build_only:
        return robot, problem
    return robot, HIPSolver(problem).setup("hip_sqp", solver_options)


def script_inputs(robot):
    """Nominal configuration and goal of the script: the end-effector position at the nominal pose shifted by GOAL_OFFSET."""
    q_nominal = optas_amd.deg2rad(NOMINAL_DEG)
    reached = np.asarray(robot.get_global_link_position(END_EFFECTOR, q_nominal)).reshape(-1)
    return q_nominal, reached + np.asarray(GOAL_OFFSET)


def main():
    robot, solver = setup_solver()
    label = robot.get_name() + "/q"
    q_nominal, p_goal = script_inputs(robot)
    solver.reset_parameters({"q_nominal": q_nominal, "p_goal": p_goal})
    # The reference seeds with the key "{name}/q", which is not a decision-variable label ("{name}/q/x"): its seed is therefore
    # zero-filled (sx_container.py:121).  Reproduced as written.
    solver.reset_initial_seed({label: q_nominal})
    answer = solver.solve()
    print("did_solve", solver.did_solve(), "evaluations", solver.number_of_iterations(), "f", solver.stats()["f"][0])
    print("q =", np.asarray(answer[label]).reshape(-1))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
