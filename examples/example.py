"""The reference's example/example.py (lines 13-60), written against optas_amd: same builder calls in the same
order; CasADiSolver(...).setup("ipopt") becomes HIPSolver(...).setup("hip_sqp").  The visualiser part is out of scope."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr
from optas_amd.solver import HIPSolver

END_EFFECTOR = "end_effector_ball"


def setup_solver(robot_name="kuka_lwr", solver_options=None, build_only=False):
    robot = optas_amd.RobotModel.builtin(robot_name)
    name = robot.get_name()
    builder = OptimizationBuilder(1, robots=robot)
    qn = builder.add_parameter("q_nominal", robot.ndof)
    pg = builder.add_parameter("p_goal", 3)
    q = builder.get_model_state(name, 0)
    p = robot.get_global_link_position(END_EFFECTOR, q)
    builder.add_equality_constraint("end_goal", p, pg)
    builder.add_cost_term("nominal", sumsqr(q - qn))
    builder.enforce_model_limits(name)
    optimization = builder.build()
    if build_only:
        return robot, optimization
    return robot, HIPSolver(optimization).setup("hip_sqp", solver_options)


def main():
    robot, solver = setup_solver()
    name = robot.get_name()
    q_nominal = optas_amd.deg2rad([0, 45, 0, -90, 0, -45, 0])
    p_nominal = robot.get_global_link_position(END_EFFECTOR, q_nominal)
    p_goal = np.asarray(p_nominal).reshape(-1) + np.array([0.0, 0.3, -0.2])
    solver.reset_parameters({"q_nominal": q_nominal, "p_goal": p_goal})
    # the reference passes the key f"{name}/q", which is not a decision-variable label ("{name}/q/x"), so its seed is
    # zero-filled (sx_container.py:121); kept as written
    solver.reset_initial_seed({f"{name}/q": q_nominal})
    solution = solver.solve()
    print("did_solve", solver.did_solve(), "evaluations", solver.number_of_iterations(), "f", solver.stats()["f"][0])
    print("q =", np.asarray(solution[f"{name}/q"]).reshape(-1))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
