"""The reference's example/planar_ik.py (lines 10-60) written against optas_amd: position IK of the planar 3-DoF arm with joint bounds and
a bound on the end-effector heading.  The problem matches none of the hand-written kernel families (the heading row is an atan2 of the
orientation), so HIPSolver compiles it to an instruction tape and the GPU interprets it (OH_PROBLEM_TAPE).  The heading is taken from the
rotation matrix (atan2(R10, R00)) instead of the quaternion the reference uses (2 atan2(qz, qw)): the same angle for a planar arm."""
import os

import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import atan2, sumsqr
from optas_amd.solver import HIPSolver

PLANAR_KIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "planar_3dof.kin.json")


def setup_solver(build_only=False, solver_options=None):
    robot = optas_amd.RobotModel(urdf_filename=PLANAR_KIN)
    name, link_ee = robot.get_name(), "end"
    builder = OptimizationBuilder(T=1, robots=[robot])
    q_T = builder.get_model_states(name)
    x_T = [1.2, 0.2]
    q_0 = [np.pi / 2.0, 0.0, 0.0]
    lim = 160.0 * np.pi
    fk = robot.get_global_link_position(link_ee, q_T)
    R = robot.get_global_link_rotation(link_ee, builder.get_model_state(name, 0))
    phi = atan2(R[1, 0], R[0, 0])
    builder.add_cost_term("cost", sumsqr(q_T - q_0))
    builder.add_equality_constraint("FK", fk[0:2], x_T)
    builder.add_bound_inequality_constraint("joint", [0.0, -lim, -lim], q_T, [np.pi, lim, lim])
    builder.add_bound_inequality_constraint("task", -70.0 * (np.pi / 180.0), phi, 0.0)
    optimization = builder.build()
    if build_only:
        return robot, optimization
    return robot, HIPSolver(optimization).setup("hip_sqp", solver_options)


def main():
    robot, solver = setup_solver()
    name = robot.get_name()
    solver.reset_initial_seed({f"{name}/q/x": [np.pi / 2.0, 0.0, 0.0]})
    sol = solver.solve()
    q = np.asarray(sol[f"{name}/q"]).reshape(-1)
    print("did_solve", solver.did_solve(), "evaluations", solver.number_of_iterations())
    print(q * (180.0 / np.pi))
    print(np.asarray(robot.get_global_link_position("end", q)).reshape(-1))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
