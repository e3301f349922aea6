"""The reference's example/planar_idk.py (lines 10-59) written against optas_amd: differential kinematics of the planar 3-DoF arm as a QP
(class QuadraticCostLinearConstraints: the reference solves it with CVXOPT), through the dense-QP family of HIPSolver.
min ||dq||^2  s.t.  J_xy(q) dq = dx,  |dq_i| <= 0.1,  -70 deg <= phi(q) + dt * J_phi(q) dq <= 0."""
import os

import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import atan2, sumsqr
from optas_amd.solver import HIPSolver

PLANAR_KIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "planar_3dof.kin.json")


def setup_solver(build_only=False, solver_options=None):
    robot = optas_amd.RobotModel(urdf_filename=PLANAR_KIN, time_derivs=[1])
    name, link_ee = robot.get_name(), "end"
    builder = OptimizationBuilder(T=1, robots=[robot], derivs_align=True)
    dq = builder.get_model_states(name, time_deriv=1)
    q = builder.add_parameter("q", robot.ndof)
    J = robot.get_global_link_linear_jacobian_function(link_ee)
    quat = robot.get_global_link_quaternion_function(link_ee)
    phi = lambda qq: 2.0 * atan2(quat(qq)[2], quat(qq)[3])
    J_phi = robot.get_global_link_angular_geometric_jacobian_function(link=link_ee)
    dx, dt, lim = [0.01, 0.0], 0.01, 0.1
    builder.add_cost_term("cost", sumsqr(dq))
    builder.add_equality_constraint("FDK", (J(q)[0:2, :]) @ dq, dx)
    builder.add_bound_inequality_constraint("joint", [-lim] * 3, dq, [lim] * 3)
    builder.add_bound_inequality_constraint("task", -70 * (np.pi / 180.0), phi(q) + dt * (J_phi(q)[2, :]) @ dq, 0.0)
    optimization = builder.build()
    if build_only:
        return robot, optimization
    return robot, HIPSolver(optimization).setup("hip_sqp", solver_options)


def main():
    robot, solver = setup_solver()
    q_t = [2.39, -2.55, -0.46]
    solver.reset_initial_seed({f"{robot.get_name()}/dq/x": [0.0, 0.0, 0.0]})
    solver.reset_parameters({"q": q_t})
    solution = solver.solve()
    print(np.asarray(solution[f"{robot.get_name()}/dq"]).reshape(-1))
    Jxy = np.asarray(robot.get_global_link_linear_jacobian("end", np.asarray(q_t)))[0:2]
    print(np.linalg.pinv(Jxy) @ np.array([0.01, 0.0]))  # the reference prints the same comparison (:58)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
