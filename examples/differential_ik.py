"""Velocity-level inverse kinematics as a small QP (the problem type of the reference's example/experiment1.py:14-148, "ExprIK/IK1"):
one decision block dq (ndof), parameter qc; cost  w1 ||dq||^2 + w2 ||J(qc) dq - v_goal||^2;  joint limits on qc + dt dq and a height
band on the end-effector after the step.  The problem class is QuadraticCostLinearConstraints, i.e. what the reference hands to OSQP;
here it goes through the dense-QP family of HIPSolver (its data P, q, M, c is read off the problem per instance)."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr, vertcat
from optas_amd.solver import HIPSolver


class DifferentialIK:
    def __init__(self, eff_link="end_effector_ball", planar_direction=(1.0, 0.0), dt=0.1, max_speed=0.1, height_band=(0.025, 0.15), solver_options=None,
                 build_only=False):
        self.robot = optas_amd.RobotModel.builtin("kuka_lwr", time_derivs=[1])
        self.name = self.robot.get_name()
        self.dt = dt
        b = OptimizationBuilder(1, robots=self.robot, derivs_align=True)
        dq = b.get_model_state(self.name, 0, time_deriv=1)
        qc = b.add_parameter("qc", self.robot.ndof)
        twist = self.robot.get_global_link_geometric_jacobian(eff_link, qc) @ dq  # 6 x 1, linear in dq
        goal_twist = vertcat(max_speed * np.asarray(planar_direction, dtype=float), np.zeros(4))
        b.add_cost_term("min_qd", 1.0 * sumsqr(dq))
        b.add_cost_term("eff_motion", 1000.0 * sumsqr(twist - goal_twist))
        q_next = qc + dt * dq
        b.add_leq_inequality_constraint("lower_qlim", self.robot.lower_actuated_joint_limits, q_next)
        b.add_leq_inequality_constraint("upper_qlim", q_next, self.robot.upper_actuated_joint_limits)
        p_next = self.robot.get_global_link_position(eff_link, qc) + dt * twist[:3]
        b.add_leq_inequality_constraint("lower_zlim", height_band[0], p_next[2])
        b.add_leq_inequality_constraint("upper_zlim", p_next[2], height_band[1])
        self.optimization = b.build()
        self.solver = None if build_only else HIPSolver(self.optimization).setup("hip_sqp", solver_options)

    def step(self, qc):
        """One control tick: the joint velocity and the configuration after dt."""
        self.solver.reset_parameters({"qc": qc})
        sol = self.solver.solve()
        dq = np.asarray(sol[f"{self.name}/dq"]).reshape(-1)
        return dq, np.asarray(qc, dtype=float) + self.dt * dq


def main():
    ik = DifferentialIK(height_band=(0.0, 2.0))
    q = optas_amd.deg2rad([0, 30, 0, -90, 0, 60, 0])
    for k in range(5):
        dq, q = ik.step(q)
        print(f"tick {k}: |dq| = {np.linalg.norm(dq):.4f}, did_solve = {ik.solver.did_solve()}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
