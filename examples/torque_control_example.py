"""The tracking controller of the reference's example/torque_control_example.py (:19-104) against the mirror front end: one decision block
dq (7 joint velocities of the KUKA med7, T = 1), parameters qc (current configuration) and pg (goal position + xyzw quaternion); the
end-effector step of one control period seen from the current end-effector frame is matched with the goal (position weight 1e3, orientation
weight 10 through the reference's own Quaternion.getrotm), joint speed is penalised, and the squared position error is bounded per axis
(1e-6, 1e-8, 1e-8).  Cost quadratic, rows squares of affine expressions: class QuadraticCostNonlinearConstraints, which the reference hands to
CasADi's sqpmethod.  HIPSolver recognises the rows as bands |e| <= sqrt(c) (lowering._band_rows) and solves the equivalent QP in the dense-QP
family, P, q, M, c read off the problem's tape on the device; a batch of (qc, pg) pairs is one launch.  The PyBullet loop of the script (:107-198)
is replaced by integrating the commanded velocity."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import optas_amd as optas  # noqa: E402
from optas_amd.spatialmath import I3, Quaternion, skew  # noqa: E402


class TrackingController:
    def __init__(self, dt, solver_options=None, build_only=False):
        link_ee = "lbr_link_ee"
        kuka = optas.RobotModel.builtin("med7", time_derivs=[1])  # joint velocities only
        self.kuka, self.kuka_name, self.dt = kuka, kuka.get_name(), dt
        builder = optas.OptimizationBuilder(1, robots=[kuka], derivs_align=True)
        qc = builder.add_parameter("qc", kuka.ndof)
        pg = builder.add_parameter("pg", 7)
        dq = builder.get_model_state(self.kuka_name, t=0, time_deriv=1)

        dp = kuka.get_global_link_geometric_jacobian(link_ee, qc) @ dq  # end-effector twist
        pc = kuka.get_global_link_position(link_ee, qc)
        Rc = kuka.get_global_link_rotation(link_ee, qc)

        # the step of one period, expressed in the current end-effector frame
        p = Rc.T @ dt @ dp[:3]
        R = Rc.T @ (skew(dp[3:]) * dt + I3())

        # the goal in the same frame
        Rg = Quaternion(pg[3], pg[4], pg[5], pg[6]).getrotm()
        pg_ee = -Rc.T @ pc + Rc.T @ pg[:3]
        Rg_ee = Rc.T @ Rg

        diffp = p - pg_ee[:3]
        diffR = Rg_ee.T @ R

        builder.add_cost_term("match_p", diffp.T @ optas.diag([1e3, 1e3, 1e3]) @ diffp)
        builder.add_cost_term("min_dq", 0.01 * optas.sumsqr(dq))
        builder.add_cost_term("match_r", 1e1 * optas.sumsqr(diffR - I3()))

        builder.add_leq_inequality_constraint("eff_x", diffp[0] * diffp[0], 1e-6)
        builder.add_leq_inequality_constraint("eff_y", diffp[1] * diffp[1], 1e-8)
        builder.add_leq_inequality_constraint("eff_z", diffp[2] * diffp[2], 1e-8)

        self.optimization = builder.build()
        self.solver = None if build_only else optas.HIPSolver(self.optimization).setup("hip_sqp", solver_options)

    def compute_target_velocity(self, qc, pg):
        self.solver.reset_parameters({"qc": qc, "pg": pg})
        solution = self.solver.solve()
        return np.asarray(solution[f"{self.kuka_name}/dq"]).reshape(-1)


def main(ticks=25):
    dt = 1.0 / 500.0
    q = optas.deg2rad([0, 30, 0, -90, 0, 60, 0])
    ctrl = TrackingController(dt)
    start = np.asarray(ctrl.kuka.get_global_link_position("lbr_link_ee", q)).reshape(3)
    for k in range(ticks):
        goal = start + np.array([0.0, 0.0004 * (k + 1), 0.0])  # the box of the script, pushed along y at 0.2 m/s
        dq = ctrl.compute_target_velocity(q, np.concatenate([goal, [0.0, 1.0, 0.0, 0.0]]))
        q = q + dt * dq
        if k % 5 == 0 or k == ticks - 1:
            err = np.asarray(ctrl.kuka.get_global_link_position("lbr_link_ee", q)).reshape(3) - goal
            print(f"tick {k}: |dq| = {np.linalg.norm(dq):.4f}, position error = {np.abs(err).max():.2e}, did_solve = {ctrl.solver.did_solve()}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
