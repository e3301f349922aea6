"""The reference's example/figure_eight_plan.py Planner.setup_solver (lines 16-113), written against
optas_amd: same builder calls in the same order; the CasADi loop at :90-96 becomes path_in_frame(...)
and CasADiSolver(...).setup("ipopt") becomes HIPSolver(...).setup("hip_sqp")."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import path_in_frame, sumsqr
from optas_amd.solver import HIPSolver


def figure_eight_local_path(T: int, Tmax: float):
    t = np.linspace(0.0, Tmax, T)
    path = np.zeros((3, T))
    path[0, :] = 0.2 * np.sin(t * np.pi * 0.5)
    path[1, :] = 0.1 * np.sin(t * np.pi)
    return t, path


def setup_solver(robot_name="kuka_lwr", link_ee="end_effector_ball", T=50, Tmax=10.0, solver_options=None, build_only=False, limits=None, obstacles=None, sphere_links=None,
                 velocity_limits=None):
    t, local = figure_eight_local_path(T, Tmax)
    dt = float(t[1] - t[0])
    kuka = optas_amd.RobotModel.builtin(robot_name, time_derivs=[0, 1])
    kuka_name = kuka.get_name()
    builder = OptimizationBuilder(T=T, robots=[kuka])
    qc = builder.add_parameter("qc", kuka.ndof)
    builder.fix_configuration(kuka_name, config=qc)
    builder.fix_configuration(kuka_name, time_deriv=1)
    builder.integrate_model_states(kuka_name, time_deriv=1, dt=dt)
    Q = builder.get_model_states(kuka_name)
    pos_ee = kuka.get_global_link_position(link_ee, Q)
    pc = kuka.get_global_link_position(link_ee, qc)
    Rc = kuka.get_global_link_rotation(link_ee, qc)
    quatc = kuka.get_global_link_quaternion(link_ee, qc)
    path = path_in_frame(pc, Rc, local)
    builder.add_cost_term("ee_path", 1000.0 * sumsqr(path - pos_ee))
    dQ = builder.get_model_states(kuka_name, time_deriv=1)
    builder.add_cost_term("min_join_vel", 0.01 * sumsqr(dQ))
    builder.add_equality_constraint("no_eff_rot", kuka.get_global_link_quaternion(link_ee, Q), quatc)
    if limits is not None:  # not in the shipped script: joint limits, True = the model's own, or (lo, up)
        if limits is True:
            builder.enforce_model_limits(kuka_name)
        else:
            builder.enforce_model_limits(kuka_name, lo=limits[0], up=limits[1])
    if velocity_limits is not None:  # not in the shipped script (whose optimum exceeds the LWR's 1.92 rad/s on joint 0): True = the model's own, or (lo, up)
        if velocity_limits is True:
            builder.enforce_model_limits(kuka_name, time_deriv=1)
        else:
            builder.enforce_model_limits(kuka_name, time_deriv=1, lo=velocity_limits[0], up=velocity_limits[1])
    if obstacles is not None:  # not in the shipped script: sphere clearances (obstacle names; parameters are set at solve time)
        builder.sphere_collision_avoidance_constraints(kuka_name, list(obstacles), link_names=sphere_links)
    optimization = builder.build()
    if build_only:
        return kuka, optimization
    solver = HIPSolver(optimization).setup("hip_sqp", solver_options)
    return kuka, solver


def main():
    kuka, solver = setup_solver()
    name = kuka.get_name()
    qc = optas_amd.deg2rad([0, 30, 0, -90, 0, -30, 0])
    solver.reset_parameters({"qc": qc})
    solver.reset_initial_seed({f"{name}/q/x": np.tile(qc.reshape(-1, 1), (1, 50))})
    solution = solver.solve()
    print("did_solve", solver.did_solve(), "iterations", solver.number_of_iterations(), "f", solver.stats()["f"][0])
    plan = solver.interpolate(solution[f"{name}/q"], 10.0)
    print("q(5.0) =", plan(5.0))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
