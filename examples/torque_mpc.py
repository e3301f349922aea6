"""BASELINE configs[4]: 7-DoF torque-control MPC with the inverse dynamics as equality constraints.

The reference has no such script -- example/torque_control_example.py plans joint velocities for one step and calls
``robot.rnea`` outside the optimiser (:198-200) -- so this is the synthetic problem the survey specifies, written with the reference's own
builder calls (robot med7: ``RobotModel.rnea`` needs a fixed first joint, optas/models.py:1748-1749).  One solve = one MPC tick:
parameters are the current joint state (qc, dqc) and the end-effector goal over the horizon.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import optas_amd as optas  # noqa: E402


def build_problem(T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, effort=None, velocity_limits=None, robot=None, link="lbr_link_ee"):
    """robot / link: another arm (any chain of 2 .. 7 revolute joints behind a fixed one that RobotModel.rnea accepts) instead of the med7."""
    if robot is None:
        robot = optas.RobotModel.builtin("med7", time_derivs=[0, 1, 2])
    name, n = robot.get_name(), robot.ndof
    eff = np.array([j.limit.effort for j in robot.urdf.joints if j.type != "fixed"]) if effort is None else np.broadcast_to(np.asarray(effort, dtype=float), (n,)).copy()
    tau = optas.TaskModel("tau", n, time_derivs=[0], dlim={0: [-eff, eff]})
    builder = optas.OptimizationBuilder(T, robots=[robot], tasks=[tau], derivs_align=True)
    qc = builder.add_parameter("qc", n)
    dqc = builder.add_parameter("dqc", n)
    goal = builder.add_parameter("goal", 3, T)
    Q, dQ, ddQ = (builder.get_model_states(name, d) for d in (0, 1, 2))
    TAU = builder.get_model_states("tau", 0)
    builder.fix_configuration(name, qc)
    builder.fix_configuration(name, dqc, time_deriv=1)
    builder.integrate_model_states(name, 1, dt)
    builder.integrate_model_states(name, 2, dt)
    builder.add_equality_constraint("dynamics", lhs=robot.rnea(Q, dQ, ddQ), rhs=TAU)
    builder.enforce_model_limits("tau")
    if velocity_limits is not None:  # (lo, up) per joint or True: the model's own (enforce_model_limits(name, time_deriv=1), builder.py:471-509)
        if velocity_limits is True:
            builder.enforce_model_limits(name, time_deriv=1)
        else:
            builder.enforce_model_limits(name, time_deriv=1, lo=velocity_limits[0], up=velocity_limits[1])
    P = robot.get_global_link_position_function(link, n=T)(Q)
    builder.add_cost_term("track", w_path * optas.sumsqr(P - goal))
    builder.add_cost_term("velocity", w_vel * optas.sumsqr(dQ))
    builder.add_cost_term("effort", w_tau * optas.sumsqr(TAU))
    return robot, link, builder.build()


def figure_eight_goal(robot, link, qc, T, dt):
    """First T dt seconds of the figure of eight of figure_eight_plan.py:90-96 in the end-effector frame at qc."""
    p0 = np.asarray(robot.get_global_link_position(link, qc)).reshape(3)
    R0 = np.asarray(robot.get_global_link_rotation(link, qc))
    ts = np.arange(T) * dt
    loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
    return p0[:, None] + R0 @ loc


def main(effort=None):
    T, dt = 30, 0.1
    robot, link, opt = build_problem(T, dt, effort=effort)
    solver = optas.HIPSolver(opt).setup("hip_sqp")
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    solver.reset_parameters({"qc": qc, "dqc": np.zeros(7), "goal": figure_eight_goal(robot, link, qc, T, dt)})
    solver.reset_initial_seed({f"{robot.get_name()}/q/x": np.tile(qc[:, None], (1, T))})
    sol = solver.solve()
    tau = sol["tau/y"]
    print(f"converged={solver.did_solve()} iterations={solver.number_of_iterations()} f={solver.stats()['f'][0]:.9f} "
          f"max|tau|={np.abs(tau).max(1).round(2)}")
    return 0 if solver.did_solve() else 1


if __name__ == "__main__":
    sys.exit(main())
