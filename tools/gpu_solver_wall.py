"""GPU: wall-clock latency of Solver.solve() at B = 1, as a user of the reference calls it (reset_parameters / reset_initial_seed / solve -> dict), against the
device time of the same solve: figure-eight (config 2), torque MPC (config 5), IK (config 1), planner (generic family).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med(fn, reps=40):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = np.asarray(ts) * 1e3
    return float(np.median(ts)), float(np.percentile(ts, 90))


def devms(be):
    while not hasattr(be, "solve_ms") and not hasattr(be, "timing") and hasattr(be, "be"):
        be = be.be
    if hasattr(be, "solve_ms"):
        return float(be.solve_ms())
    return float(be.timing()["solve_ms"])


def main():
    out = {}
    from examples.figure_eight_plan import setup_solver as fig8
    robot, solver = fig8()
    name = robot.get_name()
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    seed = {f"{name}/q/x": np.tile(qc.reshape(-1, 1), (1, 50))}

    def run():
        solver.reset_parameters({"qc": qc})
        solver.reset_initial_seed(seed)
        return solver.solve()

    w = med(run)
    out["config2_figure_eight"] = {"wall_ms_p50": w[0], "wall_ms_p90": w[1], "device_ms": devms(solver.backend), "iterations": int(np.asarray(solver.stats()["iterations"]).reshape(-1)[0])}

    from examples.torque_mpc import build_problem, figure_eight_goal
    import optas_amd as optas
    T, dt = 30, 0.1
    robot, link, opt = build_problem(T, dt, effort=58.0)
    solver = optas.HIPSolver(opt).setup("hip_sqp")
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])  # (the example's own posture; at [0, 45, ...] the arm's holding torque, 58.2 N m, is above the 58 N m limit: 105 steps)
    goal = figure_eight_goal(robot, link, qc, T, dt)
    seed = {f"{robot.get_name()}/q/x": np.tile(qc[:, None], (1, T))}

    def run5():
        solver.reset_parameters({"qc": qc, "dqc": np.zeros(7), "goal": goal})
        solver.reset_initial_seed(seed)
        return solver.solve()

    w = med(run5, reps=20)
    out["config5_torque"] = {"wall_ms_p50": w[0], "wall_ms_p90": w[1], "device_ms": devms(solver.backend), "iterations": int(np.asarray(solver.stats()["iterations"]).reshape(-1)[0])}

    from examples.simple_joint_space_planner import setup_solver as planner
    robot, solver = planner()
    name = robot.get_name()
    q0 = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    qg = np.deg2rad([20, 55, -10, -70, 10, -40, 15])
    pd = {"nominal_joint_state": q0, "current_joint_state": q0, "position_goal": robot.get_global_link_position("lbr_link_ee", qg),
          "orientation_goal": robot.get_global_link_quaternion("lbr_link_ee", qg)}
    seed = {f"{name}/q/x": np.tile(q0.reshape(-1, 1), (1, 20))}

    def runp():
        solver.reset_parameters(pd)
        solver.reset_initial_seed(seed)
        return solver.solve()

    w = med(runp, reps=20)
    out["planner_tape"] = {"wall_ms_p50": w[0], "wall_ms_p90": w[1], "device_ms": devms(solver.backend), "evaluations": int(np.asarray(solver.stats()["iterations"]).reshape(-1)[0])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
