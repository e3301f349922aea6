"""Development probe: wall time per tape evaluation of the 280-variable planner (one thread per instance) for different numbers of L-BFGS pairs --
what share of an evaluation is the quasi-Newton vector work and what share the generated evaluator."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.simple_joint_space_planner import setup_solver
g = np.load(os.path.join(ROOT, "tests", "golden", "planner_golden.npz"))
P = g["p"][:1]
for pairs in (12, 4, 1):
    os.environ["OH_TAPE_LBFGS"] = str(pairs)
    robot, solver = setup_solver(solver_options={"max_iter": 3000})
    name = robot.get_name()
    solver.reset_parameters_batch({"nominal_joint_state": P[:, :7], "current_joint_state": P[:, 7:14], "position_goal": P[:, 14:17], "orientation_goal": P[:, 17:]})
    solver.reset_initial_seed_batch({f"{name}/q/x": np.stack([np.tile(g["q0"].reshape(-1, 1), (1, 20))])})
    solver.solve_batch()
    t0 = time.perf_counter()
    solver.solve_batch()
    wall = time.perf_counter() - t0
    ev = int(np.atleast_1d(solver.stats()["iterations"])[0])
    print(f"pairs {pairs}: {ev} evaluations in {wall * 1e3:.0f} ms = {wall / ev * 1e6:.0f} us per evaluation, status {solver.stats()['status']}", flush=True)
