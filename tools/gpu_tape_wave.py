"""The 280-variable planner (examples/simple_joint_space_planner.py) through the generic tape family: wavefront-per-instance path (default)
against the thread-per-instance paths (option tape_wave = 0).  python tools/gpu_tape_wave.py [wave|thread]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1] if len(sys.argv) > 1 else "wave"
if mode == "thread":
    os.environ["OH_DEBUG_OPTIONS"] = "tape_wave=0"
from examples.simple_joint_space_planner import setup_solver
g = np.load(os.path.join(ROOT, "tests", "golden", "planner_golden.npz"))
t0 = time.time()
robot, solver = setup_solver(solver_options={"max_iter": 400000})
print(mode, "setup s", round(time.time() - t0, 2), flush=True)
name = robot.get_name()
P = g["p"]; B = len(P)
for rep in range(2):
    solver.reset_parameters_batch({"nominal_joint_state": P[:, :7], "current_joint_state": P[:, 7:14], "position_goal": P[:, 14:17], "orientation_goal": P[:, 17:]})
    solver.reset_initial_seed_batch({f"{name}/q/x": np.stack([np.tile(g["q0"].reshape(-1, 1), (1, 20))] * B)})
    t0 = time.time()
    sols = solver.solve_batch()
    st = solver.stats()
    print(mode, "B", B, "wall s", round(time.time() - t0, 3), "device ms", st.get("solve_ms"), "status", st["status"], "evals", st.get("iter_count"), "f", st["f"], "gold", g["f"], flush=True)
be = solver.backend
try:
    print("flags", {k: be.flag(k) for k in ("tape_wave", "tape_levels", "tape_passes")})
except Exception as e:
    print("flags n/a", e)
