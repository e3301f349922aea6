"""Development probe: one velocity-limited batch under rocprofv3 --kernel-trace (per-launch durations of k_step_lg as the batch drains)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(5)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
r = solver.solve_batch_arrays(x0, qcs)
print("converged", (r.status == 0).mean(), solver.backend.timing())
