import os, sys
import numpy as np
sys.path.insert(0,'/root/repo')
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = 24576
rng = np.random.default_rng(9)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
r = solver.solve_batch_arrays(x0, qcs)
top=np.argsort(-r.iters)[:5]
print(os.environ.get("OH_HYB_SWITCH"), np.bincount(r.status,minlength=3), "iters top", r.iters[top], "idx", top, "mean", r.iters.mean())
