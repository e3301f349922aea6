import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
for B in (65536,):
    rng = np.random.default_rng(5)
    qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    for _ in range(3):
        r = solver.solve_batch_arrays(x0, qcs)
        tm = solver.backend.timing()
        print("B", B, "ms", round(tm["solve_ms"], 2), "rate", round(B / tm["solve_ms"] * 1e3), "conv", (r.status == 0).mean(), "it p50/max", np.median(r.iters), r.iters.max(), "f sum", repr(float(r.f.sum())))
