"""Development probe: velocity-limited figure-eight, 16 384 instances: optima reached with the hybrid switch at 1e-5 and at 1e-4 x w_path."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = 16384
rng = np.random.default_rng(5)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
out = {}
for sw in ("1e-5", "2e-5", "3e-5", "5e-5", "1e-4"):
    os.environ["OH_HYB_SWITCH"] = sw
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    r = solver.solve_batch_arrays(x0, qcs)
    out[sw] = r
    print(sw, "converged", (r.status == 0).mean(), "iters p50", np.median(r.iters), "mean", r.iters.mean(), "f mean", r.f.mean(), flush=True)
a = out["1e-5"]
for sw in ("2e-5", "3e-5", "5e-5", "1e-4"):
    d = out[sw].f - a.f
    print(sw, "same optimum (1e-8 rel):", (np.abs(d) <= 1e-8 * a.f).mean(), " higher:", (d > 1e-8 * a.f).sum(), " lower:", (d < -1e-8 * a.f).sum())
