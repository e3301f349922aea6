"""Development probe: tighter tolerances than the default 1e-6 on the trajectory families (does an end game exist below the default?)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = 32768
for vel in (False, True):
    for tol in (1e-7, 1e-8, 1e-9, 1e-10):
        rng = np.random.default_rng(3)
        qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
        kuka, solver = setup_solver(velocity_limits=True if vel else None, solver_options={"max_iter": 600, "tol": tol})
        x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
        r = solver.solve_batch_arrays(x0, qcs)
        ok = r.status == 0
        print(f"figure-eight vel={vel} tol={tol:g}: converged {ok.mean():.5f} iters p50 {np.median(r.iters):.0f} p99 {np.percentile(r.iters, 99):.0f} max {r.iters.max()} stat max {r.kkt[ok, 0].max() if ok.any() else float('nan'):.2e}"
              f" not converged stat median {np.median(r.kkt[~ok, 0]) if (~ok).any() else 0:.2e}", flush=True)
        solver.backend.close()
