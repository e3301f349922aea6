"""Development probe: horizon lengths around the limits of the persistent kernels (64 free knots) for the plain and the velocity-limited figure-eight."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
for vel in (False, True):
    for T in (3, 4, 10, 33, 64, 65, 66, 67, 100, 128):
        for B in (1, 300, 20000):
            rng = np.random.default_rng(T * 7 + B)
            qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
            try:
                kuka, solver = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True if vel else None, solver_options={"max_iter": 600, "tol": 1e-6})
                x0 = np.zeros((B, solver.opt.nx)); x0[:, : 7 * T] = np.repeat(qcs, T, axis=0).reshape(B, 7 * T)
                r = solver.solve_batch_arrays(x0, qcs)
                tm = solver.backend.timing()
                ok = r.status == 0
                print(f"vel={vel} T={T} B={B}: converged {ok.mean():.4f} iters p50 {np.median(r.iters):.0f} max {r.iters.max()} tail {tm['tail_iterations'] > 0} launched {tm['iterations_launched']} finite {np.isfinite(r.x).all()}", flush=True)
                solver.backend.close()
            except Exception as e:
                print(f"vel={vel} T={T} B={B}: {type(e).__name__}: {str(e)[:150]}", flush=True)
