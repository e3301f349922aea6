"""Development probe: one handle solving batches of changing sizes (pool re-carving, stale per-instance state) against fresh handles."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
rng = np.random.default_rng(123)
for vel in (None, True):
    kuka, reused = setup_solver(velocity_limits=vel, solver_options={"max_iter": 600, "tol": 1e-6})
    bad = 0
    for trial, B in enumerate([3000, 1, 70000, 17, 20000, 1024, 40000, 5, 20000]):
        qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
        x0 = np.zeros((B, reused.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
        r = reused.solve_batch_arrays(x0, qcs)
        _, fresh = setup_solver(velocity_limits=vel, solver_options={"max_iter": 600, "tol": 1e-6})
        f = fresh.solve_batch_arrays(x0, qcs)
        fresh.backend.close()
        same_bits = np.array_equal(r.x, f.x) and np.array_equal(r.iters, f.iters)
        ok = (r.status == 0).mean()
        df = np.abs(r.f - f.f).max()
        print(f"vel={vel} trial {trial} B={B}: converged {ok:.5f} identical to a fresh handle {same_bits} max |df| {df:.2e}", flush=True)
        bad += (not same_bits)
    reused.backend.close()
    print("vel", vel, "mismatches", bad)
