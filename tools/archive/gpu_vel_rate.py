"""Development probe: throughput of the figure-eight family with joint-velocity limit rows (enforce_model_limits(time_deriv=1))."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
for B in (1024, 16384, 65536):
    rng = np.random.default_rng(5)
    qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    be = solver.backend
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    inner = getattr(be, "be", be)
    r = solver.solve_batch_arrays(x0, qcs)
    if hasattr(inner, "set_profiling"):
        inner.set_profiling(True)
    t = time.perf_counter(); r = solver.solve_batch_arrays(x0, qcs); wall = time.perf_counter() - t
    tm = be.timing() if hasattr(be, "timing") else {}
    print(f"B={B}: device {tm.get('solve_ms', float('nan')):.1f} ms wall {wall*1e3:.1f} ms -> {B/(tm.get('solve_ms', wall*1e3)*1e-3):.0f} solves/s; converged {(r.status==0).mean():.4f} iters p50 {np.median(r.iters):.0f} p90 {np.percentile(r.iters,90):.0f} p99 {np.percentile(r.iters,99):.0f} p99.9 {np.percentile(r.iters,99.9):.0f} max {r.iters.max()} "
          f"launched {tm.get('iterations_launched')} compactions {tm.get('compactions')} eval {tm.get('eval_ms',0):.1f} couple {tm.get('couple_ms',0):.1f} step {tm.get('step_ms',0):.1f}")
    be.close()
