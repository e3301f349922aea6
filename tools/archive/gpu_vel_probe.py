"""Development probe: the velocity-limited figure-eight batch of DESIGN 7.4 (seed 5, 16 384 instances) under the Hessian modes: which instances
hit the iteration cap, and what their stationarity is when they do."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(5)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
for hess in ("hybrid", "gauss_newton"):
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6, "hessian": hess})
    be = solver.backend
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    r = solver.solve_batch_arrays(x0, qcs)
    r = solver.solve_batch_arrays(x0, qcs)
    tm = be.timing()
    bad = np.flatnonzero(r.status != 0)
    print(f"{hess}: device {tm.get('solve_ms', 0):.1f} ms; converged {(r.status==0).mean():.5f}; iters p50 {np.median(r.iters):.0f} p90 {np.percentile(r.iters,90):.0f} p99 {np.percentile(r.iters,99):.0f} max {r.iters.max()}; "
          f"cap hitters {bad.tolist()[:12]} stat {r.kkt[bad,0][:12]} launched {tm.get('iterations_launched')}")
    alone = [solver.solve_batch_arrays(x0[b:b+1], qcs[b:b+1]) for b in bad[:6]]
    print("   the same instances alone: iters", [int(a.iters[0]) for a in alone], "status", [int(a.status[0]) for a in alone])
    be.close()
