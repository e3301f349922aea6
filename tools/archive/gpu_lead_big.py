"""Development probe: the lead-joint family (figure_eight_plan_6dof.py) at a large batch and tight tolerance."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan_6dof import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
for B, tol in ((131072, 1e-6), (32768, 1e-9)):
    kuka, solver = setup_solver(solver_options={"max_iter": 300, "tol": tol})
    be = solver.backend
    rng = np.random.default_rng(9)
    qc = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    lead = qc[:, :1] + 0.1 * np.sin(np.linspace(0, np.pi, 50))[None] * rng.uniform(-1, 1, (B, 1))
    lead[:, :2] = qc[:, :1]
    p = np.ascontiguousarray(np.concatenate([qc[:, 1:], qc[:, :1], lead], 1))
    x0 = np.zeros((B, 594)); x0[:, :300] = np.tile(qc[:, 1:], (1, 50))
    inner = getattr(be, "be", be)
    r = inner.solve(x0, p)
    ok = r.status == 0
    print(f"lead family B={B} tol={tol:g}: converged {ok.mean():.5f} iters p50 {np.median(r.iters):.0f} p99 {np.percentile(r.iters, 99):.0f} max {r.iters.max()} finite {np.isfinite(r.x).all()} ms {inner.timing()['solve_ms']:.1f}", flush=True)
    inner.close()
