"""Development probe: the instances of the limits + spheres figure-eight batch that hit the iteration cap."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
from examples.dual_arm import SPHERE_LINKS
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = 16384
rng = np.random.default_rng(51)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
obs = np.concatenate([[0.45, 0.1 * i, 0.25 + 0.1 * i, 0.08] for i in range(2)])
p = np.concatenate([qcs, np.full((B, 4), 0.08), np.tile(obs, (B, 1))], 1)
for mi in (600, 3000):
    kuka, solver = setup_solver(limits=True, obstacles=["obs0", "obs1"], sphere_links=SPHERE_LINKS, solver_options={"max_iter": mi, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    r = solver.solve_batch_arrays(x0, p)
    bad = np.flatnonzero(r.status != 0)
    it = r.iters
    print("max_iter", mi, "not converged", bad, "kkt", r.kkt[bad], "iters p50/p99/p99.9/max", np.median(it), np.percentile(it, 99), np.percentile(it, 99.9), it.max(), "top iters", np.sort(it)[-8:], "ms", solver.backend.timing()["solve_ms"])
    solver.backend.close()
