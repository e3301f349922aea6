"""Rates of the velocity-limited figure-eight (enforce_model_limits(name, time_deriv=1) on BASELINE config 2) over batch sizes: device time of the
second solve on a handle (the first loads the compiled kernels), convergence, steps."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver  # noqa: E402

QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
out = []
for B in (1, 1024, 16384, 65536, 262144):
    rng = np.random.default_rng(5)
    qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx))
    x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    ms = []
    for _ in range(3):
        r = solver.solve_batch_arrays(x0, qcs)
        ms.append(solver.backend.timing()["solve_ms"])
    out.append({"batch": B, "device_ms": float(np.median(ms[1:])), "solves_per_s": B / float(np.median(ms[1:])) * 1e3, "converged_frac": float((r.status == 0).mean()),
                "iters_p50": float(np.median(r.iters)), "iters_p99": float(np.percentile(r.iters, 99)), "iters_max": int(r.iters.max()),
                "stationarity_max": float(r.kkt[:, 0].max()), "feasibility_max": float(r.kkt[:, 1].max())})
    solver.backend.close()
print(json.dumps({"config": "figure_eight_plan.py T=50 with joint-velocity limits (686 extra rows), perturbed qc +-0.1, tol 1e-6; persistent kernel k_tail_vel for the whole batch",
                  "sizes": out}))
