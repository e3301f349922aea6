"""Development probe: the one cap hitter of 20 000 velocity-limited T = 100 instances (batched kernels: beyond the persistent kernel's 64 knots) with and without compaction."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
T = 100; B = 20000
rng = np.random.default_rng(T * 7 + B)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
x0 = np.zeros((B, 1393)); x0[:, : 7 * T] = np.repeat(qcs, T, axis=0).reshape(B, 7 * T)
for tag, env in (("default", {}), ("no compaction", {"OH_DEBUG_OPTIONS": "compaction=0"}), ("first 12000", {"N": "12000"}), ("first 11000..11999 only", {"LO": "10900", "N": "11100"})):
    for k in ("OH_DEBUG_OPTIONS",):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in env.items() if k.startswith("OH_")})
    lo, n = int(env.get("LO", 0)), int(env.get("N", B))
    kuka, solver = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6, "hessian": os.environ.get("HESS", "hybrid")})
    r = solver.solve_batch_arrays(x0[lo:n, : solver.opt.nx], qcs[lo:n])
    i = 10961 - lo
    print(tag, "status", np.bincount(r.status, minlength=3), "instance 10961: status", r.status[i], "iters", r.iters[i], "f", r.f[i], "stat", r.kkt[i, 0], "compactions", solver.backend.timing()["compactions"], "device ms", solver.backend.timing()["solve_ms"], "launched", solver.backend.timing()["iterations_launched"], flush=True)
    solver.backend.close()
