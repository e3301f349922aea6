"""Cost of the restart compaction on handles with inequality rows outside the persistent kernels (round 4): velocity-limited figure-eight T = 100 and
config 4 synthetic, default vs option compaction = 0.  python tools/gpu_compaction_cost.py"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "vel":
    from examples.figure_eight_plan import setup_solver
    B = int(sys.argv[2]); T = 100
    QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    rng = np.random.default_rng(T * 7 + B)
    qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    x0 = np.zeros((B, 1393)); x0[:, : 7 * T] = np.repeat(qcs, T, axis=0).reshape(B, 7 * T)
    kuka, solver = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6, "hessian": "hybrid"})
    r = solver.solve_batch_arrays(x0[:, : solver.opt.nx], qcs)
    r = solver.solve_batch_arrays(x0[:, : solver.opt.nx], qcs)
    tm = solver.backend.timing()
    print("vel T=100 B", B, "options", os.environ.get("OH_DEBUG_OPTIONS", "default"), "status", np.bincount(r.status, minlength=3), "device ms", round(tm["solve_ms"], 2), "launched", tm["iterations_launched"], "compactions", tm["compactions"], "f sum", repr(float(r.f.sum())))
else:
    for B in (8192, 65536):
        for c in (None, "0"):
            env = dict(os.environ)
            if c is not None: env["OH_DEBUG_OPTIONS"] = "compaction=" + c
            subprocess.run([sys.executable, __file__, "vel", str(B)], env=env)
    for B in (1024, 4096, 16384):
        for c in (None, "0"):
            env = dict(os.environ)
            if c is not None: env["OH_DEBUG_OPTIONS"] = "compaction=" + c
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_cfg4_trace.py"), str(B)], env=env, capture_output=True, text=True).stdout.strip().splitlines()
            print("config 4 compaction", c or "default", out[-1])
