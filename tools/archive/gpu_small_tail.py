"""A/B probe: config 4 synthetic (256 / 1024 arms) and the velocity-limited figure-eight (1024) with the host looking at the running count every
iteration (OH_SPARSE_CHECK_BELOW=0) or every 8th iteration once at most 2048 instances run."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    out = bench_configs.run_configs(sample=0, only="config4")
    from examples.figure_eight_plan import setup_solver
    QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    B = 1024
    qcs = QC0[None] + np.random.default_rng(5).uniform(-0.1, 0.1, (B, 7))
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    solver.solve_batch_arrays(x0, qcs); r = solver.solve_batch_arrays(x0, qcs)
    print(json.dumps({**{k: round(v["device_ms"], 2) for k, v in out.items()}, "vel_limited_1024_ms": round(solver.backend.timing()["solve_ms"], 2), "vel_iters_max": int(r.iters.max())}))
else:
    for rep in range(2):
        for mode in ("0", "2048"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, OH_SPARSE_CHECK_BELOW=mode), capture_output=True, text=True)
            print("OH_SPARSE_CHECK_BELOW=" + mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:])
