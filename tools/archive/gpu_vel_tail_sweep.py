"""Development probe: hand-over threshold of the persistent kernel for velocity-limited handles."""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from examples.figure_eight_plan import setup_solver
    QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    for B in (65536, 262144):
        rng = np.random.default_rng(5)
        qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
        kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
        x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
        r = solver.solve_batch_arrays(x0, qcs); r = solver.solve_batch_arrays(x0, qcs)
        tm = solver.backend.timing()
        print(f"  B={B}: device {tm['solve_ms']:.1f} ms -> {B / tm['solve_ms'] * 1e3:.0f} solves/s conv {(r.status == 0).mean():.5f} launched {tm['iterations_launched']} compactions {tm['compactions']}", flush=True)
        solver.backend.close()
else:
    for thr in ("16384", "32768", "65536", "131072"):
        print("OH_TAIL_THRESHOLD", thr, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, OH_TAIL_THRESHOLD=thr))
