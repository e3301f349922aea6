"""Development probe: one velocity-limited instance of the seed-5 batch solved alone with growing iteration caps (the state after k steps)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
b = int(sys.argv[1]); caps = [int(a) for a in sys.argv[2:]] or list(range(20, 80, 2))
rng = np.random.default_rng(5)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (16384, 7))
os.environ["OH_COMPACTION"] = "0"
for cap in caps:
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": cap, "tol": 1e-6})
    x0 = np.zeros((1, solver.opt.nx)); x0[:, :350] = np.tile(qcs[b], 50)
    r = solver.solve_batch_arrays(x0, qcs[b:b+1])
    tm = solver.backend.timing()
    print(cap, "status", r.status[0], "iters", r.iters[0], "f %.12f" % r.f[0], "kkt", r.kkt[0], "rejects", tm.get("rejected_steps"), flush=True)
    solver.backend.close()
