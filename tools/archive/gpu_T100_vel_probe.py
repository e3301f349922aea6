import os, sys
import numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
T=100; B=20000
rng = np.random.default_rng(T * 7 + B)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
kuka, solver = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
x0 = np.zeros((B, solver.opt.nx)); x0[:, : 7 * T] = np.repeat(qcs, T, axis=0).reshape(B, 7 * T)
r = solver.solve_batch_arrays(x0, qcs)
tm = solver.backend.timing()
top=np.argsort(-r.iters)[:6]
print(os.environ.get("OH_LG_SPLIT"), "status", np.bincount(r.status,minlength=3), "iters top", r.iters[top], "idx", top, "stat", r.kkt[top,0], "ms", tm["solve_ms"], "launched", tm["iterations_launched"])
b = int(top[0])
for tag, env in (("alone", {}), ("alone, no compaction", {"OH_COMPACTION": "0"})):
    os.environ.update(env)
    k2, s2 = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True, solver_options={"max_iter": 1500, "tol": 1e-6})
    r1 = s2.solve_batch_arrays(x0[b:b + 1], qcs[b:b + 1])
    print(tag, "status", r1.status, "iters", r1.iters, "stat", r1.kkt[:, 0], "f", r1.f, "f in batch", r.f[b])
np.savez(os.path.join(ROOT, "gpurun_out", "t100_vel_hard.npz"), qc=qcs[b], x=r.x[b], f=r.f[b])
