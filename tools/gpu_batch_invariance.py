"""Batch invariance of handles with inequality rows outside the persistent kernels (verdict r03 item 3): 20 000 velocity-limited T = 100 figure-eight
instances solved with the default compaction (round 5: every array moves with the instance), with the restart compaction of round 4, and without any compaction (= every instance as if alone: without compaction an instance's
arithmetic does not depend on the batch, tests/test_gpu_velocity_limits.py).  python tools/gpu_batch_invariance.py [B]"""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
T = 100; B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(T * 7 + 20000)
qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
x0 = np.zeros((B, 1393)); x0[:, : 7 * T] = np.repeat(qcs, T, axis=0).reshape(B, 7 * T)
res = {}
for tag, env in (("compaction", None), ("none", "compaction=0"), ("restart", "compact_move_all=0")):
    os.environ.pop("OH_DEBUG_OPTIONS", None)
    if env is not None: os.environ["OH_DEBUG_OPTIONS"] = env
    kuka, solver = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6, "hessian": "hybrid"})
    r = solver.solve_batch_arrays(x0[:, : solver.opt.nx], qcs)
    tm = solver.backend.timing()
    res[tag] = (np.array(r.status), np.array(r.iters), np.array(r.f), np.array(r.x), tm["solve_ms"], tm["compactions"])
    solver.backend.close()
a, n = res["compaction"], res["none"]
rel = np.abs(a[2] - n[2]) / np.maximum(1.0, np.abs(n[2]))
out = {"B": B, "converged": [int((a[0] == 0).sum()), int((n[0] == 0).sum())], "device_ms": [a[4], n[4]], "compactions": a[5],
       "same_iterates_bitwise": int((np.abs(a[3] - n[3]).max(1) == 0).sum()), "same_step_count": int((a[1] == n[1]).sum()),
       "same_optimum_1e-9": int((rel <= 1e-9).sum()), "other_optimum": int((rel > 1e-6).sum()), "max_rel_f_diff": float(rel.max()),
       "iters_max": [int(a[1].max()), int(n[1].max())]}
r_ = res["restart"]
rel_r = np.abs(r_[2] - n[2]) / np.maximum(1.0, np.abs(n[2]))
out["restart_compaction_round4"] = {"device_ms": r_[4], "compactions": r_[5], "same_iterates_bitwise": int((np.abs(r_[3] - n[3]).max(1) == 0).sum()),
                                    "same_optimum_1e-9": int((rel_r <= 1e-9).sum()), "converged": int((r_[0] == 0).sum())}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "batch_invariance.json"), "w"), indent=1)
