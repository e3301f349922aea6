"""GPU: the joint-space planner on the generic tape family with and without the cost metric (oh_tape_set_metric): evaluations and device time for the 4 golden
instances and for 256 / 4096 perturbed ones; penalty and pair sweeps on the metric handle.  Run through gpurun; prints one JSON object."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from examples.simple_joint_space_planner import setup_solver  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "planner_golden.npz"))
    rng = np.random.default_rng(2024 + 6)
    out = {}

    def instances(B):
        idx = np.arange(B) % 4
        P = g["p"][idx].copy()
        if B > 4:
            P[:, :14] += rng.uniform(-0.05, 0.05, (B, 14))
            P[:, 14:17] += rng.uniform(-0.02, 0.02, (B, 3))
        x0 = np.zeros((B, 280))
        x0[:, :140] = np.tile(g["q0"].reshape(-1, 1), (1, 20)).reshape(-1)[None, :]
        return np.ascontiguousarray(x0), np.ascontiguousarray(P)

    sets = {B: instances(B) for B in (4, 256, 4096)}
    cfgs = [("metric_default", {}), ("no_metric", {"metric": False})]
    for rho0 in (3e3, 3e4, 1e5):
        cfgs.append((f"metric_rho{rho0:g}", {"rho0": rho0}))
    for tag, opts in cfgs:
        _, solver = setup_solver(solver_options={"max_iter": 400000, **opts})
        be = solver.backend
        row = {}
        for B, (x0, P) in sets.items():
            if B == 4096 and tag not in ("metric_default", "no_metric"):
                continue
            be.solve(x0, P)
            ms = []
            for _ in range(3):
                r = be.solve(x0, P)
                ms.append(float(be.solve_ms()))
            it = np.asarray(r.iters)
            row[str(B)] = {"device_ms": float(np.median(ms)), "evals_p50": float(np.median(it)), "evals_max": int(it.max()), "evals_mean": float(it.mean()),
                           "converged": float((np.asarray(r.status) == 0).mean()), "f_sum": float(np.sum(r.f))}
            if B == 4:
                row[str(B)]["evals"] = it.tolist()
                row[str(B)]["f_rel_golden"] = [float(abs(a - b) / b) for a, b in zip(r.f, g["f"])]
        row["flags"] = {k: be.flag(k) for k in ("tape_wave", "tape_metric", "tape_regs_lds")}
        out[tag] = row
        be.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
