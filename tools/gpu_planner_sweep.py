"""Planner on the eliminated tape: limited-memory pairs x initial penalty -> evaluations and device time (4 golden instances, 256 perturbed ones)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.simple_joint_space_planner import setup_solver  # noqa: E402
from optas_amd.backend import tape_backend  # noqa: E402
from optas_amd.tape import compile_problem  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "planner_golden.npz"))
_, opt = setup_solver(build_only=True)
tp = compile_problem(opt)
P, nb = g["p"], len(g["p"])
x0 = np.zeros((nb, tp.nx))
x0[:, :140] = np.tile(g["q0"].reshape(-1, 1), (1, 20)).reshape(-1)[None, :]
rng = np.random.default_rng(20260933)
B = 256
idx = np.arange(B) % nb
Pn = P[idx].copy()
Pn[:, :14] += rng.uniform(-0.05, 0.05, (B, 14))
Pn[:, 14:17] += rng.uniform(-0.02, 0.02, (B, 3))
out = []
for m in (12, 24, 32, 48, 64):
    for rho0 in (10.0, 100.0, 1000.0):
        be = tape_backend(tp, max_iter=400000, rho0=rho0, options={"tape_lbfgs": m})  # (defaults since this sweep: 32 pairs, rho0 = 1000 on an eliminated tape)
        be.solve(x0, P)
        r = be.solve(x0, P)
        ms4 = be.solve_ms()
        rb = be.solve(x0[idx], Pn)
        msb = be.solve_ms()
        row = {"pairs": m, "rho0": rho0, "golden_evals": r.iters.tolist(), "golden_ok": bool((r.status == 0).all()), "golden_ms": ms4,
               "golden_f_rel": float(np.abs(r.f - g["f"]).max() / g["f"].max()), "b256_ms": msb, "b256_ok": float((rb.status == 0).mean()), "b256_evals_p50": float(np.median(rb.iters)),
               "regs_lds": be.flag("tape_regs_lds"), "wave": be.flag("tape_wave")}
        print(json.dumps(row), flush=True)
        out.append(row)
        be.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/planner_sweep.json", "w"), indent=1)
