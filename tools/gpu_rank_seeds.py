"""The headline batch of every rank of an 8-GPU run (bench.make_inputs(B, rank), rank = 0 .. 7) on one GPU: convergence, step counts, and a sample against the host port."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402
from oracle import cpu_port  # noqa: E402

dt, lp = bench.local_path()
chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
B = 262144
be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-8, hessian=2).set_options(pipe=0)
out = []
for rank in range(8):
    x0, qc = bench.make_inputs(B, rank)
    r = be.solve(x0, qc)
    ms = be.timing()["solve_ms"]
    idx = np.sort(np.random.default_rng(rank).choice(B, 8192, replace=False))
    _, fp, _, _, stp = cpu_port.solve(chain, bench.T, dt, lp, x0[idx], qc[idx], tol=1e-8, threads=bench.usable_cores())
    rel = np.abs(r.f[idx] - fp) / np.abs(fp)
    out.append({"rank": rank, "converged_frac": float((r.status == 0).mean()), "iters_p50": float(np.median(r.iters)), "iters_max": int(r.iters.max()), "device_ms": ms,
                "solves_per_s": B / ms * 1e3, "sample": 8192, "misses_1e-9_vs_host_port": int((rel > 1e-9).sum()), "max_rel": float(rel.max())})
    print(json.dumps(out[-1]), flush=True)
json.dump(out, open("gpurun_out/r06_rank_seeds.json", "w"), indent=1)
