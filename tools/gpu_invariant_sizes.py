"""What `batch_invariant` costs at every batch size (device ms of the solve, default schedule beside it), one box."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel

dt, lp = bench.local_path()
chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
x0a, qca = bench.make_inputs(262144, 0)
out = {}
for B in (1, 64, 1024, 4096, 16384, 65536, 131072, 262144):
    x0, qc = x0a[:B], qca[:B]
    row = {}
    for name, opts in (("default", {}), ("batch_invariant", {"batch_invariant": 1})):
        be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2).set_options(opts)
        ms = []
        for _ in range(4):
            r = be.solve(x0, qc)
            ms.append(be.timing()["solve_ms"])
        t = be.timing()
        be.close()
        row[name] = {"ms": float(np.median(ms[1:])), "launches": t["iterations_launched"], "compactions": t["compactions"], "converged": float((r.status == 0).mean())}
    row["cost"] = row["batch_invariant"]["ms"] / row["default"]["ms"]
    out[str(B)] = row
    print(B, row, flush=True)
json.dump(out, open("gpurun_out/invariant_sizes.json", "w"), indent=1)
