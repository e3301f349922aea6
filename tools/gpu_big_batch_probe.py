"""Development probe: which instances of a very large batch go wrong on the fused-coupling path."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench, optas_amd
from optas_amd.backend import FigureEightBackend
B = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
dt, lp = bench.local_path()
chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
x0, qc = bench.make_inputs(B, 0)
for mi in (1, 2, 300):
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=mi, tol=1e-6, hessian=2)
    be.max_batch = None
    r = be.solve(x0, qc)
    ok = r.status == 0
    edges = np.linspace(0, B, 13).astype(int)
    print("max_iter", mi, "converged", ok.mean(), "per twelfth:", [round(float(ok[a:b].mean()), 3) for a, b in zip(edges[:-1], edges[1:])],
          "f finite", np.isfinite(r.f).mean(), "f median per twelfth", [round(float(np.median(r.f[a:b])), 2) for a, b in zip(edges[:-1], edges[1:])], flush=True)
    be.close()
