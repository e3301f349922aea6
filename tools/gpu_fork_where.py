"""Which part of the default schedule sends the four fork instances of tools/gpu_fork_rate.py elsewhere?  The 262 144 batch with one ingredient removed at a time."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel

B = 262144
x0, qc = bench.make_inputs(B, 0)
dt, lp = bench.local_path()
chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
forks = [63286, 100029, 168410, 244951]
out = {}
ref = None
for name, opts in (("invariant", {"batch_invariant": 1}), ("default", {}), ("one_stream", {"streams": 1}), ("no_carry", {"compact_carry": 0}), ("no_tail", {"tail_threshold": 0}),
                   ("no_sort", {"compact_sort": 0}), ("no_compaction", {"compaction": 0}), ("no_compaction_no_tail", {"compaction": 0, "tail_threshold": 0})):
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2).set_options(opts)
    r = be.solve(x0, qc)
    ms = be.timing()["solve_ms"]
    be.close()
    if ref is None:
        ref = r.f.copy()
    d = np.abs(r.f - ref) > 1e-9 * np.abs(ref)
    out[name] = {"ms": ms, "different_optimum": int(d.sum()), "instances": np.nonzero(d)[0][:12].tolist(), "forks_f": [float(r.f[k]) for k in forks], "forks_iters": [int(r.iters[k]) for k in forks],
                 "forks_kkt0": [float(r.kkt[k, 0]) for k in forks]}
    print(name, out[name], flush=True)
json.dump(out, open("gpurun_out/fork_where.json", "w"), indent=1)
