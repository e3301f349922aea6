"""A/B of the dual-damping sweep (option dual_sweep, k_step_zc<N, true>): device ms interleaved on one box, and the answers bit for bit against the single-pass sweep."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dt, lp = bench.local_path()
chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
x0, qc = bench.make_inputs(B, 0)
out = {"batch": B}
from optas_amd import _lib
try:
    out["kernel_info"] = {n: _lib.kernel_info(n) for n in ("k_step_zc", "k_step_zc_dual")}
except Exception as e:  # noqa: BLE001
    out["kernel_info"] = str(e)
print(out, flush=True)
for base_name, base in (("default", {}), ("one_stream", {"streams": 1}), ("batch_invariant", {"batch_invariant": 1})):
    bes = {k: FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2).set_options({**base, "dual_sweep": k}) for k in (0, 1, 2)}
    ms = {0: [], 1: [], 2: []}
    res = {}
    for rep in range(4):
        for k in (0, 1, 2):
            res[k] = bes[k].solve(x0, qc)
            if rep:
                ms[k].append(bes[k].timing()["solve_ms"])
    same = [bool(np.array_equal(res[0].x, res[k].x) and np.array_equal(res[0].f, res[k].f) and np.array_equal(res[0].iters, res[k].iters) and np.array_equal(res[0].status, res[k].status)) for k in (1, 2)]
    out[base_name] = {"single_ms": ms[0], "dual_hinted_ms": ms[1], "dual_after_failure_ms": ms[2], "bit_identical": same}
    print(base_name, out[base_name], flush=True)
    for k in (0, 1, 2):
        bes[k].close()
json.dump(out, open("gpurun_out/dual_sweep.json", "w"), indent=1)
