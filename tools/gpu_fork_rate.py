"""Round 5: how often does an instance's optimum depend on the path it took?  All 262 144 instances of the headline batch through (a) the default schedule
of the big batch, (b) `batch_invariant`, (c) the persistent kernel from the start (chunks of 16 384 = what an instance meets 'alone' or in a small batch),
(d) the compiled host port (oracle/cpu_port, other arithmetic: x86 FMA contraction and libm).  Counts instances whose objectives differ by more than 1e-9
relative, pairwise, and lists them."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    x0, qc = bench.make_inputs(B, 0)
    dt, lp = bench.local_path()
    chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
    mk = lambda: FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=tol, hessian=2)
    f = {}
    be = mk()
    r = be.solve(x0, qc)
    f["default"], conv = r.f.copy(), float((r.status == 0).mean())
    be.set_option("batch_invariant", 1)
    f["invariant"] = be.solve(x0, qc).f.copy()
    be.set_option("batch_invariant", 0)
    parts = [be.solve(x0[lo : lo + 16384], qc[lo : lo + 16384]).f for lo in range(0, B, 16384)]
    f["persistent"] = np.concatenate(parts)
    be.close()
    from oracle import cpu_port

    _, fp, _, _, stp = cpu_port.solve(chain, bench.T, dt, lp, x0, qc, tol=tol, threads=bench.usable_cores())
    f["host_port"] = fp
    out = {"batch": B, "tol": tol, "converged_default": conv, "host_port_converged": float((stp == 0).mean()), "pairs": {}}
    names = list(f)
    for i, a in enumerate(names):
        for b in names[i + 1 :]:
            d = np.abs(f[a] - f[b]) > 1e-9 * np.abs(f[b])
            out["pairs"][f"{a} vs {b}"] = {"different_optimum": int(d.sum()), "frac": float(d.mean()), "instances": np.nonzero(d)[0][:16].tolist(),
                                           "f": [[float(f[a][k]), float(f[b][k])] for k in np.nonzero(d)[0][:8]]}
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/fork_rate{tag}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
