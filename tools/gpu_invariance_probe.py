"""Round 5: is an instance's path a function of the instance alone?  The same instances through (a) the persistent kernel from the start (a batch below
the tail threshold), (b) the batched launches alone (no tail, no compaction), (c) the default path of a large batch (batched launches, carried
compactions, hand-over to the persistent kernel).  Prints how many of the sampled instances agree bit for bit / in objective, and the device times."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402


def run(options, x0, qc):
    dt, lp = bench.local_path()
    chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2).set_options(options)
    r = be.solve(x0, qc)
    r = be.solve(x0, qc)
    ms = be.timing()["solve_ms"]
    be.close()
    return r, ms


def main():
    nbig = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    x0, qc = bench.make_inputs(nbig, 0)
    idx = np.sort(np.random.default_rng(7).choice(nbig, 256, replace=False))
    out = {}
    ra, ms = run({}, x0[idx], qc[idx])
    out["tail_256_ms"] = ms
    rb, ms = run({"batch_invariant": 1}, x0[idx], qc[idx])
    out["batched_256_ms"] = ms
    out["tail_vs_batched_bit_identical"] = int(sum(np.array_equal(ra.x[i], rb.x[i]) for i in range(256)))
    out["tail_vs_batched_same_f_1e-9"] = int((np.abs(ra.f - rb.f) <= 1e-9 * np.abs(ra.f)).sum())
    out["tail_vs_batched_iters_equal"] = int((ra.iters == rb.iters).sum())
    rd, ms = run({"batch_invariant": 1}, x0, qc)
    out["big_invariant_ms"] = ms
    out["big_invariant_vs_alone_invariant_bit_identical"] = int(sum(np.array_equal(rd.x[b], rb.x[i]) for i, b in enumerate(idx)))
    rc, ms = run({}, x0, qc)
    out["big_batch"] = nbig
    out["big_ms"] = ms
    out["big_vs_alone_bit_identical"] = int(sum(np.array_equal(rc.x[b], ra.x[i]) for i, b in enumerate(idx)))
    out["big_vs_alone_same_f_1e-9"] = int((np.abs(rc.f[idx] - ra.f) <= 1e-9 * np.abs(ra.f)).sum())
    out["big_vs_alone_iters_equal"] = int((rc.iters[idx] == ra.iters).sum())
    for B in (1, 1024, 16384):
        _, ms = run({}, x0[:B], qc[:B])
        out[f"tail_{B}_ms"] = ms
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/invariance_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
