"""Round 5 (last session): what `batch_invariant` costs now that its compaction moves everything (`invariant_compact_frac`).  One box, B instances of the
headline workload: the default schedule, batch_invariant without compaction (the option as it shipped first), and with the moving compaction at several
fractions, with and without the two-stream split; checks that every batch_invariant variant returns the same bits (x, f, iterations, status) and that
256 sampled instances equal themselves solved alone on another handle."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402


def run(options, x0, qc, reps=3):
    dt, lp = bench.local_path()
    chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2).set_options(options)
    ms = []
    for _ in range(reps):
        r = be.solve(x0, qc)
        t = be.timing()
        ms.append(t["solve_ms"])
    be.close()
    return r, min(ms), t


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    x0, qc = bench.make_inputs(B, 0)
    out = {"batch": B}
    r0, ms, t = run({}, x0, qc)
    out["default"] = {"ms": ms, "launches": t.get("iterations_launched"), "compactions": t.get("compactions")}
    ref = None
    for name, opts in (
        ("invariant_no_compaction", {"batch_invariant": 1, "invariant_compact_frac": 0, "invariant_split": 0}),
        ("move_all_0.5", {"batch_invariant": 1, "invariant_move_slim": 0}),
        ("move_slim_0.5_one_stream", {"batch_invariant": 1, "invariant_split": 0}),
        ("move_slim_0.5", {"batch_invariant": 1}),
        ("move_slim_0.6", {"batch_invariant": 1, "invariant_compact_frac": 0.6}),
        ("move_slim_0.7", {"batch_invariant": 1, "invariant_compact_frac": 0.7}),
        ("move_live_0.5", {"batch_invariant": 1, "invariant_move_live": 1}),
        ("move_live_0.6", {"batch_invariant": 1, "invariant_move_live": 1, "invariant_compact_frac": 0.6}),
        ("move_live_0.7", {"batch_invariant": 1, "invariant_move_live": 1, "invariant_compact_frac": 0.7}),
        ("move_live_0.8", {"batch_invariant": 1, "invariant_move_live": 1, "invariant_compact_frac": 0.8}),
    ):
        r, ms, t = run(opts, x0, qc, reps=2)
        e = {"ms": ms, "launches": t.get("iterations_launched"), "compactions": t.get("compactions"), "converged": float((r.status == 0).mean())}
        if ref is None:
            ref = r
        else:
            e["bit_identical_to_no_compaction"] = bool(
                np.array_equal(r.x, ref.x) and np.array_equal(r.f, ref.f) and np.array_equal(r.iters, ref.iters) and np.array_equal(r.status, ref.status)
            )
        out[name] = e
        print(name, e, flush=True)
    idx = np.sort(np.random.default_rng(11).choice(B, 256, replace=False))
    ra, ms, _ = run({"batch_invariant": 1}, x0[idx], qc[idx], reps=1)
    out["sample_alone_vs_in_batch_bit_identical"] = int(sum(np.array_equal(ra.x[i], ref.x[b]) and ra.iters[i] == ref.iters[b] for i, b in enumerate(idx)))
    out["default_vs_invariant_same_f_1e-9"] = float((np.abs(r0.f - ref.f) <= 1e-9 * np.abs(ref.f)).mean())
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/invariant_compaction.json", "w"), indent=1)


if __name__ == "__main__":
    main()
