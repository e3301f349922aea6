import sys, os; sys.path.insert(0,".")
import numpy as np, bench, optas_amd
from optas_amd.backend import FigureEightBackend
dt, lp = bench.local_path()
chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
B = 20000
rng = np.random.default_rng(9)
qc = np.deg2rad(bench.QC0_DEG)[None, :] + rng.uniform(-0.4, 0.4, (B, 7))   # harder: more rejections, more polish / stale paths
x0 = np.concatenate([np.repeat(qc, bench.T, axis=0).reshape(B, 7 * bench.T), np.zeros((B, 7 * (bench.T - 1)))], axis=1)
res = {}
for tag, env in (("off", {"OH_COMPACTION": "0", "OH_TAIL_THRESHOLD": "0"}), ("carry", {"OH_TAIL_THRESHOLD": "64", "OH_COMPACT_FRAC": "0.97"}), ("restart", {"OH_TAIL_THRESHOLD": "64", "OH_COMPACT_CARRY": "0", "OH_COMPACT_FRAC": "0.97"})):
    for k in ("OH_COMPACTION", "OH_TAIL_THRESHOLD", "OH_COMPACT_FRAC", "OH_COMPACT_CARRY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6)
    r = be.solve(x0, qc); res[tag] = r; be.close()
    print(tag, "converged", (r.status == 0).mean(), "steps mean", r.iters.mean(), "max", r.iters.max(), "f mean", r.f[r.status == 0].mean())
ref = res["off"]
for tag in ("carry", "restart"):
    r = res[tag]
    ok = (ref.status == 0) & (r.status == 0)
    same = np.abs(r.f - ref.f) <= 1e-9 * np.abs(ref.f)
    print(tag, "status equal", (r.status == ref.status).mean(), "same f", same[ok].mean(), "max |dx| among same", np.abs(r.x[ok & same] - ref.x[ok & same]).max(),
          "median |d iters|", np.median(np.abs(r.iters - ref.iters)[ok & same]), "feas max", r.kkt[ok, 1].max(), "stat max", r.kkt[ok, 0].max())
