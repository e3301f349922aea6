"""Development probe: wide perturbations of the figure-eight workload (qc0 + U(-a, a)^7) through the current kernels: convergence, KKT, iteration tail."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("OPTAS_HIP_CACHE", os.path.join(ROOT, ".optas_hip_cache"))
import optas_amd, bench
from optas_amd.backend import FigureEightBackend
dt, lp = bench.local_path()
be = FigureEightBackend(optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK), 50, dt, lp, max_iter=300, tol=1e-6)
B = int(os.environ.get("STRESS_B", "65536"))
qc0 = np.deg2rad(bench.QC0_DEG)
for a in (0.1, 0.2, 0.3, 0.4):
    rng = np.random.default_rng(int(1000 * a))
    qc = qc0 + rng.uniform(-a, a, (B, 7))
    x0 = np.zeros((B, 693)); x0[:, :350] = np.tile(qc, (1, 50))
    r = be.solve(x0, qc)
    ok = r.status == 0
    print(f"+-{a}: status {np.bincount(r.status, minlength=3)} iters p50 {np.median(r.iters):.0f} p99 {np.percentile(r.iters, 99):.0f} max {r.iters.max()} "
          f"stationarity max {r.kkt[ok, 0].max():.2e} feasibility max {r.kkt[ok, 1].max():.2e} f range {r.f[ok].min():.3f}..{r.f[ok].max():.3f} finite {np.isfinite(r.x).all()}")
