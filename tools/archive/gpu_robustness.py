"""Convergence of the figure-eight family under larger perturbations of the start configuration than the bench uses (+-0.1 rad):
converged fraction, step counts, objective.  python tools/gpu_robustness.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import optas_amd  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402


def main():
    dt, lp = bench.local_path()
    chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6)
    B = 16384
    for amp in (0.1, 0.2, 0.3, 0.5, 0.8):
        rng = np.random.default_rng(5)
        qc = np.deg2rad(bench.QC0_DEG)[None, :] + rng.uniform(-amp, amp, (B, 7))
        x0 = np.concatenate([np.repeat(qc, bench.T, axis=0).reshape(B, 7 * bench.T), np.zeros((B, 7 * (bench.T - 1)))], axis=1)
        r = be.solve(x0, qc)
        ok = r.status == 0
        print(f"amp {amp}: converged {ok.mean():.4f} (max_iter {np.mean(r.status == 1):.4f}, numerical {np.mean(r.status == 2):.4f}) "
              f"steps mean {r.iters[ok].mean():.1f} p50 {np.median(r.iters[ok]):.0f} p99 {np.percentile(r.iters[ok], 99):.0f} max {r.iters.max()} "
              f"f mean {r.f[ok].mean():.4f} feas max {r.kkt[ok, 1].max():.2e}")


if __name__ == "__main__":
    main()
