"""Hunt for slow instances of the torque family's interior point: several batches of 8192 perturbed initial configurations (config 5), the
iteration histogram of each and the initial configurations of everything above 100 (argv[2]) iterations -> gpurun_out/tq_stragglers.npz"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd.backend import TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

med7 = RobotModel.builtin("med7")
link, T, dt, B = "lbr_link_ee", 30, 0.1, 8192
qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
ts = np.arange(T) * dt
loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
slow_qc, slow_it = [], []
CUT = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    rng = np.random.default_rng(1000 + seed)
    qc = qn + rng.uniform(-0.1, 0.1, (B, 7))
    pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
    x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
    Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                   np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                   np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
    goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
    p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
    x0 = np.zeros((B, 4 * 7 * T))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    r = be.solve(x0, p)
    it = np.asarray(r.iters)
    print(seed, "ms", round(be.timing()["solve_ms"], 1), "conv", float((np.asarray(r.status) == 0).mean()), "p50", np.median(it), "p99", np.percentile(it, 99), "p99.9", np.percentile(it, 99.9),
          "max", it.max(), "n>100", int((it > 100).sum()), "n>34/40/50/60/80", [int((it > k).sum()) for k in (34, 40, 50, 60, 80)], "launched", be.timing()["iterations_launched"], flush=True)
    for b in np.flatnonzero(it > CUT):
        slow_qc.append(qc[b]); slow_it.append(it[b])
np.savez(os.path.join(ROOT, "gpurun_out", "tq_stragglers.npz"), qc=np.array(slow_qc), iters=np.array(slow_it))
