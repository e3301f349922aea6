"""Development probe: the slowest instances of config 5's batch (8192 torque-MPC problems) with their inputs -> gpurun_out/tq_slow.npz."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
SEED = 20260927
rng = np.random.default_rng(SEED + 5)
med7 = RobotModel.builtin("med7")
link, T, dt = "lbr_link_ee", 30, 0.1
qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
ts = np.arange(T) * dt
loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
B = 8192
qc = qn + rng.uniform(-0.1, 0.1, (B, 7))
pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
               np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
               np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
x0 = np.zeros((B, 28 * T))
x0[:, : 7 * T] = np.tile(qc, (1, T))
be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
r = be.solve(x0, p)
order = np.argsort(-r.iters)[:24]
print("iters sorted top", r.iters[order], "p50", np.median(r.iters), "p90", np.percentile(r.iters, 90), "p99", np.percentile(r.iters, 99), "p99.9", np.percentile(r.iters, 99.9))
print("hist >100:", (r.iters > 100).sum(), ">150:", (r.iters > 150).sum(), ">200:", (r.iters > 200).sum(), ">300:", (r.iters > 300).sum())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "tq_slow.npz"), idx=order, iters=r.iters[order], qc=qc[order], goal=goal[order], p=p[order], f=r.f[order], x=r.x[order], all_iters=r.iters)
