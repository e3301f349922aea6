"""Development probe: the instances of the B = 8192 torque batch that take the most steps (their qc goes to gpurun_out/ for a verbose run of the numpy port)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd
from optas_amd.backend import TorqueBackend
link = "lbr_link_ee"
robot = optas_amd.RobotModel.builtin("med7")
T, B = 30, int(os.environ.get("TQ_B", "8192"))
lim = float(os.environ.get("TQ_LIM", "58"))
rng = np.random.default_rng(20260927)
qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
qc = qn[None] + rng.uniform(-0.1, 0.1, (B, 7))
pose, _ = robot._kin(link).fk_jac(qc, want_jac=False)
x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1), np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
               np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
ts = np.arange(T) * 0.1
loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
p = np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1)
x0 = np.zeros((B, 840)); x0[:, :210] = np.tile(qc, (1, T))
be = TorqueBackend(robot.kinematic_chain(link), robot.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-lim, tau_up=lim, max_iter=1000)
res = be.solve(x0, p)
it = res.iters
order = np.argsort(-it)[:8]
print("iters hist", np.percentile(it, [50, 90, 99, 99.9]), it.max())
tau = np.abs(res.x[:, 630:]).max(1)
print("stragglers", order, it[order], "max|tau|", tau[order], "f", res.f[order])
med = np.argsort(np.abs(it - np.median(it)))[:2]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "tq_stragglers.npz"), qc=qc[order], goal=goal[order], iters=it[order], f=res.f[order], qc_med=qc[med], goal_med=goal[med], iters_med=it[med])
