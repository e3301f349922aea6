"""Development probe: the B = 8192 property-test batch of the torque family at several effort limits / iteration caps."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd
from optas_amd.backend import TorqueBackend
link = "lbr_link_ee"
robot = optas_amd.RobotModel.builtin("med7")
T, B = 30, int(os.environ.get("TQ_B", "8192"))
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
CASES = [tuple(float(v) for v in c.split(":")) for c in os.environ.get("TQ_CASES", "100:300,58:300,58:1000,55:1000").split(",")]
for lim, mi in CASES:
    mi = int(mi)
    be = TorqueBackend(robot.kinematic_chain(link), robot.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-lim, tau_up=lim, max_iter=mi)
    res = be.solve(x0, p)
    res = be.solve(x0, p)  # second pass: buffers and code objects warm
    tm = be.timing()
    it = res.iters
    tau = np.abs(res.x[:, 630:]).max(1)
    print(f"lim {lim} max_iter {mi}: status {np.bincount(res.status, minlength=3)} iters p50 {np.median(it)} p90 {np.percentile(it,90)} p99 {np.percentile(it,99)} max {it.max()} "
          f"active {(tau > lim - 1e-6).mean():.3f} kkt max {res.kkt[res.status==0].max(0)} ms {tm['solve_ms']:.1f} launched {tm['iterations_launched']}")
    be.close()
