"""Closed-loop torque MPC (oh_tq_rollout): a warm tick against the cold solve from the same plant state, digits shown."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

med7 = RobotModel.builtin("med7")
link, T, dt = "lbr_link_ee", 30, 0.1
B, n_ticks = 256, 30
rng = np.random.default_rng(5)
qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0]) + rng.uniform(-0.1, 0.1, (B, 7))
pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1), np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
               np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
ts = np.arange(n_ticks + T) * dt
loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(n_ticks + T)])
table = np.ascontiguousarray(pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc))
be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
states, tau0, f, it, st = be.rollout(np.concatenate([qc, np.zeros((B, 7))], 1), table, n_ticks)
print("rollout ok", _lib.status_ok(st).mean(), "device ms", be.timing()["solve_ms"], "steps p50 cold", np.median(it[0]), "warm", np.median(it[1:]))
for k in (1, 10, 25):
    p = np.ascontiguousarray(np.concatenate([states[k], table[:, k : k + T].reshape(B, -1)], 1))
    x0 = np.zeros((B, 4 * 7 * T))
    x0[:, 2 * 7 * T : 2 * 7 * T + 7] = -states[k, :, 7:] / dt
    c = be.solve(x0, p)
    ok = _lib.status_ok(c.status)
    rel = np.abs(c.f - f[k]) / np.abs(c.f)
    print("tick", k, "cold ok", ok.mean(), "cold steps p50", np.median(c.iters[ok]), "warm steps p50", np.median(it[k]), "f warm[:3]", f[k, :3], "f cold[:3]", c.f[:3], "rel max", rel[ok].max(), "median", np.median(rel[ok]))
