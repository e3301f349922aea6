"""Closed loop of the torque family (oh_tq_rollout, 8192 plants x 20 warm-started ticks) under settings of the interior point's constants.
python tools/gpu_tq_rollout_sweep.py "{'tq_mu_dec': 0.1}" ...  (options of the handle; 'mu_warm' is the rollout's own argument)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

med7 = RobotModel.builtin("med7")
link, T, dt, B, n_ticks = "lbr_link_ee", 30, 0.1, 8192, 20
qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
rng = np.random.default_rng(20260927)
qc = qn + rng.uniform(-0.1, 0.1, (B, 7))
pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
               np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
               np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
ts2 = np.arange(n_ticks + T) * dt
loc2 = np.stack([0.2 * np.sin(ts2 * np.pi * 0.5), 0.1 * np.sin(ts2 * np.pi), np.zeros(n_ticks + T)])
table = np.ascontiguousarray(pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc2))
st0 = np.concatenate([qc, np.zeros((B, 7))], 1)
f0 = None
for a in sys.argv[1:] or ["{}"]:
    kw = dict(eval(a))
    mu_warm = kw.pop("mu_warm", 1e-6)
    be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
    for k, v in kw.items():
        be.set_option(k, v)
    be.rollout(st0[:64], np.ascontiguousarray(table[:64, : 2 + T]), 2)
    states, tau0, fr, itr, stt = be.rollout(st0, table, n_ticks, mu_warm=mu_warm)
    ms = be.timing()["solve_ms"]
    if f0 is None:
        f0 = fr
    print(json.dumps({"setting": eval(a), "device_ms": round(ms, 1), "ticks_per_s": round(B * n_ticks / ms * 1e3), "ok": float(_lib.status_ok(stt).mean()), "cold_p50": float(np.median(itr[0])),
                      "warm_mean": float(itr[1:].mean()), "warm_p50": float(np.median(itr[1:])), "warm_p90": float(np.percentile(itr[1:], 90)), "warm_max": int(itr[1:].max()),
                      "f_rel_max_vs_first": float(np.max(np.abs(fr - f0) / np.abs(f0)))}), flush=True)
    be.close()
