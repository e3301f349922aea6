"""Barrier / step-rule constants of the torque family's interior point on three batches of 8192 perturbed instances (config 5): iteration
histogram, launches and device time per setting (options of the handle; one box, back to back).
python tools/gpu_tq_param_sweep.py ["{'tq_kappa_eps': 30}" ...]  -> gpurun_out/tq_param_sweep.json"""
import json
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
sets = []
for seed in range(3):
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
    sets.append((x0, p))
settings = [eval(a) for a in sys.argv[1:]] or [{}]
out = []
base_f = None
for kw in settings:
    kw = dict(kw)
    ctor = {k: kw.pop(k) for k in list(kw) if not k.startswith("tq_") and k not in ("streams",)}
    be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, **{'max_iter': 600, **ctor})
    for k, v in kw.items():
        be.set_option(k, v)
    its, ms, launched, ok, fs = [], [], [], [], []
    for x0, p in sets:
        be.solve(x0, p)
        r = be.solve(x0, p)
        its.append(np.asarray(r.iters)); ms.append(be.timing()["solve_ms"]); launched.append(be.timing()["iterations_launched"])
        ok.append(float(np.isin(np.asarray(r.status), (0, 4)).mean())); fs.append(np.asarray(r.f))
    it = np.concatenate(its); f = np.concatenate(fs)
    if base_f is None:
        base_f = f
    rel = np.abs(f - base_f) / np.abs(base_f)
    row = {"setting": {**ctor, **kw}, "device_ms": [round(m, 2) for m in ms], "launched": launched, "ok": min(ok), "mean": float(it.mean()), "p50": float(np.median(it)),
           "p90": float(np.percentile(it, 90)), "p99": float(np.percentile(it, 99)), "max": int(it.max()), "n>40/50/60/80": [int((it > k).sum()) for k in (40, 50, 60, 80)],
           "f_rel_diff>1e-6": int((rel > 1e-6).sum()), "f_worse>1e-6": int(((f - base_f) / np.abs(base_f) > 1e-6).sum())}
    out.append(row)
    print(json.dumps(row), flush=True)
    be.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tq_param_sweep.json"), "w"), indent=1)
