"""Timing of the torque family only (config 5): python tools/gpu_tq_time.py [B ...] -> one JSON line per batch size (device ms, launches, iteration
percentiles).  Used with OPTAS_HIP_LIBRARY to compare tuning variants of the library (tools/gpu_tq_waves.sh)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_tq_ipm_probe import instances  # noqa: E402
from optas_amd.backend import TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

med7 = RobotModel.builtin("med7")
T = 30
for B in [int(a) for a in sys.argv[1:]] or [8192, 1024, 1]:
    qc, goal, x0, p = instances(med7, B)
    be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
    be.solve(x0, p)
    ms = []
    for _ in range(3):
        r = be.solve(x0, p)
        tm = be.timing()
        ms.append(tm["solve_ms"])
    it, st = np.asarray(r.iters), np.asarray(r.status)
    print(json.dumps({"lib": os.environ.get("OPTAS_HIP_LIBRARY", "default"), "B": B, "device_ms": ms, "launched": tm["iterations_launched"], "converged": float((st == 0).mean()),
                      "it_p50": float(np.median(it)), "it_max": int(it.max()), "f_sum": float(np.sum(r.f))}), flush=True)
    be.close()
