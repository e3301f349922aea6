"""Rates of the dense-QP family: the velocity-IK tick of examples/differential_ik.py through HIPSolver (B = 1, wall clock,
includes reading P, q, M, c off the problem) and the kernel alone on a resident batch of such QPs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd  # noqa: E402
from examples.differential_ik import DifferentialIK  # noqa: E402
from optas_amd import _lib  # noqa: E402

ik = DifferentialIK(height_band=(0.0, 2.0))
q = optas_amd.deg2rad([0, 30, 0, -90, 0, 60, 0])
for _ in range(5):
    dq, q = ik.step(q)
n_ticks = 200
t0 = time.perf_counter()
for _ in range(n_ticks):
    dq, q = ik.step(q)
tick_ms = (time.perf_counter() - t0) / n_ticks * 1e3
o = ik.optimization
host = DifferentialIK(height_band=(0.0, 2.0), solver_options={"device_assembly": False})
qh = optas_amd.deg2rad([0, 30, 0, -90, 0, 60, 0])
for _ in range(5):
    _, qh = host.step(qh)
t0 = time.perf_counter()
for _ in range(50):
    _, qh = host.step(qh)
tick_host_ms = (time.perf_counter() - t0) / 50 * 1e3
rng = np.random.default_rng(20260927)
B = 65536
qs = optas_amd.deg2rad([0, 30, 0, -90, 0, 60, 0])[None] + rng.uniform(-0.2, 0.2, (B, 7))
be = ik.solver.backend.be  # QP handle with the problem's tape attached: p = qc
x0 = np.zeros((B, o.nx))
bufs = [_lib.DeviceBuffer(a.nbytes) for a in (x0, qs)]
bufs[0].upload(x0)
bufs[1].upload(qs)
d_x, d_f, d_k, d_i, d_s = _lib.DeviceBuffer(x0.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B), _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)
ms = []
for _ in range(4):
    be.solve_device(B, bufs[0], bufs[1], d_x, d_f, d_k, d_i, d_s)
    ms.append(be.solve_ms())
st, it = d_s.download(np.int32, (B,)), d_i.download(np.int32, (B,))
print(json.dumps({"config": "(f)3 velocity-IK QP (examples/differential_ik.py: 7 variables, 16 inequality rows; P, q, M, c read off the problem's tape on the device)",
                  "tape_instructions": int(len(be.tape.op)), "tick_wall_ms_b1": tick_ms, "ticks_per_s_b1": 1e3 / tick_ms,
                  "tick_wall_ms_b1_host_assembly": tick_host_ms, "batch": B, "device_ms": float(np.median(ms[1:])),
                  "qp_solves_per_s": B / float(np.median(ms[1:])) * 1e3, "converged_frac": float((st == 0).mean()), "iters_p50": float(np.median(it)),
                  "iters_max": int(it.max())}))
