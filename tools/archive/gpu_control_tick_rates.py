"""Rates of the tracking controller of examples/torque_control_example.py (the reference's example/torque_control_example.py, lowered as a banded
QP): one tick through HIPSolver (B = 1, wall clock, P, q, M, c read off the problem's tape on the device) and a resident batch of (qc, pg) pairs;
a 64-instance sample of the batch is compared with the exact active-set minimiser of the literal problem (oracle/problems.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd  # noqa: E402
from examples.torque_control_example import TrackingController  # noqa: E402
from optas_amd import _lib  # noqa: E402
from oracle.problems import TorqueControlNLP, band_qp_exact  # noqa: E402
from oracle.robot import OracleRobot  # noqa: E402

dt = 1.0 / 500.0
ctrl = TrackingController(dt)
nlp = TorqueControlNLP(OracleRobot(os.path.join(os.path.dirname(optas_amd.__file__), "robots", "med7.kin.json")))
q0 = optas_amd.deg2rad([0, 30, 0, -90, 0, 60, 0])
start = np.asarray(ctrl.kuka.get_global_link_position("lbr_link_ee", q0)).reshape(3)
q = q0.copy()
down = np.array([0.0, 1.0, 0.0, 0.0])


def tick(k, q):
    goal = start + np.array([0.0, 0.0004 * (k + 1), 0.0])
    return q + dt * ctrl.compute_target_velocity(q, np.concatenate([goal, down]))


for k in range(5):
    q = tick(k, q)
n_ticks = 200
t0 = time.perf_counter()
for k in range(5, 5 + n_ticks):
    q = tick(k, q)
tick_ms = (time.perf_counter() - t0) / n_ticks * 1e3
err = np.asarray(ctrl.kuka.get_global_link_position("lbr_link_ee", q)).reshape(3) - (start + np.array([0.0, 0.0004 * (5 + n_ticks), 0.0]))

rng = np.random.default_rng(20260928)
B = 65536
qs = q0[None] + rng.uniform(-0.3, 0.3, (B, 7))
pcs = np.asarray(ctrl.kuka.get_global_link_position("lbr_link_ee", qs.T)).reshape(3, B).T
P = np.concatenate([qs, pcs + rng.uniform(-0.003, 0.003, (B, 3)), np.tile(down, (B, 1))], axis=1)
be = ctrl.solver.backend.be
x0 = np.zeros((B, 7))
bufs = [_lib.DeviceBuffer(a.nbytes) for a in (x0, P)]
bufs[0].upload(x0)
bufs[1].upload(P)
d_x, d_f, d_k, d_i, d_s = _lib.DeviceBuffer(x0.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B), _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)
ms = []
for _ in range(4):
    be.solve_device(B, bufs[0], bufs[1], d_x, d_f, d_k, d_i, d_s)
    ms.append(be.solve_ms())
st, it, X = d_s.download(np.int32, (B,)), d_i.download(np.int32, (B,)), d_x.download(np.float64, (B, 7))
worst, worst_f, worst_tight, active = 0.0, 0.0, 0.0, 0
z = np.zeros(7)
sample = rng.choice(B, 64, replace=False)
tight = TrackingController(dt, solver_options={"tol": 1e-12}).solver.solve_batch_arrays(np.zeros((64, 7)), P[sample]).x.reshape(64, 7)
for k, i in enumerate(sample):
    A, b, _, _ = nlp.pieces(P[i])
    xs, _, state, _ = band_qp_exact(nlp.ddf(z, P[i]), nlp.df(z, P[i]), A, b, np.sqrt(nlp.bounds))
    worst = max(worst, float(np.abs(X[i] - xs).max() / max(1.0, np.abs(xs).max())))
    worst_tight = max(worst_tight, float(np.abs(tight[k] - xs).max() / max(1.0, np.abs(xs).max())))
    worst_f = max(worst_f, abs(nlp.f(X[i], P[i]) - nlp.f(xs, P[i])) / abs(nlp.f(xs, P[i])))
    active += sum(1 for s in state if s)
print(json.dumps({"config": "example/torque_control_example.py tracking controller (7 variables, 3 squared-error rows -> 6 band rows; dense-QP family, data read off the tape on the device)",
                  "tape_instructions": int(len(be.tape.op)), "tick_wall_ms_b1": tick_ms, "ticks_per_s_b1": 1e3 / tick_ms,
                  "closed_loop_position_error_after_205_ticks": float(np.abs(err).max()), "batch": B, "device_ms": float(np.median(ms[1:])),
                  "solves_per_s": B / float(np.median(ms[1:])) * 1e3, "converged_frac": float((st == 0).mean()), "iters_p50": float(np.median(it)),
                  "iters_max": int(it.max()), "oracle_sample": {"instances": 64, "x_max_rel_diff_to_exact_minimiser": worst, "x_max_rel_diff_at_tol_1e-12": worst_tight,
                                                               "f_max_rel_diff": worst_f, "active_band_rows": active,
                                                               "by": "oracle/problems.py:band_qp_exact on TorqueControlNLP (active-set enumeration)"}}))
