"""GPU probe of the torque-MPC family against the numpy port (development tool; the parity tests live in tests/test_gpu_torque.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd  # noqa: E402
from optas_amd.backend import TorqueBackend  # noqa: E402
from oracle.problems import TorqueMPCNLP  # noqa: E402
from oracle.robot import OracleRobot  # noqa: E402
from oracle.solvers import kkt_reference_form  # noqa: E402
from oracle.torque import TorqueProblem, solve_torque_lm  # noqa: E402

link = "lbr_link_ee"
robot = optas_amd.RobotModel.builtin("med7")
orc = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
lim = float(sys.argv[1]) if len(sys.argv) > 1 else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prob = TorqueProblem(orc, link, T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=lim)
nlp = TorqueMPCNLP(prob)
be = TorqueBackend(robot.kinematic_chain(link), robot.dynamics_tables(), T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=prob.tau_lo,
                   tau_up=prob.tau_up)
rng = np.random.default_rng(20260927)
qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])[None] + rng.uniform(-0.1, 0.1, (B, 7))
qc[0] = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
goal = np.stack([prob.goal_figure_eight(q) for q in qc])
p = np.stack([nlp.pack_p(qc[b], np.zeros(7), goal[b]) for b in range(B)])
x0 = np.stack([nlp.seed(q) for q in qc])
t0 = time.time()
res = be.solve(x0, p)
print("gpu wall", time.time() - t0, be.timing())
print("status", np.bincount(res.status, minlength=3), "iters", res.iters[:8], "mean", res.iters.mean(), "max", res.iters.max())
print("f", res.f[:4], "kkt", res.kkt[:4])
for b in range(min(B, 3)):
    r = solve_torque_lm(prob, qc[b], np.zeros(7), goal[b])
    xs = nlp.join(r["Q"], r["dQ"], r["U"], r["tau"])
    print(b, "port f", r["f"], "iters", r["iters"], "gpu f", res.f[b], "df", res.f[b] - r["f"], "dx", np.abs(res.x[b] - xs).max())
    k = kkt_reference_form(nlp, res.x[b], p[b])
    print("   literal: f", nlp.f(res.x[b], p[b]), "|a|", np.abs(nlp.a(res.x[b], p[b])).max(), "|h|", np.abs(nlp.h(res.x[b], p[b])).max(), "min k",
          nlp.k(res.x[b], p[b]).min(), {a: c for a, c in k.items() if a in ("stationarity", "feasibility", "complementarity")})
