"""Torque family with hard-pressed velocity rows (|dq| <= 0.2): device against numpy port, iteration cap by iteration cap -- where do the two part?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import MED7_KIN, SEED  # noqa: E402
from optas_amd.backend import TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402
from oracle.robot import OracleRobot  # noqa: E402
from oracle.torque import TorqueProblem  # noqa: E402
from oracle.torque_ipm import solve_torque_ipm  # noqa: E402

W = dict(w_path=1000.0, w_vel=0.1, w_tau=1e-4)
QC = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
T, vl = 30, float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
robot = RobotModel.builtin("med7")
LIM = float(sys.argv[4]) if len(sys.argv) > 4 else 58.0
prob = TorqueProblem(OracleRobot(MED7_KIN), "lbr_link_ee", T=T, dt=0.1, tau_lim=LIM, **W)
rng = np.random.default_rng(SEED + 9)
qc = QC[None] + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.05, 0.05, (5, 7))])
b = int(sys.argv[1]) if len(sys.argv) > 1 else 0
goal = prob.goal_figure_eight(qc[b])
p = np.ascontiguousarray(np.concatenate([qc[b], np.zeros(7), goal.reshape(-1)])[None])
x0 = np.zeros((1, 4 * 7 * T))
x0[0, : 7 * T] = np.tile(qc[b], T)
for k in range(1, int(sys.argv[3]) if len(sys.argv) > 3 else 60):
    be = TorqueBackend(robot.kinematic_chain("lbr_link_ee"), robot.dynamics_tables(), T=T, dt=0.1, tau_lo=-LIM, tau_up=LIM, dq_lo=-vl, dq_up=vl, max_iter=k, **W)
    r = be.solve(x0, p)
    be.close()
    s = solve_torque_ipm(prob, qc[b], np.zeros(7), goal, vlimits=(-vl, vl), max_iter=k)
    d = abs(s["f"] - r.f[0]) / abs(s["f"])
    print(k, "gpu f %.12f it %d st %d | port f %.12f it %d st %d mu_b %.2e | rel %.1e" % (r.f[0], r.iters[0], r.status[0], s["f"], s["iters"], s["status"], s["mu_b"], d), flush=True)
    if r.status[0] != 1 and s["status"] != 1:
        break
