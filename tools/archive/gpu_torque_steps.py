"""Development probe: objective after k evaluations, GPU vs numpy port (torque family)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd
from optas_amd.backend import TorqueBackend
from oracle.problems import TorqueMPCNLP
from oracle.robot import OracleRobot
from oracle.torque import TorqueProblem, solve_torque_lm
link = "lbr_link_ee"
robot = optas_amd.RobotModel.builtin("med7")
orc = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
prob = TorqueProblem(orc, link, T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4)
nlp = TorqueMPCNLP(prob)
qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
goal = prob.goal_figure_eight(qc)
p = nlp.pack_p(qc, np.zeros(7), goal)
for k in (1, 2, 3, 5, 8, 12, 14, 15, 16, 20):
    be = TorqueBackend(robot.kinematic_chain(link), robot.dynamics_tables(), T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=(None if len(sys.argv) > 1 else prob.tau_lo), tau_up=(None if len(sys.argv) > 1 else prob.tau_up), max_iter=k)
    res = be.solve(nlp.seed(qc), p)
    r = solve_torque_lm(prob, qc, np.zeros(7), goal, max_iter=k)
    print(k, "gpu", res.f[0], res.iters[0], res.status[0], "port", r["f"], r["iters"], r["status"], "rej", r["rejected"], "outers", r["outers"])
    be.close()
