import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_tq_ipm_probe import instances
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
med7 = RobotModel.builtin("med7"); T = 30
qc, goal, x0, p = instances(med7, 8192)
b = 7359
for mi in (40, 60, 70, 80, 90, 100, 120, 150, 200, 300, 400, 500, 545):
    be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=mi)
    r = be.solve(x0[b:b+1], p[b:b+1])
    lam = be.multipliers(1)
    tau = r.x[0, 3*7*T:].reshape(T, 7)
    smin = min((tau + 58).min(), (58 - tau).min())
    print(mi, "status", r.status[0], "it", r.iters[0], "f", repr(float(r.f[0])), "kkt", np.asarray(r.kkt)[0], "smin", smin, "lam max", float(np.max(lam)))
    be.close()
