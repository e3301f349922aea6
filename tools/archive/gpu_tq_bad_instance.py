import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from gpu_tq_ipm_probe import instances
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
med7 = RobotModel.builtin("med7"); T = 30
qc, goal, x0, p = instances(med7, 8192)
be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
r = be.solve(x0, p)
st = np.asarray(r.status); it = np.asarray(r.iters); kkt = np.asarray(r.kkt)
bad = np.flatnonzero(st != 0)
print("bad", bad, st[bad], it[bad], kkt[bad], r.f[bad])
top = np.argsort(-it)[:5]
print("top iters", top, it[top], st[top])
for b in bad:
    r1 = be.solve(x0[b:b+1], p[b:b+1])
    print("alone", b, r1.status, r1.iters, r1.f, np.asarray(r1.kkt))
    np.save("/root/repo/gpurun_out/tq_bad_qc.npy", qc[b])
