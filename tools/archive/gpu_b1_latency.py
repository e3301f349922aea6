"""Wall-clock latency of ONE instance through every problem family (backend.solve: host buffers in, results out), median of 30."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import optas_amd  # noqa: E402
from optas_amd.backend import FigureEightBackend, IKBackend, PointMassBackend, TapeBackend, TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402


def med(fn, reps=30):
    for _ in range(3):
        r = fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t) * 1e3), r


out = {}
kuka = RobotModel.builtin("kuka_lwr")
# config 1: IK
be = IKBackend(kuka.kinematic_chain("end_effector_ball"), kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits, max_iter=300)
qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
pg = np.asarray(kuka.get_global_link_position("end_effector_ball", qn + 0.2)).reshape(-1)
ms, r = med(lambda: be.solve(qn[None], np.concatenate([qn, pg])[None]))
out["1 IK (hand-written kernel)"] = {"ms": ms, "iters": int(r.iters[0]), "status": int(r.status[0])}
be.close()
# config 1 through the tape family
from examples.example import setup_solver as ik_setup  # noqa: E402
from optas_amd.tape import compile_problem  # noqa: E402

tb = TapeBackend(compile_problem(ik_setup(build_only=True)[1]), max_iter=2000)
ms, r = med(lambda: tb.solve(qn[None], np.concatenate([qn, pg])[None]))
out["1 IK (tape family, JIT)"] = {"ms": ms, "iters": int(r.iters[0]), "status": int(r.status[0])}
tb.close()
# config 2
dt, lp = bench.local_path()
fe = FigureEightBackend(kuka.kinematic_chain("end_effector_ball"), 50, dt, lp, max_iter=300, tol=1e-6)
x0, qc = bench.make_inputs(1, 0)
ms, r = med(lambda: fe.solve(x0, qc))
out["2 figure-eight T=50"] = {"ms": ms, "iters": int(r.iters[0]), "status": int(r.status[0])}
fe.close()
# config 3
pm = PointMassBackend()
T = 20
rng = np.random.default_rng(1)
p = np.concatenate([[0.0, 0.0, 0.0, 0.0], np.tile([1.0, 0.5], T), np.tile([0.5, 0.3], T)])[None]
try:
    ms, r = med(lambda: pm.solve(np.zeros((1, pm.nx)), p[:, : pm.np_] if p.shape[1] >= pm.np_ else np.zeros((1, pm.np_))))
    out["3 point-mass tick T=20"] = {"ms": ms, "iters": int(r.iters[0]), "status": int(r.status[0])}
except Exception as e:  # the probe's parameters are synthetic: report, do not stop
    out["3 point-mass tick T=20"] = {"error": str(e)[:80]}
pm.close()
# config 5
med7 = RobotModel.builtin("med7")
tq = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=1000)
qq = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
pose, _ = med7._kin("lbr_link_ee").fk_jac(qq[None], want_jac=False)
ts = np.arange(30) * 0.1
goal = pose[0, :3][None] + np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(30)], 1)
pt = np.concatenate([qq, np.zeros(7), goal.reshape(-1)])[None]
xt = np.zeros((1, 840))
xt[:, :210] = np.tile(qq, 30)
ms, r = med(lambda: tq.solve(xt, pt), reps=10)
out["5 torque MPC T=30"] = {"ms": ms, "iters": int(r.iters[0]), "status": int(r.status[0])}
tq.close()
print(json.dumps({"config": "latency of one instance per family (wall clock, host buffers)", **out}))
