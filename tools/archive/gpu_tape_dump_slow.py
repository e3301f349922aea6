import os, sys, numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from examples.example import setup_solver as ik_setup
from optas_amd.backend import TapeBackend
from optas_amd.models import RobotModel
from optas_amd.tape import compile_problem
kuka = RobotModel.builtin("kuka_lwr")
tp = compile_problem(ik_setup(build_only=True)[1])
rng = np.random.default_rng(20260927)
B = 65536
qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits).T)).T
p = np.ascontiguousarray(np.concatenate([qn, pg], 1))
be = TapeBackend(tp, max_iter=2000)
r = be.solve(np.ascontiguousarray(qn), p)
bad = np.nonzero(r.status != 0)[0]
slow = np.argsort(-r.iters)[:40]
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/tape_fail.npz', bad=bad, x0=qn[slow], p=p[slow], iters=r.iters[slow], status=r.status[slow], kkt=r.kkt[slow], x=r.x[slow], f=r.f[slow], idx=slow)
print(len(bad), r.iters[slow][:40], r.kkt[slow][:5])
