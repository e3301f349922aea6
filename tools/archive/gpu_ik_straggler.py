"""Development probe: the instance of the config-1 bench batch (65 536 IK problems) that hits the iteration cap."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd.backend import IKBackend
from optas_amd.models import RobotModel
SEED = 20260927
rng = np.random.default_rng(SEED)
kuka = RobotModel.builtin("kuka_lwr")
B = 65536
lo, up = kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits
qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), lo, up).T)).T
for mi in (300, 2000):
    be = IKBackend(kuka.kinematic_chain("end_effector_ball"), lo, up, max_iter=mi)
    r = be.solve(np.ascontiguousarray(qn), np.ascontiguousarray(np.concatenate([qn, pg], 1)))
    bad = np.flatnonzero(r.status != 0)
    print("max_iter", mi, "not converged", bad, "iters", r.iters[bad], "kkt", r.kkt[bad], "f", r.f[bad], "iters p50/p99/max", np.median(r.iters), np.percentile(r.iters, 99), r.iters.max())
    if len(bad):
        b = bad[0]
        print(" q", r.x[b], "\n lo", lo, "\n up", up, "\n qn", qn[b], "pg", pg[b], "p(q)", np.asarray(kuka.get_global_link_position("end_effector_ball", r.x[b])).reshape(-1))
    be.close()
np.savez(os.path.join(ROOT, "gpurun_out", "ik_straggler.npz"), qn=qn[bad] if len(bad) else qn[:1], pg=pg[bad] if len(bad) else pg[:1])
