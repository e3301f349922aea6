"""Development probe: config 1's IK through the generic tape family over tolerances (convergence fraction, evaluation counts)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.example import setup_solver as ik_setup
from examples.planar_ik import setup_solver as planar_setup
from optas_amd.backend import TapeBackend
from optas_amd.models import RobotModel
from optas_amd.tape import compile_problem
kuka = RobotModel.builtin("kuka_lwr")
tp = compile_problem(ik_setup(build_only=True)[1])
rng = np.random.default_rng(7)
B = 16384
qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits).T)).T
p = np.ascontiguousarray(np.concatenate([qn, pg], 1))
for tol, tf in ((1e-6, 1e-9), (1e-7, 1e-10), (1e-8, 1e-11), (1e-9, 1e-12)):
    be = TapeBackend(tp, max_iter=4000, tol=tol, tol_feas=tf)
    r = be.solve(np.ascontiguousarray(qn), p)
    ok = r.status == 0
    print(f"IK tol {tol:g} feas {tf:g}: converged {ok.mean():.5f} evals p50 {np.median(r.iters):.0f} p99 {np.percentile(r.iters, 99):.0f} max {r.iters.max()} stat max(all) {r.kkt[:, 0].max():.2e} feas max(all) {r.kkt[:, 1].max():.2e}", flush=True)
    be.close()
# unreachable goals: the rows cannot be met, the solver must come back (status MAX_ITER) without hanging
pg_far = pg + np.array([2.0, 0.0, 0.0])
be = TapeBackend(tp, max_iter=1500)
r = be.solve(np.ascontiguousarray(qn[:1024]), np.ascontiguousarray(np.concatenate([qn[:1024], pg_far[:1024]], 1)))
print("unreachable goals: status", np.bincount(r.status, minlength=3), "evals max", r.iters.max(), "feas min", r.kkt[:, 1].min())
