"""Development probe: config 4 synthetic (T = 100, limits + spheres, radius 0.15) for one batch size, for rocprofv3 --kernel-trace."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
from examples.dual_arm import SPHERE_LINKS, path_offsets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
radius = 0.15
rng = np.random.default_rng(20260927)
QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
T = int(os.environ.get("CFG4_T", "100"))
offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
g = _lib.oh_guards()
g.limits = 1
for j in range(7):
    g.q_lo[j], g.q_up[j] = arm.lower_actuated_joint_limits[j], arm.upper_actuated_joint_limits[j]
g.n_links, g.n_obstacles = 4, 6
if os.environ.get("RHO0"): g.rho0 = float(os.environ["RHO0"])
for l, (k, off) in enumerate(arm.link_attachments("end_effector_ball", SPHERE_LINKS)):
    g.link_joint[l] = k
    for i in range(3):
        g.link_offset[l][i] = off[i]
be = FigureEightBackend(arm.kinematic_chain("end_effector_ball"), T, 10.0 / (T - 1), offs.T, w_path=1.0, w_vel=0.01, max_iter=400, lock_orientation=False, fix_dq0=False,
                        path_in_frame=False, guards=g)
qc = QC + rng.uniform(-0.1, 0.1, (B, 7))
obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
p = np.ascontiguousarray(np.concatenate([qc, np.full((B, 4), radius), np.tile(obs_row, (B, 1))], 1))
x0 = np.ascontiguousarray(np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1))
for rep in range(3):
    r = be.solve(x0, p)
    tm = be.timing()
    print("B", B, "device ms", tm["solve_ms"], "launched", tm["iterations_launched"], "compactions", tm["compactions"], "iters p50/max", np.median(r.iters), r.iters.max(), "conv", (r.status == 0).mean())
