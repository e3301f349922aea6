"""Development probe: every kernel family at batch sizes well above what the tests use -- convergence only (32-bit offsets, grid limits, pools)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend, IKBackend, PointMassBackend, TorqueBackend
from optas_amd.models import RobotModel
rng = np.random.default_rng(7)
kuka = RobotModel.builtin("kuka_lwr")


def report(tag, r, t0, be=None):
    ok = r.status == 0
    print(f"{tag}: converged {ok.mean():.6f} iters p50 {np.median(r.iters):.0f} max {r.iters.max()} finite {np.isfinite(r.x).all()} wall {time.time() - t0:.1f} s", flush=True)


# IK
B = 1 << 20
lo, up = kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits
qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), lo, up).T)).T
be = IKBackend(kuka.chain if hasattr(kuka, "chain") else kuka.kinematic_chain("end_effector_ball"), lo, up, max_iter=300)
t0 = time.time(); report(f"IK B={B}", be.solve(np.ascontiguousarray(qn), np.ascontiguousarray(np.concatenate([qn, pg], 1))), t0); be.close()
# position tracking (dual_arm.py per arm), unguarded T = 50 and guarded T = 100
from examples.dual_arm import SPHERE_LINKS, path_offsets
QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
for T, B, guarded in ((50, 262144, False), (100, 65536, True)):
    offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
    g = None
    if guarded:
        g = _lib.oh_guards(); g.limits = 1
        for j in range(7):
            g.q_lo[j], g.q_up[j] = arm.lower_actuated_joint_limits[j], arm.upper_actuated_joint_limits[j]
        g.n_links, g.n_obstacles = 4, 6
        for l, (k, off) in enumerate(arm.link_attachments("end_effector_ball", SPHERE_LINKS)):
            g.link_joint[l] = k
            for i in range(3):
                g.link_offset[l][i] = off[i]
    be = FigureEightBackend(arm.kinematic_chain("end_effector_ball"), T, 10.0 / (T - 1), offs.T, w_path=1.0, w_vel=0.01, max_iter=400, lock_orientation=False, fix_dq0=False,
                            path_in_frame=False, guards=g)
    qc = QC + rng.uniform(-0.1, 0.1, (B, 7))
    p = qc
    if guarded:
        obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
        p = np.ascontiguousarray(np.concatenate([qc, np.full((B, 4), 0.1), np.tile(obs_row, (B, 1))], 1))
    x0 = np.ascontiguousarray(np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1))
    t0 = time.time(); report(f"position tracking T={T} guarded={guarded} B={B}", be.solve(x0, p), t0); be.close()
# torque
med7 = RobotModel.builtin("med7")
T, B = 30, 32768
qnn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
qc = qnn + rng.uniform(-0.1, 0.1, (B, 7))
pose, _ = med7._kin("lbr_link_ee").fk_jac(qc, want_jac=False)
ts = np.arange(T) * 0.1
goal = pose[:, None, :3] + np.stack([0.1 * np.sin(ts * np.pi * 0.5), 0.05 * np.sin(ts * np.pi), np.zeros(T)], 1)[None]
p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
x0 = np.zeros((B, 4 * 7 * T)); x0[:, : 7 * T] = np.tile(qc, (1, T))
be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-100.0, tau_up=100.0, max_iter=600)
t0 = time.time(); report(f"torque B={B}", be.solve(x0, p), t0); be.close()
