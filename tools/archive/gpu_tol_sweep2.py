"""Development probe: tolerances below the defaults on the other families (position tracking with guards, torque MPC, IK)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend, IKBackend, TorqueBackend
from optas_amd.models import RobotModel
from examples.dual_arm import SPHERE_LINKS, path_offsets
rng0 = np.random.default_rng(11)


def rep(tag, r):
    ok = r.status == 0
    print(f"{tag}: converged {ok.mean():.5f} iters p50 {np.median(r.iters):.0f} p99 {np.percentile(r.iters, 99):.0f} max {r.iters.max()} "
          f"not-converged stat median {np.median(r.kkt[~ok, 0]) if (~ok).any() else 0:.2e} feas {np.median(r.kkt[~ok, 1]) if (~ok).any() else 0:.2e}", flush=True)


QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
T, B = 100, 4096
offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
for tol in (1e-6, 1e-8, 1e-10):
    g = _lib.oh_guards(); g.limits = 1
    for j in range(7):
        g.q_lo[j], g.q_up[j] = arm.lower_actuated_joint_limits[j], arm.upper_actuated_joint_limits[j]
    g.n_links, g.n_obstacles = 4, 6
    for l, (k, off) in enumerate(arm.link_attachments("end_effector_ball", SPHERE_LINKS)):
        g.link_joint[l] = k
        for i in range(3):
            g.link_offset[l][i] = off[i]
    be = FigureEightBackend(arm.kinematic_chain("end_effector_ball"), T, 10.0 / (T - 1), offs.T, w_path=1.0, w_vel=0.01, max_iter=600, tol=tol, lock_orientation=False, fix_dq0=False,
                            path_in_frame=False, guards=g)
    rng = np.random.default_rng(4)
    qc = QC + rng.uniform(-0.1, 0.1, (B, 7))
    obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
    p = np.ascontiguousarray(np.concatenate([qc, np.full((B, 4), 0.15), np.tile(obs_row, (B, 1))], 1))
    x0 = np.ascontiguousarray(np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1))
    rep(f"config 4 synthetic tol={tol:g}", be.solve(x0, p)); be.close()
med7 = RobotModel.builtin("med7")
T, B = 30, 2048
qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
for tol in (1e-6, 1e-8, 1e-10):
    rng = np.random.default_rng(5)
    qc = qn + rng.uniform(-0.1, 0.1, (B, 7))
    pose, _ = med7._kin("lbr_link_ee").fk_jac(qc, want_jac=False)
    ts = np.arange(T) * 0.1
    goal = pose[:, None, :3] + np.stack([0.1 * np.sin(ts * np.pi * 0.5), 0.05 * np.sin(ts * np.pi), np.zeros(T)], 1)[None]
    p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
    x0 = np.zeros((B, 4 * 7 * T)); x0[:, : 7 * T] = np.tile(qc, (1, T))
    be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=1000, tol=tol)
    rep(f"torque tol={tol:g}", be.solve(x0, p)); be.close()
kuka = RobotModel.builtin("kuka_lwr")
lo, up = kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits
B = 65536
for tol in (1e-6, 1e-8, 1e-10):
    rng = np.random.default_rng(6)
    q0 = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
    pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(q0 + rng.uniform(-0.5, 0.5, (B, 7)), lo, up).T)).T
    try:
        be = IKBackend(kuka.kinematic_chain("end_effector_ball"), lo, up, max_iter=300, tol=tol)
    except TypeError:
        be = IKBackend(kuka.kinematic_chain("end_effector_ball"), lo, up, max_iter=300)
    rep(f"IK tol={tol:g}", be.solve(np.ascontiguousarray(q0), np.ascontiguousarray(np.concatenate([q0, pg], 1)))); be.close()
