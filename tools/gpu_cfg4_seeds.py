"""Config 4 synthetic over several seeds: the twisted-factorisation sweep (k_step_free_bb) against the cyclic-reduction kernels of rounds 2-3 (option free_bb = 0):
convergence, step counts, optimum.  python tools/gpu_cfg4_seeds.py [B] [seeds]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    from optas_amd import _lib
    from optas_amd.backend import FigureEightBackend
    from optas_amd.models import RobotModel
    from examples.dual_arm import SPHERE_LINKS, path_offsets
    B, seed = int(sys.argv[2]), int(sys.argv[3])
    T = 100
    rng = np.random.default_rng(seed)
    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
    arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
    arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
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
    obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
    p = np.ascontiguousarray(np.concatenate([qc, np.full((B, 4), 0.15), np.tile(obs_row, (B, 1))], 1))
    x0 = np.ascontiguousarray(np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1))
    r = be.solve(x0, p); r = be.solve(x0, p)
    np.save(os.path.join(ROOT, "gpurun_out", f"cfg4_f_{os.environ.get('OH_DEBUG_OPTIONS', 'free_bb=1')[-1]}_{seed}.npy"), np.stack([r.f, r.iters.astype(float), r.status.astype(float)]))
    print(json.dumps({"bb": os.environ.get("OH_DEBUG_OPTIONS", "free_bb=1")[-1], "seed": seed, "ms": be.timing()["solve_ms"], "conv": float((r.status == 0).mean()), "p50": float(np.median(r.iters)), "max": int(r.iters.max())}))
else:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        for bb in ("1", "0"):
            out = subprocess.run([sys.executable, __file__, "one", str(B), str(100 + seed)], env=dict(os.environ, OH_DEBUG_OPTIONS='free_bb=' + bb), capture_output=True, text=True)
            print((out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
        a = np.load(os.path.join(ROOT, "gpurun_out", f"cfg4_f_1_{100 + seed}.npy")); c = np.load(os.path.join(ROOT, "gpurun_out", f"cfg4_f_0_{100 + seed}.npy"))
        rel = np.abs(a[0] - c[0]) / np.maximum(1e-3, np.abs(c[0]))
        print("   same optimum (1e-8 rel):", int((rel <= 1e-8).sum()), "of", B, "| other:", int((rel > 1e-5).sum()), "| same step count:", int((a[1] == c[1]).sum()), flush=True)
