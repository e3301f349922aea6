"""Development probe: the figure-eight problem on the other built-in 7-DoF arm (med7: different joint frames and tool frame than the KUKA LWR)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
for vel in (None, True):
    for B in (1, 4096, 40000):
        kuka, solver = setup_solver(robot_name="med7", link_ee="lbr_link_ee", velocity_limits=vel, solver_options={"max_iter": 600, "tol": 1e-6})
        rng = np.random.default_rng(B)
        qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
        x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
        r = solver.solve_batch_arrays(x0, qcs)
        ok = r.status == 0
        print(f"med7 vel={vel} B={B}: converged {ok.mean():.5f} iters p50 {np.median(r.iters):.0f} max {r.iters.max()} f median {np.median(r.f):.4f} kkt max {r.kkt[ok].max(0)}", flush=True)
        if B == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from conftest import MED7_KIN
            from oracle.robot import OracleRobot
            from oracle.structured import StructuredFigureEight, solve_structured_lm
            orc = OracleRobot(MED7_KIN)
            prob = StructuredFigureEight(orc, "lbr_link_ee", T=50)
            vl = np.asarray(orc.velocity_actuated_joint_limits)
            s = solve_structured_lm(prob, qcs[0], max_iter=600, tol=1e-6, vlimits=(-vl, vl) if vel else None)
            print("   numpy port:", s["status"], s["iters"], s["f"], "GPU f", r.f[0], "iters", r.iters[0], "rel diff", abs(s["f"] - r.f[0]) / s["f"])
        solver.backend.close()
