"""Development probe: handles with inequality rows under wider perturbations than the tests use -- convergence fraction, cap hitters, step tail.
(round 3: after the end-game changes of retract_tol / lm_accept)"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = int(os.environ.get("STRESS_B", "16384"))


def run(tag, amp, seed, **kw):
    rng = np.random.default_rng(seed)
    qcs = QC0[None] + rng.uniform(-amp, amp, (B, 7))
    kuka, solver = setup_solver(solver_options={"max_iter": 600, "tol": 1e-6}, **kw)
    x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    p = qcs
    if "obstacles" in kw:
        n_o = len(kw["obstacles"]); n_l = len(kw["sphere_links"])
        obs = np.concatenate([[0.45, 0.1 * i, 0.25 + 0.1 * i, 0.08] for i in range(n_o)])
        p = np.concatenate([qcs, np.full((B, n_l), 0.08), np.tile(obs, (B, 1))], 1)
    r = solver.solve_batch_arrays(x0, p)
    tm = solver.backend.timing()
    ok = r.status == 0
    print(f"{tag} +-{amp}: status {np.bincount(r.status, minlength=4)} conv {ok.mean():.5f} iters p50 {np.median(r.iters):.0f} p99 {np.percentile(r.iters, 99):.0f} max {r.iters.max()} "
          f"stat max {r.kkt[ok, 0].max():.2e} feas max {r.kkt[ok, 1].max():.2e} device {tm['solve_ms']:.1f} ms launched {tm['iterations_launched']}", flush=True)
    solver.backend.close()


for amp, seed in ((0.1, 11), (0.2, 12), (0.3, 13)):
    run("velocity limits (model)", amp, seed, velocity_limits=True)
vl = np.full(7, 1.2)
run("velocity limits 1.2 rad/s", 0.1, 21, velocity_limits=(-vl, vl))
run("joint limits", 0.2, 31, limits=True)
run("joint + velocity limits", 0.2, 41, limits=True, velocity_limits=True)
from examples.dual_arm import SPHERE_LINKS
run("limits + spheres", 0.1, 51, limits=True, obstacles=["obs0", "obs1"], sphere_links=SPHERE_LINKS)
