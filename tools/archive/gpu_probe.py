"""Diagnostic run on a GPU box (not a test): FK parity, first solves, kernel timings."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import optas_amd
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from oracle.robot import OracleRobot
from oracle.problems import FigureEightNLP
from oracle.solvers import kkt_reference_form
from oracle.structured import FoldedChain

KIN = os.path.join(os.path.dirname(optas_amd.__file__), "robots", "kuka_lwr.kin.json")
LINK = "end_effector_ball"
print("devices:", _lib.device_count())
robot = optas_amd.RobotModel.builtin("kuka_lwr")
orc = OracleRobot(KIN)
rng = np.random.default_rng(20260927)

# --- K1 parity -------------------------------------------------------------------------------
N = 2048
Q = rng.uniform(-2.9, 2.9, (N, 7))
kin = robot._kin(LINK)
pose, J = kin.fk_jac(Q)
fc = FoldedChain(orc, LINK)
e, Re, Jp, Jw = fc.jac(Q)
print("K1 pos err", np.abs(pose[:, :3] - e).max(), "J err", max(np.abs(J[:, :3] - Jp).max(), np.abs(J[:, 3:] - Jw).max()))
qerr = 0
for i in range(64):
    qerr = max(qerr, np.abs(pose[i, 3:] - orc.get_global_link_quaternion(LINK, Q[i])).max())
    Jo = orc.get_global_link_geometric_jacobian(LINK, Q[i])
    qerr = max(qerr, np.abs(J[i] - Jo).max())
print("K1 literal-oracle err (quat signed, J)", qerr)

# --- solves ------------------------------------------------------------------------------------
T = 50
nlp = FigureEightNLP(orc, LINK, T=T)
ch = robot.kinematic_chain(LINK)
for hess in (0, 1):
    be = FigureEightBackend(ch, T, nlp.dt, nlp.local_path.T, max_iter=300, tol=1e-6, hessian=hess)
    qc0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    B = 16
    qc = np.tile(qc0, (B, 1)); qc[1:] += rng.uniform(-0.1, 0.1, (B - 1, 7))
    x0 = np.stack([nlp.seed(q) for q in qc])
    t0 = time.time(); r = be.solve(x0, qc); t1 = time.time()
    print(f"hessian={hess} B={B} solve wall {t1-t0:.3f}s")
    print(" f", np.round(r.f, 9)); print(" iters", r.iters, "status", r.status); print(" stat", r.kkt[:, 0], " feas", r.kkt[:, 1].max())
    print(" f[0] - 8.498170214656 =", r.f[0] - 8.498170214656)
    print(" oracle f(x*)", nlp.f(r.x[0], qc[0]) - r.f[0], "lin eq resid", np.abs(nlp.a(r.x[0], qc[0])).max(), "quat resid", np.abs(nlp.h(r.x[0], qc[0])).max())
    k = kkt_reference_form(nlp, r.x[0], qc[0]); print(" reference-form KKT:", {a: k[a] for a in ("stationarity", "feasibility", "complementarity")})
    lam = be.multipliers(B)
    g = nlp.df(r.x[0], qc[0]); Jh = nlp.dh(r.x[0], qc[0])
    # stationarity in q-block with HIP multipliers (linear-row multipliers eliminated: check via projection on null of A)
    print(" |lam_h| max", np.abs(lam[0]).max())
    if hess == 0:
        be.set_profiling(True)
        for Bb in (1, 256, 4096, 32768):
            qcb = np.tile(qc0, (Bb, 1)) + rng.uniform(-0.1, 0.1, (Bb, 7))
            x0b = np.repeat(qcb, T, axis=0).reshape(Bb, T * 7)
            x0b = np.concatenate([x0b, np.zeros((Bb, 7 * (T - 1)))], axis=1)
            be.solve(x0b[: min(Bb, 8)], qcb[: min(Bb, 8)])
            t0 = time.time(); rb = be.solve(x0b, qcb); t1 = time.time()
            tm = be.timing()
            print(f" B={Bb}: wall {t1-t0:.3f}s device {tm['solve_ms']:.1f} ms  eval {tm['eval_ms']:.1f} ms/{tm['eval_launches']}  step {tm['step_ms']:.1f} ms/{tm['step_launches']}  "
                  f"solves/s(device) {Bb/(tm['solve_ms']*1e-3):.0f}  conv {np.mean(rb.status==0):.3f} iters p50/p90/max {np.percentile(rb.iters,50):.0f}/{np.percentile(rb.iters,90):.0f}/{rb.iters.max()}")
    be.close()
