"""Development probe: the tracking controller of examples/torque_control_example.py through the generic tape family (what a real optas problem of
this class meets when it comes in through casadi_tape.py) against the exact minimiser of the literal problem."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import optas_amd
from examples.torque_control_example import TrackingController
from optas_amd.backend import TapeBackend
from optas_amd.tape import compile_problem
from oracle.problems import TorqueControlNLP, band_qp_exact
from oracle.robot import OracleRobot
ctrl = TrackingController(1.0 / 500.0, build_only=True)
nlp = TorqueControlNLP(OracleRobot(os.path.join(os.path.dirname(optas_amd.__file__), "robots", "med7.kin.json")))
rng = np.random.default_rng(3)
q0 = np.deg2rad([0, 30, 0, -90, 0, 60, 0])
P = []
for _ in range(64):
    qc = q0 + rng.uniform(-0.3, 0.3, 7)
    pc = np.asarray(nlp.robot.get_global_link_position(nlp.link, qc)).reshape(3)
    P.append(np.concatenate([qc, pc + rng.uniform(-0.003, 0.003, 3), [0.0, 1.0, 0.0, 0.0]]))
P = np.array(P)
for tag, kw in (("default", {}), ("rho0=1e6", {"rho0": 1e6}), ("rho0=1e9", {"rho0": 1e9})):
    be = TapeBackend(compile_problem(ctrl.optimization), max_iter=20000, **kw)
    r = be.solve(np.zeros((64, 7)), P)
    z = np.zeros(7)
    worst = 0.0
    for i in range(64):
        A, b, _, _ = nlp.pieces(P[i])
        xs = band_qp_exact(nlp.ddf(z, P[i]), nlp.df(z, P[i]), A, b, np.sqrt(nlp.bounds))[0]
        worst = max(worst, np.abs(r.x[i] - xs).max() / max(1.0, np.abs(xs).max()))
    print(tag, "status", np.bincount(r.status, minlength=3), "evals p50", np.median(r.iters), "max", r.iters.max(), "x rel diff to exact max", worst, "feas", r.kkt[:, 1].max())
    be.close()
