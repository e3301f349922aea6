"""Development probe: the small-launch kernels of round 2 on hostile inputs (NaN parameters, infeasible rows): they must return a status, not hang."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd.backend import PointMassBackend, QPBackend
# QP: infeasible rows x >= 1 and x <= 0, NaN in P, indefinite P; wave kernel (B <= 64) and thread kernel (B = 100)
for B in (3, 100):
    n, m = 2, 2
    rows = []
    for k in range(B):
        P = np.eye(2); q = np.zeros(2); M = np.array([[1.0, 0.0], [-1.0, 0.0]]); c = np.array([-1.0, 0.0])
        if k % 3 == 1: P = np.array([[np.nan, 0.0], [0.0, 1.0]])
        if k % 3 == 2: P = -np.eye(2); M = np.zeros((2, 2)); c = np.ones(2)
        rows.append(QPBackend.pack(P, q, M, c, np.zeros((0, 2)), np.zeros(0)))
    be = QPBackend(n, m, 0)
    r = be.solve(np.zeros((B, n)), np.stack(rows))
    print("QP B=%d status" % B, np.bincount(r.status, minlength=3), "iters", r.iters[:3])
    be.close()
# point mass: NaN start, start inside the obstacle, goal outside the box
for B, mode in ((4, "wave"), (4, "thread")):
    os.environ["OH_PM_WAVE_MAX"] = "0" if mode == "thread" else "100000"
    T = 20
    obs = np.array([[0.15 * np.sin(np.pi * (0.05 * t) - np.pi), 0.15 * np.cos(np.pi * (0.05 * t) - np.pi) + 0.15] for t in range(T)])
    P = np.zeros((B, 4 + 4 * T))
    starts = np.array([[np.nan, 0.0], [obs[0, 0], obs[0, 1]], [-1.0, 0.5], [1.4, 1.4]])
    for k in range(B):
        P[k, :2] = starts[k]
        P[k, 4 : 4 + 2 * T] = np.tile([5.0, 5.0] if k == 3 else [1.0, 1.0], T)
        P[k, 4 + 2 * T :] = obs.reshape(-1)
    be = PointMassBackend(tol=1e-8)
    r = be.solve(np.zeros((B, 4 * T)), P)
    print("point mass", mode, "status", r.status, "iters", r.iters, "finite x", np.isfinite(r.x).all(1))
    be.close()
