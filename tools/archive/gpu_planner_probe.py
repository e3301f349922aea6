"""point_mass_planner.py variant of the point-mass family: GPU kernel vs numpy port on the script's instance and a few random ones."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optas_amd.backend import PointMassBackend
from oracle.pointmass_ipm import solve_pointmass_ipm
T, dt = 45, 0.1
wv, wa = 0.01 / T, 0.005 / T
be = PointMassBackend(T=T, dt=dt, w_acc=wa, ylim=1.5, vlim=1.0, safe=0.3, max_iter=200, tol=1e-9, track_final_only=True, w_vel=wv, fix_final_velocity=True)
rng = np.random.default_rng(0)
inits = np.array([[-1.0, -1.0], [-1.2, -0.4], [0.9, -1.1], [-0.5, 1.2]])
goals = np.array([[1.0, 1.0], [1.0, 0.7], [-1.0, 1.0], [0.8, -1.0]])
P = []
for i, g in zip(inits, goals):
    P.append(np.concatenate([i, np.zeros(2), np.tile(g, T), np.zeros(2 * T)]))
r = be.solve(np.zeros((4, 4 * T)), np.array(P))
print("gpu", r.status, r.iters, r.f, r.kkt[:, 0].max(), r.kkt[:, 1].max())
for b in range(4):
    s = solve_pointmass_ipm(T, dt, wa, 1.5, 1.0, 0.09, inits[b], np.zeros(2), np.tile(goals[b][:, None], (1, T)), np.zeros((2, T)), tol=1e-9, max_iter=200,
                            track_final_only=True, w_vel=wv, fix_final_velocity=True)
    Y = r.x[b, : 2 * T].reshape(T, 2).T
    print(b, "port", s["status"], s["iters"], s["f"], "df", r.f[b] - s["f"], "dY", np.abs(Y - s["Y"]).max(), "vT", r.x[b, 2 * T + 2 * (T - 1):2 * T + 2 * T])
