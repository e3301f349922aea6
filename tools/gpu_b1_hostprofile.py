"""Where does the host time of ONE Solver.solve() go (config 2, B = 1)?  cProfile of 2000 calls + wall / device medians."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from examples.figure_eight_plan import setup_solver as fig8

robot, solver = fig8()
name = robot.get_name()
qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
seed = {f"{name}/q/x": np.tile(qc.reshape(-1, 1), (1, 50))}


def run():
    solver.reset_parameters({"qc": qc})
    solver.reset_initial_seed(seed)
    return solver.solve()


for _ in range(50):
    run()
ts = []
for _ in range(400):
    t0 = time.perf_counter()
    run()
    ts.append(time.perf_counter() - t0)
be = solver.backend
while not hasattr(be, "timing") and hasattr(be, "be"):
    be = be.be
print("wall ms p50", 1e3 * np.median(ts), "device ms", be.timing()["solve_ms"])
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
