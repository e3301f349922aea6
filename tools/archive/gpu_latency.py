"""Latency/throughput of small batches (k_tail path) on a GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import optas_amd
from optas_amd.backend import FigureEightBackend
import bench
dt, lp = bench.local_path()
robot = optas_amd.RobotModel.builtin("kuka_lwr")
be = FigureEightBackend(robot.kinematic_chain("end_effector_ball"), 50, dt, lp, max_iter=300, tol=1e-6)
BATCHES = tuple(int(v) for v in os.environ.get("LAT_BATCHES", "1,64,512,1024,2048").split(","))
SPEC = os.environ.get("LAT_SPECIALIZE", "1") != "0"
if SPEC:
    be.specialize()
for B in BATCHES:
    x0, qc = bench.make_inputs(B, 0)
    if B == 1:
        qc[0] = np.deg2rad(bench.QC0_DEG); x0[0, :350] = np.tile(qc[0], 50)
    for _ in range(int(os.environ.get("LAT_REPEATS", "1"))):
        be.solve(x0, qc)
    t0 = time.perf_counter(); r = be.solve(x0, qc); t1 = time.perf_counter()
    tm = be.timing()
    print(f"B={B}: wall {1e3*(t1-t0):.2f} ms device {tm['solve_ms']:.2f} ms iters mean {r.iters.mean():.1f} max {r.iters.max()} "
          f"-> {tm['solve_ms']*1e3/max(1,r.iters.max()):.1f} us per iteration of the slowest instance; conv {np.mean(r.status==0):.3f}; solves/s {B/(tm['solve_ms']*1e-3):.0f}")
