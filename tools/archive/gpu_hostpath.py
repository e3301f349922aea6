"""PCIe-inclusive rate of the host-buffer entry point (oh_solve) vs the resident path (oh_solve_device)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import optas_amd, bench
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
dt, lp = bench.local_path()
robot = optas_amd.RobotModel.builtin("kuka_lwr")
be = FigureEightBackend(robot.kinematic_chain("end_effector_ball"), 50, dt, lp, max_iter=300, tol=1e-6)
for B in (4096, 65536):
    x0, qc = bench.make_inputs(B, 0)
    be.solve(x0[:64], qc[:64])
    t0 = time.perf_counter(); r = be.solve(x0, qc); t1 = time.perf_counter()
    dev = be.timing()["solve_ms"]
    print(f"B={B}: oh_solve (host buffers, H2D+D2H of {2*x0.nbytes/1e6:.0f} MB) wall {1e3*(t1-t0):.1f} ms -> {B/(t1-t0):.0f} solves/s ; device part {dev:.1f} ms -> {B/(dev*1e-3):.0f} solves/s")
