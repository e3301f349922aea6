"""Experiment: do two half-batches on two handles/streams (two host threads) overlap compute-bound k_eval with memory-bound k_couple/k_step?"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import optas_amd
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
import bench
dt, lp = bench.local_path()
robot = optas_amd.RobotModel.builtin("kuka_lwr")
def make(B, rank):
    be = FigureEightBackend(robot.kinematic_chain("end_effector_ball"), 50, dt, lp, max_iter=300, tol=1e-6)
    x0, qc = bench.make_inputs(B, rank)
    bufs = dict(x0=_lib.DeviceBuffer(x0.nbytes).upload(x0), p=_lib.DeviceBuffer(qc.nbytes).upload(qc), x=_lib.DeviceBuffer(x0.nbytes),
                f=_lib.DeviceBuffer(B*8), k=_lib.DeviceBuffer(B*24), it=_lib.DeviceBuffer(B*4), st=_lib.DeviceBuffer(B*4))
    return be, bufs, B
def run(h):
    be, b, B = h
    be.solve_device(B, b["x0"], b["p"], b["x"], b["f"], b["k"], b["it"], b["st"])
for nth, B in ((1, 131072), (2, 65536), (4, 32768), (1, 65536)):
    hs = [make(B, r) for r in range(nth)]
    for h in hs: run(h)
    t0 = time.perf_counter()
    for rep in range(3):
        ths = [threading.Thread(target=run, args=(h,)) for h in hs]
        for t in ths: t.start()
        for t in ths: t.join()
    dtm = (time.perf_counter() - t0) / 3
    print(f"{nth} thread(s) x B={B}: {dtm*1e3:.1f} ms per round -> {nth*B/dtm:.0f} solves/s")
    for be, b, _ in hs:
        be.close()
        for v in b.values(): v.free()
