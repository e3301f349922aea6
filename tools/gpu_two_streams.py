"""Headline batch split over S handles (one stream and one host thread each) against one handle: do the latency-bound phases of one part (persistent
tail, small launches, host round trips, compactions) hide behind the bandwidth-bound launches of the other?  python tools/gpu_two_streams.py [B]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import optas_amd  # noqa: E402
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dt, lp = bench.local_path()
robot = optas_amd.RobotModel.builtin("kuka_lwr")
x0, qc = bench.make_inputs(B, 0)


def part(lo, hi):
    be = FigureEightBackend(robot.kinematic_chain(bench.LINK), bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2)
    n = hi - lo
    bufs = [_lib.DeviceBuffer(x0[lo:hi].nbytes).upload(np.ascontiguousarray(x0[lo:hi])), _lib.DeviceBuffer(qc[lo:hi].nbytes).upload(np.ascontiguousarray(qc[lo:hi])),
            _lib.DeviceBuffer(x0[lo:hi].nbytes), _lib.DeviceBuffer(n * 8), _lib.DeviceBuffer(n * 24), _lib.DeviceBuffer(n * 4), _lib.DeviceBuffer(n * 4)]
    return be, n, bufs


for S in (1, 2, 3, 4, 1):
    parts = [part(B * i // S, B * (i + 1) // S) for i in range(S)]

    def run(k, reps):
        be, n, bufs = parts[k]
        for _ in range(reps):
            be.solve_device(n, *bufs)

    for reps, timed in ((2, False), (4, True)):
        _lib.check(lib.oh_device_synchronize(), "sync")
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(k, reps)) for k in range(S)]
        [t.start() for t in th]
        [t.join() for t in th]
        _lib.check(lib.oh_device_synchronize(), "sync")
        el = time.perf_counter() - t0
        if timed:
            st = np.concatenate([p[2][6].download(np.int32, (p[1],)) for p in parts])
            print(f"handles {S}: {B * reps / el / 1e6:.3f} M solves/s, {el / reps * 1e3:.2f} ms per pass of {B}, converged {float((st == 0).mean()):.4f}", flush=True)
    for be, n, bufs in parts:
        be.close()
