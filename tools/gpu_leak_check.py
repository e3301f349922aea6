"""Device memory after repeated create / solve / destroy cycles, by path: a leak shows as a falling hipMemGetInfo free figure."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20


dt, lp = bench.local_path()
chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
B = 70000
x0, qc = bench.make_inputs(B, 0)
for name, opts, mult in (("one stream, no pipeline", {"pipe": 0, "streams": 1}, False), ("split on two streams", {"pipe": 0}, False), ("pipelined", {}, False), ("pipelined + multipliers", {}, True),
                         ("small batch 2048", {"small": 1}, False)):
    hist = []
    for k in range(5):
        be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-8, hessian=2)
        small = opts.get("small")
        be.set_options({k2: v for k2, v in opts.items() if k2 != "small"})
        n = 2048 if small else B
        r = be.solve(x0[:n], qc[:n])
        if mult:
            be.multipliers(n)
            be.set_option("pipe", 0)  # ... and the same handle once more, split on two streams
            be.solve(x0[:n], qc[:n])
            be.set_option("batch_invariant", 1)
            be.solve(x0[:n], qc[:n])
        be.close()
        hist.append(free_mb())
    print(f"{name:28s} free MB after each cycle: {[round(h) for h in hist]}  drift {hist[-1] - hist[1]:.0f} MB", flush=True)
