"""PCIe-inclusive rate of oh_solve (pageable host buffers) over the chunk size of the two-lane pipeline (a lane-count option was tried in round 6 and not kept: HISTORY)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
dt, lp = bench.local_path()
chain = RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
B = 262144
x0, qc = bench.make_inputs(B, 0)
lib = _lib.load()
hx, hf, hk = np.zeros((B, x0.shape[1])), np.zeros(B), np.zeros((B, 3))
hi, hs = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
for lanes in (2,):
    for chunk in (16384, 32768, 65536):
        be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-8, hessian=2).set_options(pipe_chunk=chunk)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            _lib.check(lib.oh_solve(be.handle, B, _lib._ptr(x0), _lib._ptr(qc), _lib._ptr(hx), _lib._ptr(hf), _lib._ptr(hk), _lib._ptr(hi), _lib._ptr(hs)), "oh_solve")
            ts.append(time.perf_counter() - t0)
        be.close()
        print(f"lanes {lanes} chunk {chunk}: {B / min(ts[1:]) / 1e6:.3f} M solves/s ({1e3 * min(ts[1:]):.1f} ms), converged {float((hs == 0).mean()):.4f}", flush=True)
