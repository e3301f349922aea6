"""Development probe: one batch on one handle against the same batch split over several handles whose solves run concurrently from host threads
(each handle has its own stream): do the VALU-bound and the HBM-bound kernels of different sub-batches overlap on the GPU?"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("OPTAS_HIP_CACHE", os.path.join(ROOT, ".optas_hip_cache"))
import optas_amd
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
import bench

B = int(os.environ.get("TS_B", "262144"))
dt, lp = bench.local_path()
chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK)
x0, qc = bench.make_inputs(B, 0)

def make(n):
    be = FigureEightBackend(chain, 50, dt, lp, max_iter=300, tol=1e-6)
    be.specialize()
    bufs = dict(x0=_lib.DeviceBuffer(n * x0.shape[1] * 8), p=_lib.DeviceBuffer(n * 7 * 8), x=_lib.DeviceBuffer(n * x0.shape[1] * 8), f=_lib.DeviceBuffer(n * 8),
                k=_lib.DeviceBuffer(n * 24), it=_lib.DeviceBuffer(n * 4), st=_lib.DeviceBuffer(n * 4))
    return be, bufs

def run(be, bufs, n):
    be.solve_device(n, bufs["x0"], bufs["p"], bufs["x"], bufs["f"], bufs["k"], bufs["it"], bufs["st"])

for parts in (1, 2, 3, 4):
    n = B // parts
    hs = [make(n) for _ in range(parts)]
    for i, (be, bufs) in enumerate(hs):
        bufs["x0"].upload(np.ascontiguousarray(x0[i * n : (i + 1) * n]))
        bufs["p"].upload(np.ascontiguousarray(qc[i * n : (i + 1) * n]))
        run(be, bufs, n)  # warm-up
    best = 1e9
    for rep in range(3):
        _lib.check(_lib.load().oh_device_synchronize(), "sync")
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(be, bufs, n)) for be, bufs in hs]
        for t in th: t.start()
        for t in th: t.join()
        _lib.check(_lib.load().oh_device_synchronize(), "sync")
        best = min(best, time.perf_counter() - t0)
    st = np.concatenate([bufs["st"].download(np.int32, (n,)) for _, bufs in hs])
    print(f"{parts} handle(s) x {n}: wall {1e3 * best:.1f} ms -> {parts * n / best:.0f} solves/s, converged {np.mean(st == 0):.4f}")
    for be, bufs in hs:
        be.close()
        for b in bufs.values(): b.free()
