"""Development probe: K1 in the SoA layout (solver-internal, what bench.py reports) and in the reference layout of the ABI (q[n][ndof], pose[n][7], J[n][6][ndof])."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("OPTAS_HIP_CACHE", os.path.join(ROOT, ".optas_hip_cache"))
import optas_amd, bench
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
dt, lp = bench.local_path()
be = FigureEightBackend(optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(bench.LINK), 50, dt, lp)
n = 1 << 22
rng = np.random.default_rng(1)
q = rng.uniform(-2.9, 2.9, (n, 7))
d_q, d_qs = _lib.DeviceBuffer(n * 56), _lib.DeviceBuffer(n * 56)
d_q.upload(np.ascontiguousarray(q)); d_qs.upload(np.ascontiguousarray(q.T))
d_pose, d_J = _lib.DeviceBuffer(n * 56), _lib.DeviceBuffer(n * 336)
for name, fn, dq in (("SoA", be.fk_jac_soa_device, d_qs), ("AoS (reference layout)", be.fk_jac_device, d_q)):
    fn(n, dq, d_pose, d_J)
    ms = []
    for _ in range(5):
        be.event_timer_start(); fn(n, dq, d_pose, d_J); ms.append(be.event_timer_stop())
    t = float(np.mean(ms))
    print(f"{name}: {t:.3f} ms for {n} units -> {n * 448 / (t * 1e-3) / 1e12:.2f} TB/s of algorithmic bytes ({n * 448 / (t * 1e-3) / 8e12:.3f} of the HBM roofline)")
