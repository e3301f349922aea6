"""Development probe: the generated tape kernel with its work set in LDS against the global buffer, over batch sizes (config 1 through the tape family)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.example import setup_solver as ik_setup
from optas_amd import _lib
from optas_amd.backend import TapeBackend
from optas_amd.models import RobotModel
from optas_amd.tape import compile_problem
kuka = RobotModel.builtin("kuka_lwr")
tp = compile_problem(ik_setup(build_only=True)[1])
rng = np.random.default_rng(20260927)
for B in (1, 64, 512, 2048, 4096, 8192, 32768, 65536, 131072):
    qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
    pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits).T)).T
    p = np.ascontiguousarray(np.concatenate([qn, pg], 1))
    out = []
    for mode in ("0", "1000000"):
        os.environ["OH_TAPE_LDS_MAX"] = mode
        be = TapeBackend(tp, max_iter=2000)
        bufs = [_lib.DeviceBuffer(a.nbytes) for a in (qn, p)]
        bufs[0].upload(np.ascontiguousarray(qn)); bufs[1].upload(p)
        d = [_lib.DeviceBuffer(qn.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B), _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)]
        ms = []
        for _ in range(4):
            be.solve_device(B, bufs[0], bufs[1], *d)
            ms.append(be.solve_ms())
        out.append(float(np.median(ms[1:])))
        for b in bufs + d:
            b.free()
        be.close()
    print(f"B={B}: global work {out[0]:.2f} ms, LDS work {out[1]:.2f} ms")
