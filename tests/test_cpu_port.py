"""oracle/cpu_port (the compiled CPU baseline of bench.py: the HIP kernels' state machine built for the host cores) against the
numpy restatement on the headline workload: same step counts, same optima.  Builds the library with hipcc (host part only runs)."""
import numpy as np

import bench
import optas_amd
from conftest import KUKA_KIN
from oracle import cpu_port
from oracle.robot import OracleRobot
from oracle.structured import StructuredFigureEight, solve_structured_lm


def test_compiled_port_matches_numpy_restatement():
    cpu_port.build()
    dt, lp = bench.local_path()
    chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain("end_effector_ball")
    B = 12
    x0, qc = bench.make_inputs(B, 0)
    for hessian, mode in (("hybrid", 2), ("gauss_newton", 0)):
        x, f, kkt, it, st = cpu_port.solve(chain, 50, dt, lp, x0, qc, hessian=mode, threads=3)
        assert (st == 0).all() and (kkt[:, 0] <= 1e-6).all() and (kkt[:, 1] <= 1e-9).all()
        prob = StructuredFigureEight(OracleRobot(KUKA_KIN), "end_effector_ball", T=50)
        for b in range(0, B, 3):
            s = solve_structured_lm(prob, qc[b], max_iter=300, tol=1e-6, hessian=hessian)
            # same optimum; the step at which the reduced gradient first dips under the hybrid switch (1e-2, approached as 1.2e-2, 7.4e-3 ...)
            # and the fate of a borderline over-relaxed step depend on rounding, which differs between libm and the kernels' own sincos
            assert abs(int(it[b]) - s["iters"]) <= 3 and abs(f[b] - s["f"]) <= 1e-9 * abs(s["f"])
            assert np.abs(x[b, :350].reshape(50, 7) - s["Q"]).max() < 1e-4
    x1, f1, _, it1, _ = cpu_port.solve(chain, 50, dt, lp, x0, qc, hessian=2, threads=1)
    x3, f3, _, it3, _ = cpu_port.solve(chain, 50, dt, lp, x0, qc, hessian=2, threads=3)
    assert np.array_equal(x1, x3) and np.array_equal(it1, it3)  # threading only partitions instances
