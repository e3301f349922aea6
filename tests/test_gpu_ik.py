"""BASELINE config 1 (example/example.py) on the GPU: OH_PROBLEM_IK through HIPSolver / the C ABI against the oracle.
Tolerances: objective 1e-7 and solution 1e-6 vs the golden optimum (scipy SLSQP in the reference wiring == the port),
reference-form KKT stationarity <= 1e-6, feasibility <= 1e-9; iteration counts equal the numpy port's (same state
machine) up to +-2 evaluations."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, KUKA_KIN, SEED
from optas_amd.backend import IKBackend
from optas_amd.models import RobotModel
from oracle.ik_al import solve_ik_al
from oracle.problems import IKExampleNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import FoldedChain

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.example import END_EFFECTOR, setup_solver  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    kuka = OracleRobot(KUKA_KIN)
    return IKExampleNLP(kuka, END_EFFECTOR), FoldedChain(kuka, END_EFFECTOR), np.load(os.path.join(GOLDEN, "ik_golden.npz"))


def test_reference_script_flow_and_known_answer(hip_lib, oracle):
    ik, ch, g = oracle
    robot, solver = setup_solver()
    name = robot.get_name()
    q_nominal = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    p_goal = np.asarray(robot.get_global_link_position(END_EFFECTOR, q_nominal)).reshape(-1) + np.array([0.0, 0.3, -0.2])
    solver.reset_parameters({"q_nominal": q_nominal, "p_goal": p_goal})
    solver.reset_initial_seed({f"{name}/q": q_nominal})  # not a decision-variable label: zero seed, like the reference
    sol = solver.solve()
    assert solver.did_solve() and abs(solver.stats()["f"][0] - 0.29579887518) < 1e-8  # SURVEY App. D / golden
    q = np.asarray(sol[f"{name}/q"]).reshape(-1)
    assert sol[f"{name}/q"].shape == (7, 1) and np.abs(q - g["x"][0]).max() < 1e-6
    p = np.concatenate([q_nominal, p_goal])
    assert np.allclose(p, g["p"][0], atol=1e-14)
    k = kkt_reference_form(ik, q, p, active_tol=1e-7)
    assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-9
    # diagnostics through the Solver interface (solver.py:167-237, 269-314)
    assert abs(solver.evaluate_cost({f"{name}/q/x": q}, {"q_nominal": q_nominal, "p_goal": p_goal}) - solver.stats()["f"][0]) < 1e-12
    # error_on_fail (solver.py:133-134): one evaluation cannot converge
    from optas_amd.solver import HIPSolver

    s2 = HIPSolver(solver.opt, error_on_fail=True).setup("hip_sqp", {"max_iter": 2})
    s2.reset_parameters({"q_nominal": q_nominal, "p_goal": p_goal})
    with pytest.raises(RuntimeError, match="Solver failed!"):
        s2.solve()


def test_golden_batch_bit_for_bit_state_machine(hip_lib, oracle):
    ik, ch, g = oracle
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(END_EFFECTOR)
    be = IKBackend(chain, ik.lo, ik.up)
    res = be.solve(g["x0"], g["p"])
    mu, zlo, zup = be.multipliers(len(g["p"]))
    assert (res.status == 0).all()
    assert np.abs(res.f - g["f"]).max() < 1e-7 and np.abs(res.x - g["x"]).max() < 1e-6
    for b, (p, x0) in enumerate(zip(g["p"], g["x0"])):
        r = solve_ik_al(ch, x0, p[:7], p[7:], ik.lo, ik.up, tol=1e-6, tol_feas=1e-9, max_iter=200)
        assert abs(int(res.iters[b]) - r["iterations"]) <= 2
        assert np.abs(res.x[b] - r["x"]).max() < 1e-8 and np.abs(mu[b] - r["lam_h"]).max() < 1e-5
        assert np.abs(zlo[b] - r["z_lo"]).max() < 1e-5 and np.abs(zup[b] - r["z_up"]).max() < 1e-5
        assert int((zlo[b] > 0).sum() + (zup[b] > 0).sum()) == g["nactive"][b]
        # stationarity of the reference form from the returned multipliers
        lam = np.concatenate([zlo[b], zup[b], np.maximum(mu[b], 0), np.maximum(-mu[b], 0)])
        assert np.abs(ik.df(res.x[b], p) - ik.dv(res.x[b], p).T @ lam).max() < 1e-6
        assert res.kkt[b, 0] < 1e-6 and res.kkt[b, 1] < 1e-9 and abs(ik.f(res.x[b], p) - res.f[b]) < 1e-12


def test_large_batch_properties(hip_lib, oracle):
    """65536 random goals: every converged instance is feasible to 1e-9, inside the limits, and stationary; goals
    generated from configurations inside the limits are reachable, so nearly all converge."""
    ik, ch, g = oracle
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(END_EFFECTOR)
    be = IKBackend(chain, ik.lo, ik.up, max_iter=300)
    rng = np.random.default_rng(SEED)
    B = 65536
    qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
    pg, _, _, _ = ch.fk(np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), ik.lo, ik.up))  # reachable by construction
    p = np.concatenate([qn, pg], 1)
    res = be.solve(qn, p)
    ok = res.status == 0
    assert ok.mean() > 0.995
    assert (res.x >= ik.lo - 0).all() and (res.x <= ik.up + 0).all()
    e2, _, _, _ = ch.fk(res.x[ok])
    assert np.abs(pg[ok] - e2).max() < 1e-9 and res.kkt[ok, 0].max() < 1e-6
    assert np.abs(np.sum((res.x - qn) ** 2, 1) - res.f).max() < 1e-12
    # spot-check 16 instances against the numpy port
    for b in rng.integers(0, B, 16):
        r = solve_ik_al(ch, qn[b], qn[b], pg[b], ik.lo, ik.up, tol=1e-6, tol_feas=1e-9, max_iter=300)
        if r["status"] == 0 and ok[b]:
            assert abs(r["f"] - res.f[b]) < 1e-7
    print("ik batch: %.3f ms device for %d instances, iterations p50 %d max %d, converged %.4f" % (be.solve_ms(), B, np.median(res.iters), res.iters.max(), ok.mean()))
