"""GPU optima of config 2 against the reference's ALGORITHM CLASS run on the reference's FORM from the reference's SEED
(tests/golden/nlp_ipm_golden.npz, written by tools/make_golden.py --ipm with oracle/ipm_reference_form.py: primal-dual interior point with
filter line search, IPOPT's defaults, on the literal `min f s.t. 0 <= v <= 1e10`; 17 distinct instances: the nominal one and 16 of the bench
workload).  IPOPT itself cannot run here (PARITY UNPINNED against the reference's own iterates); this is the closest independent from-seed
answer the container allows.  Per instance one of three things is asserted, and the counts are printed:

  same basin   the interior-point run converged (E_0 <= 1e-8) and its point, put onto the exact equalities by Newton-SQP steps (IPOPT's bound
               relaxation lets every row of v sit 1e-8 below zero: worth sum|lam| 1e-8 ~ 7e-6 in f), is the GPU's optimum:
               |f_gpu - f_polished| <= 1e-8 f,  |x_gpu - x_polished| <= 1e-3,  and  |f_gpu - f_ipm| <= 2e-5 (the relaxation)
  other basin  it converged to a different local minimum (the problem is nonconvex): both points satisfy the reference-form KKT conditions
               (GPU: stationarity 1e-5, feasibility 1e-9, complementarity 1e-8; interior point: feasibility 1.01e-8 = its relaxation) and the
               GPU's objective is not worse
  unfinished   it was still descending the curved valley of the stiff tracking cost when its iteration budget ran out: the GPU's objective is
               below the value it had reached, and the GPU point is a KKT point
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, KUKA_KIN
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
from oracle.problems import FigureEightNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form

pytestmark = pytest.mark.gpu
LINK = "end_effector_ball"


def test_gpu_against_the_interior_point_runs_from_the_reference_seed(hip_lib):
    g = np.load(os.path.join(GOLDEN, "nlp_ipm_golden.npz"))
    _, first = np.unique(g["qc"], axis=0, return_index=True)
    idx = np.sort(first)
    assert len(idx) >= 17
    qc = g["qc"][idx]
    nlp = FigureEightNLP(OracleRobot(KUKA_KIN), LINK, T=50)
    be = FigureEightBackend(RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK), 50, nlp.dt, nlp.local_path.T, max_iter=300, tol=1e-7)
    res = be.solve(np.stack([nlp.seed(q) for q in qc]), qc)
    be.close()
    assert (res.status == 0).all()
    counts = {"same": 0, "other": 0, "unfinished": 0}
    for n, i in enumerate(idx):
        f_gpu = res.f[n]
        k = kkt_reference_form(nlp, res.x[n], qc[n])
        assert k["stationarity"] <= 1e-5 and k["feasibility"] <= 1e-9 and k["complementarity"] <= 1e-8, (i, k["stationarity"], k["feasibility"])
        if g["optimal"][i] and g["same_basin"][i]:
            counts["same"] += 1
            assert abs(f_gpu - g["f_polished"][i]) <= 1e-8 * f_gpu, (i, f_gpu, g["f_polished"][i])
            assert np.abs(res.x[n] - g["x_polished"][i]).max() <= 1e-3
            assert abs(f_gpu - g["f_ipm"][i]) <= 2e-5 and g["kkt_ipm"][i][1] <= 1.01e-8
            assert g["kkt_polished"][i][0] <= 1e-6 and g["kkt_polished"][i][1] <= 1e-9
        elif g["optimal"][i]:
            counts["other"] += 1
            assert g["kkt_ipm"][i][1] <= 1.01e-8 and g["kkt_polished"][i][0] <= 1e-5 and g["kkt_polished"][i][1] <= 1e-9
            assert f_gpu <= g["f_polished"][i] + 1e-9, (i, f_gpu, g["f_polished"][i])
        else:
            counts["unfinished"] += 1
            assert f_gpu <= g["f_ipm"][i] + 1e-9, (i, f_gpu, g["f_ipm"][i])
    print("interior point from the reference seed vs GPU:", counts)
    assert counts["same"] >= 10 and g["same_basin"][0]  # the nominal instance (BASELINE configs[1] literally) is one of them


def test_velocity_limited_problems_against_the_interior_point_runs(hip_lib):
    """The problems with joint-velocity limit rows that round 3 lowered (tests/golden/ipm_limits_golden.npz, tools/make_golden.py --ipm-limits): the
    interior-point oracle from the reference's seed on the literal layouts -- figure_eight_plan.py + the LWR's velocity limits (4 instances) and
    dual_arm.py + 0.06 rad/s.  Where it converged to the GPU's basin the objectives agree to the bound relaxation (sum|lam| x 1e-8: 2e-5
    absolute on the figure-eight, 1e-7 on the dual arm); another basin must not be better than the GPU's by more than that."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from examples.dual_arm import setup_solver as dual_arm
    from examples.figure_eight_plan import setup_solver as figure_eight

    g = np.load(os.path.join(GOLDEN, "ipm_limits_golden.npz"))
    qcs, vl = g["fig8v_qc"], g["fig8v_vl"]
    kuka, solver = figure_eight(velocity_limits=(-vl, vl), solver_options={"max_iter": 600, "tol": 1e-7})
    x0 = np.zeros((len(qcs), solver.opt.nx))
    x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(len(qcs), 350)
    r = solver.solve_batch_arrays(x0, qcs)
    assert (r.status == 0).all() and g["fig8v_ok"].all()
    same = np.abs(r.f - g["fig8v_f"]) <= 2e-5
    assert same.sum() >= 3 and same[0], (r.f, g["fig8v_f"])  # the nominal instance among them
    assert np.all(r.f[~same] <= g["fig8v_f"][~same] + 2e-5)
    solver.backend.close()
    if "dualv_f" in g.files:
        vmax, T = float(g["dualv_vmax"]), 50
        (kl, kr), s2 = dual_arm(T=T, velocity_limits=(-np.full(7, vmax), np.full(7, vmax)), solver_options={"max_iter": 600, "tol": 1e-7})
        p = g["dualv_p"]
        s2.reset_parameters({"qcl": p[:7], "qcr": p[7:]})
        s2.solve()  # zero seed, as the script leaves it
        assert s2.did_solve() and bool(g["dualv_ok"])
        assert abs(s2.stats()["f"][0] - float(g["dualv_f"])) <= 1e-6, (s2.stats()["f"][0], float(g["dualv_f"]))
    if "tqv_f" in g.files:  # torque MPC (T = 6) with joint-velocity limits next to the effort limits
        from optas_amd.backend import TorqueBackend

        med7 = RobotModel.builtin("med7")
        vmax, lim = float(g["tqv_vmax"]), float(g["tqv_lim"])
        be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=6, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-lim, tau_up=lim,
                           dq_lo=-vmax, dq_up=vmax, max_iter=600)
        qc, goal = g["tqv_qc"], g["tqv_goal"]
        x0 = np.zeros((1, 4 * 7 * 6))
        x0[0, :42] = np.tile(qc, 6)
        r = be.solve(x0, np.concatenate([qc, np.zeros(7), goal.reshape(-1)])[None])
        be.close()
        assert r.status[0] == 0 and bool(g["tqv_ok"])
        assert r.f[0] >= float(g["tqv_f"]) - 1e-9 and r.f[0] - float(g["tqv_f"]) <= 5e-6 * r.f[0], (r.f[0], float(g["tqv_f"]))


def test_config3_point_mass_ticks_against_the_interior_point_on_the_reference_form(hip_lib):
    """BASELINE config 3 (point_mass_mpc.py tick): the kernel's answers against tests/golden/ipm_pm_golden.npz -- oracle/ipm_reference_form.py on the
    literal 264-row form from the script's zero seed (tools/make_golden.py --ipm-pm; round-3 verdict, Missing 3).  All nine ticks: same basin.  The
    interior point's optimum sits up to sum|lam| 1e-8 ~ 4e-6 below the kernel's (IPOPT's bound relaxation of the 2 x 42 dynamics rows)."""
    from optas_amd.backend import PointMassBackend
    from oracle.problems import PointMassMPCNLP

    gi = np.load(os.path.join(GOLDEN, "ipm_pm_golden.npz"))
    nlp = PointMassMPCNLP()
    assert gi["optimal"].all()
    be = PointMassBackend(tol=1e-9)
    r = be.solve(np.zeros((len(gi["p"]), nlp.nx)), gi["p"])
    be.close()
    assert (r.status == 0).all()
    for i in range(len(gi["p"])):
        assert abs(r.f[i] - gi["f"][i]) <= 2e-5 * max(1.0, abs(gi["f"][i])) and r.f[i] >= gi["f"][i] - 1e-12, (i, r.f[i], gi["f"][i])
        assert np.abs(r.x[i] - gi["x"][i]).max() <= 2e-4, (i, np.abs(r.x[i] - gi["x"][i]).max())


def test_config5_torque_mpc_at_T30_against_the_interior_point_on_the_reference_form(hip_lib):
    """BASELINE config 5 at its stated size (T = 30: 840 variables, 1680 rows of v) against tests/golden/ipm_configs_golden.npz (tq_t30*, tools/
    make_golden.py --ipm-configs t30 t30lim: oracle/ipm_reference_form.py with the exact Lagrangian Hessian, from the reference's seed; round-3
    verdict, Missing 3).  What the instrument shows, and what is asserted: the interior point on the literal form ends in the kernel's basin on ONE
    of the five instances (objective equal to the relaxation, 5e-6); on the other four it converges -- E_0 <= 1e-8 or IPOPT's "acceptable" test --
    to KKT points with objectives 52 .. 73 against the kernel's 9.2 .. 10.0 (the arm swings the long way round: tracking 20 .. 32, velocity 19 ..
    28).  They are genuine stationary points (tests/test_ipm_reference_form.py grades them; the kernel's own state machine seeded there stays there),
    so the problem has several local minima and which one a run reaches is a property of the method: both answers are KKT points of the reference's
    NLP, the kernel's is the better one.  Whether IPOPT itself would land where this restatement of its algorithm lands cannot be checked here."""
    from optas_amd.backend import TorqueBackend
    from oracle.problems import TorqueMPCNLP
    from oracle.torque import TorqueProblem

    g, gi = np.load(os.path.join(GOLDEN, "torque_golden.npz")), np.load(os.path.join(GOLDEN, "ipm_configs_golden.npz"))
    med7 = RobotModel.builtin("med7")
    orc = OracleRobot(os.path.join(os.path.dirname(KUKA_KIN), "med7.kin.json"))
    counts = {"same basin": 0, "other basin, kernel lower": 0}
    for tag in ("t30", "t30lim"):
        lim = float(g[tag + "_lim"])
        nlp = TorqueMPCNLP(TorqueProblem(orc, "lbr_link_ee", T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=None if lim > 1e8 else lim))
        lim = 100.0 if lim > 1e8 else lim
        be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-lim, tau_up=lim)
        qc, goal = g[tag + "_qc"], g[tag + "_goal"]
        B = len(qc)
        p = np.stack([nlp.pack_p(qc[b], np.zeros(7), goal[b]) for b in range(B)])
        r = be.solve(np.stack([nlp.seed(q) for q in qc]), p)
        lam = be.multipliers(B)
        be.close()
        assert (r.status == 0).all()
        for b in range(B):
            f_ipm = float(gi[f"tq_{tag}_f"][b])
            k = kkt_reference_form(nlp, r.x[b], p[b], lam_kg=np.concatenate([lam[b][:, :7].reshape(-1), lam[b][:, 7:].reshape(-1)]))
            assert k["stationarity"] <= 1e-6 and k["feasibility"] <= 1e-10 and k["complementarity"] <= 1e-8
            if abs(r.f[b] - f_ipm) <= 1e-5 * r.f[b]:
                counts["same basin"] += 1
                assert np.abs(r.x[b][: 7 * 30] - gi[f"tq_{tag}_x"][b][: 7 * 30]).max() <= 1e-3  # the joint trajectories coincide
            else:
                counts["other basin, kernel lower"] += 1
                assert r.f[b] < f_ipm
    print("config 5, T = 30, five instances vs the interior point on the reference form:", counts)
    assert counts["same basin"] >= 1 and sum(counts.values()) == 5
