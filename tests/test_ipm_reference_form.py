"""oracle/ipm_reference_form.py -- the reference's algorithm class (IPOPT: primal-dual interior point, filter line search) on the reference's
problem form (min f s.t. 0 <= v <= 1e10, equalities as (e, -e) pairs), from the reference's seeds.  Pins:
  * the reference's only numeric solver assertion (Booth -> (1, 3), tests/test_solver.py:46-54), with and without its dummy bound rows
  * config 1 (example.py) and config 3 (point_mass_mpc.py) answers the SLSQP-wired goldens hold (tools/make_golden.py), reached here by a
    different algorithm: to the bound relaxation at IPOPT's default (1e-8 per row), to 1e-9 with the relaxation tightened
  * config 2 at T = 5: the dense-SQP golden, from the seed
  * the vectorised figure-eight NLP the T = 50 runs use = the literal per-knot restatement, member by member
The T = 50 runs themselves (24 instances, ~10 minutes) are stored in tests/golden/nlp_ipm_golden.npz (tools/make_golden.py --ipm) and compared
with the GPU in tests/test_gpu_ipm_parity.py."""
import os

import numpy as np

from conftest import GOLDEN, KUKA_KIN, MED7_KIN
from oracle.ipm_reference_form import solve_ipm
from oracle.problems import BoothNLP, FastFigureEightNLP, FigureEightNLP, IKExampleNLP, PointMassMPCNLP, TorqueMPCNLP
from oracle.robot import OracleRobot
from oracle.solvers import dense_sqp, kkt_reference_form

LINK = "end_effector_ball"


def test_booth_known_answer():
    r = solve_ipm(BoothNLP(), np.zeros(2), np.array([2.0, 7.0]))
    assert r["status"] == "optimal" and np.abs(r["x"] - [1.0, 3.0]).max() <= 1e-8  # tests/test_solver.py:46-54

    class Bounded(BoothNLP):  # the OSQP variant of the reference test adds dummy rows -1e9 <= x <= 1e9 (tests/test_solver.py:56-70)
        nk = 4

        def k(self, x, p):
            return np.concatenate([x + 1e9, 1e9 - x])

        def dk(self, x, p):
            return np.concatenate([np.eye(2), -np.eye(2)])

    r = solve_ipm(Bounded(), np.zeros(2), np.array([2.0, 7.0]))
    assert r["status"] == "optimal" and np.abs(r["x"] - [1.0, 3.0]).max() <= 1e-6
    # active bound rows: x <= 0.5 cuts the optimum off; KKT in the reference's form
    class Cut(BoothNLP):
        nk = 1

        def k(self, x, p):
            return np.array([0.5 - x[0]])

        def dk(self, x, p):
            return np.array([[-1.0, 0.0]])

    nlp = Cut()
    r = solve_ipm(nlp, np.zeros(2), np.array([2.0, 7.0]), relax=1e-12)
    k = kkt_reference_form(nlp, r["x"], np.array([2.0, 7.0]))
    assert abs(r["x"][0] - 0.5) <= 1e-7 and k["stationarity"] <= 1e-6 and r["lam_v"][0] > 0.1  # multiplier in the reference's sign


def test_config1_ik_from_both_seeds(golden_nlp):
    nlp = IKExampleNLP(OracleRobot(KUKA_KIN))
    p = golden_nlp["ik_p"]
    for seed in (np.zeros(7), p[:7]):  # the script's effective seed (zero fill, sx_container.py:121) and the nominal pose
        r = solve_ipm(nlp, seed, p)
        assert r["status"] == "optimal"
        assert np.abs(r["x"] - golden_nlp["ik_x"]).max() <= 1e-6 and abs(r["f"] - float(golden_nlp["ik_f"])) <= 1e-6  # IPOPT's default relaxation
    r = solve_ipm(nlp, np.zeros(7), p, relax=1e-12)
    assert abs(r["f"] - float(golden_nlp["ik_f"])) <= 1e-9 and np.abs(r["x"] - golden_nlp["ik_x"]).max() <= 1e-7  # (the golden is an SLSQP answer)
    k = kkt_reference_form(nlp, r["x"], p)
    assert k["stationarity"] <= 1e-7 and k["feasibility"] <= 1e-9 and k["complementarity"] <= 1e-7


def test_config3_point_mass_ticks():
    g = np.load(os.path.join(GOLDEN, "pm_golden.npz"))
    nlp = PointMassMPCNLP()
    for i in (0, 1, 4):
        r = solve_ipm(nlp, np.zeros(nlp.nx), g["p"][i])
        assert r["status"] == "optimal"
        # IPOPT's default bound relaxation lets each of the 42 dynamics rows move by 1e-8: worth sum|lam| 1e-8 ~ 4e-6 in f on these ticks
        assert abs(r["f"] - g["f"][i]) <= 2e-5 * max(1.0, g["f"][i]), (i, r["f"], g["f"][i])
        # KKT of the reference form with the method's own multipliers lam_v >= 0 of v >= 0: grad f = Jv^T lam_v, lam_v . v ~ mu
        v, lam = nlp.v(r["x"], g["p"][i]), r["lam_v"]
        assert v.min() >= -1.01e-8 and lam.min() >= 0.0
        assert np.abs(nlp.df(r["x"], g["p"][i]) - nlp.dv(r["x"], g["p"][i]).T @ lam).max() <= 1e-6
        assert np.abs(lam * np.minimum(v, 1.0)).max() <= 1e-6


def test_config2_short_horizon_from_the_seed(golden_nlp):
    orc = OracleRobot(KUKA_KIN)
    T = 5
    nlp = FastFigureEightNLP(orc, LINK, T=T, Tmax=10.0 * (T - 1) / 49.0)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    r = solve_ipm(nlp, nlp.seed(qc), qc)
    assert r["status"] == "optimal"
    k = kkt_reference_form(nlp, r["x"], qc)
    assert k["stationarity"] <= 1e-6 and k["feasibility"] <= 1.01e-8  # the (e, -e) pairs hold to the bound relaxation, as IPOPT's would
    d = dense_sqp(nlp, r["x"], qc, max_iter=10, tol=1e-10)  # onto the exact equalities
    assert d["converged"] and abs(d["f"] - float(golden_nlp["fig8_T5_f"])) <= 1e-9 and np.abs(d["x"] - golden_nlp["fig8_T5_x"]).max() <= 1e-5
    assert abs(r["f"] - d["f"]) <= 1e-4  # what the relaxation is worth in the objective


def test_vectorised_figure_eight_equals_the_literal_restatement():
    orc = OracleRobot(KUKA_KIN)
    for T in (5, 9):
        a, b = FigureEightNLP(orc, LINK, T=T), FastFigureEightNLP(orc, LINK, T=T)
        rng = np.random.default_rng(T)
        qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0]) + rng.uniform(-0.1, 0.1, 7)
        x = a.seed(qc) + 0.3 * rng.standard_normal(a.nx)
        lam = rng.standard_normal(a.nh)
        assert abs(a.f(x, qc) - b.f(x, qc)) <= 1e-12 * abs(a.f(x, qc))
        assert np.abs(a.df(x, qc) - b.df(x, qc)).max() <= 1e-12 * np.abs(a.df(x, qc)).max()
        assert np.abs(a.h(x, qc) - b.h(x, qc)).max() <= 1e-15 and np.abs(a.dh(x, qc) - b.dh(x, qc)).max() <= 1e-15
        for gn in (False, True):
            Ha, Hb = a.hess_lagrangian(x, qc, lam, gn), b.hess_lagrangian(x, qc, lam, gn)
            assert np.abs(Ha - Hb).max() <= 1e-12 * np.abs(Ha).max()


def test_interior_point_answers_on_configs_4_and_5_match_the_other_solvers():
    """tests/golden/ipm_configs_golden.npz (tools/make_golden.py --ipm-configs, ~1 h of CPU): oracle/ipm_reference_form.py from the reference's
    seeds on config 4 as shipped (dual_arm.py, 1386 variables, zero seed) and on config 5 at T = 6 (168 variables, 336 rows, with and without
    binding effort rows).  Its optima against what the other independent solvers found for the same instances: the dense-SQP / SLSQP-wired
    answer of config 4 (tests/test_dual_arm.py's 0.00480191855905) and the trust-constr / L-BFGS-B / SLSQP answers of torque_golden.npz.
    The interior-point objective sits sum|lam| x 1e-8 below the exactly feasible optimum (IPOPT's bound_relax_factor lets every row end 1e-8
    below zero): 1e-6 relative on config 5 (multipliers of the dynamics rows ~ 50 N m), 2e-8 on config 4."""
    import os

    from conftest import GOLDEN

    g = np.load(os.path.join(GOLDEN, "ipm_configs_golden.npz"))
    t = np.load(os.path.join(GOLDEN, "torque_golden.npz"))
    assert bool(g["dual_optimal"]) and int(g["dual_iters"]) < 100
    assert abs(float(g["dual_f"]) - 0.00480191855905) <= 1e-9  # 1386-variable dual_arm.py as shipped, from the zero seed
    for tag in ("t6", "t6lim"):
        if f"tq_{tag}_f" not in g.files:
            continue
        f_ipm, f_other = g[f"tq_{tag}_f"], t[tag + "_f"][: len(g[f"tq_{tag}_f"])]
        assert np.all(f_ipm <= f_other + 1e-9) and np.all(f_other - f_ipm <= 2e-6 * f_other), (tag, f_ipm, f_other)
        assert np.all(g[f"tq_{tag}_iters"] < 500)
    assert "tq_t6lim_f" in g.files


def test_velocity_limited_goldens_against_the_numpy_ports():
    """tests/golden/ipm_limits_golden.npz (tools/make_golden.py --ipm-limits): the interior-point answers for the velocity-limited figure-eight and
    dual arm against the numpy ports of the kernels' state machines (oracle/structured.py, oracle/guarded.py): two solvers that share neither
    algorithm nor formulation (reference layout with the (e, -e) rows and slack bounds against eliminated velocities on the manifold)."""
    import os

    from conftest import GOLDEN, KUKA_KIN
    from oracle.guarded import Guards, solve_free_al
    from oracle.problems import dual_arm_offsets
    from oracle.robot import OracleRobot
    from oracle.structured import FoldedChain, StructuredFigureEight, solve_structured_lm

    g = np.load(os.path.join(GOLDEN, "ipm_limits_golden.npz"))
    kuka = OracleRobot(KUKA_KIN)
    prob = StructuredFigureEight(kuka, "end_effector_ball", T=50)
    vl = g["fig8v_vl"]
    assert g["fig8v_ok"].all() and (g["fig8v_iters"] < 200).all()
    n_same = 0
    for qc, f_ipm in zip(g["fig8v_qc"], g["fig8v_f"]):
        s = solve_structured_lm(prob, qc, max_iter=600, tol=1e-7, vlimits=(-vl, vl))
        assert s["status"] == 0
        n_same += abs(s["f"] - f_ipm) <= 2e-5  # IPOPT's bound relaxation: sum|lam| x 1e-8
        assert s["f"] <= f_ipm + 2e-5
    assert n_same >= 3
    if "dualv_f" in g.files:
        vmax, T, p = float(g["dualv_vmax"]), 50, g["dualv_p"]
        f = 0.0
        for arm, y, qc in (("l", -0.25, p[:7]), ("r", 0.25, p[7:])):
            rob = OracleRobot(KUKA_KIN, name="kuka" + arm)
            rob.add_base_frame("global_world", xyz=[0.0, y, 0.0])
            s = solve_free_al(FoldedChain(rob, "end_effector_ball"), T, 10.0 / (T - 1), dual_arm_offsets(T)[arm].T, qc, Guards(), Q0=np.zeros((T, 7)), rho0=10.0,
                              exact=False, vlimits=(-np.full(7, vmax), np.full(7, vmax)), max_iter=600, tol=1e-7)
            assert s["status"] == 0
            f += s["f"]
        assert bool(g["dualv_ok"]) and abs(f - float(g["dualv_f"])) <= 1e-6, (f, float(g["dualv_f"]))
    if "tqv_f" in g.files:  # config 5 at T = 6 with 0.25 rad/s next to the effort limits (tools/make_golden.py --ipm-torque-velocity)
        from conftest import MED7_KIN
        from oracle.torque import TorqueProblem, solve_torque_lm

        vmax = float(g["tqv_vmax"])
        prob = TorqueProblem(OracleRobot(MED7_KIN), "lbr_link_ee", T=6, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=float(g["tqv_lim"]))
        s = solve_torque_lm(prob, g["tqv_qc"], np.zeros(7), g["tqv_goal"], vlimits=(-vmax, vmax), max_iter=600)
        assert s["status"] == 0 and bool(g["tqv_ok"]) and np.abs(s["dQ"]).max() <= vmax + 1e-8
        assert s["f"] >= float(g["tqv_f"]) - 1e-9 and s["f"] - float(g["tqv_f"]) <= 5e-6 * s["f"], (s["f"], float(g["tqv_f"]))  # below by the bound relaxation


def test_interior_point_endpoints_of_config5_at_T30_are_kkt_points():
    """tests/golden/ipm_configs_golden.npz, tq_t30* (tools/make_golden.py --ipm-configs t30 t30lim, ~45 minutes: oracle/ipm_reference_form.py with
    the exact Lagrangian Hessian of oracle/problems.py:TorqueMPCNLP on BASELINE config 5 at its stated size, from the reference's seed).  The stored
    endpoints are graded here on the literal layout: all five are KKT points of the reference's NLP to the method's own relaxation (rows of v 1e-8
    below zero at most); ONE has the objective of the kernels' optimum, four are other local minima with objectives 5 .. 7 x higher -- recorded, not
    asserted away (tests/test_gpu_ipm_parity.py compares the GPU with them)."""
    from oracle.torque import TorqueProblem

    g, gi = np.load(os.path.join(GOLDEN, "torque_golden.npz")), np.load(os.path.join(GOLDEN, "ipm_configs_golden.npz"))
    if "tq_t30_x" not in gi.files:
        pytest.skip("T = 30 interior-point goldens not generated")
    med7 = OracleRobot(MED7_KIN)
    same = 0
    for tag in ("t30", "t30lim"):
        lim = float(g[tag + "_lim"])
        nlp = TorqueMPCNLP(TorqueProblem(med7, "lbr_link_ee", T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=None if lim > 1e8 else lim))
        for b in range(len(g[tag + "_qc"])):
            x, p = gi[f"tq_{tag}_x"][b], nlp.pack_p(g[tag + "_qc"][b], np.zeros(7), g[tag + "_goal"][b])
            assert abs(nlp.f(x, p) - float(gi[f"tq_{tag}_f"][b])) <= 1e-9 * nlp.f(x, p)
            assert np.abs(nlp.a(x, p)).max() <= 1.01e-8 and np.abs(nlp.h(x, p)).max() <= 1e-7 and nlp.k(x, p).min() >= -1.01e-8
            k = kkt_reference_form(nlp, x, p, active_tol=1e-6)
            assert k["stationarity"] <= 1e-4 and k["feasibility"] <= 1e-7, (tag, b, k["stationarity"], k["feasibility"])
            same += abs(float(gi[f"tq_{tag}_f"][b]) - float(g[tag + "_f"][b])) <= 1e-5 * float(g[tag + "_f"][b])
    assert same == 1


def test_what_an_interior_point_run_capped_at_ipopts_default_3000_iterations_returns():
    """Round-3 verdict, Next 2(ii): `setup("ipopt")` with default options stops at max_iter = 3000.  tools/make_golden.py --ipm-cap3000 re-ran the
    instances of nlp_ipm_golden.npz that needed more: 5 of the 17 distinct T = 50 instances stop at the cap with status "max_iter" -- what IPOPT
    reports as "Maximum Number of Iterations Exceeded", i.e. did_solve() == False in the reference (solver.py:407-412) -- at objectives 0.001 ..
    0.15 above the optimum the kernels return (1e-4 .. 2e-2 relative) and optimality errors E_0 between 0.15 and 1e4.  Recorded, not an assertion
    about IPOPT itself (this is a restatement of its published algorithm; the real code cannot run here)."""
    g = np.load(os.path.join(GOLDEN, "nlp_ipm_golden.npz"))
    if "f_cap3000" not in g.files:
        pytest.skip("capped runs not generated")
    capped = ~np.isnan(g["f_cap3000"])
    distinct = {tuple(np.round(q, 12)) for q in g["qc"][capped]}
    assert len(distinct) == 5 and (g["iters"][capped] > 3000).all() and (g["status_cap3000"][capped] == "max_iter").all()
    assert (g["E0_cap3000"][capped] > 0.1).all()  # nowhere near tol = 1e-8, nor acceptable_tol = 1e-6
    gap = g["f_cap3000"][capped] - g["f_struct"][capped]
    assert (gap > 1e-4).all() and (gap < 0.2).all()  # feasible-ish points above the optimum of the kernels
