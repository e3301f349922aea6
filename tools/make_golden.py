"""Generate the committed golden fixtures under tests/golden/ from the CPU oracle (run in the build
container; the GPU box only reads the .npz files).

  python tools/make_golden.py

Sources of truth: scipy.spatial.transform.Rotation for the SE(3)/quaternion primitives (the
reference's own test oracle, tests/test_spatialmath.py), the literal numpy restatement in oracle/ for
FK / Jacobians, and oracle.solvers for the NLP solutions (dense_sqp on the literal reference layout,
scipy SLSQP wired like ScipyMinimizeSolver).  Seeds are fixed; rerunning reproduces the files.
"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation as Rot

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.problems import FigureEightNLP, IKExampleNLP  # noqa: E402
from oracle.robot import OracleRobot  # noqa: E402
from oracle.solvers import dense_sqp, kkt_reference_form, scipy_minimize  # noqa: E402
from oracle.structured import StructuredFigureEight, solve_structured_lm  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
SEED = 20260927


def spatialmath_golden():
    rng = np.random.default_rng(SEED)
    n = 64
    theta = rng.uniform(-np.pi, np.pi, n)
    axis = rng.uniform(-1, 1, (n, 3))
    rpy = rng.uniform(-np.pi, np.pi, (n, 3))
    rpy[:, 1] = rng.uniform(-0.49 * np.pi, 0.49 * np.pi, n)
    out = {
        "theta": theta,
        "axis": axis,
        "rpy": rpy,
        "angvec2r": np.stack([Rot.from_rotvec(theta[i] * axis[i] / np.linalg.norm(axis[i])).as_matrix() for i in range(n)]),
        "rotx": np.stack([Rot.from_euler("x", t).as_matrix() for t in theta]),
        "roty": np.stack([Rot.from_euler("y", t).as_matrix() for t in theta]),
        "rotz": np.stack([Rot.from_euler("z", t).as_matrix() for t in theta]),
        # URDF fixed-axis roll-pitch-yaw == scipy extrinsic "xyz"
        "rpy2r": np.stack([Rot.from_euler("xyz", r).as_matrix() for r in rpy]),
        "quat_fromrpy": np.stack([Rot.from_euler("xyz", r).as_quat() for r in rpy]),
        "quat_fromangvec": np.stack([Rot.from_rotvec(theta[i] * axis[i] / np.linalg.norm(axis[i])).as_quat() for i in range(n)]),
    }
    np.savez(os.path.join(G, "spatialmath_golden.npz"), **out)


def fk_golden():
    rng = np.random.default_rng(SEED + 1)
    out = {}
    cases = [
        ("kuka_lwr", os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"), "end_effector_ball"),
        ("kuka_lwr_mid", os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"), "lwr_arm_4_link"),
        ("med7", os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"), "lbr_link_ee"),
        ("tester", os.path.join(G, "tester_robot.kin.json"), "eff"),
    ]
    for tag, kin, link in cases:
        r = OracleRobot(kin)
        n = 48
        lo = np.maximum(r.lower_actuated_joint_limits, -3.0)
        up = np.minimum(r.upper_actuated_joint_limits, 3.0)
        Q = rng.uniform(lo, up, (n, r.ndof))
        pose = np.zeros((n, 7))
        J = np.zeros((n, 6, r.ndof))
        for i in range(n):
            pose[i, :3] = r.get_global_link_position(link, Q[i])
            pose[i, 3:] = r.get_global_link_quaternion(link, Q[i])
            J[i] = r.get_global_link_geometric_jacobian(link, Q[i])
        out[f"{tag}_q"], out[f"{tag}_pose"], out[f"{tag}_J"] = Q, pose, J
        out[f"{tag}_link"] = np.array(link)
    np.savez(os.path.join(G, "fk_golden.npz"), **out)


def nlp_golden():
    kuka = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    link = "end_effector_ball"
    out = {}
    # config 1 (example/example.py): scipy SLSQP, reference wiring, seed = zeros (the script's seed key
    # "kuka/q" does not exist in the container, so dict2vec zero-fills: sx_container.py:121)
    ik = IKExampleNLP(kuka, link)
    qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    pg = kuka.get_global_link_position(link, qn) + np.array([0.0, 0.3, -0.2])
    p = np.concatenate([qn, pg])
    r = scipy_minimize(ik, np.zeros(7), p, method="SLSQP", tol=1e-12, options={"maxiter": 500})
    k = kkt_reference_form(ik, r.x, p)
    print("ik", r.fun, r.nit, k["stationarity"], k["feasibility"])
    out["ik_p"], out["ik_x"], out["ik_f"] = p, r.x, r.fun
    # config 2 (figure_eight_plan.py), nominal qc: independent dense SQP on the literal layout
    T = 50
    nlp = FigureEightNLP(kuka, link, T=T)
    qc0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    t0 = time.time()
    d = dense_sqp(nlp, nlp.seed(qc0), qc0, tol=1e-11)
    k = kkt_reference_form(nlp, d["x"], qc0)
    print("fig8 dense", d["f"], d["iters"], d["converged"], k["stationarity"], k["feasibility"], time.time() - t0)
    out["fig8_qc"], out["fig8_x"], out["fig8_f"] = qc0, d["x"], d["f"]
    out["fig8_mu_h"] = k["mu_h"]
    # perturbed instances (qc + U(-0.1,0.1)^7, rng seed 20260927): structured oracle,
    # each verified in reference form
    rng = np.random.default_rng(SEED)
    prob = StructuredFigureEight(kuka, link, T=T)
    qcs, xs, fs = [], [], []
    for i in range(6):
        qc = qc0 + rng.uniform(-0.1, 0.1, 7)
        s = solve_structured_lm(prob, qc, max_iter=400, tol=1e-9)
        Q = s["Q"]
        x = nlp.join(Q.T, (np.diff(Q, axis=0) / nlp.dt).T)
        k = kkt_reference_form(nlp, x, qc)
        print("fig8 pert", i, s["f"], s["iters"], s["stat"], k["stationarity"], k["feasibility"])
        qcs.append(qc); xs.append(x); fs.append(nlp.f(x, qc))
    out["fig8_pert_qc"], out["fig8_pert_x"], out["fig8_pert_f"] = np.array(qcs), np.array(xs), np.array(fs)
    # small horizon for quick parity
    for Ts in (5, 12):
        nl = FigureEightNLP(kuka, link, T=Ts, Tmax=10.0 * (Ts - 1) / 49.0)
        d = dense_sqp(nl, nl.seed(qc0), qc0, tol=1e-11)
        print("fig8 T", Ts, d["f"], d["iters"], d["converged"])
        out[f"fig8_T{Ts}_x"], out[f"fig8_T{Ts}_f"] = d["x"], d["f"]
    np.savez(os.path.join(G, "nlp_golden.npz"), **out)




def pm_golden():
    """BASELINE config 3 (point_mass_mpc.py): one MPC tick + 8 random initial states, reference wiring (scipy SLSQP on
    v >= 0) and the IPM port; both must agree."""
    from oracle.pointmass_ipm import solve_pointmass_ipm
    from oracle.problems import PointMassMPCNLP, point_mass_tick_parameters

    nlp = PointMassMPCNLP()
    rng = np.random.default_rng(SEED)
    P = [point_mass_tick_parameters()]
    _, _, _, obs = nlp.split_p(P[0])
    for _ in range(8):
        while True:
            c = rng.uniform(-1.2, 1.2, 2)
            if np.linalg.norm(c - obs[:, 0]) > 0.35:
                break
        goal = np.stack([np.clip(c[j] + (1 - c[j]) * np.arange(20) / 19.0, -1.5, 1.5) for j in range(2)])
        P.append(PointMassMPCNLP.pack_p(c, np.zeros(2), goal, obs))
    X, F = [], []
    for p in P:
        r = scipy_minimize(nlp, np.zeros(nlp.nx), p, method="SLSQP", tol=1e-13, options={"maxiter": 1000})
        curr, dcurr, goal, ob = nlp.split_p(p)
        i = solve_pointmass_ipm(20, 0.05, nlp.w, 1.5, 1.0, nlp.safe_sq, curr, dcurr, goal, ob, tol=1e-9)
        k = kkt_reference_form(nlp, r.x, p)
        print("pm", r.fun, r.nit, r.success, i["f"], i["iters"], abs(r.fun - i["f"]), k["stationarity"])
        best = r.x if (r.success and r.fun <= i["f"] + 1e-7) else np.concatenate([i["Y"].T.reshape(-1), i["V"].T.reshape(-1)])
        X.append(best)
        F.append(nlp.f(best, p))
    np.savez(os.path.join(G, "pm_golden.npz"), p=np.array(P), x=np.array(X), f=np.array(F))


def ipm_pm_golden():
    """oracle/ipm_reference_form.py (the reference's algorithm class on the literal 264-row form, zero seed as the script's first tick has it) on the
    nine point-mass ticks of pm_golden.npz -> tests/golden/ipm_pm_golden.npz (f, x, iterations, status); seconds.  The GPU test compares the kernel's
    answers with these directly (round-3 verdict, Missing 3: config 3 had interior-point runs on the CPU side only)."""
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import PointMassMPCNLP

    g = np.load(os.path.join(G, "pm_golden.npz"))
    nlp = PointMassMPCNLP()
    X, F, IT, OK = [], [], [], []
    for i, p in enumerate(g["p"]):
        r = solve_ipm(nlp, np.zeros(nlp.nx), p)
        print("pm ipm", i, r["status"], r["iters"], r["f"], "golden", g["f"][i], flush=True)
        X.append(r["x"]); F.append(r["f"]); IT.append(r["iters"]); OK.append(r["status"] == "optimal")
    np.savez(os.path.join(G, "ipm_pm_golden.npz"), p=g["p"], x=np.array(X), f=np.array(F), iters=np.array(IT), optimal=np.array(OK))


def ik_golden():
    """BASELINE config 1 (example/example.py): the script's instance (zero seed, see examples/example.py) + 23 random
    instances, 8 of them with the nominal configuration pushed against joint limits so that bound rows are active.
    Reference wiring (scipy SLSQP on v >= 0, solver.py:652-679) and the augmented-Lagrangian port; stored optimum = the
    one both agree on (agree=1) or the lower-cost KKT point of the two (agree=0; the problem is non-convex)."""
    from oracle.ik_al import solve_ik_al
    from oracle.structured import FoldedChain

    kuka = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    link = "end_effector_ball"
    ik = IKExampleNLP(kuka, link)
    ch = FoldedChain(kuka, link)
    rng = np.random.default_rng(SEED + 1)
    qn0 = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    P = [np.concatenate([qn0, kuka.get_global_link_position(link, qn0) + np.array([0.0, 0.3, -0.2])])]
    X0 = [np.zeros(7)]
    for i in range(23):
        qn = qn0 + rng.uniform(-0.3, 0.3, 7)
        if i >= 15:
            j = rng.integers(0, 7, 2)
            s = rng.choice([-1.0, 1.0], 2)
            qn[j] = np.where(s > 0, ik.up[j] - 0.02, ik.lo[j] + 0.02)
        pg = kuka.get_global_link_position(link, qn) + rng.uniform(-0.2, 0.2, 3)
        P.append(np.concatenate([qn, pg]))
        X0.append(qn.copy())
    X, F, AG, NACT = [], [], [], []
    for p, x0 in zip(P, X0):
        r = scipy_minimize(ik, x0, p, method="SLSQP", tol=1e-13, options={"maxiter": 1000})
        a = solve_ik_al(ch, x0, p[:7], p[7:], ik.lo, ik.up, tol=1e-9, tol_feas=1e-11, max_iter=400)
        ks = kkt_reference_form(ik, r.x, p, active_tol=1e-7)
        agree = bool(r.success and np.abs(r.x - a["x"]).max() < 1e-6)
        slsqp_ok = r.success and ks["stationarity"] < 1e-6 and ks["feasibility"] < 1e-9
        best = a["x"] if (a["status"] == 0 and (not slsqp_ok or a["f"] <= r.fun + 1e-9)) else r.x
        nact = int(((best - ik.lo) < 1e-9).sum() + ((ik.up - best) < 1e-9).sum())
        print("ik", r.fun, r.success, a["f"], a["status"], a["iterations"], "agree", agree, "active", nact)
        X.append(best)
        F.append(ik.f(best, p))
        AG.append(agree)
        NACT.append(nact)
    np.savez(os.path.join(G, "ik_golden.npz"), p=np.array(P), x0=np.array(X0), x=np.array(X), f=np.array(F), agree=np.array(AG),
             nactive=np.array(NACT), lo=ik.lo, up=ik.up)


def guard_golden():
    """Synthetic config 4 (dual_arm.py + joint limits + sphere clearances), per arm: scipy SLSQP on the
    reduced problem (q_0 eliminated, dq condensed; constraints g >= 0 as the reference's ScipyMinimizeSolver passes them,
    solver.py:672-679) against the augmented-Lagrangian port.  T = 20 both arms (qcr perturbed), T = 50 left arm."""
    from scipy.optimize import minimize

    from oracle.guarded import Guards, guard_values, solve_free_al
    from oracle.problems import dual_arm_offsets
    from oracle.structured import FoldedChain

    kin = os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json")
    links = ["end_effector_ball", "lwr_arm_7_link", "lwr_arm_5_link", "lwr_arm_6_link"]
    obs = np.array([[0.55, 0.0, 0.1 * (i + 1)] for i in range(6)])
    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    out = {"links": np.array(links), "obs": obs, "link_radius": 0.15, "obs_radius": 0.1}
    for tag, T, arm, y, qc in (("T20l", 20, "l", -0.25, QC), ("T20r", 20, "r", 0.25, QC + 0.02), ("T50l", 50, "l", -0.25, QC)):
        rob = OracleRobot(kin, name="kuka" + arm)
        rob.add_base_frame("global_world", xyz=[0.0, y, 0.0])
        ch = FoldedChain(rob, "end_effector_ball")
        off = dual_arm_offsets(T)[arm].T
        dt = 10.0 / (T - 1)
        kap = 0.01 / dt**2
        GU = Guards(lo=rob.lower_actuated_joint_limits, up=rob.upper_actuated_joint_limits, links=links, link_radii=np.full(4, 0.15), obs_pos=obs,
                   obs_radii=np.full(6, 0.1))
        path = ch.fk(qc[None])[0][0] + off

        def unpack(x):
            return np.vstack([qc[None], x.reshape(T - 1, 7)])

        def fun(x):
            Q = unpack(x)
            e, _, Jp, _ = ch.jac(Q)
            r = path - e
            d = np.diff(Q, axis=0)
            g = -2 * np.einsum("tki,tk->ti", Jp, r)
            g[1:] += 2 * kap * d
            g[:-1] -= 2 * kap * d
            return np.sum(r * r) + kap * np.sum(d * d), g[1:].reshape(-1)

        def con(x):
            return guard_values(ch, unpack(x), GU)[0][1:].reshape(-1)

        def jac(x):
            dg = guard_values(ch, unpack(x), GU)[1][1:]
            J = np.zeros((T - 1, dg.shape[1], T - 1, 7))
            for t in range(T - 1):
                J[t, :, t, :] = dg[t]
            return J.reshape((T - 1) * dg.shape[1], (T - 1) * 7)

        t0 = time.time()
        r = minimize(fun, np.tile(qc, (T - 1, 1)).reshape(-1), jac=True, method="SLSQP", constraints=[{"type": "ineq", "fun": con, "jac": jac}],
                     tol=1e-13, options={"maxiter": 1000})
        a = solve_free_al(ch, T, dt, off, qc, GU, Q0=np.tile(qc, (T, 1)), rho0=10.0, exact=False, tol=1e-9, tol_feas=1e-11, max_iter=600)
        Qs = unpack(r.x)
        print("guard", tag, r.fun, r.nit, r.success, a["f"], a["iters"], a["status"], "dQ", np.abs(Qs - a["Q"]).max(), "active", int((a["lam"] > 0).sum()),
              round(time.time() - t0, 1))
        assert r.success and a["status"] == 0 and abs(r.fun - a["f"]) < 1e-8
        out[tag + "_qc"], out[tag + "_Q"], out[tag + "_f"], out[tag + "_lam"], out[tag + "_Q_slsqp"] = qc, a["Q"], a["f"], a["lam"], Qs
    np.savez(os.path.join(G, "guard_golden.npz"), **out)


def torque_golden(slow=True):
    """BASELINE configs[4] (torque MPC, med7): optima of the numpy port (oracle/torque.py) next to two solvers that share nothing with it but the
    literal functions: scipy L-BFGS-B on the reduced problem in ddq (no effort rows active) and scipy trust-constr wired like the reference's
    ScipyMinimizeSolver (solver.py:680-712: k, a, g, h passed separately) on the literal 28T-variable layout.  trust-constr needs 6-8 minutes per
    T = 6 instance, so it only runs here."""
    import scipy.optimize

    from oracle.problems import TorqueMPCNLP
    from oracle.torque import TorqueProblem, costate_gradient, rnea_batch, rnea_jacobian, solve_torque_lm

    rob = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    link = "lbr_link_ee"
    rng = np.random.default_rng(SEED + 7)
    qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    out = {}

    def lbfgs(prob, qc, goal):
        T = prob.T

        def F(u):
            U = u.reshape(T, 7)
            Q, dQ = prob.rollout(qc, np.zeros(7), U)
            tau = rnea_batch(prob.tb, Q, dQ, U)
            J = rnea_jacobian(prob.tb, Q, dQ, U)
            e, _, Jp, _ = prob.chain.jac(Q)
            rr = e - goal
            f = prob.w_path * np.sum(rr * rr) + prob.w_vel * np.sum(dQ * dQ) + prob.w_tau * np.sum(tau * tau)
            g = np.einsum("ti,tid->td", 2 * prob.w_tau * tau, J)
            g[:, :7] += 2 * prob.w_path * np.einsum("tki,tk->ti", Jp, rr)
            g[:, 7:14] += 2 * prob.w_vel * dQ
            return f, costate_gradient(g, prob.dt).reshape(-1)

        r = scipy.optimize.minimize(F, np.zeros(T * 7), jac=True, method="L-BFGS-B",
                                    options={"maxiter": 40000, "maxfun": 80000, "ftol": 1e-15, "gtol": 1e-8, "maxcor": 30})
        return float(r.fun)

    cases = [("t6", 6, None, 1), ("t6lim", 6, 55.0, 1), ("t30", 30, None, 3), ("t30lim", 30, 55.0, 2)]
    for tag, T, lim, n in cases:
        prob = TorqueProblem(rob, link, T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=lim)
        nlp = TorqueMPCNLP(prob)
        QC, GOAL, X, F_, IT, FL, FT, LAM = [], [], [], [], [], [], [], []
        for i in range(n):
            qc = qn + (rng.uniform(-0.1, 0.1, 7) if i else 0.0)
            goal = prob.goal_figure_eight(qc)
            t0 = time.time()
            r = solve_torque_ipm(prob, qc, np.zeros(7), goal)
            assert r["status"] == 0
            al = solve_torque_lm(prob, qc, np.zeros(7), goal)
            assert al["status"] == 0 and abs(al["f"] - r["f"]) < 1e-8 * r["f"], (al["f"], r["f"])
            g.setdefault(tag + "_f_al", np.zeros(len(g[tag + "_qc"])))[i] = al["f"]
            x = nlp.join(r["Q"], r["dQ"], r["U"], r["tau"])
            k = kkt_reference_form(nlp, x, nlp.pack_p(qc, np.zeros(7), goal))
            assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-8, k
            fl = lbfgs(prob, qc, goal) if lim is None else np.nan
            ft = np.nan
            if T == 6 and slow:
                rt = scipy_minimize(nlp, nlp.seed(qc), nlp.pack_p(qc, np.zeros(7), goal), method="trust-constr", tol=1e-9, options={"maxiter": 3000})
                ft = float(rt.fun)
            print("torque", tag, i, "port", r["f"], r["iters"], "lbfgs", fl, "trust-constr", ft, "kkt", k["stationarity"], round(time.time() - t0, 1))
            if lim is None:
                assert abs(fl - r["f"]) < 1e-8 * max(1.0, r["f"])
            if np.isfinite(ft):
                assert abs(ft - r["f"]) < 1e-7 * max(1.0, r["f"])
            QC.append(qc); GOAL.append(goal); X.append(x); F_.append(r["f"]); IT.append(r["iters"]); FL.append(fl); FT.append(ft); LAM.append(r["lam"])
        out[tag + "_qc"], out[tag + "_goal"], out[tag + "_x"], out[tag + "_f"], out[tag + "_iters"] = np.stack(QC), np.stack(GOAL), np.stack(X), np.array(F_), np.array(IT)
        out[tag + "_f_lbfgs"], out[tag + "_f_trust_constr"], out[tag + "_lam"] = np.array(FL), np.array(FT), np.stack(LAM)
        out[tag + "_lim"] = np.array(1e9 if lim is None else lim)
    np.savez(os.path.join(G, "torque_golden.npz"), **out)



def torque_slsqp_reduced(prob, qc, goal, maxiter=4000):
    """Independent pin for the torque problem WITH active effort rows (L-BFGS-B cannot take them, trust-constr on the literal layout needs
    hours at T = 30): scipy SLSQP on the problem reduced to the control sequence U (states rolled out, tau = rnea(q(U), dq(U), U)), the 2 T n
    effort rows as nonlinear inequalities with their exact Jacobian (chain rule through the Euler roll-out).  Shares the literal functions
    with the port, not the algorithm (dense SQP vs. augmented Lagrangian + Riccati)."""
    import scipy.optimize

    from oracle.torque import costate_gradient, rnea_batch, rnea_jacobian

    T, n, dt = prob.T, prob.n, prob.dt

    def parts(u):
        U = u.reshape(T, n)
        Q, dQ = prob.rollout(qc, np.zeros(n), U)
        return U, Q, dQ, rnea_batch(prob.tb, Q, dQ, U), rnea_jacobian(prob.tb, Q, dQ, U)

    def F(u):
        U, Q, dQ, tau, J = parts(u)
        e, _, Jp, _ = prob.chain.jac(Q)
        rr = e - goal
        f = prob.w_path * np.sum(rr * rr) + prob.w_vel * np.sum(dQ * dQ) + prob.w_tau * np.sum(tau * tau)
        g = np.einsum("ti,tid->td", 2 * prob.w_tau * tau, J)
        g[:, :n] += 2 * prob.w_path * np.einsum("tki,tk->ti", Jp, rr)
        g[:, n : 2 * n] += 2 * prob.w_vel * dQ
        return f, costate_gradient(g, dt).reshape(-1)

    def tau_and_jac(u):
        U, Q, dQ, tau, J = parts(u)
        Jt = np.zeros((T, n, T, n))
        for t in range(T):
            Jt[t, :, t, :] = J[t][:, 2 * n :]
            for s in range(t):  # q_t = ... + (t - 1 - s) dt^2 u_s, dq_t = ... + dt u_s
                Jt[t, :, s, :] = J[t][:, :n] * ((t - 1 - s) * dt * dt) + J[t][:, n : 2 * n] * dt
        return tau.reshape(-1), Jt.reshape(T * n, T * n)

    lo, up = np.tile(prob.tau_lo, T), np.tile(prob.tau_up, T)
    cons = [{"type": "ineq", "fun": lambda u: np.concatenate([tau_and_jac(u)[0] - lo, up - tau_and_jac(u)[0]]),
             "jac": lambda u: np.concatenate([tau_and_jac(u)[1], -tau_and_jac(u)[1]])}]
    r = scipy.optimize.minimize(F, np.zeros(T * n), jac=True, method="SLSQP", constraints=cons, options={"maxiter": maxiter, "ftol": 1e-14})
    return float(r.fun), int(r.status), int(r.nit), float(np.abs(tau_and_jac(r.x)[0]).max())


def torque_golden_add_slsqp():
    """t30lim_f_slsqp for the stored instances with active effort rows (~45 s each)."""
    from oracle.torque import TorqueProblem

    rob = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    path = os.path.join(G, "torque_golden.npz")
    g = dict(np.load(path))
    for tag in ("t6lim", "t30lim"):
        lim = float(g[tag + "_lim"])
        T = g[tag + "_goal"].shape[1]
        prob = TorqueProblem(rob, "lbr_link_ee", T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=lim)
        out = []
        for i, (qc, goal) in enumerate(zip(g[tag + "_qc"], g[tag + "_goal"])):
            f, status, nit, tmax = torque_slsqp_reduced(prob, qc, goal)
            print("torque", tag, i, "slsqp", f, "status", status, "iterations", nit, "port", float(g[tag + "_f"][i]), "max |tau|", tmax)
            assert status == 0 and abs(f - float(g[tag + "_f"][i])) < 1e-7 * f and tmax > lim - 1e-6  # converged, agrees, rows active
            out.append(f)
        g[tag + "_f_slsqp"] = np.array(out)
    np.savez(path, **g)


def torque_golden_refresh_port():
    """Re-run only the numpy port on the stored instances of torque_golden.npz (after a change to its state machine): x, f, iters, lam are
    replaced, the independent solvers' objectives (L-BFGS-B, trust-constr: ~20 minutes to regenerate) are kept and must still agree.
    Round 4: the port is the interior point of oracle/torque_ipm.py; the augmented-Lagrangian machine of rounds 1-3 (oracle/torque.py:solve_torque_lm)
    joins the independent solvers as "_f_al"."""
    from oracle.problems import TorqueMPCNLP
    from oracle.torque import TorqueProblem, solve_torque_lm
    from oracle.torque_ipm import solve_torque_ipm

    rob = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    path = os.path.join(G, "torque_golden.npz")
    g = dict(np.load(path))
    for tag in ("t6", "t6lim", "t30", "t30lim"):
        lim = float(g[tag + "_lim"])
        T = g[tag + "_goal"].shape[1]
        prob = TorqueProblem(rob, "lbr_link_ee", T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=None if lim > 1e8 else lim)
        nlp = TorqueMPCNLP(prob)
        for i, (qc, goal) in enumerate(zip(g[tag + "_qc"], g[tag + "_goal"])):
            r = solve_torque_ipm(prob, qc, np.zeros(7), goal)
            assert r["status"] == 0
            al = solve_torque_lm(prob, qc, np.zeros(7), goal)
            assert al["status"] == 0 and abs(al["f"] - r["f"]) < 1e-8 * r["f"], (al["f"], r["f"])
            g.setdefault(tag + "_f_al", np.zeros(len(g[tag + "_qc"])))[i] = al["f"]
            x = nlp.join(r["Q"], r["dQ"], r["U"], r["tau"])
            k = kkt_reference_form(nlp, x, nlp.pack_p(qc, np.zeros(7), goal))
            assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-8, k
            for other in ("_f_lbfgs", "_f_trust_constr", "_f_slsqp"):
                if tag + other not in g:
                    continue
                fo = float(g[tag + other][i])
                if np.isfinite(fo):
                    assert abs(fo - r["f"]) < 1e-7 * max(1.0, r["f"]), (tag, i, other, fo, r["f"])
            print("torque", tag, i, "port", r["f"], "was", float(g[tag + "_f"][i]), "iters", r["iters"], "was", int(g[tag + "_iters"][i]), "kkt", k["stationarity"])
            g[tag + "_x"][i], g[tag + "_f"][i], g[tag + "_iters"][i], g[tag + "_lam"][i] = x, r["f"], r["iters"], r["lam"][:, :14]
    np.savez(path, **g)


def fig8_perturbed_dense_golden(n=8):
    """Config 2, PERTURBED instances (the bench workload: qc0 + U(-0.1, 0.1)^7) solved by the independent dense Newton-SQP on the literal
    693-variable layout with the literal rank-3 quaternion rows (oracle.solvers.dense_sqp: SVD null space, exact Lagrangian Hessian, l1 merit
    -- it shares no structure with the Riccati / retraction path).  From the reference's seed it may settle in another local minimum than the
    structured solver (the problem is nonconvex); therefore two runs per instance: (a) from the seed, (b) from the structured optimum displaced
    by 1e-3 -- (b) must come back to the same point, which pins the structured optimum as a strict local minimum under an independent
    algorithm; where (a) reaches the same basin the objective is pinned from the seed as well.  ~2-4 minutes per instance."""
    kuka = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    link, T = "end_effector_ball", 50
    nlp = FigureEightNLP(kuka, link, T=T)
    prob = StructuredFigureEight(kuka, link, T=T)
    qc0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    rng = np.random.default_rng(SEED)  # the same draws as fig8_pert_* of nlp_golden.npz (first 6) and two more
    out = {k: [] for k in ("qc", "x_struct", "f_struct", "f_dense_seed", "f_dense_near", "dx_near", "same_basin", "kkt_dense_near")}
    for i in range(n):
        qc = qc0 + rng.uniform(-0.1, 0.1, 7)
        t0 = time.time()
        s = solve_structured_lm(prob, qc, max_iter=400, tol=1e-9)
        xs = nlp.join(s["Q"].T, (np.diff(s["Q"], axis=0) / nlp.dt).T)
        da = dense_sqp(nlp, nlp.seed(qc), qc, tol=1e-10, max_iter=200)
        rs = np.random.default_rng(SEED + 100 + i)
        db_ = dense_sqp(nlp, xs + 1e-3 * rs.standard_normal(nlp.nx), qc, tol=1e-10, max_iter=100)
        same = abs(da["f"] - s["f"]) <= 1e-7 * max(1.0, s["f"])
        print("fig8 pert dense", i, "struct", s["f"], s["iters"], "| dense from seed", da["f"], da["iters"], da["converged"], "same basin" if same else "OTHER BASIN",
              "| dense from near", db_["f"], db_["iters"], db_["converged"], "dx", np.abs(db_["x"] - xs).max(), round(time.time() - t0, 1))
        assert db_["converged"] and abs(db_["f"] - s["f"]) <= 1e-8 * max(1.0, s["f"]), "the structured optimum is not where the dense SQP converges to"
        out["qc"].append(qc); out["x_struct"].append(xs); out["f_struct"].append(s["f"]); out["f_dense_seed"].append(da["f"] if da["converged"] else np.nan)
        out["f_dense_near"].append(db_["f"]); out["dx_near"].append(np.abs(db_["x"] - xs).max()); out["same_basin"].append(bool(same and da["converged"]))
        out["kkt_dense_near"].append(db_["kkt_stat"])
    np.savez(os.path.join(G, "nlp_pert_dense_golden.npz"), **{k: np.array(v) for k, v in out.items()})


def _ipm_one(args):
    """One instance of fig8_ipm_golden (runs in a worker process)."""
    i, qc, max_iter = args if len(args) == 3 else (*args, 1500)
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import FastFigureEightNLP

    kuka = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    link, T = "end_effector_ball", 50
    nlp = FastFigureEightNLP(kuka, link, T=T)
    prob = StructuredFigureEight(kuka, link, T=T)
    t0 = time.time()
    r = solve_ipm(nlp, nlp.seed(qc), qc, max_iter=max_iter)
    k = kkt_reference_form(nlp, r["x"], qc)
    d = dense_sqp(nlp, r["x"], qc, max_iter=10, tol=1e-10)
    kp = kkt_reference_form(nlp, d["x"], qc)
    s = solve_structured_lm(prob, qc, max_iter=400, tol=1e-9)
    xs = nlp.join(s["Q"].T, (np.diff(s["Q"], axis=0) / nlp.dt).T)
    same = abs(d["f"] - s["f"]) <= 1e-8 * max(1.0, s["f"])
    print(f"fig8 ipm {i:2d}: ipm f={r['f']:.10f} {r['status']} it={r['iters']} E0={r['E0']:.1e} | polished f={d['f']:.12f} ({d['iters']} steps, moved {np.abs(d['x'] - r['x']).max():.1e}) "
          f"| structured f={s['f']:.12f} | {'same basin' if same else 'OTHER BASIN'} | {time.time() - t0:.0f} s", flush=True)
    return (qc, r["x"], r["f"], r["iters"], r["E0"], r["status"] == "optimal", [k["stationarity"], k["feasibility"], k["complementarity"]], d["x"], d["f"],
            [kp["stationarity"], kp["feasibility"], kp["complementarity"]], s["f"], bool(same), np.abs(d["x"] - xs).max(), time.time() - t0)


def fig8_ipm_golden(n_bench=16, workers=6):
    """Config 2 solved by the reference's ALGORITHM CLASS on the reference's FORM, from the reference's SEED (round-2 verdict, Next 2):
    oracle/ipm_reference_form.py -- primal-dual interior point with filter line search (Waechter & Biegler 2006, IPOPT's defaults) on the
    literal `min f s.t. 0 <= v <= 1e10`, equalities as (e, -e) pairs, exact Lagrangian Hessian.  Instances: the nominal one, the 8 perturbed
    ones of nlp_pert_dense_golden.npz and the first `n_bench` of the bench workload (bench.make_inputs(., 0)).  Recorded per instance: where the
    interior-point method stops (x, f, iterations, its own error E_0, reference-form KKT residuals); the same point polished by Newton-SQP steps
    on the exact equalities (the bound relaxation 1e-8 of IPOPT lets each equality move by 1e-8, worth sum|lam| 1e-8 ~ 1e-5 in f: the polish
    removes that); the structured optimum (oracle/structured.py) from the same seed; whether the two are the same basin.  Slow instances crawl
    along the curved valley of the stiff tracking cost for hundreds of iterations (the walk the structured solver's second-order correction
    removes): up to 20 minutes each, hence one worker process per instance."""
    import multiprocessing as mp

    import bench

    qc0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    qcs = [qc0] + list(np.load(os.path.join(G, "nlp_pert_dense_golden.npz"))["qc"]) + list(bench.make_inputs(n_bench, 0)[1])
    keys = ("qc", "x_ipm", "f_ipm", "iters", "E0", "optimal", "kkt_ipm", "x_polished", "f_polished", "kkt_polished", "f_struct", "same_basin", "dx_struct", "seconds")
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # before the workers import numpy: one BLAS thread each
        os.environ[var] = "1"
    with mp.get_context("spawn").Pool(workers) as pool:
        rows = pool.map(_ipm_one, list(enumerate(qcs)), chunksize=1)
    np.savez(os.path.join(G, "nlp_ipm_golden.npz"), **{k2: np.array([row[j] for row in rows]) for j, k2 in enumerate(keys)})


def fig8_ipm_extend(max_iter=12000, workers=6):
    """Second pass of fig8_ipm_golden: the instances whose interior-point run was still crawling along the valley at 1500 iterations get
    `max_iter` (a few thousand iterations of ~0.1 s)."""
    import multiprocessing as mp

    path = os.path.join(G, "nlp_ipm_golden.npz")
    g = {k: v.copy() for k, v in np.load(path).items()}
    todo, seen = [], {}
    for i in np.where(~g["optimal"])[0]:
        key = g["qc"][i].tobytes()
        if key not in seen:
            seen[key] = i
            todo.append((int(i), g["qc"][i], max_iter))
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    with mp.get_context("spawn").Pool(workers) as pool:
        rows = pool.map(_ipm_one, todo, chunksize=1)
    keys = ("qc", "x_ipm", "f_ipm", "iters", "E0", "optimal", "kkt_ipm", "x_polished", "f_polished", "kkt_polished", "f_struct", "same_basin", "dx_struct", "seconds")
    for (i, qc, _), row in zip(todo, rows):
        for j in range(len(g["qc"])):
            if g["qc"][j].tobytes() == qc.tobytes():
                for k2, val in zip(keys, row):
                    g[k2][j] = val
    np.savez(path, **g)


def _ipm_cap_one(args):
    i, qc = args
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import FastFigureEightNLP

    kuka = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    nlp = FastFigureEightNLP(kuka, "end_effector_ball", T=50)
    t0 = time.time()
    r = solve_ipm(nlp, nlp.seed(qc), qc, max_iter=3000)
    print(f"fig8 ipm capped at 3000: instance {i}: {r['status']} it={r['iters']} f={r['f']:.8f} E0={r['E0']:.2e} {round(time.time() - t0)} s", flush=True)
    return i, r["f"], r["E0"], r["status"]


def fig8_ipm_cap3000(workers=5):
    """What `setup("ipopt")` with DEFAULT options would hand back (round-3 verdict, Next 2(ii)): IPOPT's max_iter is 3000.  The instances of
    nlp_ipm_golden.npz whose interior-point run needed more than that are re-run with the cap; objective, optimality error and status at the cap
    join the file as f_cap3000 / E0_cap3000 / status_cap3000 (NaN / "" where the run had finished before)."""
    import multiprocessing as mp

    path = os.path.join(G, "nlp_ipm_golden.npz")
    g = dict(np.load(path))
    todo, seen = [], set()
    for i in np.flatnonzero(g["iters"] > 3000):
        key = tuple(np.round(g["qc"][i], 12))
        if key not in seen:
            seen.add(key)
            todo.append((int(i), g["qc"][i]))
    with mp.Pool(workers) as pool:
        rows = pool.map(_ipm_cap_one, todo, chunksize=1)
    f, e0, st = np.full(len(g["iters"]), np.nan), np.full(len(g["iters"]), np.nan), np.array([""] * len(g["iters"]), dtype="U16")
    for i, fv, ev, sv in rows:
        same = np.flatnonzero(np.all(np.abs(g["qc"] - g["qc"][i]) < 1e-12, axis=1))
        f[same], e0[same], st[same] = fv, ev, sv
    g.update(f_cap3000=f, E0_cap3000=e0, status_cap3000=st)
    np.savez(path, **g)


def planner_golden():
    """example/simple_joint_space_planner.py (nx = 280: what the generic tape family's limited-memory path is tested with): four goal poses solved by
    oracle/ipm_reference_form.py on the literal layout (scipy SLSQP in the reference's wiring reports "inequality constraints incompatible" on this
    problem and trust-constr meets a singular Jacobian -- the four quaternion rows of the final pose have rank three).  IPOPT's default bound
    relaxation (1e-8): the stored optimum may sit sum|lam| 1e-8 ~ 1e-6 below the exactly feasible one.  (A tighter relaxation is no way to a
    sharper golden: the multipliers of an (e, -e) pair grow like mu / relax, the scaled termination test of the method (s_d) then accepts
    anything -- with 1e-11 the run "converges" at the solution of the first barrier problem, f = 0.7249 instead of 0.7007.)"""
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import JointSpacePlannerNLP

    med7 = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    nlp = JointSpacePlannerNLP(med7)
    q0 = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    rng = np.random.default_rng(SEED + 8)
    P, X, F = [], [], []
    for i in range(4):
        qg = np.deg2rad([20, 55, -10, -70, 10, -40, 15]) if i == 0 else q0 + rng.uniform(-0.35, 0.35, 7)
        p = np.concatenate([q0, q0, med7.get_global_link_position("lbr_link_ee", qg), med7.get_global_link_quaternion("lbr_link_ee", qg)])
        r = solve_ipm(nlp, nlp.seed(q0), p)
        # KKT of the reference form with the method's own multipliers (they reach 1e5 on the integration rows of this problem -- the acceleration
        # cost carries 10 / dt^2 -- which is beyond what a least-squares fit of the multipliers resolves to 1e-6)
        v, lam = nlp.v(r["x"], p), r["lam_v"]
        stat = float(np.abs(nlp.df(r["x"], p) - nlp.dv(r["x"], p).T @ lam).max())
        print("planner", i, r["status"], r["iters"], r["f"], "stationarity", stat, "min v", v.min(), "max lam", lam.max(), "g min", nlp.g(r["x"], p).min())
        assert r["status"] == "optimal" and v.min() >= -1.01e-8 and stat <= 1e-6 * max(1.0, lam.max()) and lam.min() >= 0.0
        P.append(p); X.append(r["x"]); F.append(r["f"])
    np.savez(os.path.join(G, "planner_golden.npz"), p=np.array(P), x=np.array(X), f=np.array(F), q0=q0)


def ipm_configs_golden(only=None):
    """oracle/ipm_reference_form.py on the other BASELINE configs' literal NLPs, from the reference's seeds: config 4 as shipped (dual_arm.py, T = 50,
    1386 variables, zero seed as the script leaves it; ~1 minute) and config 5 at T = 6 (168 variables; the instance without and the one with binding
    effort rows of torque_golden.npz; Lagrangian Hessian by central differences of its analytic gradient: ~25 minutes each).  Compared in
    tests/test_ipm_reference_form.py with the answers scipy's SLSQP / trust-constr (reference wiring) gave for the same instances.
    Parts ("dual", "t6", "t6lim") are merged into the file one by one."""
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import DualArmNLP, TorqueMPCNLP
    from oracle.torque import TorqueProblem

    path = os.path.join(G, "ipm_configs_golden.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    parts = only or ("dual", "t6", "t6lim")  # round 4: "t30", "t30lim" = config 5 at its BASELINE size (840 variables, 1680 rows; exact Lagrangian Hessian)
    kin = os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json")
    if "dual" in parts:
        rl = OracleRobot(kin, name="kukal")
        rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
        rr = OracleRobot(kin, name="kukar")
        rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
        nlp = DualArmNLP(rl, rr, T=50)
        QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
        p = np.concatenate([QC, QC])
        t0 = time.time()
        r = solve_ipm(nlp, np.zeros(nlp.nx), p)
        print("config 4 as shipped:", r["status"], r["iters"], r["f"], round(time.time() - t0), "s", flush=True)
        out.update(dual_p=p, dual_x=r["x"], dual_f=r["f"], dual_iters=r["iters"], dual_optimal=r["status"] == "optimal")
        np.savez(path, **out)
    med7 = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    g = np.load(os.path.join(G, "torque_golden.npz"))
    for tag in ("t6", "t6lim", "t30", "t30lim"):
        if tag not in parts:
            continue
        lim = float(g[tag + "_lim"])
        prob = TorqueProblem(med7, "lbr_link_ee", T=g[tag + "_goal"].shape[1], dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=None if lim > 1e8 else lim)
        nlp = TorqueMPCNLP(prob)
        fs, its, ok = [], [], []
        for b in range(len(g[tag + "_qc"])):
            pb = nlp.pack_p(g[tag + "_qc"][b], np.zeros(7), g[tag + "_goal"][b])
            t0 = time.time()
            r = solve_ipm(nlp, nlp.seed(g[tag + "_qc"][b]), pb, max_iter=3000 if tag.startswith("t30") else 500)
            print("config 5", tag, b, r["status"], r["iters"], r["f"], "golden", g[tag + "_f"][b], round(time.time() - t0), "s", flush=True)
            fs.append(r["f"]); its.append(r["iters"]); ok.append(r["status"] == "optimal")
            if tag.startswith("t30"):
                out.setdefault(f"tq_{tag}_x", np.zeros((len(g[tag + "_qc"]), nlp.nx)))[b] = r["x"]
        out.update({f"tq_{tag}_f": np.array(fs), f"tq_{tag}_iters": np.array(its), f"tq_{tag}_optimal": np.array(ok)})
        np.savez(path, **out)



def _ipm_config4_one(args):
    i, qcl, qcr, radius, T = args
    import time as _t

    from examples.dual_arm import SPHERE_LINKS
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import GuardedDualArmNLP

    kin = os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json")
    rl = OracleRobot(kin, name="kukal")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr = OracleRobot(kin, name="kukar")
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    nlp = GuardedDualArmNLP(rl, rr, SPHERE_LINKS, 6, T=T)
    obs = np.concatenate([[0.55, 0.0, 0.1 * (j + 1), 0.1] for j in range(6)])
    p = np.concatenate([qcl, qcr, np.full(4, radius), obs, np.full(4, radius), obs])
    x0 = np.zeros(nlp.nx)
    for k, qc in enumerate((qcl, qcr)):
        x0[k * nlp.nx1 : k * nlp.nx1 + 7 * T] = np.tile(qc, T)
    t0 = _t.time()
    r = solve_ipm(nlp, x0, p, max_iter=3000)
    k = kkt_reference_form(nlp, r["x"], p, active_tol=1e-6)
    print(f"config 4 synthetic {i}: {r['status']} it={r['iters']} f={r['f']:.12f} E0={r['E0']:.1e} kkt=({k['stationarity']:.1e}, {k['feasibility']:.1e}, "
          f"{k['complementarity']:.1e}) {round(_t.time() - t0)} s", flush=True)
    return p, r["x"], r["f"], r["iters"], r["status"] == "optimal", np.array([k["stationarity"], k["feasibility"], k["complementarity"]])


def ipm_config4_golden(n_dual=2, T=100, radius=0.15, workers=2):
    """oracle/ipm_reference_form.py (the reference's algorithm class on the reference's form, exact Lagrangian Hessian) on BASELINE config 4 at its
    stated size: dual_arm.py with T = 100, enforce_model_limits and sphere_collision_avoidance_constraints on both arms (4 links x 6
    obstacles, link radius 0.15): 2786 variables, 10 400 rows of v per dual-arm instance, seed Q = qc at every knot (what the GPU tests and bench.py
    seed).  n_dual dual-arm instances = 2 n_dual arm solves; ~1 hour of CPU.  -> tests/golden/ipm_config4_golden.npz"""
    import multiprocessing as mp

    from examples.dual_arm import SPHERE_LINKS
    from oracle.guarded import Guards, guard_values
    from oracle.structured import FoldedChain

    rng = np.random.default_rng(SEED + 44)
    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    # At radius 0.15 the nominal configuration clears the obstacle column by 1.6e-3 only, and q_0 = qc is pinned (fix_configuration): about half
    # of the perturbed arms START inside a clearance, which makes the NLP infeasible as posed (its knot-0 sphere rows are negative constants; the
    # kernels skip them, an interior-point method can only report infeasibility).  The golden instances are drawn among the feasible ones.
    kin = os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json")
    chains = []
    for name, y in (("kukal", -0.25), ("kukar", 0.25)):
        rob = OracleRobot(kin, name=name)
        rob.add_base_frame("global_world", xyz=[0.0, y, 0.0])
        chains.append(FoldedChain(rob, "end_effector_ball"))
    Gd = Guards(lo=None, up=None, links=SPHERE_LINKS, link_radii=np.full(4, radius), obs_pos=np.array([[0.55, 0.0, 0.1 * (j + 1)] for j in range(6)]), obs_radii=np.full(6, 0.1))
    jobs, drawn = [], 0
    while len(jobs) < n_dual:
        qcl, qcr = QC + rng.uniform(-0.1, 0.1, 7), QC + rng.uniform(-0.1, 0.1, 7)
        drawn += 1
        if min(guard_values(ch, q[None], Gd)[0].min() for ch, q in zip(chains, (qcl, qcr))) >= 1e-4:
            jobs.append((len(jobs), qcl, qcr, radius, T))
    print(f"config 4 goldens: {n_dual} feasible dual-arm instances among the first {drawn} drawn", flush=True)
    with mp.Pool(workers) as pool:
        rows = pool.map(_ipm_config4_one, jobs, chunksize=1)
    keys = ("p", "x", "f", "iters", "optimal", "kkt")
    np.savez(os.path.join(G, "ipm_config4_golden.npz"), T=T, radius=radius, **{k: np.array([row[j] for row in rows]) for j, k in enumerate(keys)})


def ipm_limits_golden():
    """oracle/ipm_reference_form.py from the reference's seeds on the problems with joint-velocity limit rows (enforce_model_limits(name, time_deriv=1),
    builder.py:471-509) that round 3 lowered: figure_eight_plan.py + the LWR's own velocity limits (T = 50, 693 variables, 1114 + 686 rows; the
    nominal instance and three perturbed ones) and dual_arm.py + 0.06 rad/s on every joint (T = 50, 1386 variables, 1372 k rows).  ~10 minutes."""
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import GuardedDualArmNLP, LimitedFigureEightNLP

    kin = os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json")
    kuka = OracleRobot(kin)
    vl = np.asarray(kuka.velocity_actuated_joint_limits)
    nlp = LimitedFigureEightNLP(kuka, "end_effector_ball", vlo=-vl, vup=vl, T=50)
    QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    rng = np.random.default_rng(SEED + 77)
    qcs = QC0[None] + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.1, 0.1, (3, 7))])
    out = {"fig8v_qc": qcs, "fig8v_vl": vl}
    fs, its, ok = [], [], []
    for b, qc in enumerate(qcs):
        t0 = time.time()
        r = solve_ipm(nlp, nlp.seed(qc), qc, max_iter=1500)
        k = kkt_reference_form(nlp, r["x"], qc, active_tol=1e-6)
        dq = np.abs(r["x"][350:]).reshape(49, 7).max(0)
        print("fig8 + velocity limits", b, r["status"], r["iters"], r["f"], "max|dq|/limit", (dq / vl).max(), "kkt", k["stationarity"], k["feasibility"], round(time.time() - t0), "s", flush=True)
        fs.append(r["f"]); its.append(r["iters"]); ok.append(r["status"] in ("optimal", "acceptable"))
    out.update(fig8v_f=np.array(fs), fig8v_iters=np.array(its), fig8v_ok=np.array(ok))
    np.savez(os.path.join(G, "ipm_limits_golden.npz"), **out)
    rl = OracleRobot(kin, name="kukal")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr = OracleRobot(kin, name="kukar")
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    vmax = 0.06
    nlp = GuardedDualArmNLP(rl, rr, [], 0, T=50, limits=False, vlimits=(-np.full(7, vmax), np.full(7, vmax)))
    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    p = np.concatenate([QC, QC + 0.02])
    x0 = np.zeros(nlp.nx)  # the script leaves the seed at zero
    t0 = time.time()
    r = solve_ipm(nlp, x0, p, max_iter=1500)
    print("dual arm + velocity limits:", r["status"], r["iters"], r["f"], round(time.time() - t0), "s", flush=True)
    out.update(dualv_p=p, dualv_vmax=vmax, dualv_f=r["f"], dualv_iters=r["iters"], dualv_ok=r["status"] in ("optimal", "acceptable"))
    np.savez(os.path.join(G, "ipm_limits_golden.npz"), **out)


def ipm_torque_velocity_golden():
    """oracle/ipm_reference_form.py on config 5 at T = 6 with joint-velocity limits 0.25 rad/s next to the effort limits (the rows round 3 lowered
    for the torque family); Lagrangian Hessian by central differences of the analytic gradient, ~25 minutes."""
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import TorqueMPCNLP
    from oracle.torque import TorqueProblem

    med7 = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    g = np.load(os.path.join(G, "torque_golden.npz"))
    vmax = 0.25
    prob = TorqueProblem(med7, "lbr_link_ee", T=6, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=float(g["t6lim_lim"]))
    nlp = TorqueMPCNLP(prob, vlimits=(-vmax, vmax))
    qc, goal = g["t6lim_qc"][0], g["t6lim_goal"][0]
    p = nlp.pack_p(qc, np.zeros(7), goal)
    t0 = time.time()
    r = solve_ipm(nlp, nlp.seed(qc), p, max_iter=500)
    dq = np.abs(nlp.split(r["x"])[1]).max()
    print("config 5 + velocity limits:", r["status"], r["iters"], r["f"], "max|dq|", dq, round(time.time() - t0), "s", flush=True)
    path = os.path.join(G, "ipm_limits_golden.npz")
    out = dict(np.load(path))
    out.update(tqv_qc=qc, tqv_goal=goal, tqv_vmax=vmax, tqv_lim=float(g["t6lim_lim"]), tqv_f=r["f"], tqv_iters=r["iters"], tqv_ok=r["status"] in ("optimal", "acceptable"))
    np.savez(path, **out)


if __name__ == "__main__":
    if "--ipm-torque-velocity" in sys.argv:
        ipm_torque_velocity_golden()
        sys.exit(0)
    if "--ipm-limits" in sys.argv:
        ipm_limits_golden()
        sys.exit(0)
    if "--ipm-pm" in sys.argv:
        ipm_pm_golden()
        sys.exit(0)
    if "--ipm-config4" in sys.argv:
        ipm_config4_golden()
        sys.exit(0)
    if "--ipm-configs" in sys.argv:
        ipm_configs_golden([a for a in sys.argv[2:] if not a.startswith("-")] or None)
        sys.exit(0)
    if "--planner" in sys.argv:
        planner_golden()
        sys.exit(0)
    if "--ipm-cap3000" in sys.argv:
        fig8_ipm_cap3000()
        sys.exit(0)
    if "--ipm-extend" in sys.argv:
        fig8_ipm_extend()
        sys.exit(0)
    if "--ipm" in sys.argv:  # ~10 minutes
        fig8_ipm_golden()
        sys.exit(0)
    if "--fig8-dense" in sys.argv:  # ~25 minutes
        fig8_perturbed_dense_golden()
        sys.exit(0)
    if "--torque-slsqp" in sys.argv:
        torque_golden_add_slsqp()
        sys.exit(0)
    if "--torque-port" in sys.argv:
        torque_golden_refresh_port()
        sys.exit(0)
    if "--torque" in sys.argv:  # ~20 minutes (trust-constr on the literal layout)
        torque_golden()
        sys.exit(0)
    spatialmath_golden()
    fk_golden()
    nlp_golden()
    pm_golden()
    ik_golden()
    guard_golden()  # ~1 minute: scipy SLSQP on the T = 50 guarded arm
    print("(tests/golden/torque_golden.npz: python tools/make_golden.py --torque, ~20 minutes)")
    print("golden fixtures written to", G)
