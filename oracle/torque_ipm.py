"""ORACLE (test infrastructure, not product code) -- numpy port of the round-4 state machine of csrc/oh_torque.hip for BASELINE configs[4]
(7-DoF torque MPC with RNEA dynamics equality rows): a primal-dual interior point on the stage form, the reference's own
algorithm class (optas/solver.py:355-398 hands the problem to IPOPT), in place of the augmented-Lagrangian outer loop of rounds 1-3
(oracle/torque.py:solve_torque_lm, kept as the independent second solver of the same problem).

Stage form as before: u_t = ddq_t free, (q, dq) rolled out through the Euler rows (builder.py:419-469), TAU_t = rnea(q_t, dq_t, u_t) through the
dynamics rows (models.py:1731-1884).  What remains are the inequality rows  s = [TAU - lo; up - TAU; (dq - vlo; vup - dq)] >= 0
(enforce_model_limits, builder.py:471-509).  They enter through the log barrier  -mu_b sum log s_i  with multipliers lam_i of their own
(primal-dual: the stage blocks carry Sigma = lam / s where the augmented Lagrangian carried rho on the active rows); below delta = theta mu_b the
logarithm is continued by its second-order Taylor polynomial (a relaxed barrier: the merit is finite at infeasible trial points, so an infeasible
seed or an overshooting step needs no separate restoration phase; at convergence every row sits in the logarithmic regime, so the answer is a point
of the central path with  lam_i s_i = mu_b <= tol_c).  Steps: Levenberg-Marquardt on the barrier merit, from a Riccati sweep over the stages; the
multipliers follow the linearised complementarity equation with the slack change the step really produced.

Curvature.  The torque term is a large-residual least-squares term (gravity torques of 50 N m), so the Gauss-Newton model converges linearly (rate
~0.8) and the constraint curvature  lam^T d^2 TAU  is missing from it altogether.  Once the reduced gradient is below `curv_from` the stage blocks hold
the exact Hessian of the Lagrangian -- sum_i c_i d^2 tau_i/dz^2 from oracle.torque.rnea_ctau_hessian (hand-written adjoint of the reference's
recursion, differentiated once more), 2 w_p sum_k r_k d^2 p_k/dq^2 in closed form -- and the iteration is Newton's: 2-3 steps per barrier value.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
import numpy as np

from .torque import TorqueProblem, costate_gradient, rnea_batch, rnea_ctau_hessian, rnea_jacobian


def position_curvature(chain, Q, r):
    """sum_k r_k d^2 p_k / dq^2, (T, n, n): d^2 p / dq_a dq_b = z_b x (z_a x (e - o_a)) for b <= a on a chain of revolute joints
    (the derivative of column a of the geometric Jacobian, models.py:1211-1264, with respect to a joint below it)."""
    e, _, z, o = chain.fk(Q)
    T, n = Q.shape
    K = np.zeros((T, n, n))
    for a in range(n):
        inner = np.cross(z[:, a], e - o[:, a])
        for b in range(a + 1):
            K[:, a, b] = K[:, b, a] = np.sum(r * np.cross(z[:, b], inner), 1)
    return K


def riccati_gains(H, g, mu, dt):
    """Backward sweep of oracle.torque.riccati_torque (state part damped by mu), returning the gains instead of the step:
    (K (T, n, 2n), k (T, n), ok, qk = sum_t qu_t^T k_t)."""
    T, m = g.shape
    n = m // 3
    nx = 2 * n
    A = np.eye(nx)
    A[:n, n:] = dt * np.eye(n)
    Bm = np.zeros((nx, n))
    Bm[n:] = dt * np.eye(n)
    P, p = np.zeros((nx, nx)), np.zeros(nx)
    Ks, ks = np.zeros((T, n, nx)), np.zeros((T, n))
    qk = 0.0
    for t in range(T - 1, -1, -1):
        Ht = H[t] + np.diag(np.concatenate([mu * np.ones(nx), np.zeros(n)]))
        Qxx = Ht[:nx, :nx] + A.T @ P @ A
        Qux = Ht[nx:, :nx] + Bm.T @ P @ A
        Quu = Ht[nx:, nx:] + Bm.T @ P @ Bm
        qx = g[t, :nx] + A.T @ p
        qu = g[t, nx:] + Bm.T @ p
        try:
            L = np.linalg.cholesky(Quu)
        except np.linalg.LinAlgError:
            return None, None, False, 0.0
        Ks[t] = np.linalg.solve(L.T, np.linalg.solve(L, Qux))
        ks[t] = np.linalg.solve(L.T, np.linalg.solve(L, qu))
        P = Qxx - Qux.T @ Ks[t]
        P = 0.5 * (P + P.T)
        p = qx - Qux.T @ ks[t]
        qk += float(qu @ ks[t])
    return Ks, ks, True, qk


def rollout_step(Ks, ks, alpha, dt):
    """dz (T, 3n) of the closed-loop step with the feed-forward scaled by alpha (the line search of iLQR): du_t = -alpha k_t - K_t dx_t."""
    T, n, nx = Ks.shape
    dz = np.zeros((T, 3 * n))
    dx = np.zeros(nx)
    for t in range(T):
        du = -alpha * ks[t] - Ks[t] @ dx
        dz[t, :nx], dz[t, nx:] = dx, du
        dx = np.concatenate([dx[:n] + dt * dx[n:], dx[n:] + dt * du])
    return dz


def solve_torque_ipm(prob: TorqueProblem, qc, dqc, goal, U0=None, max_iter=300, tol=1e-6, tol_c=1e-8, mu0=0.1, theta=0.01, kappa_eps=10.0, kappa_mu=0.4,
                     theta_mu=1.35, curv_from=0.1, vlimits=None, verbose=False, kappa_sig=1e10, tau_ftb=0.995, max_back=3, curv_after=3, curv_late=1.0, stall_max=25, mu_dec=1.0 / 3.0, ls_curv=True, curv_lag=3):
    """One instance.  Returns dict(U, Q, dQ, tau, f, iters, rejected, stat, status, mu_b, lam (T, rows), s (T, rows))."""
    T, n, dt = prob.T, prob.n, prob.dt
    wp, wt, wv = prob.w_path, prob.w_tau, prob.w_vel
    lo, up = prob.tau_lo, prob.tau_up
    vel = vlimits is not None
    if vel:
        vlo, vup = (np.broadcast_to(np.asarray(v, dtype=float), (n,)) for v in vlimits)
    U = np.zeros((T, n)) if U0 is None else np.array(U0, float)
    mub = mu0
    mu_min = 0.1 * tol_c
    store = {"Hc": None, "age": 0, "n_computed": 0}  # the stored curvature term and the evaluations since it was computed

    def evalp(U, prev, mub, use_curv):
        Q, dQ = prob.rollout(qc, dqc, U)
        tau = rnea_batch(prob.tb, Q, dQ, U)
        J = rnea_jacobian(prob.tb, Q, dQ, U)
        e, _, Jp, _ = prob.chain.jac(Q)
        r = e - goal
        s = np.concatenate([tau - lo, up - tau] + ([dQ - vlo, vup - dQ] if vel else []), 1)
        delta = theta * mub
        rel = s < delta
        sc = np.where(rel, delta, s)
        if prev is None:
            lam = mub / sc
        else:  # linearised complementarity  s dlam + lam ds = mu_b - lam s  with the slack change of the step as it came out
            lam_o, s_o = prev["lam"], prev["s"]
            lam = (mub - lam_o * (s - s_o)) / np.maximum(s_o, delta)
            lam = np.maximum(lam, (1.0 - tau_ftb) * lam_o)
            lam = np.clip(lam, mub / (kappa_sig * sc), kappa_sig * mub / sc)
        bco = np.where(rel, (2.0 * delta - s) / delta**2, 1.0 / sc)  # -psi'(s) / mu_b
        lam = np.where(rel, mub * bco, lam)
        sig = np.where(rel, mub / delta**2, lam / sc)
        Bt = np.where(rel, -np.log(delta) - (s - delta) / delta + 0.5 * (s - delta) ** 2 / delta**2, -np.log(sc)).sum(1)
        ftrue = wp * np.sum(r * r, 1) + wt * np.sum(tau * tau, 1) + wv * np.sum(dQ * dQ, 1)
        cf = 2.0 * wt * tau
        cb = -bco[:, :n] + bco[:, n:2 * n]
        d = 2.0 * wt + sig[:, :n] + sig[:, n:2 * n]
        gf = np.einsum("ti,tid->td", cf, J)
        gf[:, :n] += 2.0 * wp * np.einsum("tki,tk->ti", Jp, r)
        gf[:, n:2 * n] += 2.0 * wv * dQ
        gb = np.einsum("ti,tid->td", cb, J)
        H = np.einsum("ti,tid,tie->tde", d, J, J)
        H[:, :n, :n] += 2.0 * wp * np.einsum("tki,tkj->tij", Jp, Jp)
        H[:, n:2 * n, n:2 * n] += 2.0 * wv * np.eye(n)
        if vel:
            gb[:, n:2 * n] += -bco[:, 2 * n:3 * n] + bco[:, 3 * n:]
            H[:, np.arange(n, 2 * n), np.arange(n, 2 * n)] += sig[:, 2 * n:3 * n] + sig[:, 3 * n:]
        computed = False
        if use_curv:
            # round 5 (k_tq_curv / k_tq_eval3, D.curv 1 / 2): the curvature term is computed at every (curv_lag + 1)-th evaluation and the stored one
            # added in between -- near the solution it moves little from step to step (256 instances: 24.70 steps against 24.64 without the lag, the
            # term computed 3.2 times per solve instead of 11.3)
            if store["Hc"] is not None and store["age"] < curv_lag:
                H = H + store["Hc"]
                store["age"] += 1
            else:
                cH = cf - lam[:, :n] + lam[:, n:2 * n]
                Hc = rnea_ctau_hessian(prob.tb, Q, dQ, U, cH)
                Hc[:, :n, :n] += 2.0 * wp * position_curvature(prob.chain, Q, r)
                H = H + Hc
                store["Hc"], store["age"], computed = Hc, 0, True
                store["n_computed"] += 1
        else:
            store["Hc"] = None
        return {"U": U, "ftrue": float(ftrue.sum()), "B": float(Bt.sum()), "gf": gf, "gb": gb, "H": H, "s": s, "lam": lam, "nrel": int(rel.sum()), "tau": tau,
                "Q": Q, "dQ": dQ, "J": J, "curv_computed": computed}

    mu, nun = 0.0, 4.0
    iters = rejected = backtracks = 0
    cur, Ut, status, f_cur, use_curv = None, U, 1, np.inf, False
    alpha, qk, ndx, n_back, n_barrier, stall, n_restart = 1.0, 0.0, 0.0, 0, 0, 0, 0
    Ks = ks = None
    force_accept = False  # the pending "trial" is the accepted point itself, to be evaluated again under a new barrier parameter (round 6, below; k_tq_step: D.first = 1)
    while True:
        tr = evalp(Ut, None if force_accept else cur, mub, use_curv)
        f_t = tr["ftrue"] + mub * tr["B"]
        new_gains = True
        if cur is None or force_accept:
            accept, force_accept = True, False
        elif not np.isfinite(f_t):
            # the trial left the domain of the arithmetic: same gains, a tenth of the feed-forward (no damping change: the model is not to blame)
            accept, new_gains = False, False
            alpha *= 0.1
            backtracks += 1
        else:
            pred = (alpha - 0.5 * alpha * alpha) * qk + 0.5 * alpha * alpha * mu * ndx
            ratio = (f_cur - f_t) / max(pred, 1e-300)
            accept = bool(ratio > 1e-4 or (pred <= 1e-15 * abs(f_cur) and f_t <= f_cur + 1e-14 * abs(f_cur)))
            if accept:
                mu *= mu_dec if ratio > 0.9 else max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3)
                mu = 0.0 if mu < 1e-7 else mu
                nun = 4.0
            elif (alpha < 1.0 or (ls_curv and use_curv)) and n_back < max_back:
                # a step the boundary rule had shortened already: the rows near their bounds are to blame (the logarithm is far from its quadratic
                # model there), not the model of the states -- shorten the feed-forward further, same gains, same damping.  Round 5 (ls_curv): a rejected
                # full Newton step (trial evaluated with exact curvature) is treated the same way -- it gives up more barrier than it gains, a third of
                # it follows the model, and the damping (on the states) does not shorten a step that lives in the accelerations
                new_gains = False
                alpha *= 0.25
                n_back += 1
                backtracks += 1
            else:
                mu = max(mu * nun, 0.1)
                nun *= 2.0
                rejected += 1
        if not accept and tr["curv_computed"]:
            store["Hc"] = None  # a term computed at a point that was refused is not kept
        if accept or new_gains:
            n_back = 0
        if accept:
            cur, f_cur = tr, f_t
        g = cur["gf"] + mub * cur["gb"]
        stat = float(np.abs(costate_gradient(g, dt)).max())
        if verbose:
            print(f"  it {iters:3d} f={cur['ftrue']:.12f} merit={f_cur:.9f} stat={stat:.3e} mu_b={mub:.2e} mu={mu:.3g} alpha={alpha:.3g} relaxed={cur['nrel']} "
                  f"smin={cur['s'].min():.3e} curv={use_curv} {'A' if accept else ('R' if new_gains else 'B')}")
        if stat <= tol and mub <= tol_c and cur["nrel"] == 0:
            status = 0
            break
        if iters >= max_iter:
            break
        if new_gains:
            nb_before = n_barrier
            # barrier update (Waechter & Biegler 2006, eq. 7): merit and gradient are affine in mu_b while every row is in the logarithmic regime
            if accept and stat <= kappa_eps * mub and cur["nrel"] == 0 and mub > mu_min:
                mub = max(mu_min, min(kappa_mu * mub, mub**theta_mu))
                n_barrier += 1
                f_cur = cur["ftrue"] + mub * cur["B"]
                g = cur["gf"] + mub * cur["gb"]
                stat = float(np.abs(costate_gradient(g, dt)).max())
            elif accept and stat <= kappa_eps * mub and cur["nrel"] > 0 and mub > mu_min:
                # Round 6: stationary for this mu_b with rows inside the relaxed zone (slack below theta mu_b: a row whose multiplier exceeds 1 / theta, e.g. a
                # velocity limit the tracking cost pushes hard against).  Merit and gradient are not affine in mu_b there, so the update above does not apply --
                # and without one the iteration sat at this point until the cap (every step null, every null step "rejected" at rounding level, the damping
                # doubling: med7, effort 58 N m, |dq| <= 0.05).  The barrier parameter is lowered all the same and the point itself evaluated again under it
                # (a null step, accepted as it is); theta mu_b shrinks with it, so feasible rows leave the relaxed zone on the way down.
                mub = max(mu_min, min(kappa_mu * mub, mub**theta_mu))
                n_barrier += 1
                stall = 0
                Ut, force_accept = cur["U"], True
                iters += 1
                continue
            elif accept and stat <= 10.0 * tol and cur["nrel"] > 0 and mub <= mu_min and float(-cur["s"].min()) > tol_c:
                # ... and at the floor of the barrier parameter a stationary point that still violates a row by more than the complementarity tolerance has no
                # feasible neighbour: the relaxed barrier is a penalty of weight 1 / (theta^2 mu_b) by now.  The reference: IPOPT's Infeasible_Problem_Detected
                # (did_solve() False, solver.py:133-134, 407-412).
                status = 3  # OH_STATUS_INFEASIBLE
                break
            stall = 0 if n_barrier != nb_before else stall + 1
            if stall >= stall_max and cur["nrel"] == 0 and mub <= mu_min and stat <= 10.0 * tol:
                # acceptable level: stall_max steps at the floor of the barrier parameter within ten times the tolerance (the arithmetic floor of the
                # reduced gradient when an active row has a slack of 1e-8: its multiplier mu_b / s is good to 1e-6 relative); k_tq_step alike
                status = 4  # OH_STATUS_ACCEPTABLE
                break
            if stall >= stall_max and cur["nrel"] == 0 and accept:
                # watchdog: stall_max steps without reaching the barrier test -- the iterate sits far from the central path of this mu_b (slacks of the
                # active rows collapse and recover in turn).  Back to a larger barrier parameter: the path is regained there and followed down again.
                mub = min(mu0, 100.0 * mub)
                f_cur = cur["ftrue"] + mub * cur["B"]
                g = cur["gf"] + mub * cur["gb"]
                stat = float(np.abs(costate_gradient(g, dt)).max())
                stall, n_restart = 0, n_restart + 1
            use_curv = stat <= curv_from or (n_barrier >= curv_after and stat <= curv_late)
            while True:
                Ks, ks, ok, qk = riccati_gains(cur["H"], g, mu, dt)
                if ok:
                    break
                mu = max(mu * nun, 0.1)  # an indefinite stage block of the exact Hessian: more damping, same point
                nun *= 2.0
            ndx = float(np.sum(rollout_step(Ks, ks, 1.0, dt)[:, :2 * n] ** 2))
            # fraction to the boundary on the linearised rows (Waechter & Biegler 2006, eq. 15): the closed-loop step is linear in the scale alpha of
            # its feed-forward, so the largest alpha that leaves every slack 0.5 % of itself is a ratio test over the rows
            dz1 = rollout_step(Ks, ks, 1.0, dt)
            ds = np.einsum("tid,td->ti", cur["J"], dz1)
            ds = np.concatenate([ds, -ds] + ([dz1[:, n:2 * n], -dz1[:, n:2 * n]] if vel else []), 1)
            cut = (ds < 0.0) & (cur["s"] >= theta * mub)
            alpha = min(1.0, float(np.min(-tau_ftb * cur["s"][cut] / ds[cut]))) if cut.any() else 1.0
        dz = rollout_step(Ks, ks, alpha, dt)
        Ut = cur["U"] + dz[:, 2 * n:]
        iters += 1
    s = cur["s"]
    return {"U": cur["U"], "Q": cur["Q"], "dQ": cur["dQ"], "tau": cur["tau"], "f": cur["ftrue"], "iters": iters, "rejected": rejected, "backtracks": backtracks, "stat": stat, "status": status,
            "mu_b": mub, "lam": mub / np.maximum(s, 1e-300), "s": s, "curv_computed": store["n_computed"]}


def rollout_torque_ipm(prob: TorqueProblem, q0, dq0, goal_table, n_ticks, advance=1, mu_warm=1e-6, mu_dec_warm=0.1, **kw):
    """Closed-loop receding horizon of one plant, the loop oh_tq_rollout keeps on the device (pattern of example/point_mass_mpc.py:156-175: the seed of
    a tick is the previous solution): parameters of tick k = the plant's state and rows k * advance .. of its goal table; seed = the previous plan's
    accelerations shifted by `advance` knots, the last one repeated (tick 0: zeros), barrier parameter of a warm tick mu_warm; the plant takes the
    plan's state at knot `advance`.  A warm tick starts next to its optimum: the damping comes down by mu_dec_warm = 0.1 after a good step (the
    cold solve's 1/3 guards against accept / reject cycles far from it; measured on 8192 plants x 20 ticks: 10.5 against 11.0 steps per tick).  Returns dict(states (n_ticks + 1, 2 n), tau0 (n_ticks, n), f, iters, status (n_ticks,), plans: the per-tick results)."""
    T, n = prob.T, prob.n
    goal_table = np.asarray(goal_table, float)
    assert goal_table.shape == (n_ticks * advance + T, 3)
    states = np.zeros((n_ticks + 1, 2 * n))
    states[0, :n], states[0, n:] = q0, dq0
    tau0, f, iters, status, plans = np.zeros((n_ticks, n)), np.zeros(n_ticks), np.zeros(n_ticks, int), np.zeros(n_ticks, int), []
    U = None
    for k in range(n_ticks):
        tick = dict(kw)
        if U is None:
            tick["mu0"] = kw.get("mu0", 0.1)
        else:
            tick["mu0"] = mu_warm
            tick.setdefault("mu_dec", mu_dec_warm)
        r = solve_torque_ipm(prob, states[k, :n], states[k, n:], goal_table[k * advance : k * advance + T], U0=U, **tick)
        plans.append(r)
        states[k + 1, :n], states[k + 1, n:] = r["Q"][advance], r["dQ"][advance]
        tau0[k], f[k], iters[k], status[k] = r["tau"][0], r["f"], r["iters"], r["status"]
        U = np.concatenate([r["U"][advance:], np.repeat(r["U"][-1:], advance, 0)])
    return {"states": states, "tau0": tau0, "f": f, "iters": iters, "status": status, "plans": plans}
