"""ORACLE (test infrastructure, not product code) -- the reference's ALGORITHM CLASS on the reference's PROBLEM FORM.

Only ``tests/``, ``tools/make_golden.py`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

The reference's hot path hands ``min f(x, p)  s.t.  0 <= v(x, p) <= 1e10`` with ``v = [k; g; a; -a; h; -h]`` (every equality as a PAIR of
inequalities, optimization.py:27-51,292-306; bounds solver.py:355-363,391-395) to ``casadi.nlpsol("ipopt")`` with default options
(example/figure_eight_plan.py:108-110: ``setup("ipopt")``).  CasADi/IPOPT cannot be installed here: PARITY UNPINNED against the
reference's own iterates.  This module restates the published algorithm IPOPT implements -- A. Waechter, L. T. Biegler, "On the implementation
of an interior-point filter line-search algorithm for large-scale nonlinear programming", Math. Program. 106 (2006) [WB] -- with IPOPT 3.x's
documented default parameters, on exactly that form, started from the reference's seed, so that "which local minimum does an interior-point
method of IPOPT's class reach from the reference seed" has an answer produced by an algorithm that shares nothing with the retraction /
null-space / Riccati path of the HIP kernels (no elimination of the linear rows, no manifold, no Gauss-Newton: literal layout, literal
rank-3 quaternion rows, exact Lagrangian Hessian).

What follows [WB] literally (section in brackets):
  * the slack reformulation IPOPT applies to two-sided inequality rows: c(x, s) = d_c v(x) - s = 0, s_L <= s <= s_U, with the bounds relaxed by
    bound_relax_factor = 1e-8 (which is what gives the (e, -e) pairs an interior at all: s in [-1e-8, ...]) [3.5]
  * gradient-based scaling of f and of the rows of v at the starting point, g_max = 100 [3.8]
  * starting point: slacks pushed inside their bounds with kappa_1 = kappa_2 = 1e-2, bound multipliers 1, least-squares equality multipliers
    discarded above 1e3 [3.6]
  * monotone barrier update mu <- max(tol/10, min(0.2 mu, mu^1.5)) once E_mu <= 10 mu, mu_0 = 0.1, fraction to the boundary
    tau = max(0.99, 1 - mu) [2.1, 2.2, (7), (8)]; optimality error E_mu with the multiplier scaling s_d, s_c, s_max = 100 [(5), (6)]
  * the primal-dual step from the symmetric system (13) with the inertia-correcting shifts delta_w (1e-4 first, x100, then x8, 1/3 of the last
    successful one next time) and delta_c = 1e-8 mu^0.25 only if the system is singular [3.1, Alg. IC] -- the system is solved in condensed form
    (slacks and multipliers eliminated: the 693 x 693 matrix W + delta_w I + J^T D J must be positive definite, which is the inertia condition)
  * the multiplier reset (16) with kappa_Sigma = 1e10
  * the filter line search with theta = ||c||_1, the switching condition (19), Armijo (20), sufficient decrease (18), the filter augmentation
    (22), alpha_min (23) and second-order corrections (p_max = 4, kappa_soc = 0.99) [2.3, 2.4, Alg. A], constants gamma_theta = 1e-5,
    gamma_phi = 1e-8, delta = 1, s_theta = 1.1, s_phi = 2.3, eta_phi = 1e-8 (IPOPT's values)
  * termination: E_0 <= tol = 1e-8 (IPOPT default), or IPOPT's "acceptable" test (acceptable_tol 1e-6 on 15 consecutive iterations)
What is SIMPLIFIED (and why it does not matter for what this oracle is used for -- the point it converges to, graded afterwards by
kkt_reference_form): the feasibility restoration phase [3.3] is not IPOPT's interior-point solve of the l1-relaxed problem but a Gauss-Newton
descent on the violated rows followed by a slack reset; no watchdog, no automatic mu oracle, dense linear algebra (numpy) instead of MUMPS.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

INF = 1.0e10  # optimization.py:58


class SlackForm:
    """min d_f f(x)  s.t.  d_c v(x) - s = 0,  s_L <= s <= s_U   for one of the oracle's literal NLPs (oracle/problems.py) and one parameter vector."""

    def __init__(self, nlp, p, x0, g_max=100.0, relax=1e-8, scaling=True):
        self.nlp, self.p = nlp, np.asarray(p, float)
        self.nx, self.m = nlp.nx, nlp.nv
        g0 = nlp.df(x0, self.p)
        J0 = nlp.dv(x0, self.p)
        self.d_f = min(1.0, g_max / max(np.abs(g0).max(), 1e-300)) if scaling else 1.0
        rown = np.abs(J0).max(axis=1) if self.m else np.zeros(0)
        self.d_c = np.minimum(1.0, g_max / np.maximum(rown, 1e-300)) if scaling else np.ones(self.m)
        lo, up = np.zeros(self.m) * self.d_c, INF * self.d_c
        self.sL = lo - relax * np.maximum(1.0, np.abs(lo))
        self.sU = up + relax * np.maximum(1.0, np.abs(up))

    def f(self, x):
        return self.d_f * self.nlp.f(x, self.p)

    def df(self, x):
        return self.d_f * self.nlp.df(x, self.p)

    def v(self, x):
        return self.d_c * self.nlp.v(x, self.p)

    def dv(self, x):
        return self.d_c[:, None] * self.nlp.dv(x, self.p)

    def hess(self, x, lam):
        """Hessian of d_f f + lam^T (d_c v): the NLP supplies hess_lagrangian_v(x, p, sigma, lam_v) = sigma d2f + sum_i lam_v[i] d2v_i."""
        return self.nlp.hess_lagrangian_v(x, self.p, self.d_f, lam * self.d_c)


def _split_v(nlp, lam_v):
    """Multipliers of v = [k; g; a; -a; h; -h] -> (lam_k, lam_g, signed mu_a = lam(a) - lam(-a), signed mu_h)."""
    o = 0
    lk = lam_v[o : o + nlp.nk]; o += nlp.nk
    lg = lam_v[o : o + nlp.ng]; o += nlp.ng
    la = lam_v[o : o + nlp.na] - lam_v[o + nlp.na : o + 2 * nlp.na]; o += 2 * nlp.na
    lh = lam_v[o : o + nlp.nh] - lam_v[o + nlp.nh : o + 2 * nlp.nh]
    return lk, lg, la, lh


def attach_hessian(nlp):
    """Gives a literal NLP the member the interior-point method needs, ``hess_lagrangian_v(x, p, sigma, lam_v)``, from what the class has:
    ``hess_lagrangian(x, p, lam_h)`` (FigureEightNLP: f + lam_h . h, linear rows otherwise), ``ddf`` + linear rows (Booth, point-mass cost) plus
    ``ddg_dot(x, p, lam_g)`` where inequality rows are curved, or central differences of the analytic gradient of the Lagrangian (small problems)."""
    if hasattr(nlp, "hess_lagrangian_v"):
        return nlp

    def by_differences(x, p, sigma, lam_v, h=1e-6):
        def grad(xx):
            return sigma * nlp.df(xx, p) + nlp.dv(xx, p).T @ lam_v

        n = nlp.nx
        H = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = h
            H[:, j] = (grad(x + e) - grad(x - e)) / (2 * h)
        return 0.5 * (H + H.T)

    if hasattr(nlp, "hess_lagrangian") and nlp.ng == 0:  # FigureEightNLP, and with linear k rows (joint / joint-velocity limits) on top

        def hl(x, p, sigma, lam_v):
            _, _, _, mu_h = _split_v(nlp, lam_v)
            if sigma <= 0.0:
                raise ValueError("sigma must be positive")
            return sigma * nlp.hess_lagrangian(x, p, mu_h / sigma)

        nlp.hess_lagrangian_v = hl
    elif hasattr(nlp, "ddf") and nlp.nh == 0 and (nlp.ng == 0 or hasattr(nlp, "ddg_dot")):

        def hl(x, p, sigma, lam_v):
            H = sigma * np.asarray(nlp.ddf(x, p), float)
            if nlp.ng:
                H = H + nlp.ddg_dot(x, p, _split_v(nlp, lam_v)[1])
            return H

        nlp.hess_lagrangian_v = hl
    else:
        nlp.hess_lagrangian_v = by_differences
    return nlp


def _amax(a):
    return float(np.abs(a).max()) if a.size else 0.0


def solve_ipm(nlp, x0, p, tol=1e-8, max_iter=3000, mu0=0.1, scaling=True, verbose=False, acceptable_tol=1e-6, acceptable_iter=15, relax=1e-8):
    """Interior-point filter line search [WB] on the reference form.  Returns dict(x, f, iters, status, E0, lam_v (multipliers of v >= 0 in the
    reference's sign: lam >= 0 on active lower bounds), history).  `relax` is IPOPT's bound_relax_factor (default 1e-8): every row of v may end
    up to `relax` below zero, so an equality pair (e, -e) holds to |e| <= relax and the objective sits up to sum|lam| relax below the exact
    optimum; tests that compare with exactly feasible optima polish the point (oracle.solvers.dense_sqp).  Tightening `relax` instead is
    fragile: the two multipliers of an (e, -e) pair grow like mu / relax, and once they reach 1e8 the multiplier scaling s_d of the termination
    test [WB (6)] accepts any dual infeasibility -- a run with relax = 1e-11 on the joint-space planner stops at the solution of its first
    barrier problem."""
    attach_hessian(nlp)
    x = np.asarray(x0, float).copy()
    P = SlackForm(nlp, p, x, scaling=scaling, relax=relax)
    n, m = P.nx, P.m
    sL, sU = P.sL, P.sU
    # --- starting point [3.6] ---------------------------------------------------------------------------------------------
    k1 = k2 = 1e-2
    pL = np.minimum(k1 * np.maximum(1.0, np.abs(sL)), k2 * (sU - sL))
    pU = np.minimum(k1 * np.maximum(1.0, np.abs(sU)), k2 * (sU - sL))
    s = np.minimum(np.maximum(P.v(x), sL + pL), sU - pU)
    zL, zU = np.ones(m), np.ones(m)
    g, J = P.df(x), P.dv(x)
    # least-squares multipliers of c = v - s = 0: [I A^T; A 0][w; lam] = -[grad; 0] with A = [J, -I], grad = [g; -zL + zU]
    lam = np.linalg.solve(J @ J.T + np.eye(m), -(J @ g - (-zL + zU)))
    if _amax(lam) > 1e3:
        lam = np.zeros(m)
    mu = mu0
    tau = max(0.99, 1.0 - mu)
    smax, k_eps, k_mu, th_mu, k_sig = 100.0, 10.0, 0.2, 1.5, 1e10
    g_th, g_ph, dlt, s_th, s_ph, eta = 1e-5, 1e-8, 1.0, 1.1, 2.3, 1e-8
    c = P.v(x) - s
    theta0 = np.abs(c).sum()
    th_min, th_max = 1e-4 * max(1.0, theta0), 1e4 * max(1.0, theta0)
    filt = [(th_max, -np.inf)]  # entries (theta_j, phi_j): a trial is refused if theta >= theta_j and phi >= phi_j
    dw_last = 0.0
    hist, n_acc, n_tiny, status = [], 0, 0, "max_iter"
    n_resto = 0

    def barrier(xv, sv, fv=None):
        dl, du = sv - sL, sU - sv
        if (dl <= 0).any() or (du <= 0).any():
            return np.inf
        return (P.f(xv) if fv is None else fv) - mu * (np.log(dl).sum() + np.log(du).sum())

    def err(mu_):
        sd = max(smax, (np.abs(lam).sum() + np.abs(zL).sum() + np.abs(zU).sum()) / max(1, 3 * m)) / smax
        sc = max(smax, (np.abs(zL).sum() + np.abs(zU).sum()) / max(1, 2 * m)) / smax
        return max(_amax(g + J.T @ lam) / sd, _amax(-lam - zL + zU) / sd, _amax(c), _amax((s - sL) * zL - mu_) / sc, _amax((sU - s) * zU - mu_) / sc)

    for it in range(max_iter + 1):
        E0 = err(0.0)
        hist.append((it, P.f(x) / P.d_f, E0, np.abs(c).sum(), mu))
        if verbose and it % verbose == 0:
            print(f"  it {it:4d} f={P.f(x) / P.d_f:.10f} E0={E0:.2e} theta={np.abs(c).sum():.2e} mu={mu:.1e} dw={dw_last:.1e}")
        if E0 <= tol:
            status = "optimal"
            break
        n_acc = n_acc + 1 if E0 <= acceptable_tol else 0
        if n_acc >= acceptable_iter:
            status = "acceptable"
            break
        if it == max_iter:
            break
        while err(mu) <= k_eps * mu and mu > tol / 10.0:  # barrier problem solved: next mu, new filter [Alg. A, A-3]
            mu = max(tol / 10.0, min(k_mu * mu, mu**th_mu))
            tau = max(0.99, 1.0 - mu)
            filt = [(th_max, -np.inf)]
        # --- search direction [(13), 3.1] -----------------------------------------------------------------------------------
        dl, du = s - sL, sU - s
        Sig = zL / dl + zU / du
        W = P.hess(x, lam)
        rx = g + J.T @ lam
        rs = -lam - mu / dl + mu / du
        dw, dc, tries = 0.0, 0.0, 0
        while True:
            Dinv = 1.0 / (Sig + dw) + dc
            D = 1.0 / Dinv
            K = W + dw * np.eye(n) + (J.T * D) @ J
            try:
                cf = scipy.linalg.cho_factor(0.5 * (K + K.T), lower=True, check_finite=True)
                break
            except (np.linalg.LinAlgError, ValueError):
                pass
            tries += 1
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw *= 100.0 if dw_last == 0.0 else 8.0
            if dw > 1e40:
                status = "linear_algebra"
                break
        if status == "linear_algebra":
            break
        if dw > 0.0:
            dw_last = dw
        t = c + rs / (Sig + dw)
        dx = scipy.linalg.cho_solve(cf, -rx - J.T @ (D * t))
        dlam = D * (J @ dx + t)
        ds = (dlam - rs) / (Sig + dw)
        dzL = mu / dl - zL - (zL / dl) * ds
        dzU = mu / du - zU + (zU / du) * ds

        def frac(v, dv_):
            neg = dv_ < 0
            return min(1.0, float(np.min(-tau * v[neg] / dv_[neg]))) if neg.any() else 1.0

        a_max = min(frac(dl, ds), frac(du, -ds))
        a_z = min(frac(zL, dzL), frac(zU, dzU))
        # --- filter line search [2.3] ---------------------------------------------------------------------------------------
        theta = np.abs(c).sum()
        phi = barrier(x, s)
        dphi = float(g @ dx) - mu * float((ds / dl).sum()) + mu * float((ds / du).sum())
        if dphi < 0 and theta <= th_min:
            a_min = 0.05 * min(g_th, g_ph * theta / (-dphi) if theta > 0 else np.inf, dlt * theta**s_th / (-dphi) ** s_ph if theta > 0 else np.inf)
        elif dphi < 0:
            a_min = 0.05 * min(g_th, g_ph * theta / (-dphi))
        else:
            a_min = 0.05 * g_th
        a_min = max(a_min, 1e-14)

        def acceptable(th_t, ph_t, a):
            if not np.isfinite(ph_t) or th_t > th_max:
                return False, False
            if any(th_t >= tj and ph_t >= pj for tj, pj in filt):
                return False, False
            switch = dphi < 0 and a * (-dphi) ** s_ph > dlt * theta**s_th and theta <= th_min
            slack_ = 10.0 * np.finfo(float).eps * abs(phi)  # IPOPT compares barrier values up to 10 eps |phi| (Compare_le): near a solution
            if switch:                                       # the predicted decrease is below the rounding of phi itself
                return ph_t - phi <= eta * a * dphi + slack_, True
            return (th_t <= (1 - g_th) * theta) or (ph_t - phi <= -g_ph * theta + slack_), False

        a, accepted, ftype = a_max, False, False
        first = True
        # very small search directions [3.9]: a step below 10 eps relative is taken without a line search (the filter cannot tell the trial from
        # the current point); the second such step in a row means this barrier problem is solved as far as the arithmetic allows: next mu
        tiny = max(_amax(dx / (1.0 + np.abs(x))), _amax(ds / (1.0 + np.abs(s)))) < 10.0 * np.finfo(float).eps
        if tiny:
            x, s = x + a_max * dx, s + a_max * ds
            c = P.v(x) - s
            lam = lam + a_max * dlam
            zL, zU = zL + a_z * dzL, zU + a_z * dzU
            n_tiny += 1
            if n_tiny >= 2 and mu > tol / 10.0:
                mu = max(tol / 10.0, min(k_mu * mu, mu**th_mu))
                tau = max(0.99, 1.0 - mu)
                filt = [(th_max, -np.inf)]
                n_tiny = 0
            g, J = P.df(x), P.dv(x)
            dl, du = s - sL, sU - s
            zL = np.maximum(np.minimum(zL, k_sig * mu / dl), mu / (k_sig * dl))
            zU = np.maximum(np.minimum(zU, k_sig * mu / du), mu / (k_sig * du))
            continue
        n_tiny = 0
        while a >= a_min:
            xt, st = x + a * dx, s + a * ds
            ct = P.v(xt) - st
            th_t, ph_t = np.abs(ct).sum(), barrier(xt, st)
            ok, ftype = acceptable(th_t, ph_t, a)
            if ok:
                accepted = True
                break
            if first and th_t >= theta:  # second-order correction [2.4]
                c_soc, th_old = a * c + ct, theta
                for _ in range(4):
                    t2 = c_soc + rs / (Sig + dw)
                    dx2 = scipy.linalg.cho_solve(cf, -rx - J.T @ (D * t2))
                    dl2 = D * (J @ dx2 + t2)
                    ds2 = (dl2 - rs) / (Sig + dw)
                    a2 = min(frac(dl, ds2), frac(du, -ds2))
                    xs, ss = x + a2 * dx2, s + a2 * ds2
                    cs_ = P.v(xs) - ss
                    th_s, ph_s = np.abs(cs_).sum(), barrier(xs, ss)
                    ok, ftype = acceptable(th_s, ph_s, a2)
                    if ok:
                        xt, st, ct, th_t, ph_t, a, accepted = xs, ss, cs_, th_s, ph_s, a2, True
                        dlam = dl2
                        break
                    if th_s > 0.99 * th_old:
                        break
                    c_soc, th_old = a2 * c_soc + cs_, th_s
                if accepted:
                    break
            first = False
            a *= 0.5
        if accepted:
            if not ftype or not (ph_t - phi <= eta * a * dphi + 10.0 * np.finfo(float).eps * abs(phi)):  # augment the filter unless an f-type step with Armijo decrease [(22)]
                filt.append(((1 - g_th) * theta, phi - g_ph * theta))
            x, s, c = xt, st, ct
            lam = lam + a * dlam
            zL, zU = zL + a_z * dzL, zU + a_z * dzU
        else:
            # --- feasibility restoration (simplified, see the module docstring): Gauss-Newton on the rows that cannot be met by any slack
            #     inside its bounds, then the slacks are reset to the projection of v(x); the point enters the filter as in [3.3]
            filt.append(((1 - g_th) * theta, phi - g_ph * theta))
            n_resto += 1
            for _ in range(30):
                vx = P.v(x)
                viol = np.maximum(sL + 1e-9 - vx, 0.0) - np.maximum(vx - (sU - 1e-9), 0.0)
                if np.abs(viol).max() <= 1e-12:
                    break
                act = np.abs(viol) > 0
                Jx = P.dv(x)[act]
                step = np.linalg.lstsq(Jx, viol[act], rcond=1e-10)[0]
                b = 1.0
                while b > 1e-6:
                    v2 = P.v(x + b * step)
                    viol2 = np.maximum(sL + 1e-9 - v2, 0.0) - np.maximum(v2 - (sU - 1e-9), 0.0)
                    if np.abs(viol2).sum() < np.abs(viol).sum():
                        break
                    b *= 0.5
                x = x + b * step
            vx = P.v(x)
            s = np.minimum(np.maximum(vx, sL + np.minimum(pL, 1e-2 * mu + 1e-9)), sU - pU)
            c = vx - s
            zL, zU = np.maximum(zL, 1e-8), np.maximum(zU, 1e-8)
            if np.abs(c).sum() >= theta and np.abs(c).sum() > 1e-8:
                status = "restoration_failed"
                g, J = P.df(x), P.dv(x)
                break
        g, J = P.df(x), P.dv(x)
        # multiplier reset [(16)]
        dl, du = s - sL, sU - s
        zL = np.maximum(np.minimum(zL, k_sig * mu / dl), mu / (k_sig * dl))
        zU = np.maximum(np.minimum(zU, k_sig * mu / du), mu / (k_sig * du))
    # multipliers in the reference's form: L = f - lam_v^T v with lam_v >= 0 on v >= 0; here L = d_f f + lam^T (d_c v - s), stationarity in s gives
    # lam = -(zL - zU), so lam_v = -(d_c / d_f) lam
    lam_v = -(P.d_c / P.d_f) * lam
    return {"x": x, "f": nlp.f(x, P.p), "iters": it, "status": status, "E0": E0, "lam_v": lam_v, "mu": mu, "history": hist, "d_f": P.d_f, "n_restoration": n_resto}
