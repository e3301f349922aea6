"""ORACLE (test infrastructure, not product code) -- CPU solvers for the restated NLPs.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

The reference's hot path is ``CasADiSolver._solve`` -> ``casadi.nlpsol("ipopt")`` (optas/solver.py:386-398).
casadi (unpinned, setup.py:22; bundles IPOPT+MUMPS) is absent from this image and from the GPU box, so
the IPOPT iterates cannot be reproduced: PARITY UNPINNED for solver output on the KUKA problems.  What
stands in for it:

* ``scipy_minimize``: the reference's *own* alternative backend, ``ScipyMinimizeSolver`` wired as at
  solver.py:652-712 (SLSQP: ``v(x) >= 0`` as one "ineq" block with Jacobian dv; trust-constr: k/a/g/h
  split).  scipy *is* in this image.  Pinned by the reference's Booth known answer
  (tests/test_solver.py:46-54 -> x=1, y=3).
* ``dense_sqp``: an independent dense null-space Newton-SQP on the literal x layout (rank-revealing,
  so the rank-3 four-row quaternion equality is handled as written).  It shares no structure with the
  banded/Riccati HIP path, which is the point.
* ``kkt_reference_form``: evaluates the KKT residuals of ``min f s.t. 0 <= v(x) <= 1e10`` -- the form IPOPT
  is given (solver.py:355-363,391-395) -- at any candidate x*, with multipliers from non-negative least
  squares.  This is what "KKT residual vs IPOPT" is reported on.
"""
import numpy as np
import scipy.linalg
import scipy.optimize

INF = 1.0e10  # optimization.py:58


def scipy_minimize(nlp, x0, p, method="SLSQP", tol=None, options=None):
    """ScipyMinimizeSolver.setup/_solve (solver.py:619-714,786-792)."""
    kw = {"fun": lambda x: nlp.f(x, p), "method": method, "x0": np.asarray(x0, float), "jac": lambda x: nlp.df(x, p)}
    if tol is not None:
        kw["tol"] = tol
    if options is not None:
        kw["options"] = options
    if nlp.nv > 0:
        if method != "trust-constr":
            kw["constraints"] = [{"type": "ineq", "fun": lambda x: nlp.v(x, p), "jac": lambda x: nlp.dv(x, p)}]
        else:
            cons = []
            x_zero = np.zeros(nlp.nx)
            if nlp.nk:
                cons.append(scipy.optimize.LinearConstraint(nlp.dk(x_zero, p), -nlp.k(x_zero, p), INF * np.ones(nlp.nk)))
            if nlp.na:
                eq = -nlp.a(x_zero, p)
                cons.append(scipy.optimize.LinearConstraint(nlp.da(x_zero, p), eq, eq))
            if nlp.ng:
                cons.append(scipy.optimize.NonlinearConstraint(lambda x: nlp.g(x, p), np.zeros(nlp.ng), INF * np.ones(nlp.ng), jac=lambda x: nlp.dg(x, p)))
            if nlp.nh:
                cons.append(scipy.optimize.NonlinearConstraint(lambda x: nlp.h(x, p), np.zeros(nlp.nh), np.zeros(nlp.nh), jac=lambda x: nlp.dh(x, p)))
            kw["constraints"] = cons
    if method in {"Newton-CG", "dogleg", "trust-ncg", "trust-krylov", "trust-exact", "trust-constr"} and hasattr(nlp, "ddf"):
        kw["hess"] = lambda x: nlp.ddf(x, p)
    return scipy.optimize.minimize(**kw)


def _null_space_and_particular(A, rhs, rcond=1e-9):
    """Rank-revealing: returns (Z, dp, rank) with A dp = rhs in the least-squares sense, A Z = 0."""
    U, s, Vt = np.linalg.svd(A, full_matrices=True)
    tol = rcond * (s[0] if s.size else 1.0)
    r = int(np.sum(s > tol))
    dp = Vt[:r].T @ ((U[:, :r].T @ rhs) / s[:r])
    Z = Vt[r:].T
    return Z, dp, r


def dense_sqp(nlp, x0, p, max_iter=100, tol=1e-10, gauss_newton=False, verbose=False):
    """Equality-constrained dense Newton-SQP (problems without k/g rows, e.g. FigureEightNLP).

    Equalities c(x) = [a; h] = 0 are kept exactly as the reference states them; the redundant rows
    (rank-3 quaternion blocks) are resolved by an SVD null-space split.  l1 merit, backtracking.
    Returns dict(x, f, iters, kkt_stat, feas, lam_a, lam_h, converged, history).
    """
    assert nlp.nk == 0 and nlp.ng == 0
    x = np.asarray(x0, float).copy()
    lam = np.zeros(nlp.na + nlp.nh)
    nu = 1.0
    hist = []
    converged = False
    for it in range(max_iter + 1):
        g = nlp.df(x, p)
        c = np.concatenate([nlp.a(x, p), nlp.h(x, p)])
        A = np.concatenate([nlp.da(x, p), nlp.dh(x, p)], axis=0)
        # least-squares multipliers for the stationarity measure: g + A^T lam = 0
        lam_ls = np.linalg.lstsq(A.T, -g, rcond=1e-9)[0]
        stat = float(np.max(np.abs(g + A.T @ lam_ls)))
        feas = float(np.max(np.abs(c))) if c.size else 0.0
        fval = nlp.f(x, p)
        hist.append((it, fval, stat, feas))
        if verbose:
            print(f"  it {it:3d} f={fval:.12f} stat={stat:.3e} feas={feas:.3e}")
        if stat <= tol * max(1.0, np.max(np.abs(lam_ls)) if lam_ls.size else 1.0) and feas <= tol:
            lam = lam_ls
            converged = True
            break
        if it == max_iter:
            lam = lam_ls
            break
        lam = lam_ls
        H = nlp.hess_lagrangian(x, p, lam[nlp.na :], gauss_newton=gauss_newton)
        Z, dp, _ = _null_space_and_particular(A, -c)
        Hr = Z.T @ H @ Z
        Hr = 0.5 * (Hr + Hr.T)
        rhs = -Z.T @ (g + H @ dp)
        # inertia correction: shift until positive definite
        shift = 0.0
        while True:
            try:
                cf = scipy.linalg.cho_factor(Hr + shift * np.eye(Hr.shape[0]))
                break
            except np.linalg.LinAlgError:
                shift = max(10.0 * shift, 1e-6 * max(1.0, np.max(np.abs(np.diag(Hr)))))
        dz = scipy.linalg.cho_solve(cf, rhs)
        d = dp + Z @ dz
        # l1 merit line search
        nu = max(nu, 2.0 * float(np.max(np.abs(lam))) if lam.size else nu)
        phi0 = fval + nu * np.sum(np.abs(c))
        dphi = float(g @ d) - nu * np.sum(np.abs(c))
        alpha = 1.0
        while True:
            xt = x + alpha * d
            ct = np.concatenate([nlp.a(xt, p), nlp.h(xt, p)])
            phit = nlp.f(xt, p) + nu * np.sum(np.abs(ct))
            if phit <= phi0 + 1e-4 * alpha * min(dphi, 0.0) or alpha < 1e-8:
                break
            alpha *= 0.5
        x = xt
    return {
        "x": x,
        "f": nlp.f(x, p),
        "iters": it,
        "kkt_stat": stat,
        "feas": feas,
        "lam_a": lam[: nlp.na],
        "lam_h": lam[nlp.na :],
        "converged": converged,
        "history": hist,
    }


def kkt_reference_form(nlp, x, p, active_tol=1e-6, lam_kg=None):
    """KKT residuals of the problem exactly as CasADiSolver poses it: min f s.t. 0 <= v(x,p) <= 1e10
    (solver.py:346-363), v = [k; g; a; -a; h; -h], multipliers lam >= 0 on every row.

    An equality row e appears twice (e, -e) with multipliers (lam+, lam-); only mu = lam+ - lam- enters
    stationarity, so mu is fitted as a free variable and split as lam+ = max(mu,0), lam- = max(-mu,0).
    Inequality multipliers are fitted with a lower bound of 0 on the numerically active rows only.
    Returns dict(stationarity, feasibility, complementarity, lam) with lam in v's row order.

    lam_kg: multipliers >= 0 of the inequality rows [k; g] that came with x (an interior-point answer: every row carries lam_i = mu / v_i, none is
    "active" to a tolerance); they are taken as given and only the equality multipliers are fitted.
    """
    g = nlp.df(x, p)
    if nlp.nv == 0:
        return {"stationarity": float(np.max(np.abs(g))), "feasibility": 0.0, "complementarity": 0.0, "lam": np.zeros(0)}
    kg = np.concatenate([nlp.k(x, p), nlp.g(x, p)])
    Jkg = np.concatenate([nlp.dk(x, p), nlp.dg(x, p)], axis=0)
    e = np.concatenate([nlp.a(x, p), nlp.h(x, p)])
    Je = np.concatenate([nlp.da(x, p), nlp.dh(x, p)], axis=0)
    v = nlp.v(x, p)
    feas = float(max(0.0, -np.min(v)))
    if lam_kg is not None:
        lam_kg = np.asarray(lam_kg, dtype=float).reshape(-1)
        assert lam_kg.shape == kg.shape and lam_kg.min() >= 0.0
        mu = np.linalg.lstsq(Je.T, g - Jkg.T @ lam_kg, rcond=1e-12)[0] if e.size else np.zeros(0)
        stat = float(np.max(np.abs(g - Jkg.T @ lam_kg - (Je.T @ mu if e.size else 0.0))))
        mu_a, mu_h = mu[: nlp.na], mu[nlp.na :]
        lam = np.concatenate([lam_kg, np.maximum(mu_a, 0), np.maximum(-mu_a, 0), np.maximum(mu_h, 0), np.maximum(-mu_h, 0)])
        return {"stationarity": stat, "feasibility": feas, "complementarity": float(np.max(np.abs(lam * v))), "lam": lam, "mu_a": mu_a, "mu_h": mu_h}
    act = np.where(kg <= active_tol * max(1.0, float(np.max(np.abs(kg))) if kg.size else 1.0))[0]
    M = np.concatenate([Jkg[act], Je], axis=0).T  # nx x (nact + ne)
    if act.size:
        lb = np.concatenate([np.zeros(act.size), -np.inf * np.ones(e.size)])
        y = scipy.optimize.lsq_linear(M, g, bounds=(lb, np.inf), method="bvls" if M.shape[1] <= M.shape[0] else "trf").x
    else:
        y = np.linalg.lstsq(M, g, rcond=1e-10)[0]
    lam_kg = np.zeros(kg.size)
    lam_kg[act] = y[: act.size]
    mu = y[act.size :]
    stat = float(np.max(np.abs(g - M @ y)))
    mu_a, mu_h = mu[: nlp.na], mu[nlp.na :]
    lam = np.concatenate(
        [lam_kg, np.maximum(mu_a, 0), np.maximum(-mu_a, 0), np.maximum(mu_h, 0), np.maximum(-mu_h, 0)]
    )
    comp = float(np.max(np.abs(lam * v)))
    return {"stationarity": stat, "feasibility": feas, "complementarity": comp, "lam": lam, "mu_a": mu_a, "mu_h": mu_h}
