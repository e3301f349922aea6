"""ORACLE (test infrastructure, not product code) -- numpy port of the state machine the HIP path runs for the
inverse-kinematics family (BASELINE config 1, example/example.py:13-60):

    min_q  w ||q - qN||^2   s.t.  h(q) = p_goal - p_link(q) = 0  (builder.py:354 sign),   lo <= q <= up  (builder.py:471-509)

Bound-constrained augmented Lagrangian (Hestenes/Powell multiplier update) with a projected Newton inner
iteration (Bertsekas' active-set projection, exact Hessian of the augmented Lagrangian, Levenberg shift when the
reduced matrix is not positive definite, Armijo backtracking on the projected arc).  The independent cross-check is
scipy SLSQP on oracle.problems.IKExampleNLP wired as the reference's ScipyMinimizeSolver (solver.py:652-679).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
"""
import numpy as np

from .structured import FoldedChain


def _pos_jac_frames(chain: FoldedChain, q):
    e, _, z, pj = chain.fk(q[None])
    e, z, pj = e[0], z[0], pj[0]
    n = chain.ndof
    Jp = np.zeros((3, n))
    om = np.zeros((n, 3))  # angular rate per unit joint rate (0 for prismatic)
    for k in range(chain.n_chain):
        c = chain.qidx[k]
        if chain.jtype[k] == 0:
            Jp[:, c] = np.cross(z[k], e - pj[k])
            om[c] = z[k]
        else:
            Jp[:, c] = z[k]
    return e, Jp, om


def position_curvature(chain: FoldedChain, Jp, om, y):
    """C[a,b] = sum_k y_k d^2 e_k / dq_a dq_b = y . (om_a x Jp_b) for a <= b in chain order (symmetric)."""
    n = chain.ndof
    C = np.zeros((n, n))
    order = [chain.qidx[k] for k in range(chain.n_chain)]
    for ia, a in enumerate(order):
        for b in order[ia:]:
            C[a, b] = C[b, a] = float(y @ np.cross(om[a], Jp[:, b]))
    return C


def solve_ik_al(chain: FoldedChain, x0, qn, pg, lo, up, w=1.0, tol=1e-8, tol_feas=1e-10, max_iter=200, rho0=100.0, trace=None):
    """Returns dict(x, f, lam_h (signed multiplier of the h rows in the reference's v >= 0 form), z_lo, z_up,
    stationarity, feasibility, iterations (FK evaluations of accepted + rejected points), status)."""
    n = chain.ndof
    q = np.minimum(np.maximum(np.asarray(x0, dtype=np.float64), lo), up)
    lam = np.zeros(3)
    rho = rho0
    it = 0
    status = 1

    def merit(qq, e):
        hh = pg - e
        return w * np.sum((qq - qn) ** 2) + lam @ hh + 0.5 * rho * hh @ hh

    e, Jp, om = _pos_jac_frames(chain, q)
    it += 1
    h_prev = np.inf
    shift = 0.0
    while it < max_iter:
        # ---- inner: projected Newton on the augmented Lagrangian ----
        while it < max_iter:
            h = pg - e
            y = lam + rho * h
            grad = 2.0 * w * (q - qn) - Jp.T @ y
            act = ((q <= lo) & (grad > 0.0)) | ((q >= up) & (grad < 0.0))
            pgn = np.max(np.abs(np.where(act, 0.0, grad)))
            # inner tolerance tightens with feasibility (no point in polishing far from the manifold)
            if pgn <= max(tol * 0.5, min(1e-2, 0.1 * np.abs(h).max())):
                break
            H = 2.0 * w * np.eye(n) + rho * Jp.T @ Jp - position_curvature(chain, Jp, om, y)
            free = ~act
            while True:
                Hs = H + shift * np.eye(n)
                Hs[act, :] = 0.0
                Hs[:, act] = 0.0
                Hs[act, act] = 1.0
                try:
                    L = np.linalg.cholesky(Hs)
                    break
                except np.linalg.LinAlgError:
                    shift = max(10.0 * shift, 1e-3 * rho)
            d = -np.linalg.solve(L.T, np.linalg.solve(L, np.where(free, grad, 0.0)))
            m0 = merit(q, e)
            alpha = 1.0
            ok = False
            for _ in range(30):
                qt = np.minimum(np.maximum(q + alpha * d, lo), up)
                et, Jpt, omt = _pos_jac_frames(chain, qt)
                it += 1
                if merit(qt, et) <= m0 + 1e-4 * grad @ (qt - q) + 4e-16 * max(1.0, abs(m0)) + 8e-16 * np.abs(y).sum():  # (rounding of the merit and of e through y)
                    ok = True
                    break
                alpha *= 0.5
                if it >= max_iter:
                    break
            if not ok:
                shift = max(10.0 * shift, 1e-3 * rho)
                if shift > 1e12 * rho:
                    status = 2
                    break
                continue
            shift *= 0.1 if shift > 1e-12 else 0.0
            q, e, Jp, om = qt, et, Jpt, omt
            if trace is not None:
                trace.append((it, float(pgn), float(np.abs(pg - e).max()), rho))
        if status == 2:
            break
        # ---- outer: multiplier update ----
        h = pg - e
        lam = lam + rho * h
        hn = np.abs(h).max()
        grad = 2.0 * w * (q - qn) - Jp.T @ lam
        act = ((q <= lo) & (grad > 0.0)) | ((q >= up) & (grad < 0.0))
        stat = np.max(np.abs(np.where(act, 0.0, grad)))
        if hn <= tol_feas and stat <= tol:
            status = 0
            break
        if hn > 0.1 * h_prev:
            rho = min(rho * 10.0, 1e8)
        h_prev = hn
    h = pg - e
    grad = 2.0 * w * (q - qn) - Jp.T @ lam
    at_lo, at_up = (q <= lo) & (grad > 0.0), (q >= up) & (grad < 0.0)
    z_lo, z_up = np.where(at_lo, grad, 0.0), np.where(at_up, -grad, 0.0)
    stat = np.max(np.abs(np.where(at_lo | at_up, 0.0, grad)))
    return dict(x=q, f=float(w * np.sum((q - qn) ** 2)), lam_h=-lam, z_lo=z_lo, z_up=z_up, stationarity=float(stat),
                feasibility=float(np.abs(h).max()), iterations=it, status=status)
