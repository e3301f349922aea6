"""ORACLE (test infrastructure, not product code) -- numpy port of the state machine the HIP path runs for the dense QP
family (optas_amd/csrc/oh_qp.hip):   min x^T P x + q^T x   s.t.  M x + c >= 0,  A x + b = 0   (optimization.py:219-260).
Infeasible-start primal-dual interior point, slacks s = Mx + c, reduced Newton system H = 2P + M^T (lam/s) M, Schur
complement on the equality rows.  Independent cross-checks: scipy SLSQP in the reference's wiring and, for the Booth
function of the reference's own solver test (tests/test_solver.py:22-54), the known answer (1, 3).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
"""
import numpy as np


def solve_qp_ipm(P, q, M, c, A, b, x0=None, tol=1e-9, max_iter=100):
    P, q = np.atleast_2d(np.asarray(P, dtype=float)), np.asarray(q, dtype=float).reshape(-1)
    n = q.shape[0]
    M = np.asarray(M, dtype=float).reshape(-1, n)
    A = np.asarray(A, dtype=float).reshape(-1, n)
    c, b = np.asarray(c, dtype=float).reshape(-1), np.asarray(b, dtype=float).reshape(-1)
    m, me = M.shape[0], A.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=float)
    mu = 1.0
    s = np.maximum(M @ x + c, 1.0)
    lam = mu / s
    nu = np.zeros(me)
    status, it = 1, 0
    while True:
        rd = 2.0 * P @ x + q - M.T @ lam - A.T @ nu
        rp = M @ x + c - s
        re = A @ x + b
        stat = np.abs(rd).max()
        feas = max(np.abs(rp).max() if m else 0.0, np.abs(re).max() if me else 0.0)
        gap = (s * lam).max() if m else 0.0
        if not np.isfinite(stat) or not np.isfinite(feas):
            status = 2
            break
        if stat <= tol and feas <= tol and gap <= tol:
            status = 0
            break
        if it == max_iter:
            break
        H = P + P.T + M.T @ ((lam / s)[:, None] * M)
        shift = 1e-13 * max(np.abs(np.diag(H)).max(), 1.0)
        rhs = -rd + M.T @ ((mu / s - lam) - (lam / s) * rp)
        L = None
        for attempt in range(8):  # the kernel rebuilds H and retries with a 1000x larger shift
            if attempt:
                shift *= 1e3
            try:
                L = np.linalg.cholesky(H + shift * np.eye(n))
                break
            except np.linalg.LinAlgError:
                pass
        if L is None:
            status = 2
            break
        solve = lambda v: np.linalg.solve(L.T, np.linalg.solve(L, v))
        dx = solve(rhs)
        dnu = np.zeros(me)
        if me:
            Y = np.stack([solve(A[i]) for i in range(me)])
            S = A @ Y.T
            S[np.diag_indices(me)] += 1e-14 * np.maximum(1.0, np.diag(S))
            dnu = np.linalg.solve(S, -re - A @ dx)
            dx = dx + Y.T @ dnu
        ds = M @ dx + rp
        dl = (mu / s - lam) - (lam / s) * ds
        ap = min([1.0] + [-0.995 * s[i] / ds[i] for i in range(m) if ds[i] < 0])
        ad = min([1.0] + [-0.995 * lam[i] / dl[i] for i in range(m) if dl[i] < 0])
        x = x + ap * dx
        s = s + ap * ds
        lam = lam + ad * dl
        nu = nu + ad * dnu
        if m:
            am = min(ap, ad)
            sigma = 0.1 if am > 0.9 else (0.3 if am > 0.5 else 0.8)
            mu = max(sigma * float(s @ lam) / m, 1e-2 * tol)
        it += 1
    return {"x": x, "f": float(x @ P @ x + q @ x), "lam": lam, "nu": nu, "iters": it, "status": status, "kkt": (stat, feas, gap)}
