"""ORACLE (test infrastructure, not product code) -- numpy port of the HIP point-mass MPC kernel
(optas_amd/csrc/oh_pointmass.hip): primal-dual interior point on the stage form of
example/point_mass_mpc.py's Controller problem, Newton steps by a Riccati sweep.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Cross-check of its answers: oracle.solvers.scipy_minimize (SLSQP in the reference's v >= 0 wiring) and
kkt_reference_form on oracle.problems.PointMassMPCNLP.

Stage form (linear rows eliminated exactly): state x_t = (y_t, v_t), v = dy, control a_t = (v_{t+1}-v_t)/dt,
  y_{t+1} = y_t + dt v_t,  v_{t+1} = v_t + dt a_t,  x_0 = (curr, dcurr);
  f = sum_t ||goal_t - y_t||^2 + w sum_t ||a_t||^2,  w = 0.0025/T   (point_mass_mpc.py:130-136)
  inequalities on x_t, t >= 1: y in [-ylim, ylim]^2, v in [-vlim, vlim]^2, ||obs_t - y_t||^2 >= safe^2 (:110-127).
The obstacle row is nonconvex; its (negative) curvature -2 lam I is dropped from the Hessian (convexified
Newton step) -- the linearised row is an inner approximation, so the iteration stays well posed.
"""
import numpy as np


def solve_pointmass_ipm(T, dt, w, ylim, vlim, safe_sq, curr, dcurr, goal, obs, V0=None, tol=1e-8, max_iter=100, verbose=False,
                        track_final_only=False, w_vel=0.0, fix_final_velocity=False):
    """goal, obs: (2, T).  V0: optional (2, T) velocity seed.  Returns dict(Y, V, f, iters, kkt=(stat, feas, compl), status).
    The three keyword options turn the MPC tick (point_mass_mpc.py) into the planner of example/point_mass_planner.py:17-55: tracking
    cost on the last knot only, w_vel * sum ||dy_t||^2 over all knots, and the terminal row dy_{T-1} = 0, which pins the last control to
    a_{T-2} = -v_{T-2} / dt (a fixed feedback in the Riccati recursion instead of an optimised one)."""
    nI = 9  # rows per stage: 4 y-box, 4 v-box, 1 obstacle
    A = np.eye(4)
    A[0, 2] = A[1, 3] = dt
    B = np.zeros((4, 2))
    B[2, 0] = B[3, 1] = dt
    R = 2.0 * w * np.eye(2)
    V = np.zeros((2, T)) if V0 is None else np.array(V0, dtype=float)
    V[:, 0] = dcurr
    if fix_final_velocity:
        V[:, T - 1] = 0.0
    a = (V[:, 1:] - V[:, :-1]) / dt  # (2, T-1)
    wt = np.ones(T)
    if track_final_only:
        wt[: T - 1] = 0.0
    Kfix = np.zeros((2, 4))
    Kfix[0, 2] = Kfix[1, 3] = -1.0 / dt

    def rollout(a):
        X = np.zeros((4, T))
        X[:2, 0], X[2:, 0] = curr, dcurr
        for t in range(T - 1):
            X[:, t + 1] = A @ X[:, t] + B @ a[:, t]
        return X

    def cons(X):
        """c (nI, T) and Jacobian rows J (nI, T, 4) wrt x_t."""
        c = np.zeros((nI, T))
        J = np.zeros((nI, T, 4))
        y, v = X[:2], X[2:]
        for j in range(2):
            c[2 * j] = y[j] + ylim
            J[2 * j, :, j] = 1.0
            c[2 * j + 1] = ylim - y[j]
            J[2 * j + 1, :, j] = -1.0
            c[4 + 2 * j] = v[j] + vlim
            J[4 + 2 * j, :, 2 + j] = 1.0
            c[4 + 2 * j + 1] = vlim - v[j]
            J[4 + 2 * j + 1, :, 2 + j] = -1.0
        d = y - obs
        c[8] = np.sum(d * d, axis=0) - safe_sq
        J[8, :, 0], J[8, :, 1] = 2.0 * d[0], 2.0 * d[1]
        return c, J

    X = rollout(a)
    c, J = cons(X)
    mu = 0.1
    s = np.maximum(c, 1e-2)
    lam = mu / s
    status = 1
    it = 0
    for it in range(max_iter + 1):
        c, J = cons(X)
        rc = c - s
        # cost gradient wrt x_t: -2 (goal - y)
        gx = np.zeros((4, T))
        gx[:2] = -2.0 * wt[None] * (goal - X[:2])
        gx[2:] = 2.0 * w_vel * X[2:]
        # ---- KKT residuals (adjoint pass for the control gradient of the Lagrangian)
        lx = gx - np.einsum("itk,it->kt", J, lam)  # d/dx_t of f - lam^T c
        padj = lx[:, T - 1].copy()
        stat = 0.0
        for t in range(T - 2, -1, -1):
            gu = 2.0 * w * a[:, t] + B.T @ padj
            if fix_final_velocity and t == T - 2:
                padj = lx[:, t] + A.T @ padj + Kfix.T @ gu  # the last control is a function of x_{T-2}: its gradient flows into the state
            else:
                stat = max(stat, float(np.max(np.abs(gu))))
                padj = lx[:, t] + A.T @ padj
        feas = float(np.max(np.abs(rc[:, 1:])))
        compl = float(np.max(lam[:, 1:] * s[:, 1:]))
        fval = float(np.sum(wt[None] * (goal - X[:2]) ** 2) + w_vel * np.sum(X[2:] ** 2) + w * np.sum(a * a))
        if verbose:
            print(f"  it {it:3d} f={fval:.10f} stat={stat:.2e} feas={feas:.2e} compl={compl:.2e} mu={mu:.2e}")
        if stat <= tol and feas <= tol and compl <= tol:
            status = 0
            break
        if compl > 1e6 and feas > 1e3 * tol:  # no feasible plan: the residual of the slacks stalls, the multipliers diverge (csrc/oh_pointmass.hip, round 6)
            status = 3
            break
        if it == max_iter:
            break
        # ---- Newton step: barrier-modified LQR solved by Riccati
        sig = lam / s
        Q = np.zeros((T, 4, 4))
        q = np.zeros((T, 4))
        for t in range(1, T):
            Q[t, 0, 0] = Q[t, 1, 1] = 2.0 * wt[t]
            Q[t, 2, 2] = Q[t, 3, 3] = 2.0 * w_vel
            Q[t] += np.einsum("ik,i,il->kl", J[:, t], sig[:, t], J[:, t])
            q[t] = gx[:, t] - J[:, t].T @ (mu / s[:, t] - sig[:, t] * rc[:, t])
        P = Q[T - 1].copy()
        p = q[T - 1].copy()
        K = np.zeros((T - 1, 2, 4))
        k = np.zeros((T - 1, 2))
        for t in range(T - 2, -1, -1):
            Quu = R + B.T @ P @ B
            Qux = B.T @ P @ A
            qu = 2.0 * w * a[:, t] + B.T @ p
            if fix_final_velocity and t == T - 2:
                K[t], k[t] = Kfix, np.zeros(2)
                Pn = Q[t] + A.T @ P @ A + Qux.T @ Kfix + Kfix.T @ Qux + Kfix.T @ Quu @ Kfix
                p = q[t] + A.T @ p + Kfix.T @ qu
                P = 0.5 * (Pn + Pn.T)
                continue
            L = np.linalg.cholesky(Quu)
            K[t] = -np.linalg.solve(L.T, np.linalg.solve(L, Qux))
            k[t] = -np.linalg.solve(L.T, np.linalg.solve(L, qu))
            Pn = Q[t] + A.T @ P @ A + Qux.T @ K[t]
            p = q[t] + A.T @ p + Qux.T @ k[t]
            P = 0.5 * (Pn + Pn.T)
        dX = np.zeros((4, T))
        da = np.zeros((2, T - 1))
        for t in range(T - 1):
            da[:, t] = K[t] @ dX[:, t] + k[t]
            dX[:, t + 1] = A @ dX[:, t] + B @ da[:, t]
        ds = np.einsum("itk,kt->it", J, dX) + rc
        dlam = (mu / s - lam) - sig * ds
        # fraction to the boundary (stages t >= 1 only; stage 0 rows are constants)
        tau = 0.995

        def ftb(z, dz):
            m = dz < 0
            return min(1.0, float(np.min(-tau * z[m] / dz[m]))) if np.any(m) else 1.0

        ap = ftb(s[:, 1:], ds[:, 1:])
        ad = ftb(lam[:, 1:], dlam[:, 1:])
        a = a + ap * da
        X = rollout(a)
        s[:, 1:] += ap * ds[:, 1:]
        lam[:, 1:] += ad * dlam[:, 1:]
        # barrier update: centrality-driven
        gap = float(np.sum(s[:, 1:] * lam[:, 1:])) / (nI * (T - 1))
        sigma = 0.1 if min(ap, ad) > 0.9 else (0.3 if min(ap, ad) > 0.5 else 0.8)
        mu = max(sigma * gap, 1e-2 * tol)
    return {"Y": X[:2], "V": X[2:], "f": fval, "iters": it, "kkt": (stat, feas, compl), "status": status, "lam": lam, "s": s}
