"""ORACLE (test infrastructure, not product code) -- numpy port of the *structured* algorithm the HIP
path runs for the figure-eight family (velocity-condensed, per-knot null-space, block-tridiagonal
Cholesky == Riccati sweep).  Used (a) to check the HIP kernels stage-by-stage and end-to-end and (b) as
the "port" CPU baseline in bench.py.  Independent cross-check of its answers: oracle/solvers.dense_sqp
and kkt_reference_form on the literal reference layout.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Problem (SURVEY App. B.2 / example/figure_eight_plan.py:64-107), after eliminating the linear rows:
  q_0 = q_1 = qc (fix_configuration + zero initial velocity + Euler integration, builder.py:437,525-539),
  dq_t = (q_{t+1}-q_t)/dt,
  min_{q_2..q_{T-1}}  sum_t w_p ||path_t - p(q_t)||^2 + (w_v/dt^2) sum_{t=1}^{T-2} ||q_{t+1}-q_t||^2
  s.t. R(q_t) = R(qc)   (same feasible set as quat(q_t) = quat_c on the connected branch).
"""
import numpy as np

from .robot import OracleRobot
from .spatialmath import rpy2r, unit


class FoldedChain:
    """Per-actuated-joint constants with fixed joints folded in (what oh_set_constants receives)."""

    def __init__(self, robot: OracleRobot, link: str):
        root = robot.get_root()
        R_acc, p_acc = np.eye(3), np.zeros(3)
        self.R0, self.p0, self.axis, self.jtype, self.qidx = [], [], [], [], []
        # link name -> (index k of the last actuated chain joint before it, or -1; position of the link origin in the
        # frame that follows joint k's rotation): p_link = p_k + R_k off  (sphere centres, builder.py:366-417)
        self.attach = {root: (-1, np.zeros(3))}
        for name in robot.get_chain(root, link):
            j = robot.joint_map[name]
            xyz, rpy = robot.get_joint_origin(j)
            Rj = rpy2r(rpy)
            p_acc = p_acc + R_acc @ xyz
            R_acc = R_acc @ Rj
            if j.type == "fixed":
                self.attach[j.child] = (len(self.R0) - 1, p_acc.copy())
                continue
            self.attach[j.child] = (len(self.R0), np.zeros(3))
            self.R0.append(R_acc)
            self.p0.append(p_acc)
            self.axis.append(robot.get_joint_axis(j))
            self.jtype.append(0 if j.type in {"revolute", "continuous"} else 1)
            self.qidx.append(robot.get_actuated_joint_index(j.name))
            R_acc, p_acc = np.eye(3), np.zeros(3)
        self.R_tool, self.p_tool = R_acc, p_acc
        self.n_chain = len(self.R0)
        self.ndof = robot.ndof

    def link_positions(self, Q, links):
        """Positions (N, L, 3) and linear Jacobians (N, L, 3, ndof) of the origins of the named links (on this chain)."""
        Q = np.atleast_2d(Q)
        N = Q.shape[0]
        e, Re, z, pj, frames = self.fk(Q, frames=True)
        C = np.zeros((N, len(links), 3))
        J = np.zeros((N, len(links), 3, self.ndof))
        for li, name in enumerate(links):
            k, off = self.attach[name]
            if k < 0:
                C[:, li] = off
                continue
            Rk, pk = frames[k]
            C[:, li] = pk + Rk @ off
            for j in range(k + 1):
                c = self.qidx[j]
                J[:, li, :, c] = np.cross(z[:, j], C[:, li] - pj[:, j]) if self.jtype[j] == 0 else z[:, j]
        return C, J

    def fk(self, Q, frames=False):
        """Q: (N, ndof).  Returns e (N,3), R (N,3,3), z (N,nc,3), pj (N,nc,3) [, per-joint (R, p) after the joint motion]."""
        Q = np.atleast_2d(Q)
        fr = []
        N = Q.shape[0]
        R = np.tile(np.eye(3), (N, 1, 1))
        p = np.zeros((N, 3))
        zs, ps = [], []
        for k in range(self.n_chain):
            p = p + R @ self.p0[k]
            R = R @ self.R0[k]
            a = self.axis[k]
            qk = Q[:, self.qidx[k]]
            z = R @ a
            if self.jtype[k] == 0:
                K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0.0]])
                Rq = np.eye(3)[None] + np.sin(qk)[:, None, None] * K[None] + (1 - np.cos(qk))[:, None, None] * (K @ K)[None]
                R = R @ Rq
                ps.append(p.copy())
            else:
                ps.append(p.copy())
                p = p + z * qk[:, None]
            zs.append(z)
            fr.append((R.copy(), p.copy()))
        e = p + R @ self.p_tool
        Re = R @ self.R_tool
        if frames:
            return e, Re, np.stack(zs, 1), np.stack(ps, 1), fr
        return e, Re, np.stack(zs, 1), np.stack(ps, 1)

    def jac(self, Q):
        e, Re, z, pj = self.fk(Q)
        N = e.shape[0]
        Jp = np.zeros((N, 3, self.ndof))
        Jw = np.zeros((N, 3, self.ndof))
        for k in range(self.n_chain):
            c = self.qidx[k]
            if self.jtype[k] == 0:
                Jp[:, :, c] = np.cross(z[:, k], e - pj[:, k])
                Jw[:, :, c] = z[:, k]
            else:
                Jp[:, :, c] = z[:, k]
        return e, Re, Jp, Jw


def _vee_skew(A):
    return 0.5 * np.stack([A[:, 2, 1] - A[:, 1, 2], A[:, 0, 2] - A[:, 2, 0], A[:, 1, 0] - A[:, 0, 1]], 1)


class StructuredFigureEight:
    def __init__(self, robot, link, T=50, Tmax=10.0, w_path=1000.0, w_vel=0.01):
        self.chain = FoldedChain(robot, link)
        self.T, self.n = T, robot.ndof
        ts = np.linspace(0.0, Tmax, T)
        self.dt = float(ts[1] - ts[0])
        self.local_path = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)], 1)  # (T,3)
        self.w_path, self.w_vel = w_path, w_vel
        self.kappa = w_vel / self.dt**2

    def references(self, qc):
        e, Re, _, _ = self.chain.fk(qc[None])
        return e[0] + self.local_path @ Re[0].T, Re[0]

    def evaluate(self, Q, path, Rc, lam=None, exact=False):
        """Stage quantities at knots Q (T,n): cost pieces, gradient, Hessian blocks, constraint, Jacobian."""
        e, Re, Jp, Jw = self.chain.jac(Q)
        r = path - e
        phi = self.w_path * np.sum(r * r, 1)
        g = -2.0 * self.w_path * np.einsum("tki,tk->ti", Jp, r)
        W = 2.0 * self.w_path * np.einsum("tki,tkj->tij", Jp, Jp)
        A = Re @ Rc.T
        c = _vee_skew(A)
        trA = np.trace(A, axis1=1, axis2=2)
        M = 0.5 * (trA[:, None, None] * np.eye(3)[None] - A)
        Jc = M @ Jw
        if exact:
            n = self.n
            # -2 w_p sum_k r_k d2p_k/dqi dqj ,  d2p/dqi dqj = z_i x Jp_j (i<=j)
            zr = np.cross(r[:, None, :], np.swapaxes(Jw, 1, 2))  # (T,n,3) = r x z_i
            S = np.einsum("tik,tkj->tij", zr, Jp)  # (r x z_i) . Jp_j
            S = np.triu(S) + np.swapaxes(np.triu(S, 1), 1, 2)
            W = W - 2.0 * self.w_path * S
            if lam is not None:
                # lam . d2c/dqi dqj  ~  1/2 lam . (z_i x z_j)  (i<j), exact at feasible points
                zz = np.cross(np.swapaxes(Jw, 1, 2)[:, :, None, :], np.swapaxes(Jw, 1, 2)[:, None, :, :])  # (T,n,n,3)
                C = 0.5 * np.einsum("tijk,tk->tij", zz, lam)
                C = np.triu(C, 1)
                W = W + C + np.swapaxes(C, 1, 2)
        return phi, g, W, c, Jc

    def smooth_cost(self, Q):
        d = Q[2:] - Q[1:-1]
        return self.kappa * np.sum(d * d)

    def objective(self, Q, path):
        e, _, _, _ = self.chain.fk(Q)
        return self.w_path * np.sum((path - e) ** 2) + self.smooth_cost(Q)


def block_tridiag_solve(D, E, rhs, shift=0.0):
    """Solve K x = rhs, K = blocktridiag(E_{t-1}^T, D_t, E_t) SPD, by block Cholesky (the Riccati sweep).
    D: (N,m,m), E: (N-1,m,m) with K[t,t+1] = E[t].  Returns (x, ok)."""
    N, m = D.shape[0], D.shape[1]
    L = np.zeros_like(D)
    Fm = np.zeros_like(E)  # F_t = K[t+1,t] L_t^{-T}
    y = np.zeros_like(rhs)
    S = D[0] + shift * np.eye(m)
    for t in range(N):
        try:
            L[t] = np.linalg.cholesky(S)
        except np.linalg.LinAlgError:
            return None, False
        y[t] = np.linalg.solve(L[t], rhs[t] - (Fm[t - 1] @ y[t - 1] if t > 0 else 0.0))
        if t < N - 1:
            Fm[t] = np.linalg.solve(L[t], E[t]).T
            S = D[t + 1] + shift * np.eye(m) - Fm[t] @ Fm[t].T
    x = np.zeros_like(rhs)
    for t in range(N - 1, -1, -1):
        x[t] = np.linalg.solve(L[t].T, y[t] - (Fm[t].T @ x[t + 1] if t < N - 1 else 0.0))
    return x, True


def solve_structured(prob: StructuredFigureEight, qc, Q0=None, max_iter=60, tol=1e-9, exact=True, verbose=False):
    T, n = prob.T, prob.n
    path, Rc = prob.references(qc)
    Q = np.tile(qc, (T, 1)) if Q0 is None else Q0.copy()
    Q[0] = qc
    Q[1] = qc
    kap = prob.kappa
    lam = np.zeros((T, 3))
    nu = 1.0
    hist = []
    F = slice(2, T)  # free knots
    nf = T - 2
    for it in range(max_iter + 1):
        phi, g, W, c, Jc = prob.evaluate(Q, path, Rc, lam=lam, exact=exact)
        # smoothness gradient on free knots
        Gs = np.zeros((T, n))
        d = Q[2:] - Q[1:-1]  # Delta_t, t=1..T-2
        Gs[2:] += 2 * kap * d
        Gs[1:-1] -= 2 * kap * d
        G = g + Gs
        ndiag = np.full(T, 2.0)
        ndiag[T - 1] = 1.0
        # null-space split per knot
        Zs = np.zeros((T, n, n - 3))
        Ys = np.zeros((T, n, 3))
        Rf = np.zeros((T, 3, 3))
        for t in range(2, T):
            Qm, Rm = np.linalg.qr(Jc[t].T, mode="complete")
            Ys[t], Zs[t], Rf[t] = Qm[:, :3], Qm[:, 3:], Rm[:3]
        # particular step: Jc dq = -c  ->  dq_p = Y (R^T)^{-1} (-c)
        npart = np.zeros((T, n))
        for t in range(2, T):
            npart[t] = Ys[t] @ np.linalg.solve(Rf[t].T, -c[t])
        Dfull = W + (2 * kap * ndiag)[:, None, None] * np.eye(n)[None]
        # multipliers estimate (least squares per knot): G + Jc^T lam = 0 on range(Y)
        lam_ls = np.zeros((T, 3))
        for t in range(2, T):
            lam_ls[t] = np.linalg.solve(Rf[t], -Ys[t].T @ G[t])
        stat = np.max(np.abs(np.einsum("tij,ti->tj", Zs[F], G[F])))
        feas = np.max(np.abs(c[F]))
        fval = float(np.sum(phi) + prob.smooth_cost(Q))
        hist.append((it, fval, stat, feas))
        if verbose:
            print(f"  it {it:3d} f={fval:.12f} stat={stat:.3e} feas={feas:.3e}")
        if (stat <= tol and feas <= tol) or it == max_iter:
            lam = lam_ls
            break
        # reduced system
        Hn = np.einsum("tij,tj->ti", Dfull, npart)
        Hn[2:] += -2 * kap * np.concatenate([npart[3:], np.zeros((1, n))])  # coupling to t+1
        Hn[3:] += -2 * kap * npart[2:-1]  # coupling to t-1
        rhs = -np.einsum("tij,ti->tj", Zs[F], (G + Hn)[F])
        Dr = np.einsum("tia,tij,tjb->tab", Zs[F], Dfull[F], Zs[F])
        Er = -2 * kap * np.einsum("tia,tib->tab", Zs[2 : T - 1], Zs[3:T])
        shift = 0.0
        while True:
            z, ok = block_tridiag_solve(Dr, Er, rhs, shift)
            if ok:
                break
            shift = 1e-2 if shift == 0.0 else shift * 10.0
        dq = npart.copy()
        dq[F] += np.einsum("tia,ta->ti", Zs[F], z)
        # multipliers of the QP: (G + H dq) + Jc^T lam_qp = 0
        Hd = np.einsum("tij,tj->ti", Dfull, dq)
        Hd[2:] += -2 * kap * np.concatenate([dq[3:], np.zeros((1, n))])
        Hd[3:] += -2 * kap * dq[2:-1]
        lam_qp = np.zeros((T, 3))
        for t in range(2, T):
            lam_qp[t] = np.linalg.solve(Rf[t], -Ys[t].T @ (G[t] + Hd[t]))
        # l1 merit line search
        nu = max(nu, 1.5 * np.max(np.abs(lam_qp)))
        c1 = np.sum(np.abs(c[F]))
        phi0 = fval + nu * c1
        dphi = float(np.sum(G[F] * dq[F])) - nu * c1
        alpha = 1.0
        while True:
            Qt = Q.copy()
            Qt[F] += alpha * dq[F]
            _, _, _, ct, _ = prob.evaluate(Qt, path, Rc)
            ft = prob.objective(Qt, path)
            if ft + nu * np.sum(np.abs(ct[F])) <= phi0 + 1e-4 * alpha * min(dphi, 0.0) or alpha < 1e-6:
                break
            alpha *= 0.5
        if verbose:
            print(f"        alpha={alpha:.4g} shift={shift:g} nu={nu:.3g} |dq|={np.max(np.abs(dq)):.3g}")
        Q = Qt
        lam = lam_qp
    return {"Q": Q, "f": fval, "iters": it, "stat": stat, "feas": feas, "lam": lam, "history": hist, "path": path, "Rc": Rc}


# ----------------------------------------------------------------------------------------------------
# Port of the HIP state machine (k_eval / k_couple / k_step in optas_amd/csrc/oh_kernels.hip):
# feasible iterates by retraction, Levenberg-Marquardt ratio test on the objective, no merit parameter.
# ----------------------------------------------------------------------------------------------------
def _orient(prob, Q, Rc):
    e, Re, Jp, Jw = prob.chain.jac(Q)
    A = Re @ Rc.T
    c = _vee_skew(A)
    trA = np.trace(A, axis1=1, axis2=2)
    M = 0.5 * (trA[:, None, None] * np.eye(3)[None] - A)
    return c, M @ Jw


def retract(prob, Q, Rc, tol=1e-10, max_corr=4, e_tgt=None):
    """Newton corrections per knot (k_eval's loop).  Without a target: q_t <- q_t - Jc^T (Jc Jc^T)^{-1} c, the minimum-norm way back
    onto R(q_t) = Rc.  With e_tgt (T, 3), the end-effector positions the linear model of the step predicted: minimum-norm Newton steps
    on the six rows [c(q); e(q) - e_tgt] = 0 (a second-order correction: the trial point follows the curved valley of the stiff
    tracking cost instead of leaving it at second order, which is what kept the Gauss-Newton model honest only for tiny steps along
    the redundant direction).  The loop still stops on the orientation rows alone."""
    Q = Q.copy()
    settled = np.zeros(Q.shape[0], dtype=bool)  # knots whose last correction was short enough to trust without another look
    settled[:2] = True
    for _ in range(max_corr):
        e, Re, Jp, Jw = prob.chain.jac(Q)
        A = Re @ Rc.T
        c = _vee_skew(A)
        trA = np.trace(A, axis1=1, axis2=2)
        Jc = (0.5 * (trA[:, None, None] * np.eye(3)[None] - A)) @ Jw
        bad = np.where((np.max(np.abs(c), axis=1) > tol) & ~settled)[0]
        if bad.size == 0:
            break
        for t in bad:
            if e_tgt is None:
                S = Jc[t] @ Jc[t].T + 1e-14 * np.eye(3)
                dq = Jc[t].T @ np.linalg.solve(S, c[t])
            else:
                J6 = np.vstack([Jc[t], Jp[t]])
                S = J6 @ J6.T + 1e-10 * np.eye(6)
                dq = J6.T @ np.linalg.solve(S, np.concatenate([c[t], e[t] - e_tgt[t]]))
            Q[t] -= dq
            # what the step leaves behind is second order in dq, bounded by ||dq||_1^2: below the tolerance the kinematics pass that
            # would only confirm it is skipped (eval_knot, oh_figure8.h)
            settled[t] = np.abs(dq).sum() ** 2 <= tol
    return Q


LS_MAX, LS_SHRINK = 3, 0.3  # OH_LS_MAX, OH_LS_SHRINK (csrc/oh_types.h): line search on a rejected step of a handle with inequality rows


RETRACT_FLOOR = 1e-13


def retract_tol(tol_feas, pred, far, tight=True):
    """retract_tol in csrc/oh_figure8.h.  Far from the solution the violation a trial point may keep is tied to the decrease its step predicts,
    never looser than 1e-5.  In the end game the same holds below the floor min(1e-10, tol_feas): an accepted point that keeps a violation c
    carries an objective that is off by (multiplier) x c, and a step that predicts less than that is accepted or refused by the rounding of the
    retraction, not by its merit -- with inequality rows the outer loop asks for stat <= tol again after every multiplier update, with predicted
    decreases of 1e-12 against 1.5e-10 of such noise (round 3: the instances of a velocity-limited batch that sat at stat 1e-5 until the cap)."""
    floor = min(1e-10, tol_feas)
    if far:
        return min(1e-5, max(floor, 1e-3 * pred))
    return min(floor, max(RETRACT_FLOOR, 1e-2 * pred)) if tight else floor  # tight: handles with inequality rows (FigParams.tol_retract_min)


def solve_structured_lm(prob, qc, Q0=None, max_iter=300, tol=1e-6, tol_feas=1e-9, mu0=0.0, exact=False, rule="nielsen", verbose=False, hessian="hybrid", limits=None, rho0=None, guards=None, overrelax=1.5, overrelax_from=4, vlimits=None):
    """Returns dict(Q, f, iters (= steps solved: accepted + rejected), rejected, stat, feas, status).
    limits = (lo, up) or guards = oracle.guarded.Guards (joint limits and/or sphere clearances): inequality rows at the free
    knots through the augmented Lagrangian of oracle/guarded.py (k_eval_lg / k_step_lg); adds "lam" (T, NC), "meas", "outers".
    vlimits = (vlo, vup): joint-velocity rows dq_t - vlo >= 0, vup - dq_t >= 0 (enforce_model_limits(time_deriv=1), builder.py:471-509) on
    dq_t = (q_{t+1} - q_t) / dt; they couple neighbouring knots exactly like the velocity cost: in the Gauss-Newton model row j of interval
    (t, t+1) adds rho/dt^2 to the weight 2 kappa of (q_{t+1,j} - q_{t,j})^2 while it is active (k_couple_lg); adds "lam_v" (T-1, 2n).
    hessian: "gauss_newton" | "exact" | "hybrid" (Gauss-Newton until the reduced gradient of the accepted point is below
    1e-5 * w_path, exact curvature afterwards: OH_HESSIAN_HYBRID); exact=True is shorthand for "exact"."""
    if exact:
        hessian = "exact"
    guard = limits is not None or guards is not None or vlimits is not None
    hyb_switch = 1e-5 * prob.w_path
    stat_prev = np.inf
    tight = True  # FigParams.tol_retract_min below tol_retract: the end-game rules of retract_tol / lm_accept (every handle since the end of round 3)
    vel = vlimits is not None
    if guard:
        from .guarded import Guards, guard_values

        if guards is None:
            guards = Guards() if limits is None else Guards(lo=np.asarray(limits[0], dtype=float), up=np.asarray(limits[1], dtype=float))
        lam_g = np.zeros((prob.T, guards.n_rows(prob.n)))
        if vel:
            vlo, vup = (np.asarray(v, dtype=float) for v in vlimits)
            lam_v = np.zeros((prob.T - 1, 2 * prob.n))  # row i: interval (i, i+1) = dq_i; [dq - vlo (n); vup - dq (n)]
            # The velocity rows get the penalty rho * vscale: their Gauss-Newton weight on (q_{t+1} - q_t)^2 is rho vscale / dt^2, and with the
            # rows' own rho (10 w_path, chosen for the position rows) that is 2.4e5 against a tracking curvature of ~5e2 -- the merit becomes a
            # stiff piecewise quadratic and two thirds of the steps are rejected (459 steps on the nominal instance).  vscale = dt^2 / 40 puts the
            # weight at w_path / 4 (19 steps); the outer loop scales both penalties together.
            vscale = prob.dt**2 / 40.0
        rho_g = rho_next = (10.0 * prob.w_path) if rho0 is None else rho0
        omega, meas_prev, outer, outers = max(tol, 1e-2), np.inf, False, 0

    def vel_terms(Q, lam_v, rho_g):
        """Velocity rows of every interval: (values (T-1, 2n), psi per interval, sigma = d L_A / d v (T-1, n), Gauss-Newton weights
        rho (a- + a+) / dt^2 (T-1, n), measure).  Interval (0, 1) is constant (q_0 = q_1 = qc) and left out."""
        dtv = prob.dt
        v = (Q[1:] - Q[:-1]) / dtv
        gv = np.concatenate([v - vlo[None], vup[None] - v], 1)
        rho_g = rho_g * vscale
        sv = np.maximum(0.0, lam_v - rho_g * gv)
        sv[0] = 0.0
        psi = (sv * sv - lam_v * lam_v) * (1.0 / (2.0 * rho_g))  # (reciprocal, then product: as the kernels do since round 4)
        psi[0] = 0.0
        nn = Q.shape[1]
        sig = (sv[:, nn:] - sv[:, :nn]) / dtv  # d psi / d v : lower row -s_lo, upper row +s_up
        wv = rho_g * ((sv[:, :nn] > 0.0).astype(float) + (sv[:, nn:] > 0.0).astype(float)) / dtv**2
        meas = np.abs(np.minimum(gv, lam_v * (1.0 / rho_g)))
        meas[0] = 0.0
        return gv, psi.sum(1), sig, wv, float(meas.max())

    def guard_terms(Q, lam_g, rho_g):
        if guards.n_rows(prob.n) == 0:
            Tn = Q.shape[0]
            return np.zeros((Tn, 0)), np.zeros(Tn), np.zeros_like(Q), np.zeros((Tn, Q.shape[1], Q.shape[1])), 0.0
        gv, dg = guard_values(prob.chain, Q, guards)  # (T, NC), (T, NC, n)
        sv = np.maximum(0.0, lam_g - rho_g * gv)
        sv[:2] = 0.0
        psi = (sv * sv - lam_g * lam_g) * (1.0 / (2.0 * rho_g))
        psi[:2] = 0.0
        dgrad = -np.einsum("tc,tcn->tn", sv, dg)
        dW = rho_g * np.einsum("tc,tcn,tcm->tnm", (sv > 0.0).astype(float), dg, dg)
        meas = np.abs(np.minimum(gv, lam_g * (1.0 / rho_g)))
        meas[:2] = 0.0
        return gv, psi.sum(1), dgrad, dW, float(meas.max())
    T, n = prob.T, prob.n
    path, Rc = prob.references(qc)
    kap = prob.kappa
    F = slice(2, T)
    Qc = np.tile(qc, (T, 1)) if Q0 is None else Q0.copy()
    Qc[0] = qc
    Qc[1] = qc
    mu = mu0
    iters = rejected = 0
    first = True
    polish = False
    Qt = retract(prob, Qc, Rc, tol=retract_tol(tol_feas, 0.0, False, tight))
    lam = np.zeros((T, 3))
    cur = None
    status = 1
    nu_n = 2.0
    ls_count, ls_scale, z_last, gd_last, q_last = 0, 1.0, None, 0.0, 0.0
    while True:
        # ---- k_eval + k_couple on the trial point
        use_exact = hessian == "exact" or (hessian == "hybrid" and not first and stat_prev <= hyb_switch)
        if guard and guards.links:
            use_exact = False  # no curvature of the sphere rows in the exact block: Gauss-Newton models them better
        phi, g, W, c, Jc = prob.evaluate(Qt, path, Rc, lam=lam, exact=use_exact)
        if guard:
            if outer:  # multiplier refresh at the accepted point with the old penalty, evaluation with the new one
                if guards.n_rows(n):
                    gv_now, _ = guard_values(prob.chain, Qt, guards)
                    lam_g = np.maximum(0.0, lam_g - rho_g * gv_now)
                    lam_g[:2] = 0.0
                if vel:
                    lam_v = np.maximum(0.0, lam_v - rho_g * vscale * vel_terms(Qt, lam_v, rho_g)[0])
                    lam_v[0] = 0.0
                rho_g = rho_next
                outers += 1
            gv, psi_t, dgrad, dW, meas_t = guard_terms(Qt, lam_g, rho_g)
            wv_t = np.zeros((T - 1, n))
            if vel:
                _, psi_v, sig, wv_t, meas_v = vel_terms(Qt, lam_v, rho_g)
                psi_t = psi_t.copy()
                psi_t[1:] += psi_v  # interval (t-1, t) is booked on knot t, like kappa ||q_t - q_{t-1}||^2
                dgrad = dgrad.copy()
                dgrad[1:] += sig
                dgrad[:-1] -= sig
                meas_t = max(meas_t, meas_v)
            phi = phi + psi_t
            g = g + dgrad
            W = W + dW
        f_t = float(np.sum(phi) + prob.smooth_cost(Qt))
        feas_t = float(np.max(np.abs(c[F])))
        # ---- k_step phase A
        polish_request = False
        if first or (guard and outer) or polish:
            accept = True
            first = False
            polish = False
            if guard:
                outer = False
        else:
            rho = (cur["f"] - f_t) / max(pred, 1e-300)
            # (noise: what the orientation violations of the two points are worth, lm_accept in csrc/oh_figure8.h -- handles with inequality rows)
            noise = 10.0 * max(1.0, abs(cur["f"])) * (feas_t + cur["feas"]) if tight else 0.0
            accept = np.isfinite(f_t) and (rho > 1e-4 or (pred <= max(1e-15 * abs(cur["f"]), noise) and f_t <= cur["f"] + 1e-14 * abs(cur["f"]) + noise))
            # a rejected trial against an accepted point that was retracted loosely: the accepted objective is off by (multiplier) x
            # violation, and steps that predict less than that can never be accepted.  Re-evaluate the accepted point itself at the
            # floor tolerance (zero step, accepted unconditionally) before blaming the model.
            polish_request = (not accept) and cur["feas"] > 10.0 * retract_tol(tol_feas, pred, False, tight)
            if polish_request:
                pass
            elif rule == "hip":
                if accept:
                    if rho > 0.75:
                        mu = mu * 0.2 if mu > 1e-6 else 0.0
                    elif rho < 0.25:
                        mu = max(4.0 * mu, 1e-3)
                else:
                    mu = max(4.0 * mu, 1e-3)
            elif guard and (not accept) and ls_count < LS_MAX and z_last is not None and iters < max_iter // 2:
                # handles with inequality rows: line search along the rejected step before the damping is touched (oracle/guarded.py,
                # step_instance<N, true> in csrc/oh_figure8_units.h)
                ls_count += 1
                ls_scale *= LS_SHRINK
                rejected += 1
                z_ls = z_last * ls_scale
                pred = -gd_last * ls_scale + 0.5 * ls_scale * ls_scale * q_last
                Qt = cur["Q"].copy()
                Qt[F] += np.einsum("tia,ta->ti", cur["Z"][F], z_ls)
                e_tgt = cur["e"].copy()
                e_tgt[F] += np.einsum("tma,ta->tm", cur["JZ"], z_ls)
                tol_r = retract_tol(tol_feas, pred, stat_prev > hyb_switch, tight)
                Qt = retract(prob, Qt, Rc, tol=tol_r, e_tgt=e_tgt)
                if iters >= max_iter:
                    break
                iters += 1
                continue
            elif rule == "nielsen":
                if accept:
                    mu = mu * max(1.0 / 3.0, 1.0 - ((2.0 * rho - 1.0) if (rho > 1e-4 or noise == 0.0) else 0.0) ** 3)  # (noise-level step: damping unchanged)
                    if mu < 1e-7:
                        mu = 0.0
                    nu_n = 2.0
                else:
                    mu = max(mu * nu_n, 1e-3)
                    nu_n *= 2.0
            elif rule == "gentle":
                if accept:
                    if rho > 0.9:
                        mu = mu * 0.5 if mu > 1e-6 else 0.0
                    elif rho < 0.5:
                        mu = max(2.0 * mu, 1e-3)
                else:
                    mu = max(4.0 * mu, 1e-3)
            if not accept:
                rejected += 1
            if polish_request:
                polish = True
                Qt = retract(prob, cur["Q"], Rc, tol=retract_tol(tol_feas, 0.0, False, tight), e_tgt=cur["e"])
                pred = 0.0
                iters += 1
                continue
        if accept:
            ls_count, ls_scale = 0, 1.0
            Gs = np.zeros((T, n))
            d = Qt[2:] - Qt[1:-1]
            Gs[2:] += 2 * kap * d
            Gs[1:-1] -= 2 * kap * d
            G = g + Gs
            Zs = np.zeros((T, n, n - 3))
            for t in range(2, T):
                Qm, _ = np.linalg.qr(Jc[t].T, mode="complete")
                Zs[t] = Qm[:, 3:]
            ndiag = np.full(T, 2.0)
            ndiag[T - 1] = 1.0
            Dfull = W + (2 * kap * ndiag)[:, None, None] * np.eye(n)[None]
            wnext = np.zeros((T, n))  # Gauss-Newton weight of the velocity rows of interval (t, t+1), zero without them
            if guard and vel:
                wnext[:-1] = wv_t
                wsum = wnext.copy()
                wsum[1:] += wv_t
                Dfull = Dfull + np.einsum("tj,jk->tjk", wsum, np.eye(n))
            cur = {
                "Q": Qt, "f": f_t, "feas": feas_t, "Z": Zs, "meas": meas_t if guard else 0.0, "fpsi": float(np.sum(psi_t)) if guard else 0.0,
                "gt": np.einsum("tij,ti->tj", Zs[F], G[F]),
                "Dr": np.einsum("tia,tij,tjb->tab", Zs[F], Dfull[F], Zs[F]),
                "Er": -np.einsum("tia,ti,tib->tab", Zs[2 : T - 1], 2 * kap + wnext[2 : T - 1], Zs[3:T]),
            }
            e_acc, _, Jp_acc, _ = prob.chain.jac(Qt)
            cur["e"], cur["JZ"] = e_acc, np.einsum("tmi,tia->tma", Jp_acc[F], Zs[F])
            if hessian != "gauss_newton":
                for t in range(2, T):
                    lam[t] = -np.linalg.solve(Jc[t] @ Jc[t].T + 1e-14 * np.eye(3), Jc[t] @ G[t])
        # ---- k_step phase B
        stat = float(np.max(np.abs(cur["gt"])))
        stat_prev = stat
        while True:
            z, ok = block_tridiag_solve(cur["Dr"], cur["Er"], -cur["gt"], mu)
            if ok:
                break
            mu = max(4.0 * mu, 1e-2)
        if verbose:
            print(f"  steps {iters:3d} f={cur['f']:.12f} stat={stat:.3e} mu={mu:.3g} rejected={rejected}")
        if guard:
            if stat <= omega:
                if stat <= tol and cur["feas"] <= tol_feas and cur["meas"] <= tol_feas:
                    status = 0
                    break
                if iters >= max_iter:
                    break
                rho_next = min(rho_g * 10.0, 1e8) if cur["meas"] > 0.25 * meas_prev else rho_g
                meas_prev = cur["meas"]
                omega = max(tol, min(omega, 0.1 * cur["meas"]))
                outer = True
                Qt = retract(prob, cur["Q"], Rc, tol=retract_tol(tol_feas, 0.0, stat > hyb_switch, tight), e_tgt=cur["e"])  # (zero step through k_retract)
                pred = 0.0
                iters += 1
                continue
        elif stat <= tol and cur["feas"] <= tol_feas:
            status = 0
            break
        if iters >= max_iter:
            break
        # over-relaxation of Gauss-Newton steps in the crawl phase of the hybrid scheme (step_instance in csrc/oh_figure8_units.h)
        alpha = overrelax if (hessian == "hybrid" and not guard and stat > hyb_switch and iters >= overrelax_from) else 1.0
        # (plain handles, end of round 5: the kernels take g.z as -g^T M^-1 g = -sum_t |L_t^-1 r_t|^2 off their backward sweep -- the same number to rounding)
        gd_, z2_ = float(np.sum(cur["gt"] * z)), float(np.sum(z * z))
        pred = -alpha * gd_ + 0.5 * alpha * alpha * (gd_ + mu * z2_)
        z = alpha * z
        z_last, gd_last, q_last, ls_scale = z.copy(), alpha * gd_, alpha * alpha * (gd_ + mu * z2_), 1.0  # pred(s) = -s gd + s^2 q / 2 for the step s z
        Qt = cur["Q"].copy()
        Qt[F] += np.einsum("tia,ta->ti", cur["Z"][F], z)
        e_tgt = cur["e"].copy()
        e_tgt[F] += np.einsum("tma,ta->tm", cur["JZ"], z)  # predicted end-effector positions: e + (Jp Z) z
        # far from the solution the violation a trial point may keep is tied to the decrease its step predicts (retract_tol in oh_figure8.h)
        tol_r = retract_tol(tol_feas, pred, stat > hyb_switch, tight)
        Qt = retract(prob, Qt, Rc, tol=tol_r, e_tgt=e_tgt)
        iters += 1
    out = {"Q": cur["Q"], "f": cur["f"] - cur["fpsi"], "iters": iters, "rejected": rejected, "stat": stat, "feas": cur["feas"], "status": status, "path": path, "Rc": Rc}
    if guard:
        gv = guard_terms(cur["Q"], lam_g, rho_g)[0]
        lam_out = np.maximum(0.0, lam_g - rho_g * gv)
        lam_out[:2] = 0.0
        out.update(lam=lam_out, lam_stored=lam_g, meas=cur["meas"], outers=outers, g=gv)
        if vel:
            gvv = vel_terms(cur["Q"], lam_v, rho_g)[0]
            lv = np.maximum(0.0, lam_v - rho_g * vscale * gvv)
            lv[0] = 0.0
            out.update(lam_v=lv, g_v=gvv)
    return out


# ----------------------------------------------------------------------------------------------------
# Position-only tracking (dual_arm.py per arm): port of k_eval_free / k_couple_free / k_step_free.
# ----------------------------------------------------------------------------------------------------
def solve_free_lm(chain: FoldedChain, T, dt, offsets, qc, Q0=None, w_path=1.0, w_vel=0.01, fix_dq0=False, max_iter=300, tol=1e-6, exact=False, verbose=False):
    """min sum_t w_path ||p(q_t) - (p(qc) + offsets_t)||^2 + (w_vel/dt^2) sum_t ||q_{t+1}-q_t||^2, q_0 = qc (and q_1 = qc if fix_dq0).
    offsets: (T,3).  Same LM ratio test / Nielsen update / Riccati recursion as the HIP kernels."""
    n = chain.ndof
    t0 = 2 if fix_dq0 else 1
    kap = w_vel / dt**2
    e0, _, _, _ = chain.fk(qc[None])
    path = e0[0] + offsets
    Qc = np.zeros((T, n)) if Q0 is None else Q0.copy()
    Qc[:t0] = qc
    F = slice(t0, T)

    def evalp(Q):
        e, Re, Jp, Jw = chain.jac(Q)
        r = path - e
        phi = w_path * np.sum(r * r, 1)
        g = -2.0 * w_path * np.einsum("tki,tk->ti", Jp, r)
        W = 2.0 * w_path * np.einsum("tki,tkj->tij", Jp, Jp)
        if exact:
            zr = np.cross(r[:, None, :], np.swapaxes(Jw, 1, 2))
            S = np.einsum("tik,tkj->tij", zr, Jp)
            S = np.triu(S) + np.swapaxes(np.triu(S, 1), 1, 2)
            W = W - 2.0 * w_path * S
        d = Q[1:] - Q[:-1]
        f = float(np.sum(phi) + kap * np.sum(d * d))
        Gs = np.zeros_like(Q)
        Gs[1:] += 2 * kap * d
        Gs[:-1] -= 2 * kap * d
        return f, g + Gs, W

    mu, nun = 0.0, 2.0
    iters = rejected = 0
    first = True
    Qt = Qc
    cur = None
    status = 1
    while True:
        f_t, G, W = evalp(Qt)
        if first:
            accept, first = True, False
        else:
            rho = (cur["f"] - f_t) / max(pred, 1e-300)
            accept = np.isfinite(f_t) and (rho > 1e-4 or (pred <= 1e-15 * abs(cur["f"]) and f_t <= cur["f"] + 1e-14 * abs(cur["f"])))
            if accept:
                mu *= max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)
                mu = 0.0 if mu < 1e-7 else mu
                nun = 2.0
            else:
                mu = max(mu * nun, 1e-3)
                nun *= 2.0
                rejected += 1
        if accept:
            ndiag = np.full(T, 2.0)
            ndiag[T - 1] = 1.0
            cur = {"Q": Qt, "f": f_t, "G": G[F], "D": (W + (2 * kap * ndiag)[:, None, None] * np.eye(n)[None])[F]}
        stat = float(np.max(np.abs(cur["G"])))
        nf = T - t0
        Er = np.tile(-2 * kap * np.eye(n), (nf - 1, 1, 1))
        while True:
            z, ok = block_tridiag_solve(cur["D"], Er, -cur["G"], mu)
            if ok:
                break
            mu = max(4.0 * mu, 1e-2)
        if verbose:
            print(f"  steps {iters:3d} f={cur['f']:.12f} stat={stat:.3e} mu={mu:.3g}")
        if stat <= tol:
            status = 0
            break
        if iters >= max_iter:
            break
        pred = -0.5 * float(np.sum(cur["G"] * z)) + 0.5 * mu * float(np.sum(z * z))
        Qt = cur["Q"].copy()
        Qt[F] += z
        iters += 1
    return {"Q": cur["Q"], "f": cur["f"], "iters": iters, "rejected": rejected, "stat": stat, "status": status}


class LeadChain:
    """A FoldedChain of the full robot seen through its optimised joints: one parameterised joint (RobotModel(param_joints=[...]),
    models.py:286-321; example/figure_eight_plan_6dof.py) is held at theta_knots[t] when a whole trajectory (T rows) is passed and
    at theta_ref otherwise (the reference configuration qc)."""

    def __init__(self, full: FoldedChain, par: int, theta_knots, theta_ref: float):
        self.full, self.par = full, par
        self.theta_knots, self.theta_ref = np.asarray(theta_knots, dtype=float), float(theta_ref)
        self.opt = [i for i in range(full.ndof) if i != par]
        self.ndof = full.ndof - 1
        self.n_chain = full.n_chain - 1
        self._keep = [k for k in range(full.n_chain) if full.qidx[k] != par]
        self.jtype = [full.jtype[k] for k in self._keep]
        self.qidx = [self.opt.index(full.qidx[k]) for k in self._keep]
        self.attach = full.attach

    def _full(self, Q):
        Q = np.atleast_2d(Q)
        th = self.theta_knots if Q.shape[0] == self.theta_knots.shape[0] else np.full(Q.shape[0], self.theta_ref)
        out = np.zeros((Q.shape[0], self.full.ndof))
        out[:, self.opt] = Q
        out[:, self.par] = th
        return out

    def fk(self, Q, frames=False):
        r = self.full.fk(self._full(Q), frames=frames)
        return (r[0], r[1], r[2][:, self._keep], r[3][:, self._keep]) + tuple(r[4:])

    def jac(self, Q):
        e, Re, Jp, Jw = self.full.jac(self._full(Q))
        return e, Re, Jp[:, :, self.opt], Jw[:, :, self.opt]


def lead_problem(robot, link, par: int, theta_knots, theta_ref: float, **kw):
    """StructuredFigureEight over the optimised joints of a robot whose joint `par` is parameterised."""
    prob = StructuredFigureEight(robot, link, **kw)
    prob.chain = LeadChain(prob.chain, par, theta_knots, theta_ref)
    prob.n = prob.chain.ndof
    return prob
