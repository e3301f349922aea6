"""ORACLE (test infrastructure, not product code) -- BASELINE configs[4]: 7-DoF torque-control MPC with RNEA
dynamics equality constraints.  The reference ships no such script (its
example/torque_control_example.py evaluates RobotModel.rnea outside the optimiser, :198-200); the problem is the
synthetic one SURVEY App. B.5 specifies, written with the reference's own builder calls:

    robot = RobotModel(med7.urdf, time_derivs=[0, 1, 2]); tau = TaskModel("tau", 7, time_derivs=[0], dlim={0: [lo, up]})
    builder = OptimizationBuilder(T, robots=[robot], tasks=[tau], derivs_align=True)          builder.py:14-99
    qc, dqc, goal = add_parameter("qc", 7), add_parameter("dqc", 7), add_parameter("goal", 3, T)
    fix_configuration(name, qc); fix_configuration(name, dqc, time_deriv=1)                   builder.py:525-539
    integrate_model_states(name, 1, dt); integrate_model_states(name, 2, dt)                  builder.py:419-469
    add_equality_constraint("dynamics", lhs=robot.rnea(Q, dQ, ddQ), rhs=TAU)                  models.py:1731-1884
    enforce_model_limits("tau")                                                               builder.py:471-509
    add_cost_term("track", w_p * sumsqr(p_link(Q) - goal)); add_cost_term("effort", w_tau * sumsqr(TAU))

This module holds (i) a vectorised, dtype-agnostic restatement of the reference's RNEA recursion (checked against the
literal oracle.robot.rnea) whose Jacobian comes from complex-step differentiation -- no hand-written derivative code, so
it is an independent check of the tangent recursion inside the HIP kernel; (ii) ``solve_torque_lm``: a numpy port of the
state machine the HIP path runs (tau eliminated through the dynamics rows, linear rows rolled out exactly, effort limits
through the Powell-Hestenes-Rockafellar augmented Lagrangian, Gauss-Newton / Levenberg-Marquardt steps from a Riccati
sweep over the stage (q_t, dq_t | ddq_t)).  The literal NLP (x / p / v layout of the reference) is
oracle.problems.TorqueMPCNLP; independent solvers on it: oracle.solvers.scipy_minimize and kkt_reference_form.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
"""
import numpy as np

from .robot import OracleRobot, rnea_tables
from .spatialmath import rpy2r
from .structured import FoldedChain


class RneaTables:
    """Constants of RobotModel.rnea in the selection the reference makes (models.py:1742-1784; oracle.robot.rnea_tables)."""

    def __init__(self, robot: OracleRobot):
        m, cm, Icm, xyzs, rpys, axes = rnea_tables(robot)
        self.n = len(xyzs)  # bodies (the last one hangs on a fixed joint)
        self.ndof = self.n - 1
        self.m = np.asarray(m, float)
        self.cm = np.asarray(cm, float).T.copy()  # (n, 3)
        self.I = np.stack(Icm)  # (n, 3, 3)
        self.xyz = np.stack(xyzs)
        self.R0 = np.stack([rpy2r(r) for r in rpys])
        self.axis = np.stack(axes)
        self.K = np.stack([np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0.0]]) for a in self.axis])


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def rnea_batch(tb: RneaTables, q, qd, qdd):
    """models.py:1819-1880 over leading batch axes; q, qd, qdd (..., ndof) real or complex -> tau (..., ndof)."""
    q, qd, qdd = np.asarray(q), np.asarray(qd), np.asarray(qdd)
    dt_ = np.result_type(q.dtype, qd.dtype, qdd.dtype, float)
    shp = np.broadcast_shapes(q.shape, qd.shape, qdd.shape)[:-1]
    n = tb.n
    om = np.zeros(shp + (3,), dt_)
    omD = np.zeros(shp + (3,), dt_)
    vD = np.zeros(shp + (3,), dt_) + np.array([0.0, 0.0, 9.81])
    Rs, fs, ns = [], [], []
    for i in range(n):
        if i != n - 1:
            s, c = np.sin(q[..., i])[..., None, None], np.cos(q[..., i])[..., None, None]
            Rq = np.eye(3) + s * tb.K[i] + (1.0 - c) * (tb.K[i] @ tb.K[i])
            R = tb.R0[i] @ Rq  # pRi
        else:
            R = np.broadcast_to(tb.R0[i], shp + (3, 3)).astype(dt_)
        Rs.append(R)
        Rt = np.swapaxes(R, -1, -2)
        omp = (Rt @ om[..., None])[..., 0]
        omDp = (Rt @ omD[..., None])[..., 0]
        if i != n - 1:
            a = Rt @ tb.axis[i]
            aq = a * qd[..., i][..., None]
            omi = omp + aq
            omDi = omDp + _cross(omp, aq) + a * qdd[..., i][..., None]
        else:
            omi, omDi = omp, omDp
        r = tb.xyz[i]
        acc = vD + _cross(omD, r) + _cross(om, _cross(om, np.broadcast_to(r, om.shape)))
        vDi = (Rt @ acc[..., None])[..., 0]
        c_ = np.broadcast_to(tb.cm[i], om.shape)
        fi = tb.m[i] * (vDi + _cross(omDi, c_) + _cross(omi, _cross(omi, c_)))
        Io = (tb.I[i] @ omi[..., None])[..., 0]
        ni = (tb.I[i] @ omDi[..., None])[..., 0] + _cross(omi, Io)
        om, omD, vD = omi, omDi, vDi
        fs.append(fi)
        ns.append(ni)
    ifi = fs[n - 1]
    ini = ns[n - 1] + _cross(np.broadcast_to(tb.cm[n - 1], ifi.shape), fs[n - 1])
    taus = [None] * (n - 1)
    for i in range(n - 1, 0, -1):
        R = Rs[i]
        Rf = (R @ ifi[..., None])[..., 0]
        ini = ns[i - 1] + (R @ ini[..., None])[..., 0] + _cross(np.broadcast_to(tb.cm[i - 1], ifi.shape), fs[i - 1]) + _cross(
            np.broadcast_to(tb.xyz[i], ifi.shape), Rf)
        ifi = Rf + fs[i - 1]
        Rt = np.swapaxes(Rs[i - 1], -1, -2)
        taus[i - 1] = np.sum(ini * (Rt @ tb.axis[i - 1]), -1)
    return np.stack(taus, -1)


def rnea_jacobian(tb: RneaTables, q, qd, qdd, h=1e-30):
    """d tau / d (q, qd, qdd): (..., ndof, 3 ndof) by complex-step differentiation of rnea_batch (exact to rounding)."""
    q, qd, qdd = (np.asarray(a, float) for a in (q, qd, qdd))
    n = tb.ndof
    z = np.concatenate([q, qd, qdd], -1)[..., None, :] + 1j * h * np.eye(3 * n)  # (..., 3n, 3n)
    tau = rnea_batch(tb, z[..., :n], z[..., n:2 * n], z[..., 2 * n:])  # (..., 3n, n)
    return np.swapaxes(tau.imag / h, -1, -2)


class TorqueProblem:
    """Constants of one torque-MPC problem family (shared by the port and the literal NLP)."""

    def __init__(self, robot: OracleRobot, link, T=30, dt=0.1, w_path=1000.0, w_tau=1e-3, w_vel=0.0, tau_lim=None):
        self.robot, self.link, self.T, self.dt = robot, link, T, dt
        self.w_path, self.w_tau, self.w_vel = w_path, w_tau, w_vel
        self.tb = RneaTables(robot)
        self.chain = FoldedChain(robot, link)
        self.n = robot.ndof
        assert self.tb.ndof == self.n
        eff = np.array([j.limit["effort"] for j in robot.joints if j.type != "fixed"]) if tau_lim is None else np.broadcast_to(
            np.asarray(tau_lim, float), (self.n,)).copy()
        self.tau_lo, self.tau_up = -eff, eff

    def rollout(self, qc, dqc, U):
        T, dt = self.T, self.dt
        Q = np.zeros((T, self.n))
        dQ = np.zeros((T, self.n))
        Q[0], dQ[0] = qc, dqc
        for t in range(T - 1):
            Q[t + 1] = Q[t] + dt * dQ[t]
            dQ[t + 1] = dQ[t] + dt * U[t]
        return Q, dQ

    def goal_figure_eight(self, qc, scale=1.0):
        """Synthetic goal of SURVEY 8(d) C5: figure-eight offset (figure_eight_plan.py:90-96 pattern, first 3 s of it) in the
        end-effector frame at qc."""
        e, Re, _, _ = self.chain.fk(np.asarray(qc)[None])
        ts = np.arange(self.T) * self.dt
        loc = scale * np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros_like(ts)], 1)
        return e[0][None] + loc @ Re[0].T


def riccati_torque(H, g, mu, dt, mu_u=None):
    """Backward/forward sweep for  min sum_t 1/2 dz_t^T (H_t + mu I) dz_t + g_t^T dz_t,  dz_t = (dx_t (2n), du_t (n)),
    dx_{t+1} = A dx_t + B du_t, dx_0 = 0, A = [[I, dt I], [0, I]], B = [0; dt I].  Returns (dz (T, 3n), ok, qk) with
    qk = sum_t qu_t^T k_t (the damped model's optimal value is -qk/2)."""
    T, m = g.shape
    n = m // 3
    nx = 2 * n
    A = np.eye(nx)
    A[:n, n:] = dt * np.eye(n)
    Bm = np.zeros((nx, n))
    Bm[n:] = dt * np.eye(n)
    P = np.zeros((nx, nx))
    p = np.zeros(nx)
    Ks, ks = [None] * T, [None] * T
    qk = 0.0
    for t in range(T - 1, -1, -1):
        Ht = H[t] + np.diag(np.concatenate([mu * np.ones(nx), (mu if mu_u is None else mu_u) * np.ones(n)]))
        Qxx = Ht[:nx, :nx] + A.T @ P @ A
        Qux = Ht[nx:, :nx] + Bm.T @ P @ A
        Quu = Ht[nx:, nx:] + Bm.T @ P @ Bm
        qx = g[t, :nx] + A.T @ p
        qu = g[t, nx:] + Bm.T @ p
        try:
            L = np.linalg.cholesky(Quu)
        except np.linalg.LinAlgError:
            return None, False, 0.0
        K = np.linalg.solve(L.T, np.linalg.solve(L, Qux))
        k = np.linalg.solve(L.T, np.linalg.solve(L, qu))
        P = Qxx - Qux.T @ K
        P = 0.5 * (P + P.T)
        p = qx - Qux.T @ k
        Ks[t], ks[t] = K, k
        qk += float(qu @ k)
    dz = np.zeros((T, m))
    dx = np.zeros(nx)
    for t in range(T):
        du = -ks[t] - Ks[t] @ dx
        dz[t, :nx], dz[t, nx:] = dx, du
        dx = A @ dx + Bm @ du
    return dz, True, qk


def costate_gradient(g, dt):
    """Gradient of the rolled-out objective w.r.t. the free variables u_t from the stage gradients g_t = dl_t/d(x_t, u_t)."""
    T, m = g.shape
    n = m // 3
    lam = np.zeros(2 * n)
    gu = np.zeros((T, n))
    for t in range(T - 1, -1, -1):
        gu[t] = g[t, 2 * n:] + dt * lam[n:]
        gx = g[t, :2 * n]
        lam = np.concatenate([gx[:n] + lam[:n], gx[n:] + dt * lam[:n] + lam[n:]])
    return gu


def anderson_mix(hx, hf):
    """Anderson extrapolation from the stored (control sequence, Gauss-Newton step) pairs, oldest first (k_tq_step, csrc/oh_torque.hip):
    x_a = x_k + f_k - sum_j gamma_j (dx_j + df_j) with gamma = argmin |f_k - dF gamma|, through the regularised normal equations
    (the kernel factorises the <= 3 x 3 Gram matrix in registers).  None if the Gram matrix is not positive definite."""
    F, X = np.array(hf), np.array(hx)
    dF, dX = F[1:] - F[:-1], X[1:] - X[:-1]
    G = dF @ dF.T
    G = G + 1e-10 * max(float(np.max(np.diag(G))), 1e-300) * np.eye(len(dF))
    try:
        L = np.linalg.cholesky(G)
    except np.linalg.LinAlgError:
        return None
    gam = np.linalg.solve(L.T, np.linalg.solve(L, dF @ F[-1]))
    return X[-1] + F[-1] - gam @ (dX + dF)


def solve_torque_lm(prob: TorqueProblem, qc, dqc, goal, U0=None, max_iter=300, tol=1e-6, tol_feas=1e-9, rho0=None, mu0=0.0, verbose=False, damp_u=0.0, hess_extra=None,
                    aa_m=3, aa_from=1e-1, vlimits=None):
    """The state machine of csrc/oh_torque.hip in numpy (one instance).

    aa_m > 0: Anderson acceleration of the Gauss-Newton iteration.  The tracking residual does not vanish at the optimum, so Gauss-Newton
    converges linearly (rate ~0.8 on the med7 problem: 43 of the 56 steps of the nominal instance go by while the objective changes in its
    11th digit); its steps z_k = U_{k+1} - U_k are the residuals of a fixed-point iteration, and mixing the last aa_m + 1 of them (Walker & Ni
    2011) predicts where the sequence is heading.  Once the reduced gradient is below aa_from every other trial is the extrapolated point
    instead of the Levenberg-Marquardt step: accepted if it lowers the merit at all, otherwise the history is dropped and the plain step
    follows.  49 -> 27 steps on the golden instances."""
    T, n, dt = prob.T, prob.n, prob.dt
    wp, wt, wv = prob.w_path, prob.w_tau, prob.w_vel
    lo, up = prob.tau_lo, prob.tau_up
    U = np.zeros((T, n)) if U0 is None else np.array(U0, float)
    lam = np.zeros((T, 2 * n))
    # vlimits = (vlo, vup): rows dq_t - vlo >= 0, vup - dq_t >= 0 on the velocity states (enforce_model_limits(name, time_deriv=1),
    # builder.py:471-509; round 3): stage-local rows of the state, same penalty and outer loop as the effort rows; adds "lam_v" (T, 2n)
    vel = vlimits is not None
    lam_v = np.zeros((T, 2 * n))
    if vel:
        vlo, vup = (np.broadcast_to(np.asarray(v, dtype=float), (n,)) for v in vlimits)
    rho = rho_next = (1.0 if rho0 is None else rho0)
    omega = max(tol, 1e-2)
    meas_prev = np.inf

    def evalp(U, lam, rho, lam_v=lam_v):
        Q, dQ = prob.rollout(qc, dqc, U)
        tau = rnea_batch(prob.tb, Q, dQ, U)
        J = rnea_jacobian(prob.tb, Q, dQ, U)  # (T, n, 3n)
        e, _, Jp, _ = prob.chain.jac(Q)
        r = e - goal
        gv = np.concatenate([tau - lo, up - tau], 1)
        s = np.maximum(0.0, lam - rho * gv)
        psi = (s * s - lam * lam) / (2.0 * rho)
        phi = wp * np.sum(r * r, 1) + wt * np.sum(tau * tau, 1) + wv * np.sum(dQ * dQ, 1) + psi.sum(1)
        c = 2.0 * wt * tau - s[:, :n] + s[:, n:]
        d = 2.0 * wt + rho * ((s[:, :n] > 0).astype(float) + (s[:, n:] > 0).astype(float))
        g = np.einsum("ti,tid->td", c, J)
        g[:, :n] += 2.0 * wp * np.einsum("tki,tk->ti", Jp, r)
        g[:, n:2 * n] += 2.0 * wv * dQ
        H = np.einsum("ti,tid,tie->tde", d, J, J)
        H[:, :n, :n] += 2.0 * wp * np.einsum("tki,tkj->tij", Jp, Jp)
        H[:, n:2 * n, n:2 * n] += 2.0 * wv * np.eye(n)
        if hess_extra is not None:
            H = H + hess_extra(Q, dQ, U, c)
        meas = float(np.abs(np.minimum(gv, lam / rho)).max())
        gw = np.zeros((T, 0))
        if vel:
            gw = np.concatenate([dQ - vlo, vup - dQ], 1)
            sv = np.maximum(0.0, lam_v - rho * gw)
            phi = phi + ((sv * sv - lam_v * lam_v) / (2.0 * rho)).sum(1)
            g[:, n:2 * n] += -sv[:, :n] + sv[:, n:]
            act = rho * ((sv[:, :n] > 0).astype(float) + (sv[:, n:] > 0).astype(float))
            H[:, np.arange(n, 2 * n), np.arange(n, 2 * n)] += act
            meas = max(meas, float(np.abs(np.minimum(gw, lam_v / rho)).max()))
        return float(phi.sum()), g, H, np.concatenate([gv, gw], 1), meas, (Q, dQ, tau, J, c)

    mu, nun = mu0, 2.0
    iters = rejected = outers = 0
    hx, hf = [], []  # Anderson history: accepted control sequences and the steps taken from them
    aa_trial = aa_was = False
    first, outer = True, False
    Ut = U
    cur = None
    status = 1
    pred = 0.0
    while True:
        if outer:
            lam = np.maximum(0.0, lam - rho * cur["gv"][:, : 2 * n])
            if vel:
                lam_v = np.maximum(0.0, lam_v - rho * cur["gv"][:, 2 * n :])
            rho = rho_next
            outers += 1
            hx, hf = [], []  # the merit function changes
        f_t, g, H, gv, meas_t, traj = evalp(Ut, lam, rho, lam_v)
        if first or outer:
            accept, first, outer = True, False, False
            aa_trial = aa_was = False
        elif aa_trial:
            accept = bool(np.isfinite(f_t) and f_t < cur["f"])
            aa_trial, aa_was = False, True
            if not accept:
                hx, hf = [], []
                rejected += 1
        else:
            aa_was = False
            ratio = (cur["f"] - f_t) / max(pred, 1e-300)
            accept = np.isfinite(f_t) and (ratio > 1e-4 or (pred <= 1e-15 * abs(cur["f"]) and f_t <= cur["f"] + 1e-14 * abs(cur["f"])))
            if accept:
                mu *= max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3)
                mu = 0.0 if mu < 1e-7 else mu
                nun = 2.0
            else:
                mu = max(mu * nun, 1e-3)
                nun *= 2.0
                rejected += 1
        if accept:
            cur = {"U": Ut, "f": f_t, "g": g, "H": H, "gv": gv, "meas": meas_t, "traj": traj}
        stat = float(np.max(np.abs(costate_gradient(cur["g"], dt))))
        if verbose:
            print(f"  steps {iters:3d} f={cur['f']:.12f} stat={stat:.3e} meas={cur['meas']:.3e} mu={mu:.3g} rho={rho:.1e} omega={omega:.1e} outers={outers}")
        if stat <= omega:
            meas = cur["meas"]
            if stat <= tol and meas <= tol_feas:
                status = 0
                break
            if iters >= max_iter:
                break
            rho_next = min(rho * 10.0, 1e8) if meas > 0.25 * meas_prev else rho
            meas_prev = meas
            omega = max(tol, min(omega, 0.1 * meas))
            outer = True
            Ut = cur["U"]
            iters += 1
            continue
        if iters >= max_iter:
            break
        while True:
            dz, ok, qk = riccati_torque(cur["H"], cur["g"], mu, dt, damp_u * mu)
            if ok:
                break
            mu = max(4.0 * mu, 1e-2)
        pred = 0.5 * qk + 0.5 * mu * float(np.sum(dz[:, :2 * n] ** 2)) + 0.5 * damp_u * mu * float(np.sum(dz[:, 2 * n:] ** 2))
        Ut = cur["U"] + dz[:, 2 * n:]
        if aa_m > 0 and stat < aa_from:
            hx.append(cur["U"].reshape(-1).copy())
            hf.append(dz[:, 2 * n:].reshape(-1).copy())
            hx, hf = hx[-(aa_m + 1):], hf[-(aa_m + 1):]
            if len(hx) >= 2 and not aa_was:
                xa = anderson_mix(hx, hf)
                if xa is not None:
                    Ut = xa.reshape(T, n)
                    aa_trial = True
        iters += 1
    Q, dQ, tau = cur["traj"][:3]
    gv = np.concatenate([tau - lo, up - tau], 1)
    lam_out = np.maximum(0.0, lam - rho * gv)
    e, _, _, _ = prob.chain.fk(Q)
    f_true = float(wp * np.sum((e - goal) ** 2) + wt * np.sum(tau * tau) + wv * np.sum(dQ * dQ))
    out = {"U": cur["U"], "Q": Q, "dQ": dQ, "tau": tau, "f": f_true, "iters": iters, "rejected": rejected, "outers": outers, "stat": stat,
           "meas": cur["meas"], "status": status, "lam": lam_out}
    if vel:
        out["lam_v"] = np.maximum(0.0, lam_v - rho * np.concatenate([dQ - vlo, vup - dQ], 1))
    return out


# ---- second derivatives of the inverse dynamics (round 4) ---------------------------------------------------------------------------------
# The Lagrangian of the torque problem holds the dynamics rows h = TAU - rnea(Q, dQ, ddQ) (builder.py:354, models.py:1731-1884), so its exact
# Hessian -- what the reference obtains by AD of the CasADi graph, optimization.py:8-24 -- needs  sum_i c_i d^2 tau_i / dz^2  for a multiplier
# vector c.  By the principle of virtual work  c^T tau = sum_b f_b . v_b(c) + n_b . w_b(c):  the inertial wrench of every body paired with the
# velocity the joint rates c would give it.  Both factors come out of ONE outward recursion over the bodies (the reference's forward pass next to a
# twist propagation), so the gradient is ONE inward adjoint recursion, written out by hand below; the Hessian is that gradient differentiated once
# more -- complex steps here, dual numbers in csrc/oh_torque.hip:rnea_ctau_grad -- and nothing is differenced.
def rnea_virtual_work(tb: RneaTables, q, qd, qdd, c):
    """c^T rnea(q, qd, qdd) by the outward recursion alone (checked against rnea_batch in tests/test_torque_cpu.py)."""
    return _vw_forward(tb, np.asarray(q), np.asarray(qd), np.asarray(qdd), np.asarray(c))[0]


def _rot(tb, i, qi):
    s, c = np.sin(qi)[..., None, None], np.cos(qi)[..., None, None]
    return tb.R0[i] @ (np.eye(3) + s * tb.K[i] + (1.0 - c) * (tb.K[i] @ tb.K[i]))


def _mv(M, v):
    return (M @ v[..., None])[..., 0]


def _vw_forward(tb, q, qd, qdd, c):
    dt_ = np.result_type(q.dtype, qd.dtype, qdd.dtype, c.dtype, float)
    shp = np.broadcast_shapes(q.shape, qd.shape, qdd.shape, c.shape)[:-1]
    n = tb.n
    z = np.zeros(shp + (3,), dt_)
    om, omD, vD, wc, vo = z, z, z + np.array([0.0, 0.0, 9.81]), z, z
    phi = np.zeros(shp, dt_)
    tape = []
    for i in range(n):
        mov = i != n - 1
        R = _rot(tb, i, q[..., i]) if mov else np.broadcast_to(tb.R0[i], shp + (3, 3)).astype(dt_)
        Rt = np.swapaxes(R, -1, -2)
        r = np.broadcast_to(tb.xyz[i], shp + (3,))
        omp, omDp, wcp = _mv(Rt, om), _mv(Rt, omD), _mv(Rt, wc)
        acc = vD + _cross(omD, r) + _cross(om, _cross(om, r))
        w = vo + _cross(wc, r)
        if mov:
            a = Rt @ tb.axis[i]
            aq = a * qd[..., i][..., None]
            omi = omp + aq
            omDi = omDp + _cross(omp, aq) + a * qdd[..., i][..., None]
            wci = wcp + a * c[..., i][..., None]
        else:
            a, aq = z, z
            omi, omDi, wci = omp, omDp, wcp
        vDi, voi = _mv(Rt, acc), _mv(Rt, w)
        cm = np.broadcast_to(tb.cm[i], shp + (3,))
        fi = tb.m[i] * (vDi + _cross(omDi, cm) + _cross(omi, _cross(omi, cm)))
        Io = _mv(tb.I[i], omi)
        ni = _mv(tb.I[i], omDi) + _cross(omi, Io)
        vci = voi + _cross(wci, cm)
        phi = phi + np.sum(fi * vci, -1) + np.sum(ni * wci, -1)
        tape.append((R, om, omD, wc, omp, omDp, wcp, a, aq, acc, w, omi, omDi, wci, vDi, voi, fi, ni, vci, Io))
        om, omD, vD, wc, vo = omi, omDi, vDi, wci, voi
    return phi, tape


def rnea_ctau_gradient(tb: RneaTables, q, qd, qdd, c):
    """d (c^T rnea) / d (q, qd, qdd): (..., 3 ndof) by the hand-written adjoint of _vw_forward (real or complex arguments)."""
    q, qd, qdd, c = (np.asarray(v) for v in (q, qd, qdd, c))
    _, tape = _vw_forward(tb, q, qd, qdd, c)
    n, nd = tb.n, tb.ndof
    shp = tape[0][1].shape[:-1]
    dt_ = tape[-1][11].dtype
    z = np.zeros(shp + (3,), dt_)
    b_om, b_omD, b_vD, b_wc, b_vo = z, z, z, z, z  # adjoints of body i's (om, omD, vD, wc, vo), filled by its child
    gq, gqd, gqdd = [None] * nd, [None] * nd, [None] * nd
    dot = lambda x, y: np.sum(x * y, -1)[..., None]
    for i in range(n - 1, -1, -1):
        R, om_p, omD_p, wc_p, omp, omDp, wcp, a, aq, acc, w, omi, omDi, wci, vDi, voi, fi, ni, vci, Io = tape[i]
        cm = np.broadcast_to(tb.cm[i], shp + (3,))
        r = np.broadcast_to(tb.xyz[i], shp + (3,))
        I = tb.I[i]
        m = tb.m[i]
        # local term f_i . vc_i + n_i . wc_i
        b_vo = b_vo + fi
        b_wc = b_wc + _cross(cm, fi) + ni
        b_vD = b_vD + m * vci
        b_omD = b_omD + m * _cross(cm, vci) + _mv(I.T, wci)
        b_om = b_om + m * (vci * dot(omi, cm) + cm * dot(omi, vci) - 2.0 * omi * dot(cm, vci)) + _mv(I.T, _cross(wci, omi)) + _cross(Io, wci)
        # through the step of body i
        if i != n - 1:
            b_omp = b_om + _cross(aq, b_omD)
            b_aq = b_om + _cross(b_omD, omp)
            b_a = b_aq * qd[..., i][..., None] + b_omD * qdd[..., i][..., None] + b_wc * c[..., i][..., None]
            gqd[i] = dot(b_aq, a)[..., 0]
            gqdd[i] = dot(b_omD, a)[..., 0]
            k = tb.axis[i]
            sw = _cross(omp, b_omp) + _cross(omDp, b_omD) + _cross(wcp, b_wc) + _cross(a, b_a) + _cross(vDi, b_vD) + _cross(voi, b_vo)
            gq[i] = -np.sum(sw * k, -1)
        else:
            b_omp = b_om
        b_acc = _mv(R, b_vD)
        b_w = _mv(R, b_vo)
        n_om = _mv(R, b_omp) + b_acc * dot(om_p, r) + r * dot(om_p, b_acc) - 2.0 * om_p * dot(r, b_acc)
        n_omD = _mv(R, b_omD) + _cross(r, b_acc)
        n_vD = b_acc
        n_wc = _mv(R, b_wc) + _cross(r, b_w)
        n_vo = b_w
        b_om, b_omD, b_vD, b_wc, b_vo = n_om, n_omD, n_vD, n_wc, n_vo
    return np.concatenate([np.stack(gq, -1), np.stack(gqd, -1), np.stack(gqdd, -1)], -1)


def rnea_ctau_hessian(tb: RneaTables, q, qd, qdd, c, h=1e-30):
    """sum_i c_i d^2 tau_i / d (q, qd, qdd)^2: (..., 3 ndof, 3 ndof), complex-step derivative of rnea_ctau_gradient (exact to rounding; the
    ddq-ddq block is zero: the torques are linear in the accelerations)."""
    q, qd, qdd, c = (np.asarray(v, float) for v in (q, qd, qdd, c))
    n = tb.ndof
    zz = np.concatenate([np.broadcast_to(q, np.broadcast_shapes(q.shape, qd.shape, qdd.shape)), np.broadcast_to(qd, np.broadcast_shapes(q.shape, qd.shape, qdd.shape)),
                         np.broadcast_to(qdd, np.broadcast_shapes(q.shape, qd.shape, qdd.shape))], -1)[..., None, :] + 1j * h * np.eye(3 * n)
    g = rnea_ctau_gradient(tb, zz[..., :n], zz[..., n:2 * n], zz[..., 2 * n:], c[..., None, :])  # (..., 3n directions, 3n)
    H = g.imag / h
    return 0.5 * (H + np.swapaxes(H, -1, -2))


# ---- d tau / d (q, dq, ddq) in closed form (round 4; csrc/oh_torque.hip:rnea_idsva is this, lane by lane) ------------------------------------------
# World-frame spatial vectors (w, v_O) / (n_O, f) about the world origin (Featherstone 2008).  S_l = (z_l, o_l x z_l), v_b = sum S_l dq_l,
# a_b = a_0 + sum (S_l ddq_l + v_l x S_l dq_l), W_b = I_b a_b + v_b x* I_b v_b, tau_k = S_k . sum_{b >= k} W_b  -- what RobotModel.rnea computes
# (models.py:1819-1880) -- and, by the product rule with dS_l/dq_m = S_m x S_l, dI_b/dq_m = S_m x* I_b - I_b S_m x and the Jacobi identity,
#   d tau_k / d ddq_j = S_k . I^C_m S_j,   d tau_k / d dq_j = 2 S_k . (B^C_m S_j + I^C_m Sd_j),
#   d tau_k / d q_j = S_k . (I^C_m Sdd_j + 2 B^C_m Sd_j [+ S_j x* F^C_j if k <= j]),      m = max(k, j),
# Sd_j = v_j x S_j, Sdd_j = a_j x S_j + v_j x Sd_j, 2 B_b x = (Xi_b w_x, -2 p_b x w_x) (Carpentier & Mansard 2018; Singh, Russell & Wensing 2022).
# Valid for tables that describe a rigid-body chain (unit axes that the joint-origin rotation leaves in place: the reference adds the angular
# velocity iRp @ axis, models.py:1821-1823); checked against the complex-step derivative of the literal recursion in tests/test_torque_cpu.py.
def _skew(v):
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[...,2], v[...,1]],-1), np.stack([v[...,2], z, -v[...,0]],-1), np.stack([-v[...,1], v[...,0], z],-1)],-2)

def _mx(x, y):   # motion cross  x × y
    return (_cross(x[0], y[0]), _cross(x[0], y[1]) + _cross(x[1], y[0]))
def _fx(x, f):   # force cross  x ×* f
    return (_cross(x[0], f[0]) + _cross(x[1], f[1]), _cross(x[0], f[1]))
def _inert(I, x):  # I = (m, h, A): n = A w + h × v ; f = m v − h × w
    m, h, A = I
    return (_mv(A, x[0]) + _cross(h, x[1]), m[..., None] * x[1] - _cross(h, x[0]))
def _dot6(x, f):
    return np.sum(x[0]*f[0], -1) + np.sum(x[1]*f[1], -1)

def rnea_jacobian_spatial(tb: RneaTables, q, qd, qdd):
    """-> tau (..., n), J (..., n, 3 n) = d tau / d (q, dq, ddq)."""
    q, qd, qdd = (np.asarray(a, float) for a in (q, qd, qdd))
    shp = q.shape[:-1]; n = tb.ndof; NB = tb.n
    Rw = np.broadcast_to(np.eye(3), shp + (3, 3)); ow = np.zeros(shp + (3,))
    v = (np.zeros(shp + (3,)), np.zeros(shp + (3,)))
    a = (np.zeros(shp + (3,)), np.zeros(shp + (3,)) + np.array([0, 0, 9.81]))
    S, Sd, Sdd, Ib, Xi, pP, W = [], [], [], [], [], [], []
    for i in range(NB):
        o_i = ow + _mv(Rw, np.broadcast_to(tb.xyz[i], ow.shape))
        if i < n:
            z = _mv(Rw, np.broadcast_to(tb.axis[i], ow.shape))          # velocity axis (world) = R_{parent} axis
            R_i = Rw @ _rot(tb, i, q[..., i])
            Si = (z, _cross(o_i, z))
            Sdi = _mx(v, Si)                                              # v_parent × S_i = v_i × S_i
            v = (v[0] + Si[0] * qd[..., i, None], v[1] + Si[1] * qd[..., i, None])
            a = (a[0] + Si[0] * qdd[..., i, None] + Sdi[0] * qd[..., i, None], a[1] + Si[1] * qdd[..., i, None] + Sdi[1] * qd[..., i, None])
            Sddi = tuple(x + y for x, y in zip(_mx(a, Si), _mx(v, Sdi)))
            S.append(Si); Sd.append(Sdi); Sdd.append(Sddi)
        else:
            R_i = Rw @ tb.R0[i]
        c = o_i + _mv(R_i, np.broadcast_to(tb.cm[i], ow.shape))
        Ic = R_i @ tb.I[i] @ np.swapaxes(R_i, -1, -2)
        m = np.broadcast_to(tb.m[i], shp)
        h = m[..., None] * c
        A = Ic - m[..., None, None] * (_skew(c) @ _skew(c))
        I_i = (m, h, A)
        P = _inert(I_i, v)
        Wi = tuple(x + y for x, y in zip(_inert(I_i, a), _fx(v, P)))
        Om, V, H = _skew(v[0]), _skew(v[1]), _skew(h)
        Xi_i = Om @ A - A @ Om - V @ H - H @ V - _skew(P[0])
        Ib.append(I_i); Xi.append(Xi_i); pP.append(P[1]); W.append(Wi)
        Rw, ow = R_i, o_i
    # composites, inward
    J = np.zeros(shp + (n, 3 * n)); tau = np.zeros(shp + (n,))
    mC = np.zeros(shp); hC = np.zeros(shp + (3,)); AC = np.zeros(shp + (3, 3)); XC = np.zeros(shp + (3, 3)); pC = np.zeros(shp + (3,))
    FC = (np.zeros(shp + (3,)), np.zeros(shp + (3,)))
    comp = [None] * n
    for b in range(NB - 1, -1, -1):
        mC = mC + Ib[b][0]; hC = hC + Ib[b][1]; AC = AC + Ib[b][2]; XC = XC + Xi[b]; pC = pC + pP[b]
        FC = (FC[0] + W[b][0], FC[1] + W[b][1])
        if b < n:
            comp[b] = ((mC, hC, AC), XC, pC, FC)
            tau[..., b] = _dot6(S[b], FC)
    def B2(XC, pC, x):  # 2 B^C x
        return (_mv(XC, x[0]), -2.0 * _cross(pC, x[0]))
    for j in range(n):
        for k in range(n):
            mm = max(k, j)
            IC, XC, pC, FC = comp[mm]
            u2 = _inert(IC, S[j])
            b1 = B2(XC, pC, S[j]); i1 = _inert(IC, Sd[j])
            u1 = (b1[0] + 2 * i1[0], b1[1] + 2 * i1[1])
            b0 = B2(XC, pC, Sd[j]); i0 = _inert(IC, Sdd[j])
            u0 = (b0[0] + i0[0], b0[1] + i0[1])
            if k <= j:
                e = _fx(S[j], comp[j][3])
                u0 = (u0[0] + e[0], u0[1] + e[1])
            J[..., k, j] = _dot6(S[k], u0); J[..., k, n + j] = _dot6(S[k], u1); J[..., k, 2 * n + j] = _dot6(S[k], u2)
    return tau, J

