"""ORACLE (test infrastructure, not product code) -- the reference's example problems restated as
numpy NLPs in the reference's own x / p / v layout.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Each class mirrors what ``OptimizationBuilder.build()`` hands to a ``Solver``
(optas/optimization.py:54-309): ``nx, np, nk, na, ng, nh, nv`` and callables ``f, df, k, a, g, h, v, dv``
of ``(x, p)`` with

  * x = SXContainer.vec() of the decision variables, blocks in creation order, each block
    column-major (sx_container.py:83-89; builder.py:90-99),
  * every constraint stored as ``rhs - lhs`` (builder.py:313,354),
  * v = [k; g; a; -a; h; -h] >= 0 (optimization.py:27-51,292-306).

The reference obtains derivatives by CasADi AD (optimization.py:8-24); here they are analytic and
finite-difference-checked in tests/test_oracle_problems.py.  PARITY UNPINNED at the solve level: the
reference holds no golden solution for these KUKA problems (tests/test_examples.py checks the exit code
only); what *is* pinned is listed in DESIGN.md.
"""
import numpy as np

from .robot import OracleRobot


def _hamilton_left_pure(o, quat):
    """(ox,oy,oz,0) (x) quat, xyzw storage."""
    ox, oy, oz = o
    x, y, z, w = quat
    return np.array(
        [
            ox * w + oy * z - oz * y,
            -ox * z + oy * w + oz * x,
            ox * y - oy * x + oz * w,
            -ox * x - oy * y - oz * z,
        ]
    )


class _NLPBase:
    nk = na = ng = nh = 0

    @property
    def nv(self):  # optimization.py:301
        return self.nk + self.ng + 2 * self.na + 2 * self.nh

    # default empty blocks
    def k(self, x, p):
        return np.zeros(0)

    def dk(self, x, p):
        return np.zeros((0, self.nx))

    def a(self, x, p):
        return np.zeros(0)

    def da(self, x, p):
        return np.zeros((0, self.nx))

    def g(self, x, p):
        return np.zeros(0)

    def dg(self, x, p):
        return np.zeros((0, self.nx))

    def h(self, x, p):
        return np.zeros(0)

    def dh(self, x, p):
        return np.zeros((0, self.nx))

    def v(self, x, p):  # optimization.py:47-51
        a, h = self.a(x, p), self.h(x, p)
        return np.concatenate([self.k(x, p), self.g(x, p), a, -a, h, -h])

    def dv(self, x, p):
        da, dh = self.da(x, p), self.dh(x, p)
        return np.concatenate([self.dk(x, p), self.dg(x, p), da, -da, dh, -dh], axis=0)


class BoothNLP(_NLPBase):
    """tests/test_solver.py:19-54: f=(x+a*y-b)^2+(2x+y-5)^2, p=(a,b); known answer (1,3) at (2,7)."""

    nx, np_ = 2, 2

    def f(self, x, p):
        return (x[0] + p[0] * x[1] - p[1]) ** 2 + (2.0 * x[0] + x[1] - 5.0) ** 2

    def df(self, x, p):
        r1 = x[0] + p[0] * x[1] - p[1]
        r2 = 2.0 * x[0] + x[1] - 5.0
        return np.array([2 * r1 + 4 * r2, 2 * p[0] * r1 + 2 * r2])

    def ddf(self, x, p):
        return np.array([[2 + 8.0, 2 * p[0] + 4.0], [2 * p[0] + 4.0, 2 * p[0] ** 2 + 2.0]])


class IKExampleNLP(_NLPBase):
    """example/example.py:13-60 (BASELINE config 1).

    x = "kuka/q/x" (7x1); p = ["kuka/q/p" (0 rows); "q_nominal"(7); "p_goal"(3)];
    f = ||q-qn||^2 (:30); h = p_goal - p_ee(q) (:26 with builder.py:354); k = [q-lo; up-q]
    (:33, builder.py:334-335,509).  Class QuadraticCostNonlinearConstraints.
    """

    def __init__(self, robot: OracleRobot, link="end_effector_ball"):
        self.robot, self.link = robot, link
        self.n = robot.ndof
        self.nx, self.np_ = self.n, self.n + 3
        self.nk, self.nh = 2 * self.n, 3
        self.lo = robot.lower_actuated_joint_limits
        self.up = robot.upper_actuated_joint_limits

    def f(self, x, p):
        return float(np.sum((x - p[: self.n]) ** 2))

    def df(self, x, p):
        return 2.0 * (x - p[: self.n])

    def ddf(self, x, p):
        return 2.0 * np.eye(self.n)

    def k(self, x, p):
        return np.concatenate([x - self.lo, self.up - x])

    def dk(self, x, p):
        return np.concatenate([np.eye(self.n), -np.eye(self.n)], axis=0)

    def h(self, x, p):
        return p[self.n :] - self.robot.get_global_link_position(self.link, x)

    def dh(self, x, p):
        return -self.robot.get_global_link_linear_jacobian(self.link, x)


class FigureEightNLP(_NLPBase):
    """example/figure_eight_plan.py:16-113 (BASELINE config 2).

    x = ["{name}/q/x" (ndof x T); "{name}/dq/x" (ndof x (T-1))]  (builder.py:90-99)
    p = ["{name}/q/p" (0); "{name}/dq/p" (0); "qc" (ndof)]         (:59-61)
    a = [ qc - q_0 ;  0 - dq_0 ;  -(q_t + dt*dq_t - q_{t+1}) t=0..T-2 ]   (:64-75, builder.py:437,469,354)
    h = quat_c - quat(q_t), t=0..T-1, 4 rows each                          (:105-107)
    f = 1000*sum_t ||path_t - p_ee(q_t)||^2 + 0.01*sum ||dQ||^2            (:99,103)
    path_t = p(qc) + R(qc) @ (0.2 sin(pi/2 t_s), 0.1 sin(pi t_s), 0), t_s = linspace(0,Tmax,T) (:34-37,90-96)
    """

    def __init__(self, robot: OracleRobot, link, T=50, Tmax=10.0, w_path=1000.0, w_vel=0.01):
        self.robot, self.link, self.T, self.Tmax = robot, link, T, Tmax
        self.w_path, self.w_vel = w_path, w_vel
        self.n = n = robot.ndof
        ts = np.linspace(0.0, Tmax, T)
        self.ts = ts
        self.dt = float(ts[1] - ts[0])
        self.local_path = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
        self.nq = n * T
        self.ndq = n * (T - 1)
        self.nx = self.nq + self.ndq
        self.np_ = n
        self.na = 2 * n + n * (T - 1)
        self.nh = 4 * T
        # constant Jacobian of the linear equalities
        A = np.zeros((self.na, self.nx))
        I = np.eye(n)
        A[0:n, 0:n] = -I  # qc - q_0
        A[n : 2 * n, self.nq : self.nq + n] = -I  # 0 - dq_0
        for t in range(T - 1):
            r = 2 * n + n * t
            A[r : r + n, n * t : n * t + n] = -I
            A[r : r + n, self.nq + n * t : self.nq + n * t + n] = -self.dt * I
            A[r : r + n, n * (t + 1) : n * (t + 1) + n] = I
        self._A = A

    # -- layout helpers -------------------------------------------------------------------------------
    def split(self, x):
        n, T = self.n, self.T
        Q = x[: self.nq].reshape(T, n).T  # column-major vec of (n x T)
        dQ = x[self.nq :].reshape(T - 1, n).T
        return Q, dQ

    def join(self, Q, dQ):
        return np.concatenate([Q.T.reshape(-1), dQ.T.reshape(-1)])

    def seed(self, qc):
        """Planner.reset (:117-124): Q0 = qc repeated, dQ0 = 0 (missing key zero-filled, sx_container.py:121)."""
        return self.join(np.tile(np.asarray(qc, float).reshape(-1, 1), (1, self.T)), np.zeros((self.n, self.T - 1)))

    def references(self, p):
        qc = p
        pc = self.robot.get_global_link_position(self.link, qc)
        Rc = self.robot.get_global_link_rotation(self.link, qc)
        quatc = self.robot.get_global_link_quaternion(self.link, qc)
        path = pc.reshape(3, 1) + Rc @ self.local_path
        return path, quatc

    # -- cost ---------------------------------------------------------------------------------------------
    def f(self, x, p):
        Q, dQ = self.split(x)
        path, _ = self.references(p)
        pos = self.robot.map_position(self.link, Q)
        return float(self.w_path * np.sum((path - pos) ** 2) + self.w_vel * np.sum(dQ**2))

    def df(self, x, p):
        Q, dQ = self.split(x)
        path, _ = self.references(p)
        gq = np.zeros((self.n, self.T))
        for t in range(self.T):
            Jp = self.robot.get_global_link_linear_jacobian(self.link, Q[:, t])
            r = path[:, t] - self.robot.get_global_link_position(self.link, Q[:, t])
            gq[:, t] = -2.0 * self.w_path * (Jp.T @ r)
        return self.join(gq, 2.0 * self.w_vel * dQ)

    # -- constraints -----------------------------------------------------------------------------------
    def a(self, x, p):
        b = np.zeros(self.na)
        b[: self.n] = p
        return self._A @ x + b

    def da(self, x, p):
        return self._A

    def h(self, x, p):
        Q, _ = self.split(x)
        _, quatc = self.references(p)
        return (quatc.reshape(4, 1) - self.robot.map_quaternion(self.link, Q)).T.reshape(-1)

    def dh(self, x, p):
        Q, _ = self.split(x)
        J = np.zeros((self.nh, self.nx))
        for t in range(self.T):
            J[4 * t : 4 * t + 4, self.n * t : self.n * t + self.n] = -self.robot.quaternion_jacobian(self.link, Q[:, t])
        return J

    # -- second-order information (not a reference callable for the IPOPT path: CasADi derives
    #    nlp_hess_l by AD; restated analytically, FD-checked in tests) --------------------------------------
    def hess_lagrangian(self, x, p, lam_h, gauss_newton=False):
        """Hessian wrt x of  f(x) + lam_h . h(x)   (linear rows have none)."""
        Q, dQ = self.split(x)
        path, quatc = self.references(p)
        n, T = self.n, self.T
        H = np.zeros((self.nx, self.nx))
        for t in range(T):
            q = Q[:, t]
            J = self.robot.get_global_link_geometric_jacobian(self.link, q)
            Jp, Jw = J[:3], J[3:]
            W = 2.0 * self.w_path * (Jp.T @ Jp)
            if not gauss_newton:
                r = path[:, t] - self.robot.get_global_link_position(self.link, q)
                quat = self.robot.get_global_link_quaternion(self.link, q)
                lam = lam_h[4 * t : 4 * t + 4]
                for i in range(n):
                    for j in range(i, n):
                        # d2 p / dqi dqj = z_i x (z_j x (e - p_j)) = z_i x Jp_j   (i <= j, revolute)
                        d2p = np.cross(Jw[:, i], Jp[:, j])
                        val = -2.0 * self.w_path * float(r @ d2p)
                        # d2 quat/dqi dqj = 1/2 (dz_j/dq_i,0)(x)quat + 1/4 (z_j,0)(x)(z_i,0)(x)quat, dz_j/dq_i = z_i x z_j (i<j)
                        dz = np.cross(Jw[:, i], Jw[:, j]) if i < j else np.zeros(3)
                        d2quat = 0.5 * _hamilton_left_pure(dz, quat) + 0.25 * _hamilton_left_pure(
                            Jw[:, j], _hamilton_left_pure(Jw[:, i], quat)
                        )
                        val += float(lam @ (-d2quat))  # h = quatc - quat
                        W[i, j] += val
                        if j != i:
                            W[j, i] += val
            H[n * t : n * t + n, n * t : n * t + n] = W
        H[self.nq :, self.nq :] = 2.0 * self.w_vel * np.eye(self.ndq)
        return H


class PointMassMPCNLP(_NLPBase):
    """example/point_mass_mpc.py Controller (:88-154), BASELINE config 3, SURVEY App. B.3.

    TaskModel "point_mass", dim 2, time_derivs [0,1], derivs_align=True, T=20, dt=0.05.
    x = ["point_mass/y/x" (2xT); "point_mass/dy/x" (2xT)]                                (builder.py:90-99)
    p = [curr(2); dcurr(2); vec(goal 2xT); vec(obs 2xT)]                                   (:104-107)
    k = [Y+1.5; 1.5-Y; dY+1; 1-dY]   (enforce_model_limits d=0, d=1 -> "_l", "_r" blocks)  (:110-111)
    a = [-(y_t + dt dy_t - y_{t+1}) t=0..T-2; curr - y_0; dcurr - dy_0]                    (:114-118)
    g = ||obs_t - y_t||^2 - (0.2+0.1)^2, t=0..T-1                                          (:121-127)
    f = sum ||goal - Y||^2 + (0.0025/T) sum ||(dy_{t+1}-dy_t)/dt||^2                        (:130-136)
    """

    def __init__(self, T=20, dt=0.05, ylim=1.5, dylim=1.0, safe=0.3, w_acc=0.0025):
        self.T, self.dt, self.ylim, self.dylim = T, dt, ylim, dylim
        self.safe_sq = safe**2
        self.w = w_acc / float(T)
        self.nx, self.np_ = 4 * T, 4 + 4 * T
        self.nk, self.na, self.ng = 8 * T, 2 * (T - 1) + 4, T
        n = 2
        A = np.zeros((self.na, self.nx))
        I = np.eye(n)
        for t in range(T - 1):
            r = n * t
            A[r : r + n, n * t : n * t + n] = -I
            A[r : r + n, 2 * T + n * t : 2 * T + n * t + n] = -dt * I
            A[r : r + n, n * (t + 1) : n * (t + 1) + n] = I
        r = n * (T - 1)
        A[r : r + n, 0:n] = -I  # curr - y_0
        A[r + n : r + 2 * n, 2 * T : 2 * T + n] = -I  # dcurr - dy_0
        self._A = A
        Ik = np.eye(2 * T)
        Z = np.zeros((2 * T, 2 * T))
        self._M = np.block([[Ik, Z], [-Ik, Z], [Z, Ik], [Z, -Ik]])

    def split(self, x):
        T = self.T
        return x[: 2 * T].reshape(T, 2).T, x[2 * T :].reshape(T, 2).T

    def split_p(self, p):
        T = self.T
        return p[0:2], p[2:4], p[4 : 4 + 2 * T].reshape(T, 2).T, p[4 + 2 * T :].reshape(T, 2).T

    @staticmethod
    def pack_p(curr, dcurr, goal, obs):
        return np.concatenate([np.asarray(curr, float), np.asarray(dcurr, float), np.asarray(goal, float).T.reshape(-1), np.asarray(obs, float).T.reshape(-1)])

    def f(self, x, p):
        Y, dY = self.split(x)
        _, _, goal, _ = self.split_p(p)
        dd = (dY[:, 1:] - dY[:, :-1]) / self.dt
        return float(np.sum((goal - Y) ** 2) + self.w * np.sum(dd**2))

    def df(self, x, p):
        Y, dY = self.split(x)
        _, _, goal, _ = self.split_p(p)
        gY = -2.0 * (goal - Y)
        dd = (dY[:, 1:] - dY[:, :-1]) / self.dt
        gdY = np.zeros_like(dY)
        gdY[:, 1:] += 2.0 * self.w * dd / self.dt
        gdY[:, :-1] -= 2.0 * self.w * dd / self.dt
        return np.concatenate([gY.T.reshape(-1), gdY.T.reshape(-1)])

    def ddf(self, x, p):
        T = self.T
        H = np.zeros((self.nx, self.nx))
        H[: 2 * T, : 2 * T] = 2.0 * np.eye(2 * T)
        c = 2.0 * self.w / self.dt**2
        for t in range(T - 1):
            for j in range(2):
                a, b = 2 * T + 2 * t + j, 2 * T + 2 * (t + 1) + j
                H[a, a] += c
                H[b, b] += c
                H[a, b] -= c
                H[b, a] -= c
        return H

    def k(self, x, p):
        Y, dY = self.split(x)
        y, dy = Y.T.reshape(-1), dY.T.reshape(-1)
        return np.concatenate([y + self.ylim, self.ylim - y, dy + self.dylim, self.dylim - dy])

    def dk(self, x, p):
        return self._M

    def a(self, x, p):
        curr, dcurr, _, _ = self.split_p(p)
        b = np.zeros(self.na)
        b[2 * (self.T - 1) : 2 * (self.T - 1) + 2] = curr
        b[2 * (self.T - 1) + 2 :] = dcurr
        return self._A @ x + b

    def da(self, x, p):
        return self._A

    def g(self, x, p):
        Y, _ = self.split(x)
        _, _, _, obs = self.split_p(p)
        return np.sum((obs - Y) ** 2, axis=0) - self.safe_sq

    def dg(self, x, p):
        Y, _ = self.split(x)
        _, _, _, obs = self.split_p(p)
        J = np.zeros((self.ng, self.nx))
        for t in range(self.T):
            J[t, 2 * t : 2 * t + 2] = -2.0 * (obs[:, t] - Y[:, t])
        return J


def point_mass_tick_parameters(t0=2.0, T=20, dt=0.05, curr=(-0.45, -0.35), dcurr=(0.6, 0.6), ramp=0.032):
    """The MPC tick of BASELINE.md section 5 / SURVEY App. D: moving obstacle of point_mass_mpc.py:293-306 at time
    t0, goal = straight ramp from curr."""
    obs, goal = [], []
    for i in range(T):
        ti = t0 + dt * i
        alpha = ti * np.pi - np.pi
        obs.append([0.15 * np.sin(alpha), 0.15 * np.cos(alpha) + 0.15])
        goal.append([curr[0] + ramp * i, curr[1] + ramp * i])
    return PointMassMPCNLP.pack_p(curr, dcurr, np.array(goal).T, np.array(obs).T)


def dual_arm_offsets(T=50):
    """Piecewise-linear end-effector path of example/dual_arm.py:82-113 relative to pos0 = p(qc):
    pos1 = pos0 + d1, pos2 = pos1 + d2.  Returns {'l': (3,T), 'r': (3,T)} offsets."""
    d1 = {"l": np.array([-0.1, 0.1, -0.2]), "r": np.array([-0.1, -0.1, -0.2])}
    d2 = np.array([0.0, 0.0, 0.3])
    out = {}
    for arm in ("l", "r"):
        off = np.zeros((3, T))
        for i in range(T):
            a_ = float(i) / float(T - 1)
            if a_ < 0.4:
                off[:, i] = (a_ / 0.4) * d1[arm]
            elif a_ < 0.5:
                off[:, i] = d1[arm]
            else:
                off[:, i] = d1[arm] + ((a_ - 0.5) / 0.5) * d2
        out[arm] = off
    return out


class DualArmNLP(_NLPBase):
    """example/dual_arm.py:17-129 as shipped (BASELINE config 4 without the synthetic extensions).

    Two kuka_lwr models "kukal"/"kukar" with add_base_frame("global_world", xyz=(0, -/+0.25, 0)) (:13-14, 134-141).
    x = [vec(Ql 7xT); vec(dQl 7x(T-1)); vec(Qr); vec(dQr)]   p = [qcl(7); qcr(7)]
    a = [qcl - ql_0; qcr - qr_0; -(ql_t + dt dql_t - ql_{t+1}); same for r]                 (:41-55)
    f = 0.01 (sum dQl^2 + sum dQr^2) + sum ||p_l(ql_t) - path_l,t||^2 + sum ||p_r(qr_t) - path_r,t||^2   (:76-117)
    No inequality rows, no nonlinear equalities: class NonlinearCostLinearConstraints.  The arms are separable.
    """

    def __init__(self, robot_l: OracleRobot, robot_r: OracleRobot, link="end_effector_ball", T=50, Tmax=10.0, w_dq=0.01):
        self.robots = {"l": robot_l, "r": robot_r}
        self.link, self.T = link, T
        ts = np.linspace(0.0, Tmax, T)
        self.dt = float(ts[1] - ts[0])
        self.w_dq = w_dq
        self.n = n = robot_l.ndof
        self.nx1 = n * T + n * (T - 1)
        self.nx, self.np_ = 2 * self.nx1, 2 * n
        self.na = 2 * n + 2 * n * (T - 1)
        self.offsets = dual_arm_offsets(T)
        A = np.zeros((self.na, self.nx))
        I = np.eye(n)
        for k in range(2):
            base = k * self.nx1
            A[k * n : (k + 1) * n, base : base + n] = -I
            for t in range(T - 1):
                r = 2 * n + k * n * (T - 1) + n * t
                A[r : r + n, base + n * t : base + n * t + n] = -I
                A[r : r + n, base + n * T + n * t : base + n * T + n * t + n] = -self.dt * I
                A[r : r + n, base + n * (t + 1) : base + n * (t + 1) + n] = I
        self._A = A

    def split(self, x):
        n, T = self.n, self.T
        out = {}
        for k, arm in enumerate(("l", "r")):
            xs = x[k * self.nx1 : (k + 1) * self.nx1]
            out[arm] = (xs[: n * T].reshape(T, n).T, xs[n * T :].reshape(T - 1, n).T)
        return out

    def f(self, x, p):
        s = self.split(x)
        val = 0.0
        for k, arm in enumerate(("l", "r")):
            Q, dQ = s[arm]
            qc = p[k * self.n : (k + 1) * self.n]
            path = self.robots[arm].get_global_link_position(self.link, qc).reshape(3, 1) + self.offsets[arm]
            pos = self.robots[arm].map_position(self.link, Q)
            val += self.w_dq * np.sum(dQ**2) + np.sum((pos - path) ** 2)
        return float(val)

    def df(self, x, p):
        s = self.split(x)
        parts = []
        for k, arm in enumerate(("l", "r")):
            Q, dQ = s[arm]
            qc = p[k * self.n : (k + 1) * self.n]
            path = self.robots[arm].get_global_link_position(self.link, qc).reshape(3, 1) + self.offsets[arm]
            gq = np.zeros_like(Q)
            for t in range(self.T):
                Jp = self.robots[arm].get_global_link_linear_jacobian(self.link, Q[:, t])
                gq[:, t] = 2.0 * Jp.T @ (self.robots[arm].get_global_link_position(self.link, Q[:, t]) - path[:, t])
            parts += [gq.T.reshape(-1), (2.0 * self.w_dq * dQ).T.reshape(-1)]
        return np.concatenate(parts)

    def a(self, x, p):
        b = np.zeros(self.na)
        b[: 2 * self.n] = p
        return self._A @ x + b

    def da(self, x, p):
        return self._A

    def hess_lagrangian_v(self, x, p, sigma, lam_v):
        """sigma d2f (the rows are linear): per arm and knot 2 Jp^T Jp + 2 sum_k r_k d2p_k with d2p/dq_i dq_j = z_i x Jp_j (i <= j), plus
        2 w_dq I on the velocities.  For oracle/ipm_reference_form.py; checked against differences of df in tests/test_ipm_reference_form.py."""
        s = self.split(x)
        n, T = self.n, self.T
        H = np.zeros((self.nx, self.nx))
        for k, arm in enumerate(("l", "r")):
            Q, _ = s[arm]
            qc = p[k * n : (k + 1) * n]
            rob = self.robots[arm]
            path = rob.get_global_link_position(self.link, qc).reshape(3, 1) + self.offsets[arm]
            base = k * self.nx1
            for t in range(T):
                J = rob.get_global_link_geometric_jacobian(self.link, Q[:, t])
                Jp, Jw = J[:3], J[3:]
                r = rob.get_global_link_position(self.link, Q[:, t]) - path[:, t]
                W = 2.0 * Jp.T @ Jp
                for i in range(n):
                    for j in range(i, n):
                        v = 2.0 * float(r @ np.cross(Jw[:, i], Jp[:, j]))
                        W[i, j] += v
                        if j != i:
                            W[j, i] += v
                H[base + n * t : base + n * t + n, base + n * t : base + n * t + n] = W
            H[base + n * T : base + self.nx1, base + n * T : base + self.nx1] = 2.0 * self.w_dq * np.eye(n * (T - 1))
        return sigma * H


class GuardedDualArmNLP(DualArmNLP):
    """BASELINE config 4 with the synthetic extensions of SURVEY 8(a) H4 / 8(d) C4: example/dual_arm.py plus, per arm,
    enforce_model_limits (builder.py:471-509) and sphere_collision_avoidance_constraints (builder.py:366-417).

    p = [qcl(7); qcr(7); per arm: link radii (L); per obstacle: position (3), radius (1)]      (builder.py:391-405 order)
    k = [Ql - lo; up - Ql; Qr - lo; up - Qr]   each block vec of a 7 x T array                 (rows "_l", "_r")
    g = per arm, knot-major, link, obstacle:  ||p_link(q_t) - o||^2 - (r_link + r_o)^2          (builder.py:407-415)
    """

    def __init__(self, robot_l, robot_r, links, n_obs, link="end_effector_ball", T=50, Tmax=10.0, w_dq=0.01, limits=True, vlimits=None):
        """vlimits = (vlo, vup): enforce_model_limits(name, time_deriv=1) per arm after the position limits (builder.py:471-509):
        k gains [dQl - vlo; vup - dQl; dQr - vlo; vup - dQr], each block vec of a 7 x (T-1) array."""
        super().__init__(robot_l, robot_r, link=link, T=T, Tmax=Tmax, w_dq=w_dq)
        from .structured import FoldedChain

        self.links, self.n_obs, self.limits = list(links), n_obs, limits
        self.vlimits = None if vlimits is None else (np.asarray(vlimits[0], float), np.asarray(vlimits[1], float))
        self.chains = {arm: FoldedChain(self.robots[arm], link) for arm in ("l", "r")}
        L = len(self.links)
        self.npar_arm = L + 4 * n_obs
        self.np_ = 2 * self.n + 2 * self.npar_arm
        self.nk = (4 * self.n * T if limits else 0) + (4 * self.n * (T - 1) if vlimits is not None else 0)
        self.ng = 2 * T * L * n_obs

    def arm_params(self, p, k):
        n, L = self.n, len(self.links)
        q = p[2 * n + k * self.npar_arm : 2 * n + (k + 1) * self.npar_arm]
        ob = q[L:].reshape(self.n_obs, 4)
        return q[:L], ob[:, :3], ob[:, 3]

    def _guards(self, p, k):
        from .guarded import Guards

        lr, op, orad = self.arm_params(p, k)
        return Guards(lo=None, up=None, links=self.links, link_radii=lr, obs_pos=op, obs_radii=orad)

    def k(self, x, p):
        if not self.limits and self.vlimits is None:
            return np.zeros(0)
        s = self.split(x)
        out = []
        if self.limits:
            for arm in ("l", "r"):
                Q = s[arm][0]
                lo, up = self.robots[arm].lower_actuated_joint_limits, self.robots[arm].upper_actuated_joint_limits
                out += [(Q - lo[:, None]).T.reshape(-1), (up[:, None] - Q).T.reshape(-1)]
        if self.vlimits is not None:
            vlo, vup = self.vlimits
            for arm in ("l", "r"):
                dQ = s[arm][1]
                out += [(dQ - vlo[:, None]).T.reshape(-1), (vup[:, None] - dQ).T.reshape(-1)]
        return np.concatenate(out)

    def dk(self, x, p):
        n, T = self.n, self.T
        M = np.zeros((self.nk, self.nx))
        r0 = 0
        if self.limits:
            I = np.eye(n * T)
            for k in range(2):
                base = k * self.nx1
                M[(2 * k) * n * T : (2 * k + 1) * n * T, base : base + n * T] = I
                M[(2 * k + 1) * n * T : (2 * k + 2) * n * T, base : base + n * T] = -I
            r0 = 4 * n * T
        if self.vlimits is not None:
            m = n * (T - 1)
            I = np.eye(m)
            for k in range(2):
                base = k * self.nx1 + n * T
                M[r0 + (2 * k) * m : r0 + (2 * k + 1) * m, base : base + m] = I
                M[r0 + (2 * k + 1) * m : r0 + (2 * k + 2) * m, base : base + m] = -I
        return M

    def g(self, x, p):
        from .guarded import guard_values

        s = self.split(x)
        return np.concatenate([guard_values(self.chains[arm], s[arm][0].T, self._guards(p, k))[0].reshape(-1) for k, arm in enumerate(("l", "r"))])

    def dg(self, x, p):
        from .guarded import guard_values

        n, T = self.n, self.T
        s = self.split(x)
        per = T * len(self.links) * self.n_obs
        M = np.zeros((self.ng, self.nx))
        for k, arm in enumerate(("l", "r")):
            d = guard_values(self.chains[arm], s[arm][0].T, self._guards(p, k))[1]  # (T, L*O, n)
            rows = d.shape[1]
            for t in range(T):
                M[k * per + t * rows : k * per + (t + 1) * rows, k * self.nx1 + n * t : k * self.nx1 + n * (t + 1)] = d[t]
        return M

    def hess_lagrangian_v(self, x, p, sigma, lam_v):
        """DualArmNLP's sigma d2f plus the curvature of the sphere rows, sum_i lam_g[i] d2 g_i (g = ||c_l(q_t) - o||^2 - (r_l + r_o)^2,
        builder.py:407-415; d2 g = 2 J^T J + 2 sum_k d_k d2 c_k from oracle.guarded.guard_values): what the reference's AD hands nlpsol.  The rows
        k and a are linear.  Round 4, for the interior-point goldens of config 4 at its BASELINE size (tools/make_golden.py --ipm-config4)."""
        from .guarded import guard_values

        H = super().hess_lagrangian_v(x, p[: 2 * self.n], sigma, lam_v)
        if self.ng:
            n, T = self.n, self.T
            s = self.split(x)
            per = T * len(self.links) * self.n_obs
            lam_g = np.asarray(lam_v[self.nk : self.nk + self.ng], float)
            for k, arm in enumerate(("l", "r")):
                w = lam_g[k * per : (k + 1) * per].reshape(T, -1)
                HW = guard_values(self.chains[arm], s[arm][0].T, self._guards(p, k), weights=w)[2]
                base = k * self.nx1
                for t in range(T):
                    H[base + n * t : base + n * t + n, base + n * t : base + n * t + n] += HW[t]
        return H

    def a(self, x, p):
        b = np.zeros(self.na)
        b[: 2 * self.n] = p[: 2 * self.n]
        return self._A @ x + b

    def _paths(self, p):
        return p[: 2 * self.n]

    def f(self, x, p):
        return super().f(x, p[: 2 * self.n])

    def df(self, x, p):
        return super().df(x, p[: 2 * self.n])


class LimitedFigureEightNLP(FigureEightNLP):
    """example/figure_eight_plan.py plus builder.enforce_model_limits(kuka_name) with limits (lo, up) (builder.py:471-509):
    k = [vec(Q - lo); vec(up - Q)] (rows "_l", "_r"), everything else as FigureEightNLP."""

    def __init__(self, robot, link, lo=None, up=None, vlo=None, vup=None, **kw):
        """lo/up: joint-position limits (None: no such rows); vlo/vup: joint-velocity limits, enforce_model_limits(name, time_deriv=1) added after
        the position limits: k = [vec(Q - lo); vec(up - Q); vec(dQ - vlo); vec(vup - dQ)]."""
        super().__init__(robot, link, **kw)
        f = lambda v: None if v is None else np.asarray(v, dtype=float)
        self.lo, self.up, self.vlo, self.vup = f(lo), f(up), f(vlo), f(vup)
        self.nk = (2 * self.n * self.T if self.lo is not None else 0) + (2 * self.n * (self.T - 1) if self.vlo is not None else 0)

    def k(self, x, p):
        Q, dQ = self.split(x)
        parts = []
        if self.lo is not None:
            parts += [(Q - self.lo[:, None]).T.reshape(-1), (self.up[:, None] - Q).T.reshape(-1)]
        if self.vlo is not None:
            parts += [(dQ - self.vlo[:, None]).T.reshape(-1), (self.vup[:, None] - dQ).T.reshape(-1)]
        return np.concatenate(parts)

    def dk(self, x, p):
        nq, ndq = self.n * self.T, self.n * (self.T - 1)
        M = np.zeros((self.nk, self.nx))
        r = 0
        if self.lo is not None:
            M[:nq, :nq] = np.eye(nq)
            M[nq : 2 * nq, :nq] = -np.eye(nq)
            r = 2 * nq
        if self.vlo is not None:
            M[r : r + ndq, nq:] = np.eye(ndq)
            M[r + ndq : r + 2 * ndq, nq:] = -np.eye(ndq)
        return M


class GuardedFigureEightNLP(FigureEightNLP):
    """example/figure_eight_plan.py plus enforce_model_limits (optional) and sphere_collision_avoidance_constraints (builder.py:366-417).
    p = [qc(7); link radii (L); per obstacle: position (3), radius (1)];  k as LimitedFigureEightNLP;  g knot-major, link, obstacle."""

    def __init__(self, robot, link, links, n_obs, lo=None, up=None, **kw):
        super().__init__(robot, link, **kw)
        from .structured import FoldedChain

        self.links, self.n_obs = list(links), n_obs
        self.lo = None if lo is None else np.asarray(lo, dtype=float)
        self.up = None if up is None else np.asarray(up, dtype=float)
        self.nk = 2 * self.n * self.T if lo is not None else 0
        self.ng = self.T * len(self.links) * n_obs
        self.np_ = self.n + len(self.links) + 4 * n_obs
        self.chain = FoldedChain(robot, link)

    def _guards(self, p):
        from .guarded import Guards

        L = len(self.links)
        ob = p[self.n + L :].reshape(self.n_obs, 4)
        return Guards(lo=None, up=None, links=self.links, link_radii=p[self.n : self.n + L], obs_pos=ob[:, :3], obs_radii=ob[:, 3])

    def f(self, x, p):
        return super().f(x, p[: self.n])

    def df(self, x, p):
        return super().df(x, p[: self.n])

    def a(self, x, p):
        return super().a(x, p[: self.n])

    def da(self, x, p):
        return super().da(x, p[: self.n])

    def h(self, x, p):
        return super().h(x, p[: self.n])

    def dh(self, x, p):
        return super().dh(x, p[: self.n])

    def k(self, x, p):
        if self.lo is None:
            return np.zeros(0)
        Q, _ = self.split(x)
        return np.concatenate([(Q - self.lo[:, None]).T.reshape(-1), (self.up[:, None] - Q).T.reshape(-1)])

    def dk(self, x, p):
        M = np.zeros((self.nk, self.nx))
        if self.lo is not None:
            nq = self.n * self.T
            M[:nq, :nq] = np.eye(nq)
            M[nq:, :nq] = -np.eye(nq)
        return M

    def g(self, x, p):
        from .guarded import guard_values

        Q, _ = self.split(x)
        return guard_values(self.chain, Q.T, self._guards(p))[0].reshape(-1)

    def dg(self, x, p):
        from .guarded import guard_values

        Q, _ = self.split(x)
        d = guard_values(self.chain, Q.T, self._guards(p))[1]  # (T, L*O, n)
        rows = d.shape[1]
        M = np.zeros((self.ng, self.nx))
        for t in range(self.T):
            M[t * rows : (t + 1) * rows, self.n * t : self.n * (t + 1)] = d[t]
        return M


class PointMassPlannerNLP(PointMassMPCNLP):
    """example/point_mass_planner.py Planner (:8-55): the point mass of config 3 planned once over T = 45 knots (dt = 0.1).

    p = [init(2); goal(2)];  k as PointMassMPCNLP;
    a = [-(y_t + dt dy_t - y_{t+1}); init - y_0; 0 - dy_0; 0 - dy_{T-1}]                         (:28-35)
    g = ||obs - y_t||^2 - 0.3^2 with the constant obstacle obs = (0, 0)                           (:37-42)
    f = ||goal - y_{T-1}||^2 + (0.01/T) sum ||dY||^2 + (0.005/T) sum ||(dy_{t+1}-dy_t)/dt||^2     (:43-51)
    """

    def __init__(self, T=45, dt=0.1, ylim=1.5, dylim=1.0, safe=0.3, w_vel=0.01, w_acc=0.005, obstacle=(0.0, 0.0)):
        super().__init__(T=T, dt=dt, ylim=ylim, dylim=dylim, safe=safe, w_acc=w_acc)
        self.w_vel = w_vel / float(T)
        self.obstacle = np.asarray(obstacle, dtype=float)
        self.np_ = 4
        self.na = 2 * (T - 1) + 6
        A = np.zeros((self.na, self.nx))
        A[: 2 * (T - 1) + 4] = self._A
        r = 2 * (T - 1) + 4
        A[r : r + 2, 2 * T + 2 * (T - 1) : 2 * T + 2 * T] = -np.eye(2)  # 0 - dy_{T-1}
        self._A = A

    def f(self, x, p):
        Y, dY = self.split(x)
        dd = (dY[:, 1:] - dY[:, :-1]) / self.dt
        return float(np.sum((p[2:4] - Y[:, -1]) ** 2) + self.w_vel * np.sum(dY**2) + self.w * np.sum(dd**2))

    def df(self, x, p):
        Y, dY = self.split(x)
        gY = np.zeros_like(Y)
        gY[:, -1] = -2.0 * (p[2:4] - Y[:, -1])
        dd = (dY[:, 1:] - dY[:, :-1]) / self.dt
        gdY = 2.0 * self.w_vel * dY
        gdY[:, 1:] += 2.0 * self.w * dd / self.dt
        gdY[:, :-1] -= 2.0 * self.w * dd / self.dt
        return np.concatenate([gY.T.reshape(-1), gdY.T.reshape(-1)])

    def a(self, x, p):
        b = np.zeros(self.na)
        b[2 * (self.T - 1) : 2 * (self.T - 1) + 2] = p[0:2]
        return self._A @ x + b

    def g(self, x, p):
        Y, _ = self.split(x)
        return np.sum((self.obstacle[:, None] - Y) ** 2, axis=0) - self.safe_sq

    def dg(self, x, p):
        Y, _ = self.split(x)
        J = np.zeros((self.ng, self.nx))
        for t in range(self.T):
            J[t, 2 * t : 2 * t + 2] = -2.0 * (self.obstacle - Y[:, t])
        return J


class TorqueMPCNLP(_NLPBase):
    """BASELINE configs[4]: torque-control MPC with RNEA dynamics equality rows -- not a reference
    script (torque_control_example.py:198-200 calls rnea outside the optimiser); built from the reference's builder calls as
    listed in oracle/torque.py.  med7 (RobotModel.rnea needs a fixed first joint, models.py:1748-1749), derivs_align=True:

    x = ["{r}/q/x" (n x T); "{r}/dq/x" (n x T); "{r}/ddq/x" (n x T); "tau/y/x" (n x T)]           (builder.py:45,90-99)
    p = ["{r}/q/p", "{r}/dq/p", "{r}/ddq/p" (0 rows); "qc" (n); "dqc" (n); "goal" (3 x T)]        (builder.py:96-97,263-273)
    k = [vec(TAU) - lo; up - vec(TAU)]                     enforce_model_limits("tau")            (builder.py:334-335,509)
    a = [qc - q_0; dqc - dq_0; -(q_t + dt dq_t - q_{t+1}); -(dq_t + dt ddq_t - dq_{t+1})], t = 0..T-2  (builder.py:437,469,539,354)
    h = vec(TAU - rnea(Q, dQ, ddQ))                        add_equality_constraint(lhs=rnea, rhs=TAU)   (builder.py:354)
    f = w_path sum ||p_link(q_t) - goal_t||^2 + w_vel sum ||dQ||^2 + w_tau sum ||TAU||^2
    """

    def __init__(self, prob, vlimits=None):
        """vlimits = (vlo, vup): enforce_model_limits(name, time_deriv=1) after the effort limits (builder.py:471-509): k gains
        [vec(dQ) - vlo; vup - vec(dQ)] (round 3)."""
        from .torque import rnea_batch, rnea_jacobian  # the vectorised restatement of oracle.robot.rnea and its complex-step Jacobian

        self._rnea, self._rnea_jac = rnea_batch, rnea_jacobian
        self.prob = prob
        self.robot, self.link, self.T, self.dt = prob.robot, prob.link, prob.T, prob.dt
        self.n = n = prob.n
        T = self.T
        self.nb = n * T
        self.nx = 4 * self.nb
        self.np_ = 2 * n + 3 * T
        self.nk = 2 * self.nb
        self.na = 2 * n + 2 * n * (T - 1)
        self.nh = self.nb
        nb, dt, I = self.nb, self.dt, np.eye(n)
        A = np.zeros((self.na, self.nx))
        A[0:n, 0:n] = -I
        A[n:2 * n, nb:nb + n] = -I
        for d in range(2):  # integrate_model_states(name, d + 1, dt): block d -> rows of x^(d), derivative block d + 1
            for t in range(T - 1):
                r = 2 * n + d * n * (T - 1) + n * t
                A[r:r + n, d * nb + n * t:d * nb + n * t + n] = -I
                A[r:r + n, (d + 1) * nb + n * t:(d + 1) * nb + n * t + n] = -dt * I
                A[r:r + n, d * nb + n * (t + 1):d * nb + n * (t + 1) + n] = I
        self._A = A
        self.vlimits = None if vlimits is None else tuple(np.broadcast_to(np.asarray(v, float), (n,)) for v in vlimits)
        if self.vlimits is not None:
            self.nk = 4 * self.nb
        K = np.zeros((self.nk, self.nx))
        K[:nb, 3 * nb:] = np.eye(nb)
        K[nb:2 * nb, 3 * nb:] = -np.eye(nb)
        if self.vlimits is not None:
            K[2 * nb:3 * nb, nb:2 * nb] = np.eye(nb)
            K[3 * nb:, nb:2 * nb] = -np.eye(nb)
        self._K = K

    def split(self, x):
        n, T, nb = self.n, self.T, self.nb
        return tuple(x[i * nb:(i + 1) * nb].reshape(T, n) for i in range(4))  # rows = knots (the transposes of the n x T blocks)

    def join(self, Q, dQ, ddQ, TAU):
        return np.concatenate([np.asarray(a, float).reshape(-1) for a in (Q, dQ, ddQ, TAU)])

    def split_p(self, p):
        n = self.n
        return p[:n], p[n:2 * n], p[2 * n:].reshape(self.T, 3)

    @staticmethod
    def pack_p(qc, dqc, goal):
        return np.concatenate([np.asarray(qc, float), np.asarray(dqc, float), np.asarray(goal, float).reshape(-1)])

    def seed(self, qc):
        """Q = qc at every knot, everything else zero-filled (sx_container.py:121)."""
        z = np.zeros((self.T, self.n))
        return self.join(np.tile(np.asarray(qc, float), (self.T, 1)), z, z, z)

    def f(self, x, p):
        Q, dQ, _, TAU = self.split(x)
        goal = self.split_p(p)[2]
        pos = self.robot.map_position(self.link, Q.T).T
        w = self.prob
        return float(w.w_path * np.sum((pos - goal) ** 2) + w.w_vel * np.sum(dQ**2) + w.w_tau * np.sum(TAU**2))

    def df(self, x, p):
        Q, dQ, ddQ, TAU = self.split(x)
        goal = self.split_p(p)[2]
        w = self.prob
        gq = np.zeros_like(Q)
        for t in range(self.T):
            Jp = self.robot.get_global_link_linear_jacobian(self.link, Q[t])
            gq[t] = 2.0 * w.w_path * (Jp.T @ (self.robot.get_global_link_position(self.link, Q[t]) - goal[t]))
        return self.join(gq, 2.0 * w.w_vel * dQ, np.zeros_like(ddQ), 2.0 * w.w_tau * TAU)

    def k(self, x, p):
        tau = x[3 * self.nb:]
        lo, up = np.tile(self.prob.tau_lo, self.T), np.tile(self.prob.tau_up, self.T)
        rows = [tau - lo, up - tau]
        if self.vlimits is not None:
            dq = x[self.nb:2 * self.nb]
            rows += [dq - np.tile(self.vlimits[0], self.T), np.tile(self.vlimits[1], self.T) - dq]
        return np.concatenate(rows)

    def dk(self, x, p):
        return self._K

    def a(self, x, p):
        b = np.zeros(self.na)
        b[:2 * self.n] = p[:2 * self.n]
        return self._A @ x + b

    def da(self, x, p):
        return self._A

    def h(self, x, p):
        Q, dQ, ddQ, TAU = self.split(x)
        return (TAU - self._rnea(self.prob.tb, Q, dQ, ddQ)).reshape(-1)

    def hess_lagrangian_v(self, x, p, sigma, lam_v):
        """sigma d2f + sum_i lam_v[i] d2 v_i (what the reference's AD of the CasADi graph gives nlpsol as the Hessian of the Lagrangian,
        optimization.py:8-24, solver.py:346-363), exact: the rows k and a are linear; h = TAU - rnea(Q, dQ, ddQ) contributes
        -sum_i mu_h[t, i] d2 tau_i / d(q_t, dq_t, ddq_t)^2 with the signed mu_h = lam(h) - lam(-h) (oracle.torque.rnea_ctau_hessian: hand-written
        adjoint of the recursion, differentiated once more by complex steps); the tracking term has 2 w_p (Jp^T Jp + sum_k r_k d2 p_k / dq2)."""
        from .torque import rnea_ctau_hessian
        from .torque_ipm import position_curvature

        Q, dQ, ddQ, _ = self.split(x)
        goal = self.split_p(p)[2]
        n, nb, T, w = self.n, self.nb, self.T, self.prob
        o = self.nk + self.ng + 2 * self.na
        mu_h = (np.asarray(lam_v[o:o + self.nh]) - np.asarray(lam_v[o + self.nh:o + 2 * self.nh])).reshape(T, n)
        Hd = -rnea_ctau_hessian(w.tb, Q, dQ, ddQ, mu_h)  # (T, 3n, 3n)
        e, _, Jp, _ = w.chain.jac(Q)
        Hq = 2.0 * w.w_path * sigma * (np.einsum("tki,tkj->tij", Jp, Jp) + position_curvature(w.chain, Q, e - goal))
        H = np.zeros((self.nx, self.nx))
        for t in range(T):
            idx = np.concatenate([blk * nb + n * t + np.arange(n) for blk in range(3)])
            H[np.ix_(idx, idx)] += Hd[t]
            iq = n * t + np.arange(n)
            H[np.ix_(iq, iq)] += Hq[t]
        d = np.arange(nb)
        H[nb + d, nb + d] += 2.0 * w.w_vel * sigma
        H[3 * nb + d, 3 * nb + d] += 2.0 * w.w_tau * sigma
        return H

    def dh(self, x, p):
        Q, dQ, ddQ, _ = self.split(x)
        n, nb = self.n, self.nb
        J = self._rnea_jac(self.prob.tb, Q, dQ, ddQ)  # (T, n, 3n)
        D = np.zeros((self.nh, self.nx))
        for t in range(self.T):
            r = slice(n * t, n * t + n)
            for blk in range(3):
                D[r, blk * nb + n * t:blk * nb + n * t + n] = -J[t][:, blk * n:(blk + 1) * n]
            D[r, 3 * nb + n * t:3 * nb + n * t + n] = np.eye(n)
        return D


class FastFigureEightNLP(FigureEightNLP):
    """FigureEightNLP with its callables evaluated for all T knots at once (numpy arrays over the knots instead of a Python loop over the
    literal per-knot restatement).  Same layout, same rows, same signs; every member is checked against the literal class to 1e-12 at random
    points (tests/test_ipm_reference_form.py).  It exists so that oracle/ipm_reference_form.py -- hundreds of iterations on the 693-variable
    problem, each with an exact Lagrangian Hessian -- finishes in a minute instead of an hour; it is not a different formulation."""

    def __init__(self, robot, link, **kw):
        super().__init__(robot, link, **kw)
        from .structured import FoldedChain

        self._fc = FoldedChain(robot, link)

    def _kin(self, x):
        Q, dQ = self.split(x)
        e, Re, Jp, Jw = self._fc.jac(Q.T)  # (T,3), (T,3,3), (T,3,n), (T,3,n)
        return Q, dQ, e, Jp, Jw

    def f(self, x, p):
        Q, dQ = self.split(x)
        path, _ = self.references(p)
        e = self._fc.fk(Q.T)[0]
        return float(self.w_path * np.sum((path.T - e) ** 2) + self.w_vel * np.sum(dQ**2))

    def df(self, x, p):
        Q, dQ, e, Jp, _ = self._kin(x)
        path, _ = self.references(p)
        gq = -2.0 * self.w_path * np.einsum("tkj,tk->jt", Jp, path.T - e)
        return self.join(gq, 2.0 * self.w_vel * dQ)

    def h(self, x, p):
        Q, _ = self.split(x)
        _, quatc = self.references(p)
        return (quatc.reshape(1, 4) - self.robot.quaternion_batch(self.link, Q.T)).reshape(-1)

    @staticmethod
    def _left_pure(o, quat):
        """(o, 0) (x) quat for arrays o (..., 3), quat (..., 4): Hamilton product, xyzw storage (oracle/robot.py:quaternion_jacobian)."""
        ox, oy, oz = o[..., 0], o[..., 1], o[..., 2]
        x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
        return np.stack([ox * w + oy * z - oz * y, -ox * z + oy * w + oz * x, ox * y - oy * x + oz * w, -ox * x - oy * y - oz * z], axis=-1)

    def dh(self, x, p):
        Q, _, _, _, Jw = self._kin(x)
        quat = self.robot.quaternion_batch(self.link, Q.T)  # (T, 4)
        dq = 0.5 * self._left_pure(np.moveaxis(Jw, 1, 2), quat[:, None, :])  # (T, n, 4): column j = 1/2 (z_j, 0) (x) quat
        J = np.zeros((self.nh, self.nx))
        n = self.n
        for t in range(self.T):
            J[4 * t : 4 * t + 4, n * t : n * t + n] = -dq[t].T
        return J

    def hess_lagrangian(self, x, p, lam_h, gauss_newton=False):
        Q, dQ, e, Jp, Jw = self._kin(x)
        path, _ = self.references(p)
        n, T = self.n, self.T
        W = 2.0 * self.w_path * np.einsum("tki,tkj->tij", Jp, Jp)
        if not gauss_newton:
            r = path.T - e
            quat = self.robot.quaternion_batch(self.link, Q.T)
            lam = np.asarray(lam_h, float).reshape(T, 4)
            zi, zj = Jw[:, :, :, None], Jw[:, :, None, :]  # (T,3,n,1), (T,3,1,n)
            # d2p/dqi dqj = z_i x Jp_j (i <= j)
            d2p = np.cross(np.broadcast_to(zi, (T, 3, n, n)), np.broadcast_to(Jp[:, :, None, :], (T, 3, n, n)), axis=1)
            val = -2.0 * self.w_path * np.einsum("tk,tkij->tij", r, d2p)
            # d2quat/dqi dqj = 1/2 (z_i x z_j, 0)(x)quat [i < j] + 1/4 (z_j,0)(x)(z_i,0)(x)quat
            dz = np.cross(np.broadcast_to(zi, (T, 3, n, n)), np.broadcast_to(zj, (T, 3, n, n)), axis=1)  # (T,3,i,j)
            iu = np.triu(np.ones((n, n)), 1)
            dz = dz * iu[None, None]
            qq = quat[:, None, None, :]
            t1 = 0.5 * self._left_pure(np.moveaxis(dz, 1, -1), qq)  # (T,i,j,4)
            inner = self._left_pure(np.moveaxis(Jw, 1, 2)[:, :, None, :], qq)  # (z_i,0)(x)quat: (T,i,1,4)
            t2 = 0.25 * self._left_pure(np.moveaxis(Jw, 1, 2)[:, None, :, :], inner)  # (z_j,0)(x)that: (T,i,j,4)
            val = val + np.einsum("tc,tijc->tij", lam, -(t1 + t2))
            up = np.triu(np.ones((n, n)))[None]
            val = val * up
            val = val + np.swapaxes(val * np.triu(np.ones((n, n)), 1)[None], 1, 2)
            W = W + val
        H = np.zeros((self.nx, self.nx))
        for t in range(T):
            H[n * t : n * t + n, n * t : n * t + n] = W[t]
        H[self.nq :, self.nq :] = 2.0 * self.w_vel * np.eye(self.ndq)
        return H


class JointSpacePlannerNLP(_NLPBase):
    """example/simple_joint_space_planner.py:15-73 (a problem outside BASELINE's five configs: what SURVEY 8(f) rank 1, "arbitrary user
    problems", is exercised with), literal layout.

    x = ["{name}/q/x" (n x T); "{name}/dq/x" (n x T)]                 (derivs_align, builder.py:90-99)
    p = [nominal_joint_state (n); current_joint_state (n); position_goal (3); orientation_goal (4)]          (:22-25)
    a = [qc - q_0;  -(q_t + dt dq_t - q_{t+1}), t = 0..T-2;  0 - dq_{T-1}]         (:28, :38, :66; builder.py:525-539, 437-469)
    h = [pg - p_ee(q_{T-1});  og - quat_ee(q_{T-1})]                                  (:31-35)
    g = per knot [z_ee(q_t) + zpad;  z_elbow(q_t) + zpad]                            (:41-54, add_geq: lhs - rhs)
    f = sum_t 0.1 ||q_t - qn||^2 + 0.1 ||dQ||^2 + 10 ||(dQ[:, 1:] - dQ[:, :-1]) / dt||^2                (:45, :57-64)
    """

    def __init__(self, robot: OracleRobot, ee="lbr_link_ee", elbow="lbr_link_3", T=20, duration=4.0, zpad=0.05):
        self.robot, self.ee, self.elbow, self.T, self.zpad = robot, ee, elbow, T, zpad
        self.n = n = robot.ndof
        self.dt = duration / float(T - 1)
        self.nx, self.np_ = 2 * n * T, 2 * n + 7
        self.na, self.nh, self.ng = n + n * (T - 1) + n, 7, 2 * T
        A = np.zeros((self.na, self.nx))
        I = np.eye(n)
        A[:n, :n] = -I
        for t in range(T - 1):
            r = n + n * t
            A[r : r + n, n * t : n * t + n] = -I
            A[r : r + n, n * T + n * t : n * T + n * t + n] = -self.dt * I
            A[r : r + n, n * (t + 1) : n * (t + 1) + n] = I
        A[n + n * (T - 1) :, n * T + n * (T - 1) :] = -I
        self._A = A

    def split(self, x):
        n, T = self.n, self.T
        return x[: n * T].reshape(T, n).T, x[n * T :].reshape(T, n).T

    def seed(self, q0):
        return np.concatenate([np.tile(np.asarray(q0, float), self.T), np.zeros(self.n * self.T)])

    def f(self, x, p):
        Q, dQ = self.split(x)
        qn = p[: self.n]
        return float(0.1 * np.sum((Q - qn[:, None]) ** 2) + 0.1 * np.sum(dQ**2) + 10.0 * np.sum(((dQ[:, 1:] - dQ[:, :-1]) / self.dt) ** 2))

    def df(self, x, p):
        Q, dQ = self.split(x)
        qn = p[: self.n]
        gq = 0.2 * (Q - qn[:, None])
        gd = 0.2 * dQ
        dd = (dQ[:, 1:] - dQ[:, :-1]) * (20.0 / self.dt**2)
        gd[:, 1:] += dd
        gd[:, :-1] -= dd
        return np.concatenate([gq.T.reshape(-1), gd.T.reshape(-1)])

    def a(self, x, p):
        b = np.zeros(self.na)
        b[: self.n] = p[self.n : 2 * self.n]
        return self._A @ x + b

    def da(self, x, p):
        return self._A

    def h(self, x, p):
        qF = self.split(x)[0][:, -1]
        n = self.n
        return np.concatenate([p[2 * n : 2 * n + 3] - self.robot.get_global_link_position(self.ee, qF), p[2 * n + 3 :] - self.robot.get_global_link_quaternion(self.ee, qF)])

    def dh(self, x, p):
        qF = self.split(x)[0][:, -1]
        n, T = self.n, self.T
        J = np.zeros((7, self.nx))
        J[:3, n * (T - 1) : n * T] = -self.robot.get_global_link_linear_jacobian(self.ee, qF)
        J[3:, n * (T - 1) : n * T] = -self.robot.quaternion_jacobian(self.ee, qF)
        return J

    def g(self, x, p):
        Q, _ = self.split(x)
        out = np.zeros(self.ng)
        for t in range(self.T):
            out[2 * t] = self.robot.get_global_link_position(self.ee, Q[:, t])[2] + self.zpad
            out[2 * t + 1] = self.robot.get_global_link_position(self.elbow, Q[:, t])[2] + self.zpad
        return out

    def dg(self, x, p):
        Q, _ = self.split(x)
        n = self.n
        J = np.zeros((self.ng, self.nx))
        for t in range(self.T):
            J[2 * t, n * t : n * t + n] = self.robot.get_global_link_linear_jacobian(self.ee, Q[:, t])[2]
            J[2 * t + 1, n * t : n * t + n] = self.robot.get_global_link_linear_jacobian(self.elbow, Q[:, t])[2]
        return J

    def hess_lagrangian_v(self, x, p, sigma, lam_v):
        """sigma d2f + sum_i lam_v[i] d2v_i over v = [g; a; -a; h; -h] (what oracle/ipm_reference_form.py asks for): the cost is quadratic, the linear
        rows have no curvature; a height row contributes lam d2(p_z) of its link, the final-pose rows -mu d2p and -mu d2quat with the signed
        mu = lam(h) - lam(-h).  d2p/dq_i dq_j = z_i x Jp_j (i <= j), d2quat as in FigureEightNLP.hess_lagrangian."""
        n, T = self.n, self.T
        Q, _ = self.split(x)
        lg = lam_v[: self.ng]
        o = self.ng + 2 * self.na
        mu_h = lam_v[o : o + self.nh] - lam_v[o + self.nh : o + 2 * self.nh]
        H = np.zeros((self.nx, self.nx))
        H[: n * T, : n * T] = 0.2 * np.eye(n * T)
        D = np.zeros((T, T))
        for t in range(T - 1):
            D[t, t] += 1.0
            D[t + 1, t + 1] += 1.0
            D[t, t + 1] -= 1.0
            D[t + 1, t] -= 1.0
        H[n * T :, n * T :] = 0.2 * np.eye(n * T) + np.kron(20.0 / self.dt**2 * D, np.eye(n))
        H *= sigma

        def d2p(J):
            Jp, Jw = J[:3], J[3:]
            out = np.zeros((3, n, n))
            for i in range(n):
                for j in range(i, n):
                    out[:, i, j] = out[:, j, i] = np.cross(Jw[:, i], Jp[:, j])
            return out

        for t in range(T):
            q = Q[:, t]
            W = np.zeros((n, n))
            for link, lam in ((self.ee, lg[2 * t]), (self.elbow, lg[2 * t + 1])):
                if lam != 0.0:
                    W += lam * d2p(self.robot.get_global_link_geometric_jacobian(link, q))[2]
            if t == T - 1:
                J = self.robot.get_global_link_geometric_jacobian(self.ee, q)
                W -= np.einsum("k,kij->ij", mu_h[:3], d2p(J))
                Jw = J[3:]
                quat = self.robot.get_global_link_quaternion(self.ee, q)
                for i in range(n):
                    for j in range(i, n):
                        dz = np.cross(Jw[:, i], Jw[:, j]) if i < j else np.zeros(3)
                        d2q = 0.5 * _hamilton_left_pure(dz, quat) + 0.25 * _hamilton_left_pure(Jw[:, j], _hamilton_left_pure(Jw[:, i], quat))
                        val = -float(mu_h[3:] @ d2q)
                        W[i, j] += val
                        if j != i:
                            W[j, i] += val
            H[n * t : n * t + n, n * t : n * t + n] += W
        return H


class TorqueControlNLP(_NLPBase):
    """example/torque_control_example.py TrackingController (:19-104): a T = 1 velocity-level tracking step, handed to ``sqpmethod``.

    x = "{name}/dq/x" (7x1, RobotModel(time_derivs=[1]), derivs_align=True :29-42);  p = [qc (7); pg (7: goal position, quaternion xyzw)] (:44-45)
    dp = J(qc) dq (:54-57);  p_ee = Rc' (dt dp[:3]),  R_ee = Rc' (dt skew(dp[3:]) + I) (:69-70: the step seen from the current end-effector frame)
    Rg = Quaternion(pg[3:7]).getrotm() (:73-74, spatialmath.py:426-437 -- restated entry by entry in `getrotm`, with the reference's own
    off-textbook entries [0][2], [1][2], [2][1]);  pg_ee = -Rc' pc + Rc' pg[:3],  Rg_ee = Rc' Rg (:76-77)
    diffp = p_ee - pg_ee,  diffR = Rg_ee' R_ee (:79-80)
    f = diffp' diag(1e3) diffp + 0.01 ||dq||^2 + 10 ||diffR - I||_F^2 (:82-91)
    g = [1e-6 - diffp_x^2; 1e-8 - diffp_y^2; 1e-8 - diffp_z^2] >= 0 (:93-95 with builder.py:313: rhs - lhs).
    Class QuadraticCostNonlinearConstraints (f quadratic in dq, rows quadratic).  Every quantity is affine in dq, so the derivatives are exact.
    """

    def __init__(self, robot: OracleRobot, link="lbr_link_ee", dt=1.0 / 500.0, w_p=1e3, w_dq=0.01, w_ori=1e1, bounds=(1e-6, 1e-8, 1e-8)):
        self.robot, self.link, self.dt = robot, link, float(dt)
        self.n = robot.ndof
        self.nx, self.np_ = self.n, self.n + 7
        self.ng = 3
        self.w_p, self.w_dq, self.w_ori = float(w_p), float(w_dq), float(w_ori)
        self.bounds = np.asarray(bounds, dtype=np.float64)

    @staticmethod
    def getrotm(quat):
        """Quaternion.getrotm (spatialmath.py:426-437) of xyzw numbers, entry by entry as the reference writes it."""
        x, y, z, w = (float(v) for v in quat)
        return np.array([
            [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * y + 2 * w * y],
            [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * z],
            [2 * x * z - 2 * w * y, 2 * y * z + w * w * x, 1 - 2 * x * x - 2 * y * y],
        ])

    def pieces(self, p):
        """(A, b, C, c0): diffp = A dq - b (3 rows), vec(diffR - I) = C dq + c0 (9 rows, column-major)."""
        qc, pg = np.asarray(p[: self.n], float), np.asarray(p[self.n :], float)
        J = np.asarray(self.robot.get_global_link_geometric_jacobian(self.link, qc), float)
        pc = np.asarray(self.robot.get_global_link_position(self.link, qc), float).reshape(3)
        Rc = np.asarray(self.robot.get_global_link_rotation(self.link, qc), float)
        A = Rc.T @ (self.dt * J[:3])
        b = (-Rc.T @ pc + Rc.T @ pg[:3])
        Rg_ee = Rc.T @ self.getrotm(pg[3:])
        left = Rg_ee.T @ Rc.T
        C = np.zeros((9, self.n))
        for j in range(self.n):
            C[:, j] = (left @ (self.dt * _skew(J[3:, j]))).T.reshape(-1)
        c0 = (left - np.eye(3)).T.reshape(-1)
        return A, b, C, c0

    def f(self, x, p):
        A, b, C, c0 = self.pieces(p)
        d, r = A @ x - b, C @ x + c0
        return float(self.w_p * d @ d + self.w_dq * x @ x + self.w_ori * r @ r)

    def df(self, x, p):
        A, b, C, c0 = self.pieces(p)
        return 2.0 * self.w_p * A.T @ (A @ x - b) + 2.0 * self.w_dq * x + 2.0 * self.w_ori * C.T @ (C @ x + c0)

    def ddf(self, x, p):
        A, b, C, c0 = self.pieces(p)
        return 2.0 * self.w_p * A.T @ A + 2.0 * self.w_dq * np.eye(self.n) + 2.0 * self.w_ori * C.T @ C

    def g(self, x, p):
        A, b, _, _ = self.pieces(p)
        d = A @ x - b
        return self.bounds - d * d

    def dg(self, x, p):
        A, b, _, _ = self.pieces(p)
        return -2.0 * (A @ x - b)[:, None] * A

    def ddg_dot(self, x, p, lam_g):
        A, _, _, _ = self.pieces(p)
        return -2.0 * (A.T * np.asarray(lam_g, float)) @ A


def _skew(v):
    x, y, z = v
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def band_qp_exact(H, grad0, A, b, half):
    """min 1/2 x'Hx + grad0'x  s.t. |A_i x - b_i| <= half_i, H positive definite, by enumerating the 3^m active sets (row i free, at the lower
    or at the upper edge) and keeping the feasible one whose multipliers have the right sign: the unique minimiser, with no tolerance of an
    iterative method in it.  For the few rows of TorqueControlNLP (its g rows are exactly such bands: eps_i - d_i^2 >= 0 <=> |d_i| <= sqrt eps_i)."""
    import itertools

    m, n = A.shape
    best = None
    for state in itertools.product((0, -1, 1), repeat=m):
        act = [i for i in range(m) if state[i]]
        sg = np.array([state[i] for i in act], float)
        Aa = A[act]
        K = np.block([[H, Aa.T], [Aa, np.zeros((len(act), len(act)))]])
        rhs = np.concatenate([-grad0, b[act] + sg * half[act]])
        try:
            sol = np.linalg.solve(K, rhs)
        except np.linalg.LinAlgError:
            continue
        x, nu = sol[:n], sol[n:]  # H x + grad0 + Aa' nu = 0: at an upper edge nu >= 0, at a lower edge nu <= 0
        d = A @ x - b
        if np.all(np.abs(d) <= half * (1 + 1e-9)) and np.all(nu * sg >= -1e-12 * (1 + np.abs(nu))):
            fval = 0.5 * x @ H @ x + grad0 @ x
            if best is None or fval < best[1]:
                best = (x, fval, state, nu)
    if best is None:
        raise RuntimeError("no consistent active set")
    return best
