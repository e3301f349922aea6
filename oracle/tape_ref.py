"""ORACLE (test infrastructure, not product code) -- numpy interpreter of the instruction tapes optas_amd/tape.py builds (forward values,
reverse-mode gradients of chosen registers) and the numpy port of the generic augmented-Lagrangian / BFGS solver the HIP path runs on them
(optas_amd/csrc/oh_tape.hip).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it."""
import numpy as np

OP_CONST, OP_X, OP_P, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_SIN, OP_COS, OP_ATAN2, OP_SQRT, OP_SQR = range(13)
OP_ASIN, OP_FABS, OP_FMIN, OP_FMAX, OP_LT, OP_LE, OP_EQ, OP_NE, OP_NOT, OP_AND, OP_OR, OP_IFZ = range(13, 25)
OP_EXP, OP_LOG = 25, 26


def forward(tape, x, p):
    L = tape.op.shape[0]
    v = np.zeros(L)
    for i in range(L):
        o, a, b = tape.op[i], tape.a[i], tape.b[i]
        if o == OP_CONST:
            v[i] = tape.c[i]
        elif o == OP_X:
            v[i] = x[a]
        elif o == OP_P:
            v[i] = p[a]
        elif o == OP_ADD:
            v[i] = v[a] + v[b]
        elif o == OP_SUB:
            v[i] = v[a] - v[b]
        elif o == OP_MUL:
            v[i] = v[a] * v[b]
        elif o == OP_DIV:
            v[i] = v[a] / v[b]
        elif o == OP_NEG:
            v[i] = -v[a]
        elif o == OP_SIN:
            v[i] = np.sin(v[a])
        elif o == OP_COS:
            v[i] = np.cos(v[a])
        elif o == OP_ATAN2:
            v[i] = np.arctan2(v[a], v[b])
        elif o == OP_SQRT:
            v[i] = np.sqrt(v[a])
        elif o == OP_SQR:
            v[i] = v[a] * v[a]
        elif o == OP_ASIN:
            v[i] = np.arcsin(v[a])
        elif o == OP_FABS:
            v[i] = abs(v[a])
        elif o == OP_FMIN:
            v[i] = min(v[a], v[b])
        elif o == OP_FMAX:
            v[i] = max(v[a], v[b])
        elif o == OP_LT:
            v[i] = float(v[a] < v[b])
        elif o == OP_LE:
            v[i] = float(v[a] <= v[b])
        elif o == OP_EQ:
            v[i] = float(v[a] == v[b])
        elif o == OP_NE:
            v[i] = float(v[a] != v[b])
        elif o == OP_NOT:
            v[i] = float(v[a] == 0.0)
        elif o == OP_AND:
            v[i] = float(v[a] != 0.0 and v[b] != 0.0)
        elif o == OP_OR:
            v[i] = float(v[a] != 0.0 or v[b] != 0.0)
        elif o == OP_IFZ:
            v[i] = v[b] if v[a] != 0.0 else 0.0
        elif o == OP_EXP:
            v[i] = np.exp(v[a])
        elif o == OP_LOG:
            v[i] = np.log(v[a])
    return v


def reverse(tape, v, seeds):
    """Gradient wrt x of sum_r seeds[r] * register r  (seeds: dict register -> weight)."""
    L = tape.op.shape[0]
    adj = np.zeros(L)
    for r, w in seeds.items():
        adj[r] += w
    g = np.zeros(tape.nx)
    for i in range(L - 1, -1, -1):
        w = adj[i]
        if w == 0.0:
            continue
        o, a, b = tape.op[i], tape.a[i], tape.b[i]
        if o == OP_X:
            g[a] += w
        elif o == OP_ADD:
            adj[a] += w
            adj[b] += w
        elif o == OP_SUB:
            adj[a] += w
            adj[b] -= w
        elif o == OP_MUL:
            adj[a] += w * v[b]
            adj[b] += w * v[a]
        elif o == OP_DIV:
            adj[a] += w / v[b]
            adj[b] -= w * v[a] / (v[b] * v[b])
        elif o == OP_NEG:
            adj[a] -= w
        elif o == OP_SIN:
            adj[a] += w * np.cos(v[a])
        elif o == OP_COS:
            adj[a] -= w * np.sin(v[a])
        elif o == OP_ATAN2:
            d = v[a] * v[a] + v[b] * v[b]
            adj[a] += w * v[b] / d
            adj[b] -= w * v[a] / d
        elif o == OP_SQRT:
            adj[a] += w * 0.5 / v[i]
        elif o == OP_SQR:
            adj[a] += w * 2.0 * v[a]
        elif o == OP_ASIN:
            adj[a] += w / np.sqrt(1.0 - v[a] * v[a])
        elif o == OP_FABS:
            adj[a] += w * np.sign(v[a])
        elif o == OP_FMIN:  # casadi/core/calculus.hpp: d fmin = (x <= y, !(x <= y))
            if v[a] <= v[b]:
                adj[a] += w
            else:
                adj[b] += w
        elif o == OP_FMAX:  # d fmax = (x >= y, !(x >= y))
            if v[a] >= v[b]:
                adj[a] += w
            else:
                adj[b] += w
        elif o == OP_IFZ:
            if v[a] != 0.0:
                adj[b] += w
        elif o == OP_EXP:
            adj[a] += w * v[i]
        elif o == OP_LOG:
            adj[a] += w / v[a]
        # comparisons and logic: piecewise constant, no derivative
    return g


class _LBFGS:
    """The limited-memory inverse-Hessian operator of csrc/oh_tape_solver.h (T.lbfgs = m pairs): Nocedal's two-loop recursion."""

    def __init__(self, m, h0=None):
        self.m, self.S, self.Y, self.h0 = m, [], [], h0  # h0: initial metric [n, n] (oh_tape_set_metric), None: the identity scaled by the newest pair

    def reset(self):
        self.S, self.Y = [], []

    def is_identity(self):
        return not self.S

    def direction(self, grad):
        q = grad.copy()
        al = []
        for sv, yv in zip(reversed(self.S), reversed(self.Y)):
            a = (sv @ q) / (sv @ yv)
            al.append(a)
            q = q - a * yv
        if self.h0 is not None:
            q = self.h0 @ q
        elif self.S:
            q = q * ((self.S[-1] @ self.Y[-1]) / (self.Y[-1] @ self.Y[-1]))
        for (sv, yv), a in zip(zip(self.S, self.Y), reversed(al)):
            q = q + sv * (a - (yv @ q) / (sv @ yv))
        return -q

    def update(self, sv, yv):
        self.S, self.Y = (self.S + [sv])[-self.m:], (self.Y + [yv])[-self.m:]


KEEP_METRIC = True  # the quasi-Newton metric survives the multiplier / penalty updates of the outer loop (csrc/oh_tape_solver.h)
SCALE_FIRST = True  # dense form: the first pair scales the identity before updating it


def solve_tape_al(tape, x0, p, tol=1e-6, tol_feas=1e-9, max_iter=2000, rho0=10.0, lbfgs=None, trace=None, h0=None):
    """Generic NLP on a tape: min f s.t. rows[:n_ineq] >= 0, rows[n_ineq:] = 0.  Augmented Lagrangian (PHR for the inequality rows)
    minimised by BFGS with Armijo backtracking -- the dense inverse Hessian up to 48 variables, the limited-memory form with `lbfgs` = 12 pairs
    beyond, as oh_api.hip:tape_params chooses; one forward + one reverse sweep per evaluation.  Port of k_tape_solve.  h0: the initial metric of the
    limited-memory form (oh_tape_set_metric; the product passes tape.py:quadratic_cost_metric), ignored by the dense form."""
    n, ni, ne = tape.nx, tape.n_ineq, tape.n_eq
    lbfgs = (12 if n > 48 else 0) if lbfgs is None else lbfgs
    LB = _LBFGS(lbfgs, h0) if lbfgs > 0 else None
    metric = LB is not None and h0 is not None
    rows = tape.out_rows
    lam = np.zeros(ni)
    mu = np.zeros(ne)
    rho = rho0
    x = np.array(x0, dtype=float)

    def phi(xx):
        v = forward(tape, xx, p)
        g, c = v[rows[:ni]], v[rows[ni:]]
        s = np.maximum(0.0, lam - rho * g)
        val = v[tape.out_cost] + np.sum((s * s - lam * lam) / (2.0 * rho)) + np.sum(-mu * c + 0.5 * rho * c * c)
        seeds = {int(tape.out_cost): 1.0}
        for i in range(ni):
            if s[i] > 0.0:
                seeds[int(rows[i])] = seeds.get(int(rows[i]), 0.0) - s[i]
        for i in range(ne):
            seeds[int(rows[ni + i])] = seeds.get(int(rows[ni + i]), 0.0) + (-mu[i] + rho * c[i])
        return val, reverse(tape, v, seeds), g, c, v[tape.out_cost]

    evals = 1
    val, grad, g, c, fval = phi(x)
    H = np.eye(n) if LB is None else None
    fresh = True  # dense form: H is the identity (nothing measured yet)
    omega, meas_prev = max(tol, 1e-2), np.inf
    alpha_prev = 1.0
    status = 1
    while True:
        stat = np.abs(grad).max() if n else 0.0
        if not np.isfinite(val) or not np.isfinite(stat):
            status = 2
            break
        if stat <= omega:
            meas = max(np.abs(c).max() if ne else 0.0, np.abs(np.minimum(g, lam / rho)).max() if ni else 0.0)
            if stat <= tol and meas <= tol_feas:
                status = 0
                break
            if evals >= max_iter:
                break
            mu = mu - rho * c
            lam = np.maximum(0.0, lam - rho * g)
            if meas > 0.25 * meas_prev:
                rho = min(rho * 10.0, 1e8)
                if metric:  # the pairs measured the rows' curvature under the old penalty (csrc/oh_tape_solver.h)
                    LB.reset()
            meas_prev = meas
            omega = max(tol, min(omega, 0.1 * meas))
            val, grad, g, c, fval = phi(x)
            evals += 1
            # the metric is kept: the multiplier update shifts the merit, its curvature (cost + penalty of the rows in reach) stays what the
            # pairs have measured; rebuilding it from the identity at every outer update cost 4 of every 5 evaluations on the 7-variable IK
            if not KEEP_METRIC:  # the rule before round 3, for A/B runs
                H = np.eye(n) if LB is None else None
                fresh = True
                if LB is not None:
                    LB.reset()
            continue
        if evals >= max_iter:
            break
        d = -H @ grad if LB is None else LB.direction(grad)
        slope = float(grad @ d)
        if not slope < 0.0:
            if LB is None:
                H = np.eye(n)
            else:
                LB.reset()
            fresh = True
            d = -grad
            slope = float(grad @ d)
        # a fresh (identity) metric knows nothing about the scale of the problem: keep the first step within unit length
        if LB is not None:
            fresh = LB.is_identity()
        alpha = min(1.0, 1.0 / np.abs(d).max()) if fresh else 1.0
        if metric and 1e-4 <= alpha_prev < 1.0:  # the last accepted step was a fraction of the metric's unit step: this one is tried at four times that first
            alpha = min(alpha, 4.0 * alpha_prev)  # (csrc/oh_tape_solver.h; 32 planner instances: 54 -> 40 evaluations)
        ok = False
        for _ in range(40):
            xt = x + alpha * d
            vt, gt, g_t, c_t, f_t = phi(xt)
            evals += 1
            # what the merit resolves: its own rounding plus the rows' rounding (1e-16 of quantities of order one) times their multipliers --
            # under multipliers of 30 the term -mu c moves by 3e-15 between two evaluations of the same point
            slack = 4e-16 * (max(1.0, abs(val)) + float(np.sum(np.abs(mu)) + np.sum(lam)))
            need = -1e-4 * alpha * slope
            if need > slack:
                if np.isfinite(vt) and vt <= val - need + slack:
                    ok = True
                    break
            elif np.isfinite(vt) and vt <= val - slack:  # a decrease the merit does resolve, larger than the one asked for
                ok = True
                break
            elif np.isfinite(vt) and vt <= val + slack and float(gt @ gt) <= (1.0 - 1e-4 * alpha) * float(grad @ grad) and float(gt @ gt) < float(grad @ grad):
                # the decrease asked for is below that resolution (end game under a large penalty: a gradient of 4e-6 across a curvature of 1e4
                # is worth 7e-16 of merit): the value cannot judge the step, the gradient can -- a step that keeps the merit within its rounding
                # is taken if it shrinks the gradient; one that leaves the gradient where it was is not a step (alpha -> 0 used to pass as one)
                ok = True
                break
            if metric and np.isfinite(vt) and abs(vt) < 1e300 and vt - val - alpha * slope > 0.0:
                # minimiser of the parabola through phi(0), phi'(0), phi(alpha), kept inside [0.1, 0.5] alpha
                alpha = min(0.5 * alpha, max(0.1 * alpha, -slope * alpha * alpha / (2.0 * (vt - val - alpha * slope))))
            else:
                alpha *= 0.5
            if evals >= max_iter:
                break
        if not ok:
            if trace is not None:
                trace.append({"evals": evals, "val": val, "stat": float(stat), "alpha": 0.0, "slope": slope, "rho": rho, "omega": omega, "sy": 0.0, "fresh": bool(fresh)})
            evals += 1  # the kernel re-evaluates the accepted point (its tape registers were overwritten by the rejected trials)
            if fresh or evals >= max_iter:
                break  # steepest descent cannot improve: rounding floor
            if LB is None:
                H = np.eye(n)
            else:
                LB.reset()
            fresh = True
            continue
        alpha_prev = alpha
        sv, yv = xt - x, gt - grad
        sy = float(sv @ yv)
        if trace is not None:
            trace.append({"evals": evals, "val": vt, "stat": float(np.abs(gt).max()), "alpha": alpha, "slope": slope, "rho": rho, "omega": omega, "sy": sy,
                          "fresh": bool(fresh)})
        if LB is not None:
            if sy > 1e-12 * np.linalg.norm(sv) * np.linalg.norm(yv):
                LB.update(sv, yv)
        elif sy > 1e-12 * np.linalg.norm(sv) * np.linalg.norm(yv):
            if fresh and SCALE_FIRST:
                H = (sy / float(yv @ yv)) * np.eye(n)  # Nocedal & Wright (6.20): the first pair sets the scale before it updates the identity
            fresh = False
            Hy = H @ yv
            H = H + ((sy + float(yv @ Hy)) / (sy * sy)) * np.outer(sv, sv) - (np.outer(Hy, sv) + np.outer(sv, Hy)) / sy
        x, val, grad, g, c, fval = xt, vt, gt, g_t, c_t, f_t
    return {"x": x, "f": float(fval), "lam": lam, "mu": mu, "evals": evals, "status": status, "stat": float(np.abs(grad).max()),
            "feas": float(max(np.abs(c).max() if ne else 0.0, np.maximum(0.0, -g).max() if ni else 0.0))}
