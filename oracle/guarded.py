"""ORACLE (test infrastructure, not product code) -- numpy port of the state machine the HIP path runs for the
position-tracking family with inequality rows (SURVEY 8(a) B4, B5, H4 synthetic: example/dual_arm.py per arm +
enforce_model_limits, builder.py:471-509 + sphere_collision_avoidance_constraints, builder.py:366-417):

    min  sum_t w_p ||p(q_t) - path_t||^2 + (w_v/dt^2) sum_t ||q_{t+1} - q_t||^2,   q_0 = qc
    s.t. q_t - lo >= 0,  up - q_t >= 0                                     (limits, rows "_l" / "_r")
         ||c_l(q_t) - o_j||^2 - (r_l + r_j)^2 >= 0  for link l, obstacle j (spheres; builder.py:411-415)

Inequalities enter through the Powell-Hestenes-Rockafellar augmented Lagrangian
    psi(g, lam, rho) = (max(0, lam - rho g)^2 - lam^2) / (2 rho),
the inner problem is the same Levenberg-Marquardt / Riccati iteration as solve_free_lm (oracle/structured.py) applied
to L_A, and the outer iteration is lam <- max(0, lam - rho g).  Independent cross-check: scipy SLSQP on the literal
reference layout (oracle.problems.GuardedArmNLP) and kkt_reference_form.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .structured import LS_MAX, LS_SHRINK, FoldedChain, block_tridiag_solve

AL_OMEGA = 1.0  # OH_AL_OMEGA_FREE in csrc/oh_free.hip: inner tolerance of the outer loop relative to the complementarity measure


@dataclass
class Guards:
    lo: Optional[np.ndarray] = None  # (n,) joint limits or None
    up: Optional[np.ndarray] = None
    links: Optional[List[str]] = None  # sphere links (on the chain)
    link_radii: Optional[np.ndarray] = None  # (L,)
    obs_pos: Optional[np.ndarray] = None  # (O, 3)
    obs_radii: Optional[np.ndarray] = None  # (O,)

    def n_rows(self, n):
        nl = 2 * n if self.lo is not None else 0
        ns = len(self.links) * len(self.obs_radii) if self.links else 0
        return nl + ns


def guard_values(chain: FoldedChain, Q, G: Guards, weights=None):
    """g (T, NC) and dg (T, NC, n): rows ordered [q - lo (n); up - q (n); spheres link-major, obstacle-minor].
    With weights (T, NC) also returns sum_i weights_i * Hessian(g_i) per knot, (T, n, n)."""
    T, n = Q.shape
    vals, jac = [], []
    HW = np.zeros((T, n, n))
    nl = 0
    if G.lo is not None:
        vals += [Q - G.lo[None], G.up[None] - Q]
        eye = np.tile(np.eye(n)[None], (T, 1, 1))
        jac += [eye, -eye]
        nl = 2 * n
    if G.links:
        C, J = chain.link_positions(Q, G.links)  # (T, L, 3), (T, L, 3, n)
        d = C[:, :, None, :] - G.obs_pos[None, None]  # (T, L, O, 3)
        bnd = (G.link_radii[:, None] + G.obs_radii[None, :]) ** 2
        gs = np.sum(d * d, -1) - bnd[None]
        dgs = 2.0 * np.einsum("tlok,tlkn->tlon", d, J)
        vals.append(gs.reshape(T, -1))
        jac.append(dgs.reshape(T, -1, n))
        if weights is not None:
            # Hessian(g) = 2 J^T J + 2 sum_k d_k d2c_k,  d2c/dq_a dq_b = omega_a x J[:, b] (a <= b, both up to the link's joint)
            L, O = len(G.links), len(G.obs_radii)
            wts = weights[:, nl:].reshape(T, L, O)
            _, _, z, _ = chain.fk(Q)
            om = np.zeros((T, n, 3))
            for k in range(chain.n_chain):
                if chain.jtype[k] == 0:
                    om[:, chain.qidx[k]] = z[:, k]
            order = [chain.qidx[k] for k in range(chain.n_chain)]
            wl = wts.sum(2)  # (T, L)
            yd = np.einsum("tlo,tlok->tlk", wts, d)  # sum_o w d  (T, L, 3)
            HW += 2.0 * np.einsum("tl,tlkn,tlkm->tnm", wl, J, J)
            for li in range(L):
                for ia, a in enumerate(order):
                    for b in order[ia:]:
                        v = 2.0 * np.einsum("tk,tk->t", yd[:, li], np.cross(om[:, a], J[:, li, :, b]))
                        HW[:, a, b] += v
                        if a != b:
                            HW[:, b, a] += v
    if not vals:  # no position rows at all (velocity rows only)
        g_all, dg_all = np.zeros((T, 0)), np.zeros((T, 0, n))
    else:
        g_all, dg_all = np.concatenate(vals, 1), np.concatenate(jac, 1)
    if weights is not None:
        return g_all, dg_all, HW
    return g_all, dg_all


def solve_free_al(chain: FoldedChain, T, dt, offsets, qc, guards: Guards, Q0=None, w_path=1.0, w_vel=0.01, fix_dq0=False, max_iter=400,
                  tol=1e-6, tol_feas=1e-9, rho0=1e3, verbose=False, exact=True, vlimits=None, fuse=True):
    """vlimits = (vlo, vup): joint-velocity rows dq_t - vlo >= 0, vup - dq_t >= 0 on dq_t = (q_{t+1} - q_t) / dt, t = 0 .. T-2
    (enforce_model_limits(name, time_deriv=1), builder.py:471-509; round 3: k_couple_free_vel in csrc/oh_free.hip).  Same treatment as in
    oracle/structured.py: penalty rho * vscale, the value of interval (t-1, t) is booked on knot t, its gradient enters both knots, its
    Gauss-Newton weight rho_v / dt^2 joins 2 kappa on the diagonal of both knots and in the coupling block between them (which stays
    diagonal).  Adds "lam_v" (T-1, 2n) and "g_v"."""
    n = chain.ndof
    t0 = 2 if fix_dq0 else 1
    kap = w_vel / dt**2
    e0, _, _, _ = chain.fk(qc[None])
    path = e0[0] + offsets
    Qc = np.zeros((T, n)) if Q0 is None else Q0.copy()
    Qc[:t0] = qc
    F = slice(t0, T)
    NC = guards.n_rows(n)
    lam = np.zeros((T, NC))
    rho = rho_next = rho0
    omega = max(tol, 1e-2)
    meas_prev = np.inf
    vel = vlimits is not None
    if vel:
        vlo, vup = (np.asarray(v, dtype=float) for v in vlimits)
        lam_v = np.zeros((T - 1, 2 * n))
        vscale = dt**2 / 40.0  # as in oracle/structured.py (GuardParams.vscale): the rows' Gauss-Newton weight is rho / 40

    def vel_terms(Q, lam_v, rho):
        v = (Q[1:] - Q[:-1]) / dt
        gv = np.concatenate([v - vlo[None], vup[None] - v], 1)
        rv = rho * vscale
        sv = np.maximum(0.0, lam_v - rv * gv)
        fixed = slice(0, t0 - 1)  # intervals between fixed knots carry no row the solver can move
        sv[fixed] = 0.0
        psi = (sv * sv - lam_v * lam_v) / (2.0 * rv)
        psi[fixed] = 0.0
        sig = (sv[:, n:] - sv[:, :n]) / dt
        wv = rv * ((sv[:, :n] > 0.0).astype(float) + (sv[:, n:] > 0.0).astype(float)) / dt**2
        meas = np.abs(np.minimum(gv, lam_v / rv))
        meas[fixed] = 0.0
        return gv, psi.sum(1), sig, wv, float(meas.max()) if meas.size else 0.0

    def evalp(Q, lam, rho):
        e, Re, Jp, Jw = chain.jac(Q)
        r = path - e
        phi = w_path * np.sum(r * r, 1)
        g = -2.0 * w_path * np.einsum("tki,tk->ti", Jp, r)
        W = 2.0 * w_path * np.einsum("tki,tkj->tij", Jp, Jp)
        gv, dg = guard_values(chain, Q, guards)
        s = np.maximum(0.0, lam - rho * gv)
        s[:t0] = 0.0
        if exact:
            W = W - guard_values(chain, Q, guards, weights=s)[2]
        psi = (s * s - lam * lam) / (2.0 * rho)
        psi[:t0] = 0.0
        phi = phi + psi.sum(1)
        g = g - np.einsum("tc,tcn->tn", s, dg)
        W = W + rho * np.einsum("tc,tcn,tcm->tnm", (s > 0.0).astype(float), dg, dg)
        meas = np.abs(np.minimum(gv, lam / rho))
        meas[:t0] = 0.0
        d = Q[1:] - Q[:-1]
        f = float(np.sum(phi) + kap * np.sum(d * d))
        Gs = np.zeros_like(Q)
        Gs[1:] += 2 * kap * d
        Gs[:-1] -= 2 * kap * d
        mx = float(meas.max()) if meas.size else 0.0
        wv = np.zeros((T - 1, n))
        if vel:
            _, psi_v, sig, wv, meas_v = vel_terms(Q, lam_v, rho)
            f += float(psi_v.sum())
            Gs[1:] += sig
            Gs[:-1] -= sig
            mx = max(mx, meas_v)
        return f, g + Gs, W, gv, mx, wv

    mu, nun = 0.0, 2.0
    iters = rejected = outers = 0
    first, outer = True, False
    ls_count, ls_scale, z_last, gd_last, q_last = 0, 1.0, None, 0.0, 0.0
    Qt = Qc
    cur = None
    status = 1
    pred = 0.0
    while True:
        if outer:  # multiplier update at the current point with the old penalty, then evaluate with the new one
            gv_now, _ = guard_values(chain, Qt, guards)
            lam = np.maximum(0.0, lam - rho * gv_now)
            lam[:t0] = 0.0
            if vel:
                lam_v = np.maximum(0.0, lam_v - rho * vscale * vel_terms(Qt, lam_v, rho)[0])
                lam_v[: max(t0 - 1, 0)] = 0.0
            rho = rho_next
            outers += 1
        f_t, G, W, gv, meas_t, wv_t = evalp(Qt, lam, rho)
        if first or outer:
            accept, first, outer = True, False, False
        else:
            ratio = (cur["f"] - f_t) / max(pred, 1e-300)
            accept = np.isfinite(f_t) and (ratio > 1e-4 or (pred <= 1e-15 * abs(cur["f"]) and f_t <= cur["f"] + 1e-14 * abs(cur["f"])))
            if accept:
                mu *= max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3)
                mu = 0.0 if mu < 1e-7 else mu
                nun = 2.0
            elif ls_count < LS_MAX and z_last is not None and iters < max_iter // 2:
                # line search along the rejected step before the damping is touched (free_accept in csrc/oh_free.hip): what rejects a step
                # of these problems is a row that was inactive at the accepted point and is violated at the trial -- the model cannot
                # know it, damping the whole step to 1e3 and easing it back costs a dozen steps, a shorter step along the same direction one
                ls_count += 1
                ls_scale *= LS_SHRINK
                rejected += 1
                pred = -gd_last * ls_scale + 0.5 * ls_scale * ls_scale * q_last
                Qt = cur["Q"].copy()
                Qt[F] += z_last * ls_scale
                if iters >= max_iter:
                    break
                iters += 1
                continue
            else:
                mu = max(mu * nun, 1e-3)
                nun *= 2.0
                rejected += 1
        if accept:
            ls_count, ls_scale = 0, 1.0
            ndiag = np.full(T, 2.0)
            ndiag[T - 1] = 1.0
            Dfull = W + (2 * kap * ndiag)[:, None, None] * np.eye(n)[None]
            wsum = np.zeros((T, n))
            wsum[1:] += wv_t
            wsum[:-1] += wv_t
            Dfull = Dfull + np.einsum("tj,jk->tjk", wsum, np.eye(n))
            cur = {"Q": Qt, "f": f_t, "G": G[F], "D": Dfull[F], "meas": meas_t, "Er": -np.einsum("tj,jk->tjk", 2 * kap + wv_t[t0:], np.eye(n))}
        stat = float(np.max(np.abs(cur["G"])))
        Er = cur["Er"]
        while True:
            z, ok = block_tridiag_solve(cur["D"], Er, -cur["G"], mu)
            if ok:
                break
            mu = max(4.0 * mu, 1e-2)
        if verbose:
            print(f"  steps {iters:3d} f={cur['f']:.12f} stat={stat:.3e} meas={cur['meas']:.3e} mu={mu:.3g} rho={rho:.1e} omega={omega:.1e} outers={outers}")
        if stat <= omega:
            meas = cur["meas"]
            if stat <= tol and meas <= tol_feas:
                status = 0
                break
            if iters >= max_iter:
                break
            # outer update: refresh multipliers at the next evaluation, tighten the inner tolerance (fuse: the pending step is taken along)
            rho_next = min(rho * 10.0, 1e8) if meas > 0.25 * meas_prev else rho
            meas_prev = meas
            omega = max(tol, min(omega, AL_OMEGA * meas))
            outer = True
            Qt = cur["Q"]
            if fuse:
                # round 5 (al_fuse, csrc/oh_free.hip:step_instance_free): the launch that decides on the update also takes the step the sweep has just
                # solved for; the evaluation that refreshes the multipliers looks at that point and accepts it as it is
                Qt = cur["Q"].copy()
                Qt[F] += z
            iters += 1
            continue
        if iters >= max_iter:
            break
        gd_last, z2_ = float(np.sum(cur["G"] * z)), float(np.sum(z * z))
        pred = -0.5 * gd_last + 0.5 * mu * z2_
        z_last, q_last, ls_scale = z.copy(), gd_last + mu * z2_, 1.0  # pred(s) = -s gd + s^2 q / 2 for the step s z
        Qt = cur["Q"].copy()
        Qt[F] += z
        iters += 1
    gv, _ = guard_values(chain, cur["Q"], guards)
    lam_out = np.maximum(0.0, lam - rho * gv)
    lam_out[:t0] = 0.0
    e, _, _, _ = chain.fk(cur["Q"])
    f_true = float(w_path * np.sum((path - e) ** 2) + kap * np.sum(np.diff(cur["Q"], axis=0) ** 2))
    out = {"Q": cur["Q"], "f": f_true, "iters": iters, "rejected": rejected, "outers": outers, "stat": stat, "meas": cur["meas"],
           "status": status, "lam": lam_out, "g": gv}
    if vel:
        gvv = vel_terms(cur["Q"], lam_v, rho)[0]
        lv = np.maximum(0.0, lam_v - rho * vscale * gvv)
        lv[: max(t0 - 1, 0)] = 0.0
        out.update(lam_v=lv, g_v=gvv)
    return out
