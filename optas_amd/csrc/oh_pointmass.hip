// OH_PROBLEM_POINT_MASS_MPC: example/point_mass_mpc.py Controller (:88-154), BASELINE config 3.
// One lane per MPC instance; the whole primal-dual interior-point loop runs inside one launch, Newton steps
// by a Riccati sweep over the T stages (state (y,v) in R^4, control a in R^2).  Per-instance work arrays live
// in HBM scratch laid out [row][b] (instance index fastest -> coalesced; ~6 KB per instance, so a 4096-batch
// stays resident in L2 / Infinity Cache).  Algorithm and constants mirror oracle/pointmass_ipm.py line by line.
#include <cstdlib>

#include "oh_device.h"
#include "oh_kernels.h"

#define PIDX(row) ((size_t)(row) * Bp + b)

struct PM4 {  // 4x4 helpers on row-major double[16]
  double m[16];
};

// c (9 rows) and the nonzero Jacobian entries of stage t: box rows are +-e_j, obstacle row is 2 (y - o)
OH_DEV void pm_cons(const PmParams& P, const double* x, const double ox, const double oy, double (&c)[9], double& jx, double& jy) {
  c[0] = x[0] + P.ylim; c[1] = P.ylim - x[0];
  c[2] = x[1] + P.ylim; c[3] = P.ylim - x[1];
  c[4] = x[2] + P.vlim; c[5] = P.vlim - x[2];
  c[6] = x[3] + P.vlim; c[7] = P.vlim - x[3];
  const double dx = x[0] - ox, dy = x[1] - oy;
  c[8] = dx * dx + dy * dy - P.safe_sq;
  jx = 2.0 * dx; jy = 2.0 * dy;
}
// y = J^T w for the 9 rows (J rows: +e0,-e0,+e1,-e1,+e2,-e2,+e3,-e3,(jx,jy,0,0))
OH_DEV void pm_JTw(const double (&w)[9], const double jx, const double jy, double (&y)[4]) {
  y[0] = w[0] - w[1] + jx * w[8];
  y[1] = w[2] - w[3] + jy * w[8];
  y[2] = w[4] - w[5];
  y[3] = w[6] - w[7];
}
// d = J v
OH_DEV void pm_Jv(const double* v, const double jx, const double jy, double (&d)[9]) {
  d[0] = v[0]; d[1] = -v[0]; d[2] = v[1]; d[3] = -v[1];
  d[4] = v[2]; d[5] = -v[2]; d[6] = v[3]; d[7] = -v[3];
  d[8] = jx * v[0] + jy * v[1];
}

__global__ __launch_bounds__(64) void k_pm_solve(PmParams P, PmBuffers D, const double* __restrict__ x0, const double* __restrict__ pin,
                                                 double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt,
                                                 int* __restrict__ iters_o, int* __restrict__ status_o) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int T = P.T;
  const double dt = P.dt, w = P.w_acc;
  const size_t np_ = 4 + 4 * (size_t)T, nx = 4 * (size_t)T;
  const double* pb = pin + (size_t)b * np_;
  const double* goal = pb + 4;           // [t][2]
  const double* obs = pb + 4 + 2 * T;    // [t][2]
  // work arrays (rows): a[2(T-1)], X[4T], s[9T], lam[9T], K[8(T-1)], kk[2(T-1)], dX[4T], da[2(T-1)]
  double* A_ = D.a; double* X_ = D.X; double* S_ = D.s; double* L_ = D.lam;
  double* K_ = D.K; double* k_ = D.kk; double* dX_ = D.dX; double* dA_ = D.da;

  // ---- seed: velocities from x0's dY block, v_0 = dcurr, controls = velocity differences, states by roll-out
  {
    double vprev[2] = {pb[2], pb[3]};
    double x[4] = {pb[0], pb[1], pb[2], pb[3]};
    for (int j = 0; j < 4; ++j) X_[PIDX(j)] = x[j];
    for (int t = 0; t < T - 1; ++t) {
      double v1[2] = {x0[(size_t)b * nx + 2 * T + 2 * (t + 1)], x0[(size_t)b * nx + 2 * T + 2 * (t + 1) + 1]};
      if (P.fix_vf && t == T - 2) v1[0] = v1[1] = 0.0;  // terminal row dy_{T-1} = 0 (point_mass_planner.py:34-35)
      const double a0 = (v1[0] - vprev[0]) / dt, a1 = (v1[1] - vprev[1]) / dt;
      A_[PIDX(2 * t)] = a0; A_[PIDX(2 * t + 1)] = a1;
      x[0] += dt * x[2]; x[1] += dt * x[3]; x[2] += dt * a0; x[3] += dt * a1;
      for (int j = 0; j < 4; ++j) X_[PIDX(4 * (t + 1) + j)] = x[j];
      vprev[0] = v1[0]; vprev[1] = v1[1];
    }
  }
  double mu = 0.1;
  for (int t = 0; t < T; ++t) {
    double x[4], c[9], jx, jy;
    for (int j = 0; j < 4; ++j) x[j] = X_[PIDX(4 * t + j)];
    pm_cons(P, x, obs[2 * t], obs[2 * t + 1], c, jx, jy);
    for (int i = 0; i < 9; ++i) {
      const double s = fmax(c[i], 1e-2);
      S_[PIDX(9 * t + i)] = s;
      L_[PIDX(9 * t + i)] = mu / s;
    }
  }

  struct Stage {
    double x[4], s[9], l[9], a[2], g[2], o[2];
  };
  auto load_stage = [&](const int t, const bool with_goal) {
    Stage r;
    for (int j = 0; j < 4; ++j) r.x[j] = X_[PIDX(4 * t + j)];
    for (int i = 0; i < 9; ++i) { r.s[i] = S_[PIDX(9 * t + i)]; r.l[i] = L_[PIDX(9 * t + i)]; }
    r.a[0] = r.a[1] = 0.0;
    if (t < T - 1) { r.a[0] = A_[PIDX(2 * t)]; r.a[1] = A_[PIDX(2 * t + 1)]; }
    r.g[0] = r.g[1] = 0.0;
    if (with_goal) { r.g[0] = goal[2 * t]; r.g[1] = goal[2 * t + 1]; }
    r.o[0] = obs[2 * t]; r.o[1] = obs[2 * t + 1];
    return r;
  };
  int status = OH_STATUS_MAX_ITER, it = 0;
  double stat = 0.0, feas = 0.0, compl_ = 0.0, fval = 0.0;
  for (it = 0; it <= P.max_iter; ++it) {
    // ---- backward pass: residuals (adjoint) and Riccati recursion -----------------------------------------------
    double Pm[16], pv[4], padj[4];
    stat = 0.0; feas = 0.0; compl_ = 0.0; fval = 0.0;
    // every sweep below walks the knots in a dependent chain while what it reads depends on the knot alone: the next knot's values are
    // requested before this one's arithmetic (otherwise each of the 3 x T knot steps of an iteration is a memory round trip: the whole
    // solve was 7.2 ms for any batch up to ~16 k instances)
    Stage nx_ = load_stage(T - 1, true);
    for (int t = T - 1; t >= 0; --t) {
      const Stage st = nx_;
      if (t > 0) nx_ = load_stage(t - 1, true);
      double x[4], c[9], jx, jy, s[9], lam[9];
      for (int j = 0; j < 4; ++j) x[j] = st.x[j];
      pm_cons(P, x, st.o[0], st.o[1], c, jx, jy);
      for (int i = 0; i < 9; ++i) { s[i] = st.s[i]; lam[i] = st.l[i]; }
      const double wt = (P.final_only && t < T - 1) ? 0.0 : 1.0;  // tracking on every knot (MPC) or on the last one only (planner)
      const double gx[4] = {-2.0 * wt * (st.g[0] - x[0]), -2.0 * wt * (st.g[1] - x[1]), 2.0 * P.w_vel * x[2], 2.0 * P.w_vel * x[3]};
      fval += wt * ((st.g[0] - x[0]) * (st.g[0] - x[0]) + (st.g[1] - x[1]) * (st.g[1] - x[1])) +
              P.w_vel * (x[2] * x[2] + x[3] * x[3]);
      double rc[9], sig[9], wq[9], jl[4], jq[4];
      for (int i = 0; i < 9; ++i) {
        rc[i] = c[i] - s[i];
        sig[i] = lam[i] / s[i];
        wq[i] = mu / s[i] - sig[i] * rc[i];
        if (t >= 1) { feas = fmax(feas, fabs(rc[i])); compl_ = fmax(compl_, lam[i] * s[i]); }
      }
      pm_JTw(lam, jx, jy, jl);
      pm_JTw(wq, jx, jy, jq);
      double lx[4], q[4];
      for (int j = 0; j < 4; ++j) { lx[j] = gx[j] - jl[j]; q[j] = gx[j] - jq[j]; }
      // Q_t = diag(2,2,0,0) + J^T Sig J  (stage 0 is fixed: its Q, q are never used)
      double Q[16];
      for (int j = 0; j < 16; ++j) Q[j] = 0.0;
      Q[0] = 2.0 * wt + sig[0] + sig[1] + sig[8] * jx * jx;
      Q[5] = 2.0 * wt + sig[2] + sig[3] + sig[8] * jy * jy;
      Q[1] = Q[4] = sig[8] * jx * jy;
      Q[10] = 2.0 * P.w_vel + sig[4] + sig[5];
      Q[15] = 2.0 * P.w_vel + sig[6] + sig[7];
      if (t == T - 1) {
        for (int j = 0; j < 16; ++j) Pm[j] = Q[j];
        for (int j = 0; j < 4; ++j) { pv[j] = q[j]; padj[j] = lx[j]; }
        continue;
      }
      // here Pm, pv, padj belong to stage t+1; controls a_t act between t and t+1
      const double a0 = st.a[0], a1 = st.a[1];
      fval += w * (a0 * a0 + a1 * a1);
      // control gradient of the Lagrangian: 2 w a + B^T padj, B^T z = dt (z2, z3)
      const bool pinned = P.fix_vf && t == T - 2;  // a_{T-2} = -v_{T-2} / dt is a function of the state, not a free control
      const double gu0 = 2.0 * w * a0 + dt * padj[2], gu1 = 2.0 * w * a1 + dt * padj[3];
      if (!pinned) stat = fmax(stat, fmax(fabs(gu0), fabs(gu1)));
      // Quu = R + B^T P B = 2w I + dt^2 P[2:4,2:4] ; Qux = B^T P A = dt (P[2:4,:] A) ; A = [[I, dt I],[0, I]]
      double PA[16];  // P A : column j<2 same as P, column j>=2: P[:,j] + dt P[:,j-2]
      for (int r = 0; r < 4; ++r) {
        PA[4 * r + 0] = Pm[4 * r + 0]; PA[4 * r + 1] = Pm[4 * r + 1];
        PA[4 * r + 2] = Pm[4 * r + 2] + dt * Pm[4 * r + 0];
        PA[4 * r + 3] = Pm[4 * r + 3] + dt * Pm[4 * r + 1];
      }
      const double q00 = 2.0 * w + dt * dt * Pm[10], q01 = dt * dt * Pm[11], q11 = 2.0 * w + dt * dt * Pm[15];
      double Qux[8];
      for (int j = 0; j < 4; ++j) { Qux[j] = dt * PA[8 + j]; Qux[4 + j] = dt * PA[12 + j]; }
      const double qu0 = 2.0 * w * a0 + dt * pv[2], qu1 = 2.0 * w * a1 + dt * pv[3];
      // 2x2 Cholesky solve
      // (reciprocal square roots and products: the 21 divisions of the textbook substitution were half the instructions of a knot step)
      const double i00 = rsqrt(q00), l10 = q01 * i00, i11 = rsqrt(q11 - l10 * l10);
      double Kt[8], kt[2];
      if (pinned) {
        // fixed feedback K = [0 0 -1/dt 0; 0 0 0 -1/dt], k = 0:  P = Q + A^T P A + Qux^T K + K^T Qux + K^T Quu K,  p = q + A^T p + K^T qu,
        // and the control gradient flows into the state adjoint: padj = lx + A^T padj + K^T gu
        for (int j = 0; j < 8; ++j) Kt[j] = 0.0;
        Kt[2] = Kt[4 + 3] = -1.0 / dt;
        kt[0] = kt[1] = 0.0;
        for (int j = 0; j < 8; ++j) K_[PIDX(8 * t + j)] = Kt[j];
        k_[PIDX(2 * t)] = 0.0; k_[PIDX(2 * t + 1)] = 0.0;
        const double kf = -1.0 / dt;
        double Pn[16], pn[4], pa[4];
        for (int r = 0; r < 4; ++r)
          for (int cc = 0; cc < 4; ++cc) {
            double v = PA[4 * r + cc];
            if (r >= 2) v += dt * PA[4 * (r - 2) + cc];
            v += Qux[r] * Kt[cc] + Qux[4 + r] * Kt[4 + cc];          // Qux^T K
            v += Kt[r] * Qux[cc] + Kt[4 + r] * Qux[4 + cc];          // K^T Qux
            Pn[4 * r + cc] = Q[4 * r + cc] + v;
          }
        Pn[10] += kf * kf * q00; Pn[11] += kf * kf * q01; Pn[14] += kf * kf * q01; Pn[15] += kf * kf * q11;  // K^T Quu K
        for (int r = 0; r < 4; ++r) {
          double v = pv[r], va = padj[r];
          if (r >= 2) { v += dt * pv[r - 2]; va += dt * padj[r - 2]; }
          pn[r] = q[r] + v;
          pa[r] = lx[r] + va;
        }
        pn[2] += kf * qu0; pn[3] += kf * qu1;
        pa[2] += kf * gu0; pa[3] += kf * gu1;
        for (int r = 0; r < 4; ++r)
          for (int cc = 0; cc < 4; ++cc) Pm[4 * r + cc] = 0.5 * (Pn[4 * r + cc] + Pn[4 * cc + r]);
        for (int j = 0; j < 4; ++j) { pv[j] = pn[j]; padj[j] = pa[j]; }
        continue;
      }
      {
        for (int j = 0; j < 4; ++j) {
          const double y0 = Qux[j] * i00, y1 = (Qux[4 + j] - l10 * y0) * i11;
          const double z1 = y1 * i11, z0 = (y0 - l10 * z1) * i00;
          Kt[j] = -z0; Kt[4 + j] = -z1;
        }
        const double y0 = qu0 * i00, y1 = (qu1 - l10 * y0) * i11;
        const double z1 = y1 * i11, z0 = (y0 - l10 * z1) * i00;
        kt[0] = -z0; kt[1] = -z1;
      }
      for (int j = 0; j < 8; ++j) K_[PIDX(8 * t + j)] = Kt[j];
      k_[PIDX(2 * t)] = kt[0]; k_[PIDX(2 * t + 1)] = kt[1];
      // P_t = Q_t + A^T P A + Qux^T K ;  p_t = q_t + A^T p + Qux^T k ;  (A^T M)[r] = M[r] (+ dt M[r-2] for r >= 2)
      double Pn[16], pn[4], pa[4];
      for (int r = 0; r < 4; ++r)
        for (int cc = 0; cc < 4; ++cc) {
          double v = PA[4 * r + cc];
          if (r >= 2) v += dt * PA[4 * (r - 2) + cc];
          v += Qux[r] * Kt[cc] + Qux[4 + r] * Kt[4 + cc];
          Pn[4 * r + cc] = Q[4 * r + cc] + v;
        }
      for (int r = 0; r < 4; ++r) {
        double v = pv[r], va = padj[r];
        if (r >= 2) { v += dt * pv[r - 2]; va += dt * padj[r - 2]; }
        pn[r] = q[r] + v + Qux[r] * kt[0] + Qux[4 + r] * kt[1];
        pa[r] = lx[r] + va;
      }
      for (int r = 0; r < 4; ++r)
        for (int cc = 0; cc < 4; ++cc) Pm[4 * r + cc] = 0.5 * (Pn[4 * r + cc] + Pn[4 * cc + r]);
      for (int j = 0; j < 4; ++j) { pv[j] = pn[j]; padj[j] = pa[j]; }
    }
    if (!(stat == stat) || !(fval == fval) || !(fabs(fval) < 1e300) || !(feas == feas) || !(compl_ == compl_)) { status = OH_STATUS_NUMERICAL; break; }
    if (stat <= P.tol && feas <= P.tol && compl_ <= P.tol) { status = OH_STATUS_CONVERGED; break; }
    // No feasible plan (round 6; the reference: IPOPT's Infeasible_Problem_Detected -> did_solve() False, solver.py:407-412): the obstacle is a parameter of every
    // knot, so a problem whose pinned knot is fine can still have none -- the obstacle lands where the mass cannot leave in time.  The infeasible-start
    // iteration then shows its textbook signature: the slack residual stalls at the violation it cannot remove while the multipliers of those rows leave every
    // bound (lam s from 1e-1 to 1e8 in three steps, 1e240 a dozen steps later; oracle/pointmass_ipm.py has the same rule).  A healthy run keeps lam s at the
    // size of the barrier parameter (<= 1).
    if (compl_ > 1e6 && feas > 1e3 * P.tol) { status = OH_STATUS_INFEASIBLE; break; }
    if (it == P.max_iter) break;

    // ---- forward pass 1: Newton direction, fraction-to-the-boundary step lengths -------------------------------------------------
    double ap = 1.0, ad = 1.0;
    {
      double dx[4] = {0, 0, 0, 0};
      for (int j = 0; j < 4; ++j) dX_[PIDX(j)] = 0.0;
      struct Gain {
        double K[8], k[2];
      };
      auto load_gain = [&](const int t) {
        Gain r;
        for (int j = 0; j < 8; ++j) r.K[j] = t < T - 1 ? K_[PIDX(8 * t + j)] : 0.0;
        r.k[0] = t < T - 1 ? k_[PIDX(2 * t)] : 0.0;
        r.k[1] = t < T - 1 ? k_[PIDX(2 * t + 1)] : 0.0;
        return r;
      };
      Stage nx1 = load_stage(0, false);
      Gain ng1 = load_gain(0);
      for (int t = 0; t < T; ++t) {
        const Stage st = nx1;
        const Gain gn = ng1;
        if (t + 1 < T) { nx1 = load_stage(t + 1, false); ng1 = load_gain(t + 1); }
        if (t >= 1) {
          double x[4], c[9], jx, jy, d[9];
          for (int j = 0; j < 4; ++j) x[j] = st.x[j];
          pm_cons(P, x, st.o[0], st.o[1], c, jx, jy);
          pm_Jv(dx, jx, jy, d);
          for (int i = 0; i < 9; ++i) {
            const double s = st.s[i], lam = st.l[i];
            const double ds = d[i] + (c[i] - s);
            const double dl = (mu / s - lam) - (lam / s) * ds;
            if (ds < 0.0) ap = fmin(ap, -0.995 * s / ds);
            if (dl < 0.0) ad = fmin(ad, -0.995 * lam / dl);
          }
        }
        if (t < T - 1) {
          double da0 = gn.k[0], da1 = gn.k[1];
          for (int j = 0; j < 4; ++j) { da0 += gn.K[j] * dx[j]; da1 += gn.K[4 + j] * dx[j]; }
          dA_[PIDX(2 * t)] = da0; dA_[PIDX(2 * t + 1)] = da1;
          const double n0 = dx[0] + dt * dx[2], n1 = dx[1] + dt * dx[3], n2 = dx[2] + dt * da0, n3 = dx[3] + dt * da1;
          dx[0] = n0; dx[1] = n1; dx[2] = n2; dx[3] = n3;
          for (int j = 0; j < 4; ++j) dX_[PIDX(4 * (t + 1) + j)] = dx[j];
        }
      }
    }
    // ---- forward pass 2: take the step (slacks / multipliers with the old Jacobians, states by exact roll-out) ------------------------
    double gap = 0.0;
    {
      double xn[4] = {pb[0], pb[1], pb[2], pb[3]};
      struct Delta {
        double dx[4], da[2];
      };
      auto load_delta = [&](const int t) {
        Delta r;
        for (int j = 0; j < 4; ++j) r.dx[j] = dX_[PIDX(4 * t + j)];
        r.da[0] = t < T - 1 ? dA_[PIDX(2 * t)] : 0.0;
        r.da[1] = t < T - 1 ? dA_[PIDX(2 * t + 1)] : 0.0;
        return r;
      };
      Stage nx2 = load_stage(0, false);
      Delta nd2 = load_delta(0);
      for (int t = 0; t < T; ++t) {
        const Stage st = nx2;
        const Delta dl_ = nd2;
        if (t + 1 < T) { nx2 = load_stage(t + 1, false); nd2 = load_delta(t + 1); }
        if (t >= 1) {
          double x[4], dx[4], c[9], jx, jy, d[9];
          for (int j = 0; j < 4; ++j) { x[j] = st.x[j]; dx[j] = dl_.dx[j]; }
          pm_cons(P, x, st.o[0], st.o[1], c, jx, jy);
          pm_Jv(dx, jx, jy, d);
          for (int i = 0; i < 9; ++i) {
            double s = st.s[i], lam = st.l[i];
            const double ds = d[i] + (c[i] - s);
            const double dl = (mu / s - lam) - (lam / s) * ds;
            s += ap * ds; lam += ad * dl;
            S_[PIDX(9 * t + i)] = s; L_[PIDX(9 * t + i)] = lam;
            gap += s * lam;
          }
        }
        for (int j = 0; j < 4; ++j) X_[PIDX(4 * t + j)] = xn[j];
        if (t < T - 1) {
          const double a0 = st.a[0] + ap * dl_.da[0], a1 = st.a[1] + ap * dl_.da[1];
          A_[PIDX(2 * t)] = a0; A_[PIDX(2 * t + 1)] = a1;
          const double n0 = xn[0] + dt * xn[2], n1 = xn[1] + dt * xn[3], n2 = xn[2] + dt * a0, n3 = xn[3] + dt * a1;
          xn[0] = n0; xn[1] = n1; xn[2] = n2; xn[3] = n3;
        }
      }
    }
    gap /= (double)(9 * (T - 1));
    const double am = fmin(ap, ad);
    const double sigma = (am > 0.9) ? 0.1 : ((am > 0.5) ? 0.3 : 0.8);
    mu = fmax(sigma * gap, 1e-2 * P.tol);
  }
  // ---- results in the reference layout x = [vec(Y 2xT); vec(dY 2xT)] --------------------------------------------------------------------
  if (xo) {
    double* xb = xo + (size_t)b * nx;
    for (int t = 0; t < T; ++t) {
      xb[2 * t] = X_[PIDX(4 * t)]; xb[2 * t + 1] = X_[PIDX(4 * t + 1)];
      xb[2 * T + 2 * t] = X_[PIDX(4 * t + 2)]; xb[2 * T + 2 * t + 1] = X_[PIDX(4 * t + 3)];
    }
  }
  if (fo) fo[b] = fval;
  if (kkt) { kkt[3 * (size_t)b] = stat; kkt[3 * (size_t)b + 1] = feas; kkt[3 * (size_t)b + 2] = compl_; }
  if (iters_o) iters_o[b] = it > P.max_iter ? P.max_iter : it;
  if (status_o) status_o[b] = status;
}

// ---- the same solve, ONE WAVEFRONT PER INSTANCE, one lane per knot (T <= 64) ---------------------------------------------------------------
// One lane per instance walks 3 x T knots per iteration with ~1000 dependent instructions each: 134 us per iteration whatever the batch, and
// BASELINE's 4096 plants are 64 wavefronts on 1024 SIMDs.  Here everything that belongs to a knot (constraint values, slacks, multipliers,
// the stage's Q, q and Lagrangian gradient, step lengths, the slack / multiplier update) is computed by the knot's lane from registers -- no
// work arrays at all -- and only the three recursions stay serial: the Riccati sweep, the Newton direction and the roll-out, executed by all
// lanes alike on values broadcast from the knot's lane (v_readlane), each lane keeping what belongs to its knot (gains, dx, da, x).  The
// recursions perform the thread kernel's operations in the thread kernel's order, and so does the sum behind the barrier parameter; only the
// reported objective is summed across lanes.  One iteration: ~20 us.
OH_DEV double pm_bcast(const double v, const int lane) {  // lane must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
OH_DEV double pm_wave_max(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}
OH_DEV double pm_wave_min(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
  return v;
}
OH_DEV double pm_wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__global__ __launch_bounds__(64) void k_pm_solve_wave(PmParams P, int B, const double* __restrict__ x0, const double* __restrict__ pin, double* __restrict__ xo,
                                                      double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters_o, int* __restrict__ status_o) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int T = P.T;
  const bool on = lane < T;
  const double dt = P.dt, w = P.w_acc;
  const size_t np_ = 4 + 4 * (size_t)T, nx = 4 * (size_t)T;
  const double* pb = pin + (size_t)b * np_;
  const int tk = on ? lane : T - 1;
  const double g0 = pb[4 + 2 * tk], g1 = pb[4 + 2 * tk + 1];                  // goal of this knot
  const double o0 = pb[4 + 2 * T + 2 * tk], o1 = pb[4 + 2 * T + 2 * tk + 1];  // obstacle centre at this knot
  double X[4] = {0, 0, 0, 0}, a[2] = {0, 0};
  // ---- seed (k_pm_solve): velocities from x0's dY block, controls = velocity differences, states by roll-out
  {
    double vprev[2] = {pb[2], pb[3]};
    double x[4] = {pb[0], pb[1], pb[2], pb[3]};
    if (lane == 0)
      for (int j = 0; j < 4; ++j) X[j] = x[j];
    for (int t = 0; t < T - 1; ++t) {
      double v1[2] = {x0[(size_t)b * nx + 2 * T + 2 * (t + 1)], x0[(size_t)b * nx + 2 * T + 2 * (t + 1) + 1]};
      if (P.fix_vf && t == T - 2) v1[0] = v1[1] = 0.0;
      const double a0 = (v1[0] - vprev[0]) / dt, a1 = (v1[1] - vprev[1]) / dt;
      if (lane == t) { a[0] = a0; a[1] = a1; }
      x[0] += dt * x[2]; x[1] += dt * x[3]; x[2] += dt * a0; x[3] += dt * a1;
      if (lane == t + 1)
        for (int j = 0; j < 4; ++j) X[j] = x[j];
      vprev[0] = v1[0]; vprev[1] = v1[1];
    }
  }
  double mu = 0.1;
  double s[9], lam[9];
  {
    double c[9], jx, jy;
    pm_cons(P, X, o0, o1, c, jx, jy);
    for (int i = 0; i < 9; ++i) {
      s[i] = fmax(c[i], 1e-2);
      lam[i] = mu / s[i];
    }
  }
  int status = OH_STATUS_MAX_ITER, it = 0;
  double stat = 0.0, feas = 0.0, compl_ = 0.0, fval = 0.0;
  for (it = 0; it <= P.max_iter; ++it) {
    // ---- per knot: constraints, residuals, the stage's Q, q and Lagrangian gradient --------------------------------------------------
    double c[9], jx, jy;
    pm_cons(P, X, o0, o1, c, jx, jy);
    const double wt = (P.final_only && lane < T - 1) ? 0.0 : 1.0;
    const double gx[4] = {-2.0 * wt * (g0 - X[0]), -2.0 * wt * (g1 - X[1]), 2.0 * P.w_vel * X[2], 2.0 * P.w_vel * X[3]};
    double fl = wt * ((g0 - X[0]) * (g0 - X[0]) + (g1 - X[1]) * (g1 - X[1])) + P.w_vel * (X[2] * X[2] + X[3] * X[3]);
    if (lane < T - 1) fl += w * (a[0] * a[0] + a[1] * a[1]);
    double rc[9], sig[9], mus[9], wq[9], jl[4], jq[4];
    double fe = 0.0, co = 0.0;
    for (int i = 0; i < 9; ++i) {
      rc[i] = c[i] - s[i];
      sig[i] = lam[i] / s[i];
      mus[i] = mu / s[i];
      wq[i] = mus[i] - sig[i] * rc[i];
      if (on && lane >= 1) { fe = fmax(fe, fabs(rc[i])); co = fmax(co, lam[i] * s[i]); }
    }
    pm_JTw(lam, jx, jy, jl);
    pm_JTw(wq, jx, jy, jq);
    double lx[4], q[4];
    for (int j = 0; j < 4; ++j) { lx[j] = gx[j] - jl[j]; q[j] = gx[j] - jq[j]; }
    const double Q0 = 2.0 * wt + sig[0] + sig[1] + sig[8] * jx * jx, Q5 = 2.0 * wt + sig[2] + sig[3] + sig[8] * jy * jy, Q1 = sig[8] * jx * jy;
    const double Q10 = 2.0 * P.w_vel + sig[4] + sig[5], Q15 = 2.0 * P.w_vel + sig[6] + sig[7];
    fval = pm_wave_sum(on ? fl : 0.0);
    feas = pm_wave_max(fe);
    compl_ = pm_wave_max(co);
    {  // (fmax drops NaNs: carry them separately, the thread kernel's maxima see them through the comparison chain as well)
      bool bad = false;
      for (int i = 0; i < 9; ++i) bad = bad || !(rc[i] == rc[i]) || !(lam[i] * s[i] == lam[i] * s[i]);
      if (__any(on && bad)) feas = __builtin_nan("");
    }
    // ---- backward: Riccati recursion, all lanes alike; lane t keeps K_t, k_t ----------------------------------------------------------
    double Pm[16], pv[4], padj[4];
    double Kmine[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kmine[2] = {0, 0};
    stat = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      double Q[16];
      for (int j = 0; j < 16; ++j) Q[j] = 0.0;
      Q[0] = pm_bcast(Q0, t); Q[5] = pm_bcast(Q5, t); Q[1] = Q[4] = pm_bcast(Q1, t); Q[10] = pm_bcast(Q10, t); Q[15] = pm_bcast(Q15, t);
      double qs[4], lxs[4];
      for (int j = 0; j < 4; ++j) { qs[j] = pm_bcast(q[j], t); lxs[j] = pm_bcast(lx[j], t); }
      if (t == T - 1) {
        for (int j = 0; j < 16; ++j) Pm[j] = Q[j];
        for (int j = 0; j < 4; ++j) { pv[j] = qs[j]; padj[j] = lxs[j]; }
        continue;
      }
      const double a0 = pm_bcast(a[0], t), a1 = pm_bcast(a[1], t);
      const bool pinned = P.fix_vf && t == T - 2;
      const double gu0 = 2.0 * w * a0 + dt * padj[2], gu1 = 2.0 * w * a1 + dt * padj[3];
      if (!pinned) stat = fmax(stat, fmax(fabs(gu0), fabs(gu1)));
      double PA[16];
      for (int r = 0; r < 4; ++r) {
        PA[4 * r + 0] = Pm[4 * r + 0]; PA[4 * r + 1] = Pm[4 * r + 1];
        PA[4 * r + 2] = Pm[4 * r + 2] + dt * Pm[4 * r + 0];
        PA[4 * r + 3] = Pm[4 * r + 3] + dt * Pm[4 * r + 1];
      }
      const double q00 = 2.0 * w + dt * dt * Pm[10], q01 = dt * dt * Pm[11], q11 = 2.0 * w + dt * dt * Pm[15];
      double Qux[8];
      for (int j = 0; j < 4; ++j) { Qux[j] = dt * PA[8 + j]; Qux[4 + j] = dt * PA[12 + j]; }
      const double qu0 = 2.0 * w * a0 + dt * pv[2], qu1 = 2.0 * w * a1 + dt * pv[3];
      // (reciprocal square roots and products: the 21 divisions of the textbook substitution were half the instructions of a knot step)
      const double i00 = rsqrt(q00), l10 = q01 * i00, i11 = rsqrt(q11 - l10 * l10);
      double Kt[8], kt[2];
      double Pn[16], pn[4], pa[4];
      if (pinned) {
        for (int j = 0; j < 8; ++j) Kt[j] = 0.0;
        Kt[2] = Kt[4 + 3] = -1.0 / dt;
        kt[0] = kt[1] = 0.0;
        const double kf = -1.0 / dt;
        for (int r = 0; r < 4; ++r)
          for (int cc = 0; cc < 4; ++cc) {
            double v = PA[4 * r + cc];
            if (r >= 2) v += dt * PA[4 * (r - 2) + cc];
            v += Qux[r] * Kt[cc] + Qux[4 + r] * Kt[4 + cc];
            v += Kt[r] * Qux[cc] + Kt[4 + r] * Qux[4 + cc];
            Pn[4 * r + cc] = Q[4 * r + cc] + v;
          }
        Pn[10] += kf * kf * q00; Pn[11] += kf * kf * q01; Pn[14] += kf * kf * q01; Pn[15] += kf * kf * q11;
        for (int r = 0; r < 4; ++r) {
          double v = pv[r], va = padj[r];
          if (r >= 2) { v += dt * pv[r - 2]; va += dt * padj[r - 2]; }
          pn[r] = qs[r] + v;
          pa[r] = lxs[r] + va;
        }
        pn[2] += kf * qu0; pn[3] += kf * qu1;
        pa[2] += kf * gu0; pa[3] += kf * gu1;
      } else {
        for (int j = 0; j < 4; ++j) {
          const double y0 = Qux[j] * i00, y1 = (Qux[4 + j] - l10 * y0) * i11;
          const double z1 = y1 * i11, z0 = (y0 - l10 * z1) * i00;
          Kt[j] = -z0; Kt[4 + j] = -z1;
        }
        const double y0 = qu0 * i00, y1 = (qu1 - l10 * y0) * i11;
        const double z1 = y1 * i11, z0 = (y0 - l10 * z1) * i00;
        kt[0] = -z0; kt[1] = -z1;
        for (int r = 0; r < 4; ++r)
          for (int cc = 0; cc < 4; ++cc) {
            double v = PA[4 * r + cc];
            if (r >= 2) v += dt * PA[4 * (r - 2) + cc];
            v += Qux[r] * Kt[cc] + Qux[4 + r] * Kt[4 + cc];
            Pn[4 * r + cc] = Q[4 * r + cc] + v;
          }
        for (int r = 0; r < 4; ++r) {
          double v = pv[r], va = padj[r];
          if (r >= 2) { v += dt * pv[r - 2]; va += dt * padj[r - 2]; }
          pn[r] = qs[r] + v + Qux[r] * kt[0] + Qux[4 + r] * kt[1];
          pa[r] = lxs[r] + va;
        }
      }
      if (lane == t) {
        for (int j = 0; j < 8; ++j) Kmine[j] = Kt[j];
        kmine[0] = kt[0]; kmine[1] = kt[1];
      }
      for (int r = 0; r < 4; ++r)
        for (int cc = 0; cc < 4; ++cc) Pm[4 * r + cc] = 0.5 * (Pn[4 * r + cc] + Pn[4 * cc + r]);
      for (int j = 0; j < 4; ++j) { pv[j] = pn[j]; padj[j] = pa[j]; }
    }
    if (!(stat == stat) || !(fval == fval) || !(fabs(fval) < 1e300) || !(feas == feas) || !(compl_ == compl_)) { status = OH_STATUS_NUMERICAL; break; }
    if (stat <= P.tol && feas <= P.tol && compl_ <= P.tol) { status = OH_STATUS_CONVERGED; break; }
    // No feasible plan (round 6; the reference: IPOPT's Infeasible_Problem_Detected -> did_solve() False, solver.py:407-412): the obstacle is a parameter of every
    // knot, so a problem whose pinned knot is fine can still have none -- the obstacle lands where the mass cannot leave in time.  The infeasible-start
    // iteration then shows its textbook signature: the slack residual stalls at the violation it cannot remove while the multipliers of those rows leave every
    // bound (lam s from 1e-1 to 1e8 in three steps, 1e240 a dozen steps later; oracle/pointmass_ipm.py has the same rule).  A healthy run keeps lam s at the
    // size of the barrier parameter (<= 1).
    if (compl_ > 1e6 && feas > 1e3 * P.tol) { status = OH_STATUS_INFEASIBLE; break; }
    if (it == P.max_iter) break;

    // ---- forward 1: Newton direction (serial), then the step lengths per knot ---------------------------------------------------------
    double dxm[4] = {0, 0, 0, 0}, dam[2] = {0, 0};
    {
      double dx[4] = {0, 0, 0, 0};
      for (int t = 0; t < T - 1; ++t) {
        double da0 = pm_bcast(kmine[0], t), da1 = pm_bcast(kmine[1], t);
        for (int j = 0; j < 4; ++j) { da0 += pm_bcast(Kmine[j], t) * dx[j]; da1 += pm_bcast(Kmine[4 + j], t) * dx[j]; }
        if (lane == t) { dam[0] = da0; dam[1] = da1; }
        const double n0 = dx[0] + dt * dx[2], n1 = dx[1] + dt * dx[3], n2 = dx[2] + dt * da0, n3 = dx[3] + dt * da1;
        dx[0] = n0; dx[1] = n1; dx[2] = n2; dx[3] = n3;
        if (lane == t + 1)
          for (int j = 0; j < 4; ++j) dxm[j] = dx[j];
      }
    }
    double ds[9], dl[9];
    double ap = 1.0, ad = 1.0;
    {
      double d[9];
      pm_Jv(dxm, jx, jy, d);
      for (int i = 0; i < 9; ++i) {
        ds[i] = d[i] + rc[i];
        dl[i] = (mus[i] - lam[i]) - sig[i] * ds[i];  // (the quotients of the residual phase: same operands, same values)
        if (on && lane >= 1) {
          if (ds[i] < 0.0) ap = fmin(ap, -0.995 * s[i] / ds[i]);
          if (dl[i] < 0.0) ad = fmin(ad, -0.995 * lam[i] / dl[i]);
        }
      }
    }
    ap = pm_wave_min(ap);
    ad = pm_wave_min(ad);
    // ---- forward 2: take the step; the barrier parameter from the products summed in the thread kernel's order -----------------------
    double prod[9];
    for (int i = 0; i < 9; ++i) {
      if (on && lane >= 1) {
        s[i] += ap * ds[i];
        lam[i] += ad * dl[i];
      }
      prod[i] = s[i] * lam[i];
    }
    double gap = 0.0;
    for (int t = 1; t < T; ++t)
      for (int i = 0; i < 9; ++i) gap += pm_bcast(prod[i], t);
    if (lane < T - 1) { a[0] += ap * dam[0]; a[1] += ap * dam[1]; }
    {
      double xn[4] = {pb[0], pb[1], pb[2], pb[3]};
      for (int t = 0; t < T; ++t) {
        if (lane == t)
          for (int j = 0; j < 4; ++j) X[j] = xn[j];
        if (t < T - 1) {
          const double a0 = pm_bcast(a[0], t), a1 = pm_bcast(a[1], t);
          const double n0 = xn[0] + dt * xn[2], n1 = xn[1] + dt * xn[3], n2 = xn[2] + dt * a0, n3 = xn[3] + dt * a1;
          xn[0] = n0; xn[1] = n1; xn[2] = n2; xn[3] = n3;
        }
      }
    }
    gap /= (double)(9 * (T - 1));
    const double am = fmin(ap, ad);
    const double sigma = (am > 0.9) ? 0.1 : ((am > 0.5) ? 0.3 : 0.8);
    mu = fmax(sigma * gap, 1e-2 * P.tol);
  }
  if (xo && on) {
    double* xb = xo + (size_t)b * nx;
    xb[2 * lane] = X[0]; xb[2 * lane + 1] = X[1];
    xb[2 * T + 2 * lane] = X[2]; xb[2 * T + 2 * lane + 1] = X[3];
  }
  if (lane == 0) {
    if (fo) fo[b] = fval;
    if (kkt) { kkt[3 * (size_t)b] = stat; kkt[3 * (size_t)b + 1] = feas; kkt[3 * (size_t)b + 2] = compl_; }
    if (iters_o) iters_o[b] = it > P.max_iter ? P.max_iter : it;
    if (status_o) status_o[b] = status;
  }
}

// ---- closed-loop receding horizon kept on the device (example/point_mass_mpc.py main loop, :293-306 + Controller.next_state :156-161) ----
// parameters of one tick from the current plant state: p = [curr; dcurr; goal; obs], goal[:, i] = curr + ramp * i,
// obs[:, i] = obstacle centre at time index tick * advance + i of the table
__global__ __launch_bounds__(64) void k_pm_tick_params(int B, int T, int tick, int advance, double ramp, const double* __restrict__ state,
                                                       const double* __restrict__ obs_table, double* __restrict__ p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double* pb = p + (size_t)b * (4 + 4 * (size_t)T);
  const double c0 = state[4 * (size_t)b], c1 = state[4 * (size_t)b + 1];
  pb[0] = c0; pb[1] = c1; pb[2] = state[4 * (size_t)b + 2]; pb[3] = state[4 * (size_t)b + 3];
  for (int i = 0; i < T; ++i) {
    pb[4 + 2 * i] = c0 + ramp * i;
    pb[4 + 2 * i + 1] = c1 + ramp * i;
    pb[4 + 2 * T + 2 * i] = obs_table[2 * ((size_t)tick * advance + i)];
    pb[4 + 2 * T + 2 * i + 1] = obs_table[2 * ((size_t)tick * advance + i) + 1];
  }
}
// the plant follows the plan for `advance` knots (the reference evaluates its linear interpolant at advance * dt, i.e. at a knot)
__global__ __launch_bounds__(64) void k_pm_advance(int B, int T, int advance, const double* __restrict__ x, double* __restrict__ state_next) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* xb = x + (size_t)b * 4 * T;
  state_next[4 * (size_t)b] = xb[2 * advance];
  state_next[4 * (size_t)b + 1] = xb[2 * advance + 1];
  state_next[4 * (size_t)b + 2] = xb[2 * T + 2 * advance];
  state_next[4 * (size_t)b + 3] = xb[2 * T + 2 * advance + 1];
}
void oh_launch_pm_tick_params(hipStream_t s, int B, int T, int tick, int advance, double ramp, const double* state, const double* obs_table, double* p) {
  hipLaunchKernelGGL(k_pm_tick_params, dim3((B + 63) / 64), dim3(64), 0, s, B, T, tick, advance, ramp, state, obs_table, p);
}
void oh_launch_pm_advance(hipStream_t s, int B, int T, int advance, const double* x, double* state_next) {
  hipLaunchKernelGGL(k_pm_advance, dim3((B + 63) / 64), dim3(64), 0, s, B, T, advance, x, state_next);
}

// The plant's state (y_0, dy_0) = (curr, dcurr) is pinned by equality rows, so the box rows and the obstacle row of knot 0 are constants of an instance
// (the reference writes them for every knot, point_mass_mpc.py:96-123).  A start outside the box or inside the obstacle has no feasible plan: IPOPT reports an
// infeasible problem (did_solve() False, solver.py:407-412) -- here OH_STATUS_INFEASIBLE, kkt[1] = the violation (round 5; SURVEY C3 rejects such starts).
__global__ __launch_bounds__(64) void k_pm_infeasible(PmParams P, int B, const double* __restrict__ pin, double* __restrict__ kkt, int* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* pb = pin + (size_t)b * (4 + 4 * (size_t)P.T);
  double c[9], jx, jy;
  pm_cons(P, pb, pb[4 + 2 * P.T], pb[4 + 2 * P.T + 1], c, jx, jy);
  double worst = 0.0;
  for (int i = 0; i < 9; ++i) worst = fmin(worst, c[i]);
  if (worst < -P.tol) {
    if (status) status[b] = OH_STATUS_INFEASIBLE;
    if (kkt) kkt[3 * (size_t)b + 1] = fmax(kkt[3 * (size_t)b + 1], -worst);
  }
}

void oh_launch_pm_solve(hipStream_t s, const PmParams& P, const PmBuffers& D, const double* x0, const double* p, double* x, double* f, double* kkt,
                        int* iters, int* status) {
  // a wavefront per instance while that leaves the chip room (the thread kernel issues ~8x fewer instructions per instance, but needs ~10^5
  // instances to fill the SIMDs: 1024 plants take 3.5 ms with it and 0.7 ms here)
  const int wave_max = oh_launch_opts().pm_wave_max;  // option "pm_wave_max", default 20480  // (tools/gpu_pm_sweep.py: 16 384 plants 6.8 / 5.8 ms thread / wave kernel, 32 768 9.8 / 11.3 ms)
  if (P.T <= 64 && D.B <= wave_max) hipLaunchKernelGGL(k_pm_solve_wave, dim3(D.B), dim3(64), 0, s, P, D.B, x0, p, x, f, kkt, iters, status);
  else hipLaunchKernelGGL(k_pm_solve, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, x0, p, x, f, kkt, iters, status);
  if (status || kkt) hipLaunchKernelGGL(k_pm_infeasible, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D.B, p, kkt, status);
}
