// Device functions of the figure-eight family shared by the batched kernels (k_eval / k_couple / k_step:
// one lane per knot or per instance, stage data in HBM) and by the persistent tail kernel (k_tail: one
// wavefront per instance, one lane per knot, stage data in registers).  Sharing them keeps the two paths
// arithmetically identical operation by operation.
#pragma once
#include "oh_device.h"
#include "oh_types.h"

// Orientation residual c = vee(skew(Re Rc^T)) and M = 1/2 (tr(A) I - A) with dc = M domega.
OH_DEV void orient_residual(const double* Re, const double* Rc, double* c, double* M) {
  double A[9];
  mmT3(Re, Rc, A);
  c[0] = 0.5 * (A[7] - A[5]);
  c[1] = 0.5 * (A[2] - A[6]);
  c[2] = 0.5 * (A[3] - A[1]);
  const double tr = A[0] + A[4] + A[8];
#pragma unroll
  for (int i = 0; i < 9; ++i) M[i] = -0.5 * A[i];
  M[0] += 0.5 * tr; M[4] += 0.5 * tr; M[8] += 0.5 * tr;
}

// Sphere-clearance rows of one knot (sphere_collision_avoidance_constraints, builder.py:366-417): walks the chain at q and calls
// row(l, o, g, dg) for every sphere link l and obstacle o with g = ||c_l - o||^2 - (r_l + r_o)^2 and dg = 2 J_l^T (c_l - o).
// The rows of link l are emitted right after the joint it hangs on: c_l and the columns z_j x (c_l - p_j), j <= joint(l), of its
// position Jacobian only need frames already visited.  par: [n_links + 4 n_obs][Bp] link radii, then x, y, z, r per obstacle.
template <int N, class F>
OH_DEV void sphere_rows_walk(const oh_chain* __restrict__ ch, const GuardParams& GP, const double* __restrict__ par, const size_t Bp, const int b,
                             const double (&q)[N], F&& row) {
  double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0}, p[3] = {0.0, 0.0, 0.0}, z[N][3], pj[N][3];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double tv[3];
    mv3(R, ch->p0[k], tv);
    p[0] += tv[0]; p[1] += tv[1]; p[2] += tv[2];
    if (!ch->r0ident[k]) {
      double Rn[9];
      mm3(R, ch->R0[k], Rn);
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    pj[k][0] = p[0]; pj[k][1] = p[1]; pj[k][2] = p[2];
    if (ch->jtype[k] == 0) {
      double sn, cs;
      sincos_joint(q[k], &sn, &cs);
      const int code = ch->axcode[k];
      if (code != 0) rot_principal_right(R, code, sn, cs, z[k]);
      else rot_axis_right(R, ch->axis[k], sn, cs, z[k]);
    } else {
      mv3(R, ch->axis[k], z[k]);
      p[0] += z[k][0] * q[k]; p[1] += z[k][1] * q[k]; p[2] += z[k][2] * q[k];
    }
    for (int l = 0; l < GP.n_links; ++l) {
      if (GP.link_joint[l] != k) continue;
      double c[3];
      mv3(R, GP.link_off[l], c);
      c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
      double Jl[N][3];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (j <= k) {
          if (ch->jtype[j] == 0) {
            const double dd[3] = {c[0] - pj[j][0], c[1] - pj[j][1], c[2] - pj[j][2]};
            cross3(z[j], dd, Jl[j]);
          } else {
            Jl[j][0] = z[j][0]; Jl[j][1] = z[j][1]; Jl[j][2] = z[j][2];
          }
        } else {
          Jl[j][0] = Jl[j][1] = Jl[j][2] = 0.0;
        }
      }
      const double rl = par[(size_t)l * Bp + b];
      for (int o = 0; o < GP.n_obs; ++o) {
        const size_t ob = (size_t)(GP.n_links + 4 * o) * Bp + b;
        const double d[3] = {c[0] - par[ob], c[1] - par[ob + Bp], c[2] - par[ob + 2 * Bp]};
        const double rr = rl + par[ob + 3 * Bp];
        double dg[N];
#pragma unroll
        for (int j = 0; j < N; ++j) dg[j] = 2.0 * dot3(Jl[j], d);
        row(l, o, dot3(d, d) - rr * rr, dg);
      }
    }
  }
}

// One knot: retraction onto R(q)=Rc (q is updated in place), FK chain + Jacobians, tracking cost phi,
// constraint violation cv, tracking gradient g, Hessian block W (Gauss-Newton, or exact with the
// multiplier estimate from Gprev), Householder null-space basis Z of the orientation rows, Dr = Z^T W Z.
// Null-space basis of the orientation rows from its three Householder vectors: Z = H1 H2 H3 [0; I_NZ].  The batched kernels keep only
// the vectors in HBM (3N - 3 doubles against N (N - 3) for Z) and rebuild Z with this very code where they need it, so the rebuilt
// basis equals the one eval_knot used bit for bit.
template <int N>
OH_DEV void z_from_householder(const double (&V)[3][N], double (&Z)[N][N - 3]) {
  constexpr int NZ = N - 3;
#pragma unroll
  for (int a = 0; a < NZ; ++a) {
    double col[N];
#pragma unroll
    for (int k = 0; k < N; ++k) col[k] = (k == a + 3) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 2; m >= 0; --m) {
      double d = 0.0;
#pragma unroll
      for (int k = m; k < N; ++k) d += V[m][k] * col[k];
      d *= 2.0;
#pragma unroll
      for (int k = m; k < N; ++k) col[k] -= d * V[m][k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) Z[k][a] = col[k];
  }
}
// packed storage of the vectors: V[m][k], k >= m, at row HV_OFF(N, m) + k - m of 3N - 3
#define HV_ROWS(N) (3 * (N) - 3)
#define HV_OFF(N, m) ((m) * (N) - (m) * ((m) - 1) / 2)
// rows of the model stage array: e (3), then Jp Z row-major (3 x (N - 3))
#define MDL_ROWS(N) (3 + 3 * ((N) - 3))

// Retraction tolerance of a trial point.  While the accepted point is far from stationary (reduced gradient above the hybrid switch)
// the orientation violation a trial may keep is tied to the decrease its step predicts: 1e-3 pred, two orders below where the step
// counts start to move, never looser than 1e-5.  In the end game every point is retracted to the floor: an accepted point that keeps
// a violation c carries an objective that is off by (multiplier) x c, and once the predicted decreases fall below that (they shrink
// quadratically) no accurate trial can beat it any more.  The first evaluation and restarts use the floor as well.
// tol_retract_min < tol_retract (first built for handles with inequality rows, every handle since the end of round 3): the outer loop of
// those handles asks for stat <= tol again after every multiplier update, with
// predicted decreases of 1e-12 when 1e-10 of violation is worth 1.5e-10 of objective -- steps were then accepted or refused by the rounding
// of the retraction and a few instances per 10^4 sat at stat 1e-5 until the iteration cap (round 3; reproduced in oracle/structured.py).
// There the end-game tolerance follows the prediction too: 1e-2 pred, down to tol_retract_min.
OH_DEV double retract_tol(const FigParams& P, const bool have_tgt, const double pred, const double stat) {
  return (have_tgt && stat > P.hyb_switch) ? fmin(1e-5, fmax(P.tol_retract, 1e-3 * pred)) : fmin(P.tol_retract, fmax(P.tol_retract_min, 1e-2 * pred));
}

// Velocity rows of one interval (enforce_model_limits(time_deriv=1), builder.py:471-509): v = (qb - qa) / dt, rows v - vlo >= 0, vup - v >= 0 with the
// multipliers lam[0..N) / lam[N..2N) and penalty rho.  Out: sigma_k = d L_A / d v_k / dt (enters the gradient of knot b with +, of knot a
// with -), the Gauss-Newton weight w_k = rho (active rows) / dt^2 of (qb_k - qa_k)^2, the augmented-Lagrangian value psi and the measure
// |min(g, lam / rho)|_inf.  (oracle/structured.py:vel_terms)
template <int N>
OH_DEV void velocity_rows(const GuardParams& GP, const double dt, const double rho, const double (&qa)[N], const double (&qb)[N], const double (&lam)[2 * N],
                          double (&sigma)[N], double (&w)[N], double& psi, double& meas) {
  psi = 0.0;
  meas = 0.0;
  const double irho = 1.0 / rho, i2rho = 1.0 / (2.0 * rho);  // (one division each instead of four per joint: an f64 division is ~30 instructions)
  const double idt = 1.0 / dt;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double v = (qb[k] - qa[k]) * idt;
    const double g_lo = v - GP.vlo[k], g_up = GP.vup[k] - v;
    const double s_lo = fmax(0.0, lam[k] - rho * g_lo), s_up = fmax(0.0, lam[N + k] - rho * g_up);
    psi += (s_lo * s_lo - lam[k] * lam[k]) * i2rho + (s_up * s_up - lam[N + k] * lam[N + k]) * i2rho;
    meas = fmax(meas, fmax(fabs(fmin(g_lo, lam[k] * irho)), fabs(fmin(g_up, lam[N + k] * irho))));
    sigma[k] = (s_up - s_lo) * idt;
    w[k] = rho * ((s_lo > 0.0 ? 1.0 : 0.0) + (s_up > 0.0 ? 1.0 : 0.0)) * idt * idt;
  }
}

// Hooks let the batched kernel shorten live ranges: q and g leave for HBM the moment they are final, and the Lagrangian gradient of the
// accepted point is fetched only inside the exact-curvature branch (k_eval sits at the 256-register limit of 2 waves/SIMD).
struct EvalNoHooks {
  template <int N> OH_DEV void q_final(const double (&)[N]) const {}
  template <int N> OH_DEV void g_final(const double (&)[N]) const {}
  template <int N> OH_DEV void v_final(const double (&)[3][N]) const {}
  template <int N> OH_DEV void load_G(const double (&Gprev)[N], double (&G)[N]) const {
#pragma unroll
    for (int k = 0; k < N; ++k) G[k] = Gprev[k];
  }
};
// have_tgt / e_tgt: the end-effector position the linear model of the step predicted, e_cur + (Jp Z)_cur z.  With it the retraction
// takes minimum-norm Newton steps on the six rows [c(q); e(q) - e_tgt] (a second-order correction: the trial point follows the curved
// valley of the stiff tracking cost; without it the model is honest only for tiny steps along the redundant direction and the slow
// half of the instances crawls: mean 30 -> 16 steps).  e_out, JZ_out = e and Jp Z of this knot, the next step's prediction data.
// tol_r: retraction tolerance of this evaluation (retract_tol below).
// MODE: the batched path runs the knot in two kernels so that each fits two waves per SIMD (fused, the retraction loop with the
// six-row step needs ~350 live registers): EVAL_RETRACT_ONLY stops after the loop (q through hooks.q_final), EVAL_ONLY takes q as
// retracted and is the loop's last pass plus everything after it.  Same code, same inputs: the pair is bit-identical to EVAL_FUSED
// (tail kernel, guarded / lead variants, host port).
enum { EVAL_FUSED = 0, EVAL_RETRACT_ONLY = 1, EVAL_ONLY = 2 };
template <int N, bool LEAD = false, class Hooks = EvalNoHooks, int MODE = EVAL_FUSED>
OH_DEV void eval_knot(const oh_chain* __restrict__ ch, const FigParams& P, const int t, double (&q)[N], const double (&pc)[3],
                      const double (&Rc)[9], const bool exact, const bool have_G, const double (&Gprev)[N], double& phi, double& cv, double (&g)[N],
                      double (&Dr)[(N - 3) * (N - 2) / 2], double (&Z)[N][N - 3], const bool have_tgt, const double (&e_tgt)[3], const double tol_r,
                      double (&e_out)[3], double (&JZ_out)[3][N - 3], const double lead_theta = 0.0, const Hooks hooks = Hooks()) {
  constexpr int NZ = N - 3;
  double R[9], p[3], z[N][3], pj[N][3];
  double Re[9], c[3], M[9];
  double cmax;
  double Rb[9], pb[3];  // frame after the parameterised lead joint (LEAD only)
  if constexpr (LEAD) lead_base(ch, lead_theta, Rb, pb);
  // settled: the Newton step just taken was so short that the violation it leaves is below the tolerance without looking.  A step dq
  // removes c to first order; what remains is the second-order term, bounded by ||dq||_1^2 (the second derivatives of the rows are
  // cross products of unit joint axes).  The kinematics pass that would only confirm it is skipped: by the batched retraction
  // kernel altogether (k_evalb evaluates the point anyway and reports the violation it finds), by the fused variants in that they
  // go on to the evaluation without a further correction - the accepted q is the same in both.
  bool settled = false;
  for (int it = 0;; ++it) {
    fk_chain<N, LEAD>(ch, q, R, p, z, pj, Rb, pb);
    mm3(R, ch->R_tool, Re);
    orient_residual(Re, Rc, c, M);
    cmax = fmax(fabs(c[0]), fmax(fabs(c[1]), fabs(c[2])));
    if (MODE == EVAL_ONLY || settled || cmax <= tol_r || it >= P.max_retract) break;
    if constexpr (MODE != EVAL_ONLY) {
    double dq1 = 0.0;  // ||dq||_1 of this correction
    // Newton correction q <- q - Jc^T (Jc Jc^T)^{-1} c,  Jc = M Jw,  Jw[:,k] = z_k (revolute) / 0
    double Jc[N][3];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (ch->jtype[k] == 0) mv3(M, z[k], Jc[k]);
      else { Jc[k][0] = Jc[k][1] = Jc[k][2] = 0.0; }
    }
    if (have_tgt) {
      // six rows: J6_k = [M z_k; z_k x (e - p_k)], residual [c; e - e_tgt], minimum-norm step q -= J6^T (J6 J6^T + 1e-10 I)^{-1} r6.
      // The columns are formed twice (for the normal matrix and for the update) instead of being kept: 42 doubles fewer alive.
      double e6[3], tv6[3];
      mv3(R, ch->p_tool, tv6);
      e6[0] = p[0] + tv6[0]; e6[1] = p[1] + tv6[1]; e6[2] = p[2] + tv6[2];
      auto column = [&](const int k, double (&col)[6]) {
        col[0] = Jc[k][0]; col[1] = Jc[k][1]; col[2] = Jc[k][2];
        if (ch->jtype[k] == 0) {
          const double d[3] = {e6[0] - pj[k][0], e6[1] - pj[k][1], e6[2] - pj[k][2]};
          double cp[3];
          cross3(z[k], d, cp);
          col[3] = cp[0]; col[4] = cp[1]; col[5] = cp[2];
        } else {
          col[3] = z[k][0]; col[4] = z[k][1]; col[5] = z[k][2];
        }
      };
      double S6[21];
#pragma unroll
      for (int i = 0; i < 21; ++i) S6[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) S6[tri(i, i)] = 1e-10;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double col[6];
        column(k, col);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) S6[tri(i, j)] += col[i] * col[j];
      }
      double rd6[6];
      chol_rcp<6>(S6, rd6, 0.0);  // reciprocal pivots: no f64 division in the loop (33 of them cost more than the rest of the solve)
      double y6[6] = {c[0], c[1], c[2], e6[0] - e_tgt[0], e6[1] - e_tgt[1], e6[2] - e_tgt[2]};
      fsub_rcp<6>(S6, rd6, y6);
      bsub_rcp<6>(S6, rd6, y6);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double col[6];
        column(k, col);
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += col[i] * y6[i];
        q[k] -= acc;
        dq1 += fabs(acc);
      }
    } else {
      double S[6] = {1e-14, 0, 1e-14, 0, 0, 1e-14};
#pragma unroll
      for (int k = 0; k < N; ++k) {
        S[0] += Jc[k][0] * Jc[k][0];
        S[1] += Jc[k][1] * Jc[k][0];
        S[2] += Jc[k][1] * Jc[k][1];
        S[3] += Jc[k][2] * Jc[k][0];
        S[4] += Jc[k][2] * Jc[k][1];
        S[5] += Jc[k][2] * Jc[k][2];
      }
      double rd3[3];
      chol_rcp<3>(S, rd3, 0.0);
      double y[3] = {c[0], c[1], c[2]};
      fsub_rcp<3>(S, rd3, y);
      bsub_rcp<3>(S, rd3, y);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const double acc = dot3(Jc[k], y);
        q[k] -= acc;
        dq1 += fabs(acc);
      }
    }
    settled = P.settle_k * dq1 * dq1 <= tol_r;
    if (MODE == EVAL_RETRACT_ONLY && settled) break;
    }
  }
  cv = cmax;
  hooks.q_final(q);
  if constexpr (MODE == EVAL_RETRACT_ONLY) {
    phi = 0.0;
    e_out[0] = e_out[1] = e_out[2] = 0.0;
    return;
  }

  // end-effector position, tracking residual
  double e[3], tv[3];
  mv3(R, ch->p_tool, tv);
  e[0] = p[0] + tv[0]; e[1] = p[1] + tv[1]; e[2] = p[2] + tv[2];
  e_out[0] = e[0]; e_out[1] = e[1]; e_out[2] = e[2];
  const double l[3] = {P.local_path[3 * t], P.local_path[3 * t + 1], P.local_path[3 * t + 2]};
  double r[3];
  if (P.path_in_frame) mv3(Rc, l, r);
  else { r[0] = l[0]; r[1] = l[1]; r[2] = l[2]; }
  r[0] += pc[0] - e[0]; r[1] += pc[1] - e[1]; r[2] += pc[2] - e[2];
  const double w = P.w_path;
  phi = w * dot3(r, r);

  // Jacobian columns
  double Jp[N][3], Jc[N][3];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (ch->jtype[k] == 0) {
      const double d[3] = {e[0] - pj[k][0], e[1] - pj[k][1], e[2] - pj[k][2]};
      cross3(z[k], d, Jp[k]);
      mv3(M, z[k], Jc[k]);
    } else {
      Jp[k][0] = z[k][0]; Jp[k][1] = z[k][1]; Jp[k][2] = z[k][2];
      Jc[k][0] = Jc[k][1] = Jc[k][2] = 0.0;
    }
  }
  // gradient of w ||r||^2 : -2 w Jp^T r
#pragma unroll
  for (int k = 0; k < N; ++k) g[k] = -2.0 * w * dot3(Jp[k], r);
  hooks.g_final(g);

  // multipliers of the orientation rows (exact curvature only): least squares of  G_prev + Jc^T lam = 0  with the Lagrangian
  // gradient of the last accepted point (lagged by one iteration; exact at convergence)
  double lam[3] = {0.0, 0.0, 0.0};
  if (exact && have_G) {
    double Gl[N];
    hooks.load_G(Gprev, Gl);
    double S[6] = {1e-14, 0, 1e-14, 0, 0, 1e-14};
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double Gk = Gl[k];
      S[0] += Jc[k][0] * Jc[k][0];
      S[1] += Jc[k][1] * Jc[k][0];
      S[2] += Jc[k][1] * Jc[k][1];
      S[3] += Jc[k][2] * Jc[k][0];
      S[4] += Jc[k][2] * Jc[k][1];
      S[5] += Jc[k][2] * Jc[k][2];
      lam[0] -= Jc[k][0] * Gk; lam[1] -= Jc[k][1] * Gk; lam[2] -= Jc[k][2] * Gk;
    }
    double rdl[3];
    chol_rcp<3>(S, rdl, 0.0);
    fsub_rcp<3>(S, rdl, lam);
    bsub_rcp<3>(S, rdl, lam);
  }

  // Householder QR of Jc^T (N x 3): H3 H2 H1 Jc^T = [Rf; 0];  Z = H1 H2 H3 [0; I_NZ]
  double A[3][N];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int k = 0; k < N; ++k) A[m][k] = Jc[k][m];
  double V[3][N];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    double nrm2 = 0.0;
#pragma unroll
    for (int k = m; k < N; ++k) nrm2 += A[m][k] * A[m][k];
    const double nrm = sqrt(nrm2);
    const double alpha = (A[m][m] > 0.0) ? -nrm : nrm;
    double vn2 = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      V[m][k] = (k < m) ? 0.0 : ((k == m) ? A[m][k] - alpha : A[m][k]);
      vn2 += V[m][k] * V[m][k];
    }
    const double inv = (vn2 > 1e-300) ? 1.0 / sqrt(vn2) : 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) V[m][k] *= inv;
#pragma unroll
    for (int m2 = m + 1; m2 < 3; ++m2) {
      double d = 0.0;
#pragma unroll
      for (int k = m; k < N; ++k) d += V[m][k] * A[m2][k];
      d *= 2.0;
#pragma unroll
      for (int k = m; k < N; ++k) A[m2][k] -= d * V[m][k];
    }
  }
  hooks.v_final(V);
  z_from_householder<N>(V, Z);

  // reduced block Dr = Z^T W Z (packed lower NZ x NZ) without forming the N x N block: the Gauss-Newton part is
  // 2 w (Jp Z)^T (Jp Z); the exact curvature C is projected separately, on the lanes that use it
  double JZ[3][NZ];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int a = 0; a < NZ; ++a) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) s += Jp[k][m] * Z[k][a];
      JZ[m][a] = s;
      JZ_out[m][a] = s;
    }
#pragma unroll
  for (int a = 0; a < NZ; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) Dr[tri(a, c2)] = 2.0 * w * (JZ[0][a] * JZ[0][c2] + JZ[1][a] * JZ[1][c2] + JZ[2][a] * JZ[2][c2]);
  if (exact) {
    // C_ij = -2 w r . d2p/dq_j dq_i + lam . d2c/dq_j dq_i;  d2p/dq_j dq_i = z_j x Jp_i for j <= i (revolute j),
    // d2c = 1/2 z_j x z_i (j < i), exact on the constraint manifold
    double C[N * (N + 1) / 2];
#pragma unroll
    for (int i = 0; i < N * (N + 1) / 2; ++i) C[i] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (ch->jtype[j] == 0) {
        double rz[3], lz[3];
        cross3(r, z[j], rz);    // (r x z_j) . Jp_i = r . (z_j x Jp_i)
        cross3(lam, z[j], lz);  // (lam x z_j) . z_i = lam . (z_j x z_i)
#pragma unroll
        for (int i = j; i < N; ++i) {
          double v = -2.0 * w * dot3(rz, Jp[i]);
          if (i > j && ch->jtype[i] == 0) v += 0.5 * dot3(lz, z[i]);
          C[tri(i, j)] = v;
        }
      }
    }
    double CZ[N][NZ];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += C[(i >= k) ? tri(i, k) : tri(k, i)] * Z[k][a];
        CZ[i][a] = s;
      }
#pragma unroll
    for (int a = 0; a < NZ; ++a)
#pragma unroll
      for (int c2 = 0; c2 <= a; ++c2) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += Z[k][a] * CZ[k][c2];
        Dr[tri(a, c2)] += s;
      }
  }
}

// Neighbour coupling of one knot: G = g + 2k((q0-qm) - (qp-q0)), gt = Z^T G, E = -2k Z^T Zn, merit share.
template <int N>
OH_DEV void couple_knot(const double kappa, const bool last, const double (&qm)[N], const double (&q0)[N], const double (&qp)[N],
                        const double (&g)[N], const double (&Zt)[N][N - 3], const double (&Zn)[N][N - 3], const double phi,
                        double (&G)[N], double (&gt)[N - 3], double (&E)[(N - 3) * (N - 3)], double& merit) {
  constexpr int NZ = N - 3;
  const double kap2 = 2.0 * kappa;
#pragma unroll
  for (int a = 0; a < NZ; ++a) {
    gt[a] = 0.0;
#pragma unroll
    for (int c2 = 0; c2 < NZ; ++c2) E[a * NZ + c2] = 0.0;
  }
  double sm = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double dm = q0[k] - qm[k];
    sm += dm * dm;
    double Gk = g[k] + kap2 * dm;
    if (!last) Gk -= kap2 * (qp[k] - q0[k]);
    G[k] = Gk;
#pragma unroll
    for (int a = 0; a < NZ; ++a) {
      gt[a] += Zt[k][a] * Gk;
      if (!last) {
#pragma unroll
        for (int c2 = 0; c2 < NZ; ++c2) E[a * NZ + c2] -= kap2 * Zt[k][a] * Zn[k][c2];
      }
    }
  }
  merit = phi + kappa * sm;
}


// One backward Riccati step from knot t+1 to knot t.  In: S = S_{t+1} (unfactorised), rn = r_{t+1},
// E = E_t (row-major), Ht = D_t with the diagonal shift already added, gt = g~_t.  Out: S = S_t, rn = r_t,
// and the gains of knot t+1 (Kmat row-major [row*NZ+col], kv) for z_{t+1} = -(kv + Kmat z_t).
template <int NZ>
OH_DEV bool riccati_back(double (&S)[NZ * (NZ + 1) / 2], double (&rd)[NZ], double (&rn)[NZ], const double (&E)[NZ * NZ],
                         const double (&Ht)[NZ * (NZ + 1) / 2], const double (&gt)[NZ], double (&Kmat)[NZ * NZ], double (&kv)[NZ], double* quad = nullptr) {
  // quad (optional) += r_{t+1}^T S_{t+1}^{-1} r_{t+1} = |L^{-1} r_{t+1}|^2: summed over the knots it is g^T M^{-1} g = -g.z of the step the sweep solves for
  const bool ok = chol_rcp<NZ>(S, rd, 1e-12);
  double X[NZ][NZ];  // X = L^{-1} E^T, column a from row a of E
#pragma unroll
  for (int a = 0; a < NZ; ++a) {
    double col[NZ];
#pragma unroll
    for (int c2 = 0; c2 < NZ; ++c2) col[c2] = E[a * NZ + c2];
    fsub_rcp<NZ>(S, rd, col);
#pragma unroll
    for (int c2 = 0; c2 < NZ; ++c2) X[c2][a] = col[c2];
  }
  double u[NZ];
#pragma unroll
  for (int a = 0; a < NZ; ++a) u[a] = rn[a];
  fsub_rcp<NZ>(S, rd, u);
  if (quad) {
    double qs = 0.0;
#pragma unroll
    for (int a = 0; a < NZ; ++a) qs = fma(u[a], u[a], qs);
    *quad += qs;
  }
#pragma unroll
  for (int a = 0; a < NZ; ++a) kv[a] = u[a];
  bsub_rcp<NZ>(S, rd, kv);
#pragma unroll
  for (int a = 0; a < NZ; ++a) {
    double col[NZ];
#pragma unroll
    for (int c2 = 0; c2 < NZ; ++c2) col[c2] = X[c2][a];
    bsub_rcp<NZ>(S, rd, col);
#pragma unroll
    for (int c2 = 0; c2 < NZ; ++c2) Kmat[c2 * NZ + a] = col[c2];
  }
#pragma unroll
  for (int a = 0; a < NZ; ++a) {
    double sacc = gt[a];
#pragma unroll
    for (int c2 = 0; c2 < NZ; ++c2) sacc -= X[c2][a] * u[c2];
    rn[a] = sacc;
  }
#pragma unroll
  for (int a = 0; a < NZ; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      double sacc = Ht[tri(a, c2)];
#pragma unroll
      for (int k = 0; k < NZ; ++k) sacc -= X[k][a] * X[k][c2];
      S[tri(a, c2)] = sacc;
    }
  return ok;
}

// Levenberg-Marquardt acceptance test and Nielsen damping update (shared so both paths decide alike).
struct LMState {
  double mu, nun;
};
OH_DEV bool lm_accept(const FigParams& P, const double f, const double feas, const double fc, const double pred, const double stat, LMState& s,
                      const double feas_cur = 0.0) {
  const double rho = (fc - f) / fmax(pred, 1e-300);
  // also accept steps whose predicted decrease is at rounding level of f (end game).  Handles with inequality rows (tol_retract_min <
  // tol_retract): at the level of what the violations of the two points are worth -- (multiplier ~ 10 max(1, |f|)) x (feas + feas_cur); the
  // outer loop asks for stat <= tol again after every multiplier update and the last steps predict 1e-14 while 1e-13 of violation shifts the
  // objective by 1.5e-13: without this the ratio test refuses them all (an instance in 10^4 sat at stat 1.6e-6 until the cap, round 3)
  const double noise = (P.tol_retract_min < P.tol_retract) ? 10.0 * fmax(1.0, fabs(fc)) * (feas + feas_cur) : 0.0;
  const double level = fmax(1e-15 * fabs(fc), noise);
  // a trial point whose retraction did not converge is refused; converged ones may keep up to retract_tol of violation
  const bool accept = (f == f) && (feas <= fmax(P.feas_accept, 10.0 * retract_tol(P, true, pred, stat))) && (rho > 1e-4 || (pred <= level && f <= fc + 1e-14 * fabs(fc) + noise));
  if (accept) {
    // (a step taken at noise level says nothing about the model: the damping stays where it is instead of being multiplied by 1 - (2 rho - 1)^3
    //  of a meaningless, possibly very negative rho)
    const double w3 = (noise > 0.0 && !(rho > 1e-4)) ? 0.0 : 2.0 * rho - 1.0;
    s.mu *= fmax(1.0 / 3.0, 1.0 - w3 * w3 * w3);
    if (s.mu < 1e-7) s.mu = 0.0;
    s.nun = 2.0;
  } else {
    s.mu = fmax(s.mu * s.nun, 1e-3);
    s.nun *= 2.0;
  }
  return accept;
}
