// Block-level bodies of the figure-eight kernels that read the kinematic chain in their inner loops: the two halves of the trial-knot
// evaluation and the persistent tail kernel.  oh_kernels.hip wraps them in __global__ kernels that fetch the chain through the handle's
// device pointer; oh_jit.hip compiles the same text with hiprtc behind a constexpr copy of one handle's chain (OH_CHAIN, see
// oh_figure8_units.h).  Device-only: blockIdx / threadIdx / LDS.
#pragma once
#include "oh_figure8_units.h"

// The same knot in two launches, each at two waves per SIMD (see EVAL_RETRACT_ONLY / EVAL_ONLY in oh_figure8.h): the first leaves the
// retracted trial knot in the slot, the second evaluates it.  Grid: (instance block, knot), instance block fastest.
template <int N>
__device__ void retract_block(const FigParams& P, const FigBuffers& D, const int slot) {
  eval_unit<N, false, false, EVAL_RETRACT_ONLY>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}
// scalars of knot 0 and the multipliers of the quaternion rows (every knot) of the instances that have finished: grid (instance block, knot | 1)
template <int N>
__device__ void finalize_block(const FigParams& P, const FigBuffers& D, const int only_done, double* __restrict__ f, double* __restrict__ kkt, int* __restrict__ iters,
                               int* __restrict__ status) {
  finalize_unit<N>(P, D, only_done, nullptr, f, kkt, iters, status, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
template <int N, bool ZC = false>
__device__ void evalb_block(const FigParams& P, const FigBuffers& D, const int slot) {
  if constexpr (ZC) {
    // no k_couple in the stream any more to reset the counter the sweep adds its survivors to
    if (blockIdx.x == 0 && threadIdx.x == 0) *D.n_running = 0;
    // XCD-aware 1-D grid (k_couple's): workgroup w runs on XCD w % 8, and knot t of an instance block reads the knots t - 1 and t + 1 of the
    // same block.  Consecutive workgroups of one XCD walk the knots of ONE instance block -- w -> (chunk, r), XCD = r % 8 owns instance block
    // chunk * 8 + XCD, knot = r / 8 -- so each knot row comes out of HBM once and is served to its two neighbours by that XCD's L2.
    const int Tn = P.T - P.t0;
    const int w = blockIdx.x;
    const int chunk = w / (8 * Tn), r = w - chunk * 8 * Tn;
    eval_unit<N, false, false, EVAL_ONLY, true>(P, D, slot, (chunk * 8 + (r & 7)) * blockDim.x + threadIdx.x, (r >> 3) + P.t0);
  } else {
    eval_unit<N, false, false, EVAL_ONLY, false>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
  }
}

// ---------------------------------------------------------------------------------------------
// K4 (tail / small batches): ONE WAVEFRONT PER INSTANCE, one lane per free knot (T-2 <= 64), the whole
// remaining SQP loop in a single launch with every stage quantity in registers: no HBM traffic per
// iteration, no launch or host round trip per iteration.  Knot-parallel work (eval_knot, couple_knot) runs
// on all lanes; neighbour data moves with wave shuffles; the Riccati recursion broadcasts knot l's blocks
// with v_readlane and is computed redundantly (wave-uniformly) by all lanes.  Same device functions and
// the same operation order as k_eval/k_couple/k_step.  Entered at a restart point (first == 1): the
// accepted knots are in D.q[slot].
// ---------------------------------------------------------------------------------------------
OH_DEV double bcast(const double v, const int lane) {  // lane must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// VEL (round 3): the same loop for handles whose inequality rows are joint limits and / or joint-velocity limits (oh_guards.limits, vel_limits;
// no sphere rows): the limit rows of knot t and the velocity rows of interval (t-1, t) live on lane t with their multipliers in registers, the
// neighbour's velocity contribution comes over with a shuffle, and the outer loop of the augmented Lagrangian (multiplier refresh, penalty, inner tolerance), the line search along a rejected
// step and the noise-level acceptance are those of step_head / step_instance<N, true> and couple_unit<N, true>, statement for statement.
template <int N, bool VEL = false>
__device__ void tail_block(const FigParams& P, const FigBuffers& D, const int slot, const GuardParams* GPp = nullptr, const GuardBuffers* GBp = nullptr) {
  constexpr int NZ = N - 3;
  constexpr int NP = NZ * (NZ + 1) / 2;
  // Stage data of the accepted point and the blocks the cyclic reduction exchanges live in LDS, one column per lane (= knot): [row][lane], conflict-free for the
  // lane's own column, one broadcast read for another knot's value in the serial sweeps (instead of a v_readlane pair per double).  In
  // registers they cost 190 VGPRs next to the ~350 of the fused evaluation: 512 + 256 registers with 241 spilled to scratch in round 1.
  constexpr int O_DR = 0, O_E = O_DR + NP, O_GT = O_E + NZ * NZ, O_G = O_GT + NZ, O_GF = O_G + N, O_EC = O_GF + N, O_JZ = O_EC + 3, O_PCR = O_JZ + 3 * NZ,
                ROWS = O_PCR + (NZ + 1) * NZ;
  __shared__ double sm[ROWS][64];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  if (D.status[b] >= 0) return;
  const int T = P.T;
  const int nK = T - P.t0;  // free knots, lanes 0..nK-1
  const int t = lane + P.t0;
  const bool active = lane < nK;
  const int tl = active ? t : T - 1;  // clamp addresses of idle lanes
  const bool last = (t == T - 1);
  const double kap2 = 2.0 * P.kappa;
  const oh_chain* ch = OH_CHAIN(D);

  double qt[N], qfix[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    qt[j] = D.q[slot][IDX(tl, N, j)];
    qfix[j] = D.q[slot][IDX(P.t0 - 1, N, j)];
  }
  double Rc[9], pc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = D.ref[(size_t)i * Bp + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = D.ref[(size_t)(3 + i) * Bp + b];
  const double fconst = D.fconst[b];
  LMState lm{D.mu[b], D.nun[b]};
  int iters = D.iters[b];
  bool first = true, polish = false;
  int status = -1;
  unsigned long long n_launch_equiv = 0, n_reject = 0;
  // VEL: multipliers of this lane's interval (t-1, t), outer-loop state of the instance (wave-uniform), line-search state
  double lamv[2 * N], lamq[2 * N];
  double rho_g = 0.0, rho_next = 0.0, omega = 0.0, meas_prev = 0.0, meas_cur = 0.0, fpsi_cur = 0.0, ls_gd = 0.0, ls_q = 0.0, ls_scale = 1.0;
  int outer = 0, n_outer = 0, ls_count = 0;
  double zls[NZ];  // the step as it was solved for (the line search shortens it)
  if constexpr (VEL) {
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) {
      lamv[i] = GPp->vel ? GBp->lamv[IDX(tl, 2 * N, i)] : 0.0;
      lamq[i] = GPp->limits ? GBp->lam[IDX(tl, 2 * N, i)] : 0.0;  // (NC = 2 N: no sphere rows on these handles)
    }
    rho_g = GBp->rho[b]; rho_next = GBp->rho_next[b]; omega = GBp->omega[b]; meas_prev = GBp->meas_prev[b];
    outer = GBp->outer[b]; n_outer = GBp->n_outer[b];
#pragma unroll
    for (int a = 0; a < NZ; ++a) zls[a] = 0.0;
  }

  // accepted point (per lane = per knot)
  double q_c[N], Z_c[N][NZ];
  double e_tgt[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < ROWS; ++i) sm[i][lane] = 0.0;
  double f_cur = 0.0, feas_cur = 0.0, pred = 0.0, stat = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) q_c[k] = qt[k];
  struct TailHooks {  // the Lagrangian gradient of the accepted point is fetched from LDS only inside the exact-curvature branch
    const double (*acc)[64];
    int lane;
    OH_DEV void q_final(const double (&)[N]) const {}
    OH_DEV void g_final(const double (&)[N]) const {}
    OH_DEV void v_final(const double (&)[3][N]) const {}
    OH_DEV void load_G(const double (&)[N], double (&G)[N]) const {
#pragma unroll
      for (int k = 0; k < N; ++k) G[k] = acc[O_GF + k][lane];
    }
  };
  const TailHooks hooks{sm, lane};
  const double Gdummy[N] = {};

  for (;;) {
    ++n_launch_equiv;
    // ---- evaluate the trial knots (k_eval) -----------------------------------------------------------------
    double phi = 0.0, cv = 0.0, g[N], Dr[NP], Z[N][NZ];
    const bool exact = (P.hessian == OH_HESSIAN_EXACT) || (P.hessian == OH_HESSIAN_HYBRID && !first && stat <= P.hyb_switch);
    const bool have_G = exact && !first;
    double e_new[3] = {0.0, 0.0, 0.0}, JZ_new[3][NZ];
    if (active)
      eval_knot<N, false, TailHooks>(ch, P, t, qt, pc, Rc, exact, have_G, Gdummy, phi, cv, g, Dr, Z, !first, e_tgt, retract_tol(P, !first, pred, stat), e_new,
                                     JZ_new, 0.0, hooks);
    double psi_q = 0.0, meas_q = 0.0;
    if constexpr (VEL) {
      if (GPp->limits) {  // joint-limit rows of this knot (eval_unit<N, true>)
        const GuardParams& GP = *GPp;
        const double rho_old = rho_g, rho = outer ? rho_next : rho_g;
        const double irho = 1.0 / rho, i2rho = 1.0 / (2.0 * rho);
        double dd[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          dd[j] = 0.0;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const double gval = side ? GP.up[j] - qt[j] : qt[j] - GP.lo[j];
            double lam = lamq[side * N + j];
            if (outer) {
              lam = fmax(0.0, lam - rho_old * gval);
              lamq[side * N + j] = lam;
            }
            const double sv = lam - rho * gval;
            meas_q = fmax(meas_q, fabs(fmin(gval, lam * irho)));
            if (sv > 0.0) {
              psi_q += (sv * sv - lam * lam) * i2rho;
              g[j] += side ? sv : -sv;
              dd[j] += rho;
            } else {
              psi_q -= lam * lam * i2rho;
            }
          }
        }
#pragma unroll
        for (int a = 0; a < NZ; ++a)
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) acc += dd[j] * Z[j][a] * Z[j][c2];
            Dr[tri(a, c2)] += acc;
          }
        phi += psi_q;
      }
    }
    // ---- neighbour coupling (k_couple) -----------------------------------------------------------------------
    double qm[N], qp[N], Zn[N][NZ];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double up = __shfl_up(qt[k], 1);
      qm[k] = (lane == 0) ? qfix[k] : up;
      qp[k] = __shfl_down(qt[k], 1);
#pragma unroll
      for (int a = 0; a < NZ; ++a) Zn[k][a] = __shfl_down(Z[k][a], 1);
    }
    double G[N], gt[NZ], E[NZ * NZ], merit = 0.0;
    double psi_p = 0.0, meas_p = 0.0, wn[N];
    if constexpr (VEL) {
      psi_p = psi_q;  // (what phase A sums as the augmented-Lagrangian part of the knot, and its complementarity measure)
      meas_p = meas_q;
    }
    if (VEL && GPp->vel) {
      const GuardParams& GP = *GPp;
      double psi_v = 0.0, meas_v = 0.0;
      if (outer) {  // multiplier refresh at the re-evaluated accepted point with the old penalty (vel_update_unit), then the new penalty
        const double rho_old = rho_g * GP.vscale, idt = 1.0 / P.dt;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const double v = (qt[k] - qm[k]) * idt;
          lamv[k] = fmax(0.0, lamv[k] - rho_old * (v - GP.vlo[k]));
          lamv[N + k] = fmax(0.0, lamv[N + k] - rho_old * (GP.vup[k] - v));
        }
      }
      const double rho = (outer ? rho_next : rho_g) * GP.vscale;
      double sp[N], wp[N], sn[N];
      velocity_rows<N>(GP, P.dt, rho, qm, qt, lamv, sp, wp, psi_v, meas_v);
      bool any = false;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const double s_next = __shfl_down(sp[k], 1), w_next = __shfl_down(wp[k], 1);  // interval (t, t+1): the next lane's own
        sn[k] = last ? 0.0 : s_next;
        wn[k] = last ? 0.0 : w_next;
        g[k] += sp[k] - sn[k];
        any = any || wp[k] + wn[k] > 0.0;
      }
      if (any) {
#pragma unroll
        for (int a = 0; a < NZ; ++a)
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) acc += (wp[k] + wn[k]) * Z[k][a] * Z[k][c2];
            Dr[tri(a, c2)] += acc;
          }
      }
      phi += psi_v;
      psi_p += psi_v;
      meas_p = fmax(meas_p, meas_v);
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) wn[k] = 0.0;
    }
    if (active) couple_knot<N>(P.kappa, last, qm, qt, qp, g, Z, Zn, phi, G, gt, E, merit);
    if (VEL && GPp->vel) {
      if (!last) {
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
          for (int a = 0; a < NZ; ++a)
#pragma unroll
            for (int c2 = 0; c2 < NZ; ++c2) E[a * NZ + c2] -= wn[k] * Z[k][a] * Zn[k][c2];
      }
    }
    // ---- phase A: merit in knot order, ratio test (wave-uniform) ---------------------------------------------
    double f = fconst, feas = 0.0, fpsi = 0.0, meas = 0.0;
    for (int l = 0; l < nK; ++l) {
      f += bcast(merit, l);
      feas = fmax(feas, bcast(cv, l));
      if constexpr (VEL) {
        fpsi += bcast(psi_p, l);
        meas = fmax(meas, bcast(meas_p, l));
      }
    }
    bool accept;
    bool line_search = false;
    if (first) {
      if (!(f == f) || !(fabs(f) < 1e300)) {  // non-finite seed / parameters
        status = OH_STATUS_NUMERICAL;
        f_cur = f;
        stat = f;
        break;
      }
      accept = true;
      first = false;
      if constexpr (VEL) {
        if (outer) {  // restart with a multiplier update pending: this evaluation has refreshed the multipliers
          outer = 0;
          rho_g = rho_next;
        }
      }
    } else if (VEL && outer) {
      accept = true;  // re-evaluation of the accepted point after a multiplier update: the merit function itself changed
      outer = 0;
      rho_g = rho_next;
    } else if (polish) {
      accept = true;
      polish = false;
    } else {
      const LMState lm_before = lm;
      accept = lm_accept(P, f, feas, f_cur, pred, stat, lm, feas_cur);
      if (!accept) ++n_reject;
      const bool polish_request = !accept && feas_cur > 10.0 * retract_tol(P, false, pred, 0.0);
      if constexpr (VEL) {
        if (!accept && !polish_request && ls_count < OH_LS_MAX && iters < P.max_iter / 2) {  // a shorter step along the same direction first
          lm = lm_before;
          line_search = true;
        }
      }
      if (polish_request) {  // see step_instance: re-retract the accepted point before blaming the model
        lm = lm_before;
        polish = true;
        pred = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) qt[j] = q_c[j];
#pragma unroll
        for (int m = 0; m < 3; ++m) e_tgt[m] = sm[O_EC + m][lane];
        ++iters;
        if (iters >= P.max_iter + 40) { status = OH_STATUS_MAX_ITER; break; }
        continue;
      }
    }
    if constexpr (VEL) {
      if (line_search) {
        if (iters >= P.max_iter) { status = OH_STATUS_MAX_ITER; break; }
        ++ls_count;
        ls_scale *= OH_LS_SHRINK;
        double sk = 1.0;
        for (int i = 0; i < ls_count; ++i) sk *= OH_LS_SHRINK;
        pred = -sk * ls_gd + 0.5 * sk * sk * ls_q;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          double v = q_c[j];
#pragma unroll
          for (int a = 0; a < NZ; ++a) v += Z_c[j][a] * (ls_scale * zls[a]);
          qt[j] = v;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          double v = sm[O_EC + m][lane];
#pragma unroll
          for (int a = 0; a < NZ; ++a) v += sm[O_JZ + m * NZ + a][lane] * (ls_scale * zls[a]);
          e_tgt[m] = v;
        }
        ++iters;
        continue;
      }
    }
    if (accept) {
      f_cur = f;
      feas_cur = feas;
      if constexpr (VEL) {
        fpsi_cur = fpsi;
        meas_cur = meas;
        ls_count = 0;
      }
#pragma unroll
      for (int k = 0; k < N; ++k) {
        q_c[k] = qt[k];
        sm[O_G + k][lane] = g[k];
        sm[O_GF + k][lane] = G[k];
#pragma unroll
        for (int a = 0; a < NZ; ++a) Z_c[k][a] = Z[k][a];
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) sm[O_DR + i][lane] = Dr[i];
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) sm[O_E + i][lane] = E[i];
#pragma unroll
      for (int a = 0; a < NZ; ++a) sm[O_GT + a][lane] = gt[a];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        sm[O_EC + m][lane] = e_new[m];
#pragma unroll
        for (int a = 0; a < NZ; ++a) sm[O_JZ + m * NZ + a][lane] = JZ_new[m][a];
      }
    }
    double mu = lm.mu;
    // ---- phase B: the reduced block-tridiagonal system  D_l z_l + E_l z_{l+1} + E_{l-1}^T z_{l-1} = -gt_l  by block PARALLEL CYCLIC
    // REDUCTION across the lanes (lane = knot): at stride s every equation eliminates its neighbours l -+ s using their own rows,
    //   A_l <- A_l - L_l A_{l-s}^{-1} U_{l-s} - U_l A_{l+s}^{-1} L_{l+s},  L_l <- -L_l A_{l-s}^{-1} L_{l-s},  U_l <- -U_l A_{l+s}^{-1} U_{l+s},
    //   r_l <- r_l - L_l A_{l-s}^{-1} r_{l-s} - U_l A_{l+s}^{-1} r_{l+s},
    // and after ceil(log2 nK) strides z_l = A_l^{-1} r_l.  Six dependent NZ x NZ factorisations per lane instead of the nK - 1 = 47 of the
    // serial Riccati sweep the batched k_step runs (the diagonal blocks stay Schur complements of a positive definite matrix, so the
    // plain Cholesky is stable; a failed pivot on any lane raises the damping for the whole instance as before).  Neighbour blocks
    // travel through LDS.
    stat = 0.0;
    if (active) {
#pragma unroll
      for (int a = 0; a < NZ; ++a) stat = fmax(stat, fabs(sm[O_GT + a][lane]));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) stat = fmax(stat, __shfl_xor(stat, m));
    double zmine[NZ];
    for (int attempt = 0; attempt < 40; ++attempt) {
      double A[NZ * NZ], Lw[NZ * NZ], U[NZ * NZ], r[NZ];
#pragma unroll
      for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
          const int hi = i > j ? i : j, lo = i > j ? j : i;
          A[i * NZ + j] = active ? sm[O_DR + tri(hi, lo)][lane] + (i == j ? ((last ? kap2 : 2.0 * kap2) + mu) : 0.0) : (i == j ? 1.0 : 0.0);
          U[i * NZ + j] = (active && !last) ? sm[O_E + i * NZ + j][lane] : 0.0;
          Lw[i * NZ + j] = (active && lane > 0) ? sm[O_E + j * NZ + i][lane - 1] : 0.0;
        }
#pragma unroll
      for (int a = 0; a < NZ; ++a) r[a] = active ? -sm[O_GT + a][lane] : 0.0;
      bool ok = true;
      double Lc[NP], rd[NZ];
      for (int sft = 1; sft < nK; sft <<= 1) {
#pragma unroll
        for (int i = 0; i < NZ; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) Lc[tri(i, j)] = A[i * NZ + j];
        ok = chol_rcp<NZ>(Lc, rd, 1e-12) && ok;
        // Y = A^{-1} [Lw | r], then A^{-1} U, column by column, parked in LDS for the neighbours -- in two passes through one (NZ + 1) x NZ
        // tile: with all 2 NZ + 1 columns parked at once the block needs 48.6 KB of LDS and three of them fit a CU; at 40.4 KB four do
        // (one wavefront of 512 registers per SIMD), a third more instances in flight when the kernel takes whole batches
        const int lm_ = lane - sft, lp_ = lane + sft;
        const bool hm = lm_ >= 0, hp = lp_ < 64;
        const int im = hm ? lm_ : lane, ip = hp ? lp_ : lane;
        double An[NZ * NZ], Ln[NZ * NZ], Un[NZ * NZ], rn2[NZ];
#pragma unroll
        for (int c2 = 0; c2 <= NZ; ++c2) {
          double col[NZ];
#pragma unroll
          for (int i = 0; i < NZ; ++i) col[i] = c2 < NZ ? Lw[i * NZ + c2] : r[i];
          fsub_rcp<NZ>(Lc, rd, col);
          bsub_rcp<NZ>(Lc, rd, col);
#pragma unroll
          for (int i = 0; i < NZ; ++i) sm[O_PCR + c2 * NZ + i][lane] = col[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
          double racc = r[i];
#pragma unroll
          for (int j = 0; j < NZ; ++j) {
            double aacc = A[i * NZ + j], lacc = 0.0;
#pragma unroll
            for (int k = 0; k < NZ; ++k) {
              aacc -= U[i * NZ + k] * sm[O_PCR + j * NZ + k][ip];   // A - U Y^L_{+}
              lacc -= Lw[i * NZ + k] * sm[O_PCR + j * NZ + k][im];  // -Lw Y^L_{-}
            }
            An[i * NZ + j] = aacc;
            Ln[i * NZ + j] = lacc;
          }
#pragma unroll
          for (int k = 0; k < NZ; ++k) racc -= Lw[i * NZ + k] * sm[O_PCR + NZ * NZ + k][im] + U[i * NZ + k] * sm[O_PCR + NZ * NZ + k][ip];
          rn2[i] = racc;
        }
        __syncthreads();
#pragma unroll
        for (int c2 = 0; c2 < NZ; ++c2) {  // (Lc is still the factor of the block as it stood before this level)
          double col[NZ];
#pragma unroll
          for (int i = 0; i < NZ; ++i) col[i] = U[i * NZ + c2];
          fsub_rcp<NZ>(Lc, rd, col);
          bsub_rcp<NZ>(Lc, rd, col);
#pragma unroll
          for (int i = 0; i < NZ; ++i) sm[O_PCR + c2 * NZ + i][lane] = col[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NZ; ++i)
#pragma unroll
          for (int j = 0; j < NZ; ++j) {
            double aacc = An[i * NZ + j], uacc = 0.0;
#pragma unroll
            for (int k = 0; k < NZ; ++k) {
              aacc -= Lw[i * NZ + k] * sm[O_PCR + j * NZ + k][im];  // A - Lw Y^U_{-}
              uacc -= U[i * NZ + k] * sm[O_PCR + j * NZ + k][ip];   // -U Y^U_{+}
            }
            An[i * NZ + j] = aacc;
            Un[i * NZ + j] = uacc;
          }
        __syncthreads();
        // Lw / U of a lane without that neighbour are zero, so the clamped reads above contributed nothing
#pragma unroll
        for (int i = 0; i < NZ * NZ; ++i) {
          A[i] = An[i];
          Lw[i] = hm ? Ln[i] : 0.0;
          U[i] = hp ? Un[i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) r[i] = rn2[i];
      }
#pragma unroll
      for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Lc[tri(i, j)] = 0.5 * (A[i * NZ + j] + A[j * NZ + i]);
      ok = chol_rcp<NZ>(Lc, rd, 1e-12) && ok;
#pragma unroll
      for (int a = 0; a < NZ; ++a) zmine[a] = r[a];
      fsub_rcp<NZ>(Lc, rd, zmine);
      bsub_rcp<NZ>(Lc, rd, zmine);
      if (__all(ok || !active)) break;
      mu = fmax(4.0 * mu, 1e-2);
    }
    lm.mu = mu;
    if constexpr (VEL) {
      if (!(stat == stat)) { status = OH_STATUS_NUMERICAL; break; }
      if (stat <= omega) {
        if (stat <= P.tol && feas_cur <= P.tol_feas && meas_cur <= P.tol_feas) { status = OH_STATUS_CONVERGED; break; }
        if (iters >= P.max_iter) { status = OH_STATUS_MAX_ITER; break; }
        // outer iteration: stay where we are, let the next evaluation refresh the multipliers, tighten the inner tolerance
        rho_next = (meas_cur > 0.25 * meas_prev) ? fmin(10.0 * rho_g, 1e8) : rho_g;
        meas_prev = meas_cur;
        omega = fmax(P.tol, fmin(omega, 0.1 * meas_cur));
        outer = 1;
        ++n_outer;
        pred = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) qt[j] = q_c[j];
#pragma unroll
        for (int m = 0; m < 3; ++m) e_tgt[m] = sm[O_EC + m][lane];
        ++iters;
        continue;
      }
      if (iters >= P.max_iter) { status = OH_STATUS_MAX_ITER; break; }
    } else {
    if (stat <= P.tol && feas_cur <= P.tol_feas) { status = OH_STATUS_CONVERGED; break; }
    if (iters >= P.max_iter) { status = OH_STATUS_MAX_ITER; break; }
    if (!(stat == stat)) { status = OH_STATUS_NUMERICAL; break; }
    }
    {
      double gd = 0.0, z2 = 0.0;
      if (active) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
          gd += sm[O_GT + a][lane] * zmine[a];
          z2 += zmine[a] * zmine[a];
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        gd += __shfl_xor(gd, m);
        z2 += __shfl_xor(z2, m);
      }
      const double alpha = (!VEL && P.hessian == OH_HESSIAN_HYBRID && stat > P.hyb_switch && iters >= P.relax_from) ? P.relax : 1.0;  // see step_instance
#pragma unroll
      for (int a = 0; a < NZ; ++a) zmine[a] *= alpha;
      pred = -alpha * gd + 0.5 * alpha * alpha * (gd + mu * z2);
      if constexpr (VEL) {  // for the line search along this step, should it be rejected
        ls_gd = alpha * gd;
        ls_q = alpha * alpha * (gd + mu * z2);
        ls_scale = 1.0;
#pragma unroll
        for (int a = 0; a < NZ; ++a) zls[a] = zmine[a];
      }
    }
    // next trial knots: q_cur + Z_cur z
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double v = q_c[j];
#pragma unroll
      for (int a = 0; a < NZ; ++a) v += Z_c[j][a] * zmine[a];
      qt[j] = v;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double v = sm[O_EC + m][lane];
#pragma unroll
      for (int a = 0; a < NZ; ++a) v += sm[O_JZ + m * NZ + a][lane] * zmine[a];
      e_tgt[m] = v;
    }
    ++iters;
  }

  // ---- hand the result to k_finalize ---------------------------------------------------------------------------
  if (active) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      D.q[slot][IDX(t, N, j)] = q_c[j];
      D.g[slot][IDX(t, N, j)] = sm[O_G + j][lane];
      if (P.zc) D.Gfull[slot][IDX(t, N, j)] = sm[O_GF + j][lane];  // where such a handle's k_finalize looks for the Lagrangian gradient
    }
  }
  if (lane == 0) {
    D.cur[b] = slot;
    D.f_cur[b] = f_cur;
    D.feas[b] = feas_cur;
    D.stat[b] = stat;
    D.mu[b] = lm.mu;
    D.nun[b] = lm.nun;
    D.iters[b] = iters;
    D.first[b] = 0;
    D.status[b] = status;
    atomicAdd(D.work + 2, n_launch_equiv);  // tail iterations are accounted separately from the batched launches
    if (n_reject) atomicAdd(D.work + 1, n_reject);
    if constexpr (VEL) {
      GBp->rho[b] = rho_g; GBp->rho_next[b] = rho_next; GBp->omega[b] = omega; GBp->meas_prev[b] = meas_prev;
      GBp->outer[b] = outer; GBp->n_outer[b] = n_outer; GBp->meas[b] = meas_cur; GBp->ls_count[b] = 0;
      D.fpsi[b] = fpsi_cur;
    }
  }
  if constexpr (VEL) {
    if (active) {
#pragma unroll
      for (int i = 0; i < 2 * N; ++i) {
        if (GPp->vel) GBp->lamv[IDX(t, 2 * N, i)] = lamv[i];
        if (GPp->limits) GBp->lam[IDX(t, 2 * N, i)] = lamq[i];
      }
    }
  }
}
