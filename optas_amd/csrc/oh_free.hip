// Position-only end-effector tracking (lock_orientation = 0): example/dual_arm.py as shipped (each arm),
// SURVEY App. B.4.  Same state machine, buffers and launch sequence as the orientation-locked kernels in
// oh_kernels.hip, but there are no constraint rows: the null-space basis is the identity (NZ = N), the
// reduced Hessian block is W_t itself and every coupling block is E_t = -2 kappa I, so the Riccati recursion
// simplifies to   S_t = H_t - (2k)^2 S_{t+1}^{-1},  r_t = g_t + 2k S_{t+1}^{-1} r_{t+1},
//                 z_{t+1} = -S_{t+1}^{-1} r_{t+1} + 2k S_{t+1}^{-1} z_t.
#include "oh_figure8.h"
#include "oh_kernels.h"  // OhLaunchOpts

#define IDX(t, K, k) (((size_t)(t) * (K) + (k)) * Bp + b)
// Stage blocks W_t (packed, NP) and reduced gradients (N) of this family: knot-major like everything else (a thread per instance reads coalesced), or
// instance-major [b][t][.] when every launch of the solve gives an instance a block of its own (P.inst_major, round 4: k_step_free_bb's lanes read one
// instance; knot-major that was 64 cache lines per load instruction and 27 of its 82 us).
#define DRX(t, i) (P.inst_major ? (((size_t)b * P.T + (t)) * NP + (i)) : IDX(t, NP, i))
#define GTX(t, k) (P.inst_major ? (((size_t)b * P.T + (t)) * N + (k)) : IDX(t, N, k))

// Trial slot of an instance (round 3): the slot its accepted point is NOT in.  The orientation-locked kernels alternate one slot for the whole
// batch, so an instance whose trial was rejected has to sit the next launch out (its accepted point lies where that launch writes); here every
// instance flips for itself, and a rejection costs no idle launch -- on config 4 (256 arms, 88 launches of ~147 us, every one of them pure
// latency) 26 of the 88 were such idle launches of the slowest arm.  The `slot` the host passes only says where a restart compaction laid
// the knots down, and that is where D.cur points away from (k_compact_scatter).
#define OH_FREE_SLOT(D, b) (1 - (D).cur[b])
// Inner tolerance of the augmented-Lagrangian outer loop of this family: the reduced gradient the next inner solve has to reach is
// OH_AL_OMEGA_FREE x the complementarity measure of the point where the multipliers were last refreshed.  0.1 until the end of round 3 (and still
// on the orientation-locked handles, whose tails get longer with it); with 1.0 the inner solves stop sooner and the multipliers move more often:
// config 4 synthetic 256 arms 10.3 -> 8.9 ms, 1024 arms 20.6 -> 18.5 ms (numpy port, 16 arms: mean 37.5 -> 33.6 steps).
#ifndef OH_AL_OMEGA_FREE
#define OH_AL_OMEGA_FREE 1.0
#endif
// (a slot chosen per lane: by selection -- an index into the argument struct would make the compiler keep a private copy of it)
#define SEL(arr, s) ((s) ? (arr)[1] : (arr)[0])

// The neighbour coupling folded into the evaluation (round 3; handles without velocity rows): the trial knots of t - 1 and t + 1 are the
// accepted knots plus the step -- known before their own lanes have evaluated anything -- so the launch of k_couple_free (7 us and a launch gap
// per iteration of a latency-bound solve) goes away.  Same operations in the same order as k_couple_free.
template <int N>
OH_DEV void couple_inline_free(const FigParams& P, const FigBuffers& D, const int b, const int t, const int slot, const bool first, const double (&q)[N],
                               const double (&g)[N], const double phi) {
  const int Bp = D.Bp;
  const double kap2 = 2.0 * P.kappa;
  const bool last = (t == P.T - 1);
  const double* __restrict__ qa = first ? SEL(D.q, slot) : SEL(D.q, 1 - slot);  // knots of a restart / the seed, or the accepted point
  double sm = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double qm = qa[IDX(t - 1, N, k)], qp = last ? 0.0 : qa[IDX(t + 1, N, k)];
    if (!first) {
      qm += D.zstep[IDX(t - 1, N, k)];
      if (!last) qp += D.zstep[IDX(t + 1, N, k)];
    }
    const double dm = q[k] - qm;
    sm += dm * dm;
    double G = g[k] + kap2 * dm;
    if (!last) G -= kap2 * (qp - q[k]);
    SEL(D.gt, slot)[GTX(t, k)] = G;
  }
  SEL(D.merit, slot)[(size_t)t * Bp + b] = phi + P.kappa * sm;
}

// one knot, no retraction / null space: tracking cost, gradient, Gauss-Newton (or exact) block W (packed lower)
template <int N>
OH_DEV void eval_knot_free(const oh_chain* __restrict__ ch, const FigParams& P, const int t, const double (&q)[N], const double (&pc)[3],
                           const double (&Rc)[9], double& phi, double (&g)[N], double (&W)[N * (N + 1) / 2]) {
  double R[9], p[3], z[N][3], pj[N][3];
  fk_chain<N>(ch, q, R, p, z, pj);
  double e[3], tv[3];
  mv3(R, ch->p_tool, tv);
  e[0] = p[0] + tv[0]; e[1] = p[1] + tv[1]; e[2] = p[2] + tv[2];
  const double l[3] = {P.local_path[3 * t], P.local_path[3 * t + 1], P.local_path[3 * t + 2]};
  double r[3];
  if (P.path_in_frame) mv3(Rc, l, r);
  else { r[0] = l[0]; r[1] = l[1]; r[2] = l[2]; }
  r[0] += pc[0] - e[0]; r[1] += pc[1] - e[1]; r[2] += pc[2] - e[2];
  const double w = P.w_path;
  phi = w * dot3(r, r);
  double Jp[N][3];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (ch->jtype[k] == 0) {
      const double d[3] = {e[0] - pj[k][0], e[1] - pj[k][1], e[2] - pj[k][2]};
      cross3(z[k], d, Jp[k]);
    } else {
      Jp[k][0] = z[k][0]; Jp[k][1] = z[k][1]; Jp[k][2] = z[k][2];
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) g[k] = -2.0 * w * dot3(Jp[k], r);
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) W[tri(i, j)] = 2.0 * w * dot3(Jp[i], Jp[j]);
  if (P.hessian == OH_HESSIAN_EXACT) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (ch->jtype[j] == 0) {
        double rz[3];
        cross3(r, z[j], rz);
#pragma unroll
        for (int i = j; i < N; ++i) W[tri(i, j)] += -2.0 * w * dot3(rz, Jp[i]);
      }
    }
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_eval_free(FigParams P, FigBuffers D, const int slot_batch) {
  constexpr int NP = N * (N + 1) / 2;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y + P.t0;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);
  if (D.status[b] >= 0 || D.skip[b]) return;
  const int cur = 1 - slot;
  double q[N];
  if (D.first[b]) {
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = SEL(D.q, slot)[IDX(t, N, j)];
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = SEL(D.q, cur)[IDX(t, N, j)] + D.zstep[IDX(t, N, j)];
  }
  double Rc[9], pc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = D.ref[(size_t)i * Bp + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = D.ref[(size_t)(3 + i) * Bp + b];
  double phi, g[N], W[NP];
  eval_knot_free<N>(D.chain, P, t, q, pc, Rc, phi, g, W);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    SEL(D.q, slot)[IDX(t, N, j)] = q[j];
    SEL(D.g, slot)[IDX(t, N, j)] = g[j];
  }
  SEL(D.phi, slot)[(size_t)t * Bp + b] = phi;
  SEL(D.cv, slot)[(size_t)t * Bp + b] = 0.0;
  if (P.zc_free) couple_inline_free<N>(P, D, b, t, slot, D.first[b] != 0, q, g, phi);
#pragma unroll
  for (int i = 0; i < NP; ++i) SEL(D.Dr, slot)[DRX(t, i)] = W[i];
}


// ---------------------------------------------------------------------------------------------------------------
// Inequality rows (oh_guards): joint limits and sphere clearances enter the stage cost through the
// Powell-Hestenes-Rockafellar augmented Lagrangian  psi(g, lam, rho) = (max(0, lam - rho g)^2 - lam^2) / (2 rho),
// Gauss-Newton curvature rho dg dg^T on the rows with lam - rho g > 0.  The sphere rows of link l are processed
// inside the kinematics walk, right after the joint the link hangs on: its centre c_l and the columns
// z_j x (c_l - p_j), j <= joint(l), of its position Jacobian only need the frames already visited.
// (oracle restatement: oracle/guarded.py)
// ---------------------------------------------------------------------------------------------------------------
template <int N>
OH_DEV void guard_row(const double gval, const double (&dg)[N], const double rho, const double rho_old, const bool upd, double* __restrict__ lam_ptr,
                      double& psi, double& meas, double (&g)[N], double (&W)[N * (N + 1) / 2]) {
  double lam = *lam_ptr;
  if (upd) {
    lam = fmax(0.0, lam - rho_old * gval);
    *lam_ptr = lam;
  }
  const double s = lam - rho * gval;
  meas = fmax(meas, fabs(fmin(gval, lam / rho)));
  if (s > 0.0) {
    psi += (s * s - lam * lam) / (2.0 * rho);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      g[i] -= s * dg[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) W[tri(i, j)] += rho * dg[i] * dg[j];
    }
  } else {
    psi -= lam * lam / (2.0 * rho);
  }
}

template <int N>
OH_DEV void eval_guarded_knot(const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, const int b, const int t) {
  constexpr int NP = N * (N + 1) / 2;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);
  if (D.status[b] >= 0 || D.skip[b]) return;
  const int cur = 1 - slot;
  double q[N];
  if (D.first[b]) {
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = SEL(D.q, slot)[IDX(t, N, j)];
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = SEL(D.q, cur)[IDX(t, N, j)] + D.zstep[IDX(t, N, j)];
  }
  const bool upd = GB.outer[b] != 0;
  const double rho_old = GB.rho[b];
  const double rho = upd ? GB.rho_next[b] : rho_old;
  const oh_chain* __restrict__ ch = D.chain;
  const int NC = GP.NC;
  const int nl = GP.limits ? 2 * N : 0;
  // requested before the kinematics walk so that the walk hides them: the first eight obstacles (the same for every link); further obstacles and the
  // multipliers are fetched chunk by chunk inside the walk
  double ox0[8], oy0[8], oz0[8], or0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int o = u < GP.n_obs ? u : 0;
    const size_t ob = (size_t)(GP.n_links + 4 * o) * Bp + b;
    ox0[u] = GB.par[ob];
    oy0[u] = GB.par[ob + Bp];
    oz0[u] = GB.par[ob + 2 * (size_t)Bp];
    or0[u] = GB.par[ob + 3 * (size_t)Bp];
  }
  double g[N], W[NP], psi = 0.0, meas = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) g[i] = 0.0;
#pragma unroll
  for (int i = 0; i < NP; ++i) W[i] = 0.0;
  // ---- kinematics walk (same arithmetic as fk_chain) with the sphere rows hooked in ----
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
      const double rl = GB.par[(size_t)l * Bp + b];
      // Obstacles in chunks of eight: multipliers and obstacle parameters of a chunk are requested together, then the rows are formed, then the
      // refreshed multipliers are stored.  (Row by row -- load the multiplier, use it, perhaps store it -- every row waited for its own load: the store
      // may alias the next load as far as the compiler can tell.  38 rows x a memory latency were most of the kernel's 40 us.)
      for (int o0 = 0; o0 < GP.n_obs; o0 += 8) {
        double lm8[8], ox[8], oy[8], oz[8], orad[8];
        if (o0 == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            ox[u] = ox0[u]; oy[u] = oy0[u]; oz[u] = oz0[u]; orad[u] = or0[u];
            lm8[u] = GB.lam[IDX(t, NC, nl + l * GP.n_obs + (u < GP.n_obs ? u : 0))];
          }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int o = (o0 + u < GP.n_obs) ? o0 + u : o0;
            const size_t ob = (size_t)(GP.n_links + 4 * o) * Bp + b;
            lm8[u] = GB.lam[IDX(t, NC, nl + l * GP.n_obs + o)];
            ox[u] = GB.par[ob];
            oy[u] = GB.par[ob + Bp];
            oz[u] = GB.par[ob + 2 * (size_t)Bp];
            orad[u] = GB.par[ob + 3 * (size_t)Bp];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (o0 + u < GP.n_obs) {
            const double d[3] = {c[0] - ox[u], c[1] - oy[u], c[2] - oz[u]};
            const double rr = rl + orad[u];
            const double gval = dot3(d, d) - rr * rr;
            double dg[N];
#pragma unroll
            for (int j = 0; j < N; ++j) dg[j] = 2.0 * dot3(Jl[j], d);
            guard_row<N>(gval, dg, rho, rho_old, upd, &lm8[u], psi, meas, g, W);
          }
        }
        if (upd) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (o0 + u < GP.n_obs) GB.lam[IDX(t, NC, nl + l * GP.n_obs + o0 + u)] = lm8[u];
        }
      }
    }
  }
  if (GP.limits) {
    double lml[2 * N];
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) lml[i] = GB.lam[IDX(t, NC, i)];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      // rows q_j - lo_j and up_j - q_j: gradients +e_j / -e_j
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const double gval = side ? GP.up[j] - q[j] : q[j] - GP.lo[j];
        double lam = lml[side * N + j];
        if (upd) {
          lam = fmax(0.0, lam - rho_old * gval);
          lml[side * N + j] = lam;
        }
        const double s = lam - rho * gval;
        meas = fmax(meas, fabs(fmin(gval, lam / rho)));
        if (s > 0.0) {
          psi += (s * s - lam * lam) / (2.0 * rho);
          g[j] += side ? s : -s;
          W[tri(j, j)] += rho;
        } else {
          psi -= lam * lam / (2.0 * rho);
        }
      }
    }
    if (upd) {
#pragma unroll
      for (int i = 0; i < 2 * N; ++i) GB.lam[IDX(t, NC, i)] = lml[i];
    }
  }
  // ---- tracking terms (as eval_knot_free, Gauss-Newton) ----
  double e[3], tv[3];
  mv3(R, ch->p_tool, tv);
  e[0] = p[0] + tv[0]; e[1] = p[1] + tv[1]; e[2] = p[2] + tv[2];
  const double lp[3] = {P.local_path[3 * t], P.local_path[3 * t + 1], P.local_path[3 * t + 2]};
  double r[3];
  if (P.path_in_frame) {
    double Rc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = D.ref[(size_t)(3 + i) * Bp + b];
    mv3(Rc, lp, r);
  } else {
    r[0] = lp[0]; r[1] = lp[1]; r[2] = lp[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) r[i] += D.ref[(size_t)i * Bp + b] - e[i];
  const double w = P.w_path;
  double Jp[N][3];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (ch->jtype[k] == 0) {
      const double dd[3] = {e[0] - pj[k][0], e[1] - pj[k][1], e[2] - pj[k][2]};
      cross3(z[k], dd, Jp[k]);
    } else {
      Jp[k][0] = z[k][0]; Jp[k][1] = z[k][1]; Jp[k][2] = z[k][2];
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    g[i] += -2.0 * w * dot3(Jp[i], r);
#pragma unroll
    for (int j = 0; j <= i; ++j) W[tri(i, j)] += 2.0 * w * dot3(Jp[i], Jp[j]);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    SEL(D.q, slot)[IDX(t, N, j)] = q[j];
    SEL(D.g, slot)[IDX(t, N, j)] = g[j];
  }
  SEL(D.phi, slot)[(size_t)t * Bp + b] = w * dot3(r, r) + psi;
  if (P.zc_free) couple_inline_free<N>(P, D, b, t, slot, D.first[b] != 0, q, g, w * dot3(r, r) + psi);
  SEL(GB.psi, slot)[(size_t)t * Bp + b] = psi;
  SEL(D.cv, slot)[(size_t)t * Bp + b] = meas;
#pragma unroll
  for (int i = 0; i < NP; ++i) SEL(D.Dr, slot)[DRX(t, i)] = W[i];
}

template <int N>
__global__ __launch_bounds__(256) void k_eval_guarded(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot_batch) {
  eval_guarded_knot<N>(P, D, GP, GB, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}

// guard parameters of every instance into SoA, multipliers and outer-loop state reset
__global__ __launch_bounds__(64) void k_setup_guards(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const double* __restrict__ pin,
                                                     const int N) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int npar = GP.n_links + 4 * GP.n_obs;
  for (int i = 0; i < npar; ++i) GB.par[(size_t)i * Bp + b] = pin[(size_t)b * P.np + N + i];
  for (int t = 0; t < P.T; ++t)
    for (int i = 0; i < GP.NC; ++i) GB.lam[((size_t)t * GP.NC + i) * Bp + b] = 0.0;
  if (GP.vel)
    for (int t = 0; t < P.T; ++t)
      for (int i = 0; i < 2 * N; ++i) GB.lamv[((size_t)t * 2 * N + i) * Bp + b] = 0.0;
  GB.rho[b] = GP.rho0;
  GB.rho_next[b] = GP.rho0;
  GB.omega[b] = fmax(P.tol, 1e-2);
  GB.meas_prev[b] = 1e300;
  GB.outer[b] = 0;
  GB.n_outer[b] = 0;
  D.fpsi[b] = 0.0;
  if (GB.meas) GB.meas[b] = 0.0;
}

template <int N>
__global__ __launch_bounds__(256) void k_couple_free(FigParams P, FigBuffers D, const int slot_batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y + P.t0;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);
  if (D.status[b] >= 0 || D.skip[b]) return;
  const double* __restrict__ qs = SEL(D.q, slot);
  const double kap2 = 2.0 * P.kappa;
  const bool last = (t == P.T - 1);
  double sm = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double qm = qs[IDX(t - 1, N, k)];
    const double q0 = qs[IDX(t, N, k)];
    const double dm = q0 - qm;
    sm += dm * dm;
    double G = SEL(D.g, slot)[IDX(t, N, k)] + kap2 * dm;
    if (!last) G -= kap2 * (qs[IDX(t + 1, N, k)] - q0);
    SEL(D.gt, slot)[GTX(t, k)] = G;
  }
  SEL(D.merit, slot)[(size_t)t * Bp + b] = SEL(D.phi, slot)[(size_t)t * Bp + b] + P.kappa * sm;
}

// Joint-velocity rows (oh_guards.vel_limits; enforce_model_limits(name, time_deriv=1), builder.py:471-509) on dq_t = (q_{t+1} - q_t) / dt, round 3.
// They couple neighbouring knots exactly like the velocity cost: interval (t-1, t) is booked on knot t, its augmented-Lagrangian gradient
// enters both knots, its Gauss-Newton weight w = rho_v (active rows) / dt^2 joins 2 kappa on the diagonal of both knots and in the coupling
// block between them, which stays diagonal: E_t = -diag(2 kappa + w_t) -- the sweeps below carry that vector (SEL(D.E, slot), N rows per knot)
// instead of the scalar.  (oracle restatement: oracle/guarded.py:solve_free_al(vlimits=...); locked family: couple_unit<N, true>)
template <int N>
__global__ __launch_bounds__(256) void k_vel_update_free(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot_batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y + P.t0;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);
  if (D.status[b] >= 0 || D.skip[b] || !GB.outer[b]) return;
  // multiplier refresh at an outer update, at the re-evaluated accepted point with the old penalty (a launch of its own: no lane reads a
  // neighbour's row block while it is rewritten)
  const double rho = GB.rho[b] * GP.vscale, idt = 1.0 / P.dt;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double v = (SEL(D.q, slot)[IDX(t, N, k)] - SEL(D.q, slot)[IDX(t - 1, N, k)]) * idt;
    double* l_lo = GB.lamv + IDX(t, 2 * N, k);
    double* l_up = GB.lamv + IDX(t, 2 * N, N + k);
    *l_lo = fmax(0.0, *l_lo - rho * (v - GP.vlo[k]));
    *l_up = fmax(0.0, *l_up - rho * (GP.vup[k] - v));
  }
}
template <int N>
__global__ __launch_bounds__(256) void k_couple_free_vel(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot_batch) {
  constexpr int NP = N * (N + 1) / 2;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y + P.t0;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);
  if (D.status[b] >= 0 || D.skip[b]) return;
  const double* __restrict__ qs = SEL(D.q, slot);
  const double kap2 = 2.0 * P.kappa;
  const bool last = (t == P.T - 1);
  double qm[N], q0[N], qp[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    qm[k] = qs[IDX(t - 1, N, k)];
    q0[k] = qs[IDX(t, N, k)];
    qp[k] = last ? 0.0 : qs[IDX(t + 1, N, k)];
  }
  const double rho = (GB.outer[b] ? GB.rho_next[b] : GB.rho[b]) * GP.vscale;
  double lam[2 * N], sp[N], wp[N], sn[N], wn[N], psi_p, meas_p, psi_n, meas_n;
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) lam[i] = GB.lamv[IDX(t, 2 * N, i)];
  velocity_rows<N>(GP, P.dt, rho, qm, q0, lam, sp, wp, psi_p, meas_p);
  if (!last) {
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) lam[i] = GB.lamv[IDX(t + 1, 2 * N, i)];
    velocity_rows<N>(GP, P.dt, rho, q0, qp, lam, sn, wn, psi_n, meas_n);
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) sn[k] = wn[k] = 0.0;
  }
  double sm = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double dm = q0[k] - qm[k];
    sm += dm * dm;
    double G = SEL(D.g, slot)[IDX(t, N, k)] + kap2 * dm + sp[k] - sn[k];
    if (!last) G -= kap2 * (qp[k] - q0[k]);
    SEL(D.gt, slot)[GTX(t, k)] = G;
    if (wp[k] + wn[k] > 0.0) SEL(D.Dr, slot)[DRX(t, tri(k, k))] += wp[k] + wn[k];
    SEL(D.E, slot)[IDX(t, N, k)] = kap2 + wn[k];  // the coupling of knots t and t+1, as the sweeps use it
  }
  SEL(D.merit, slot)[(size_t)t * Bp + b] = SEL(D.phi, slot)[(size_t)t * Bp + b] + psi_p + P.kappa * sm;
  SEL(GB.psi, slot)[(size_t)t * Bp + b] += psi_p;
  SEL(D.cv, slot)[(size_t)t * Bp + b] = fmax(SEL(D.cv, slot)[(size_t)t * Bp + b], meas_p);
}

// S^{-1} (packed lower) from the packed Cholesky factor L and reciprocal pivots: Li = L^{-1}, Sinv = Li^T Li
template <int M>
OH_DEV void spd_inverse(const double (&L)[M * (M + 1) / 2], const double (&rd)[M], double (&Sinv)[M * (M + 1) / 2]) {
  double Li[M * (M + 1) / 2];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    Li[tri(j, j)] = rd[j];
#pragma unroll
    for (int i = j + 1; i < M; ++i) {
      double v = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) v -= L[tri(i, k)] * Li[tri(k, j)];
      Li[tri(i, j)] = v * rd[i];
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = i; k < M; ++k) v += Li[tri(k, i)] * Li[tri(k, j)];
      Sinv[tri(i, j)] = v;
    }
}
template <int M>
OH_DEV void symv(const double (&A)[M * (M + 1) / 2], const double (&x)[M], double (&y)[M]) {
#pragma unroll
  for (int i = 0; i < M; ++i) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) v += A[(i >= k) ? tri(i, k) : tri(k, i)] * x[k];
    y[i] = v;
  }
}

// Ratio test and bookkeeping of one instance at the head of K3 (shared by the serial sweep and the cyclic-reduction kernel below).
// f, fpsi, meas: merit, penalty part and violation of the trial slot ts summed over the knots.  Returns 0 when the instance stops here, 1 when
// the step is to be solved for, 2 when the rejected step is to be tried again shorter (line search, OH_LS_MAX: the caller scales D.zstep).
template <int N, bool GUARD>
OH_DEV int free_accept(const FigParams& P, const FigBuffers& D, const GuardBuffers& GB, const int b, const int ts, const double f, const double fpsi,
                        const double meas, int& cur, LMState& lm) {
  bool accept, line_search = false;
  if (D.first[b]) {
    if (!(f == f) || !(fabs(f) < 1e300)) {  // non-finite seed / parameters: report, do not iterate
      D.status[b] = OH_STATUS_NUMERICAL;
      D.cur[b] = ts;
      D.f_cur[b] = f;
      D.stat[b] = f;
      return 0;
    }
    accept = true;
    D.first[b] = 0;
    if constexpr (GUARD) {
      // restart after a batch compaction with a multiplier update pending: this evaluation has refreshed the multipliers
      if (GB.outer[b]) {
        GB.outer[b] = 0;
        GB.rho[b] = GB.rho_next[b];
      }
    }
  } else if (GUARD && GB.outer[b]) {
    // re-evaluation of the current point after a multiplier update: the merit function itself changed
    accept = true;
    GB.outer[b] = 0;
    GB.rho[b] = GB.rho_next[b];
  } else {
    const LMState lm_before = lm;
    accept = lm_accept(P, f, 0.0, D.f_cur[b], D.pred[b], 0.0, lm);
    if constexpr (GUARD) {
      // a shorter step along the same direction first: the damping stays where it was.  (Only in the first half of the iteration budget: a
      // rejected trial costs a skipped launch, and an instance that never converges should not take the batch twice as long to find out.)
      if (!accept && GB.ls_count[b] < OH_LS_MAX && D.iters[b] < P.max_iter / 2) {
        lm = lm_before;
        line_search = true;
      }
    }
    D.nun[b] = lm.nun;
  }
  if (accept) {
    cur = ts;
    D.f_cur[b] = f;
    D.feas[b] = meas;
    if constexpr (GUARD) {
      D.fpsi[b] = fpsi;
      GB.ls_count[b] = 0;
    }
  }
  D.cur[b] = cur;
  if (!accept) atomicAdd(D.work + 1, 1ULL);  // (no idle launch: the next trial goes back into this instance's own trial slot, OH_FREE_SLOT)
  return line_search ? 2 : 1;
}
// the bookkeeping of a line-search trial (the caller has scaled D.zstep by OH_LS_SHRINK); returns whether the instance goes on
template <bool GUARD>
OH_DEV bool free_line_search(const FigParams& P, const FigBuffers& D, const GuardBuffers& GB, const int b) {
  if constexpr (GUARD) {
    const int k = GB.ls_count[b] + 1;
    GB.ls_count[b] = k;
    double sk = 1.0;
    for (int i = 0; i < k; ++i) sk *= OH_LS_SHRINK;
    D.pred[b] = -sk * GB.ls_gd[b] + 0.5 * sk * sk * GB.ls_q[b];
  }
  const int iters = D.iters[b];
  if (iters >= P.max_iter) {
    D.status[b] = OH_STATUS_MAX_ITER;
    return false;
  }
  D.iters[b] = iters + 1;
  return true;
}

// What happens after the factorisation, given the reduced gradient norm: -1 take the step, 0 the instance stops, 1 it goes on from where
// it stands (outer iteration of the augmented Lagrangian: the caller zeroes the step).
template <int N, bool GUARD>
OH_DEV int free_decide(const FigParams& P, const FigBuffers& D, const GuardBuffers& GB, const int b, const double stat, const double mu, const int iters) {
  D.stat[b] = stat;
  if (!(stat == stat)) {
    D.status[b] = OH_STATUS_NUMERICAL;
    D.mu[b] = mu;
    return 0;
  }
  if constexpr (GUARD) {
    if (stat <= GB.omega[b]) {
      const double meas = D.feas[b];
      if (stat <= P.tol && meas <= P.tol_feas) {
        D.status[b] = OH_STATUS_CONVERGED;
        D.mu[b] = mu;
        return 0;
      }
      if (iters >= P.max_iter) {
        D.status[b] = OH_STATUS_MAX_ITER;
        D.mu[b] = mu;
        return 0;
      }
      // outer iteration: stay where we are, let the next evaluation refresh the multipliers, tighten the inner tolerance
      const double rho = GB.rho[b];
      GB.rho_next[b] = (meas > 0.25 * GB.meas_prev[b]) ? fmin(10.0 * rho, 1e8) : rho;
      GB.meas_prev[b] = meas;
      GB.omega[b] = fmax(P.tol, fmin(GB.omega[b], OH_AL_OMEGA_FREE * meas));
      GB.outer[b] = 1;
      GB.n_outer[b] += 1;
      D.pred[b] = 0.0;
      D.mu[b] = mu;
      D.iters[b] = iters + 1;
      return 1;
    }
  } else {
    if (stat <= P.tol) {
      D.status[b] = OH_STATUS_CONVERGED;
      D.mu[b] = mu;
      return 0;
    }
  }
  if (iters >= P.max_iter) {
    D.status[b] = OH_STATUS_MAX_ITER;
    D.mu[b] = mu;
    return 0;
  }
  return -1;
}

template <int N, bool GUARD, bool VEL = false>
OH_DEV bool step_instance_free(const FigParams& P, const FigBuffers& D, const GuardBuffers& GB, const int b, const int ts) {
  constexpr int NP = N * (N + 1) / 2;
  const int Bp = D.Bp;
  const int T = P.T;
  const double kap2 = 2.0 * P.kappa;
  int cur = 1 - ts;
  LMState lm{D.mu[b], D.nun[b]};
  const int iters = D.iters[b];
  {
    double f = D.fconst[b], fpsi = 0.0, meas = 0.0;
    for (int t = P.t0; t < T; ++t) {
      f += SEL(D.merit, ts)[(size_t)t * Bp + b];
      if constexpr (GUARD) {
        fpsi += SEL(GB.psi, ts)[(size_t)t * Bp + b];
        meas = fmax(meas, SEL(D.cv, ts)[(size_t)t * Bp + b]);
      }
    }
    const int act = free_accept<N, GUARD>(P, D, GB, b, ts, f, fpsi, meas, cur, lm);
    if (act == 0) return false;
    if (act == 2) {
      for (int t = P.t0; t < T; ++t) {
#pragma unroll
        for (int a = 0; a < N; ++a) D.zstep[IDX(t, N, a)] *= OH_LS_SHRINK;
      }
      return free_line_search<GUARD>(P, D, GB, b);
    }
  }
  double mu = lm.mu;
  const double* __restrict__ Drc = SEL(D.Dr, cur);
  const double* __restrict__ gtc = SEL(D.gt, cur);
  const double* __restrict__ Ec = cur ? D.E[1] : D.E[0];  // velocity rows: the coupling vectors (VEL)
  double stat = 0.0;
  double S[NP], rd[N], rn[N];
  bool factored = false;
  for (int attempt = 0; attempt < 40; ++attempt) {
    bool ok = true;
    stat = 0.0;
    {
      const int t = T - 1;
#pragma unroll
      for (int i = 0; i < NP; ++i) S[i] = Drc[DRX(t, i)];
#pragma unroll
      for (int a = 0; a < N; ++a) {
        S[tri(a, a)] += kap2 + mu;
        rn[a] = gtc[GTX(t, a)];
        stat = fmax(stat, fabs(rn[a]));
      }
    }
    // software pipeline: the blocks of knot t-1 are requested before the factorisation of knot t+1's Schur complement, so
    // that the only serial dependency of the sweep (S_{t+1} -> S_t) is not stretched by a memory round trip per knot
    double Hn[NP], gn[N];
    if (T - 2 >= P.t0) {
#pragma unroll
      for (int i = 0; i < NP; ++i) Hn[i] = Drc[DRX(T - 2, i)];
#pragma unroll
      for (int a = 0; a < N; ++a) gn[a] = gtc[GTX(T - 2, a)];
    }
    for (int t = T - 2; t >= P.t0; --t) {
      double Ht[NP], gt[N];
#pragma unroll
      for (int i = 0; i < NP; ++i) Ht[i] = Hn[i];
#pragma unroll
      for (int a = 0; a < N; ++a) {
        gt[a] = gn[a];
        stat = fmax(stat, fabs(gt[a]));
        Ht[tri(a, a)] += 2.0 * kap2 + mu;
      }
      {
        const int tp = (t > P.t0) ? t - 1 : t;
#pragma unroll
        for (int i = 0; i < NP; ++i) Hn[i] = Drc[DRX(tp, i)];
#pragma unroll
        for (int a = 0; a < N; ++a) gn[a] = gtc[GTX(tp, a)];
        __builtin_amdgcn_sched_barrier(0);
      }
      ok = chol_rcp<N>(S, rd, 1e-12) && ok;
      double Sinv[NP], wv[N];
      spd_inverse<N>(S, rd, Sinv);
      symv<N>(Sinv, rn, wv);  // S_{t+1}^{-1} r_{t+1}
#pragma unroll
      for (int i = 0; i < NP; ++i) D.Kmat[IDX(t + 1, N * N, i)] = Sinv[i];
      if constexpr (VEL) {
        // E_t = -diag(d), d = 2 kappa + w_t (velocity rows of interval (t, t+1)): S_t = H_t - D S^{-1} D, r_t = g_t + D S^{-1} r
        double dv[N];
#pragma unroll
        for (int a = 0; a < N; ++a) dv[a] = Ec[IDX(t, N, a)];
#pragma unroll
        for (int a = 0; a < N; ++a) {
          D.kvec[IDX(t + 1, N, a)] = wv[a];
          rn[a] = gt[a] + dv[a] * wv[a];
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) S[tri(i, j)] = Ht[tri(i, j)] - dv[i] * dv[j] * Sinv[tri(i, j)];
      } else {
#pragma unroll
      for (int a = 0; a < N; ++a) {
        D.kvec[IDX(t + 1, N, a)] = wv[a];
        rn[a] = gt[a] + kap2 * wv[a];
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) S[i] = Ht[i] - kap2 * kap2 * Sinv[i];
      }
    }
    ok = chol_rcp<N>(S, rd, 1e-12) && ok;
    if (ok) {
      factored = true;
      break;
    }
    mu = fmax(4.0 * mu, 1e-2);
  }
  if (!factored) stat = __builtin_nan("");  // no damping (up to 4^40) made the matrix factorisable: free_decide reports NUMERICAL
  {
    const int r = free_decide<N, GUARD>(P, D, GB, b, stat, mu, iters);
    // multiplier update (r == 1).  Until round 5 the instance stayed put for that launch (zero step); now it also takes the step the sweep has just
    // solved for -- small, the inner iteration has converged to omega -- and the evaluation that refreshes the multipliers looks at that point and
    // accepts it as it is (free_accept): 7 of an arm's ~27 launches were updates without a step (port, 64 arms: 26.6 -> 24.8 steps, slowest 48 -> 44)
    if (r == 1 && !P.al_fuse) {
      for (int t = P.t0; t < T; ++t) {
#pragma unroll
        for (int a = 0; a < N; ++a) D.zstep[IDX(t, N, a)] = 0.0;
      }
    }
    if (r == 0 || (r == 1 && !P.al_fuse)) return r != 0;
  }
  {
    double zz[N];
#pragma unroll
    for (int a = 0; a < N; ++a) zz[a] = -rn[a];
    fsub_rcp<N>(S, rd, zz);
    bsub_rcp<N>(S, rd, zz);
    double gd = 0.0, z2 = 0.0;
    double Kn[NP], kn[N];
    if (P.t0 + 1 < T) {
#pragma unroll
      for (int i = 0; i < NP; ++i) Kn[i] = D.Kmat[IDX(P.t0 + 1, N * N, i)];
#pragma unroll
      for (int a = 0; a < N; ++a) kn[a] = D.kvec[IDX(P.t0 + 1, N, a)];
    }
    for (int t = P.t0; t < T; ++t) {
      if (t > P.t0) {
        double Sinv[NP], kv[N], y[N];
#pragma unroll
        for (int i = 0; i < NP; ++i) Sinv[i] = Kn[i];
#pragma unroll
        for (int a = 0; a < N; ++a) kv[a] = kn[a];
        {
          const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
          for (int i = 0; i < NP; ++i) Kn[i] = D.Kmat[IDX(tn, N * N, i)];
#pragma unroll
          for (int a = 0; a < N; ++a) kn[a] = D.kvec[IDX(tn, N, a)];
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (VEL) {  // z_t = S_t^{-1} (D_{t-1} z_{t-1}) - S_t^{-1} r_t
#pragma unroll
          for (int a = 0; a < N; ++a) zz[a] *= Ec[IDX(t - 1, N, a)];
          symv<N>(Sinv, zz, y);
#pragma unroll
          for (int a = 0; a < N; ++a) zz[a] = y[a] - kv[a];
        } else {
        symv<N>(Sinv, zz, y);
#pragma unroll
        for (int a = 0; a < N; ++a) zz[a] = kap2 * y[a] - kv[a];
        }
      }
#pragma unroll
      for (int a = 0; a < N; ++a) {
        D.zstep[IDX(t, N, a)] = zz[a];
        gd += gtc[GTX(t, a)] * zz[a];
        z2 += zz[a] * zz[a];
      }
    }
    D.pred[b] = -0.5 * gd + 0.5 * mu * z2;
    if constexpr (GUARD) {
      GB.ls_gd[b] = gd;
      GB.ls_q[b] = gd + mu * z2;
    }
  }
  D.mu[b] = mu;
  D.iters[b] = iters + 1;
  return true;
}

template <int N, bool GUARD, bool VEL = false>
__global__ __launch_bounds__(64) void k_step_free(FigParams P, FigBuffers D, GuardBuffers GB, const int slot_batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = (b < D.B) ? OH_FREE_SLOT(D, b) : slot_batch;
  const bool alive = (b < D.B) && (D.status[b] < 0);
  const bool skipping = alive && D.skip[b];
  const bool running = alive && !skipping;
  {
    const unsigned long long m = __ballot(running);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(D.work, (unsigned long long)__popcll(m));
  }
  bool still = skipping;
  if (skipping) D.skip[b] = 0;
  if (running) still = step_instance_free<N, GUARD, VEL>(P, D, GB, b, slot);
  const unsigned long long m2 = __ballot(still);
  if ((threadIdx.x & 63) == 0 && m2) atomicAdd(D.n_running, __popcll(m2));
}

// K3 of a batch that cannot fill the chip: one BLOCK per instance, one thread per knot.  The serial sweep above is one lane's dependent walk
// over 2 x T knots with a 7 x 7 factorisation at each (T = 100: ~280 us per launch however few instances there are; BASELINE's config 4 is
// 256 arms per GPU, four wavefronts).  Here the block-tridiagonal system  A_t z_t - 2k z_{t-1} - 2k z_{t+1} = -g_t  is solved by block
// parallel cyclic reduction across the lanes: ceil(log2 T) levels; at each every lane factorises its diagonal block, solves for its two
// coupling blocks and its right-hand side and eliminates the neighbours at distance 2^l from its row.  What a lane needs of its neighbours
// goes through LDS in two passes (A^{-1} L and A^{-1} r, then A^{-1} U: 56 rows x NT lanes = 57 KB at NT = 128).  All pivots are positive
// definite iff the matrix is (Haynsworth: each level is a Schur complement onto the lanes of one parity class), so the damping loop raises mu
// exactly when the serial sweep would.  The ratio test and the outer-loop decisions are the serial kernel's (free_accept / free_decide, lane 0).
// The stage arrays are read knot-major (a lane's doubles lie a row apart): instances that share a line are dealt to the same XCD.
template <int N, bool GUARD, int NT, bool VEL = false>
__global__ __launch_bounds__(NT) void k_step_free_pcr(FigParams P, FigBuffers D, GuardBuffers GB, const int slot_batch) {
  constexpr int NP = N * (N + 1) / 2;
  constexpr int O_R = N * N;  // exchange tile: rows [0, N*N) one block (column-major), rows [N*N, N*N + N) one vector
  __shared__ double sm[N * N + N][NT];
  __shared__ double red[3][2];
  __shared__ int ctl[2];
  __shared__ double ctld;
  const int per = (D.B + 7) / 8;  // blocks are dealt round-robin to the 8 XCDs: XCD x takes the instances [x per, (x + 1) per)
  const int b = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);  // (uniform over the block: one instance)
  const int lane = threadIdx.x;
  const int Bp = D.Bp;
  const int T = P.T;
  const int nK = T - P.t0;
  const int t = P.t0 + lane;
  const bool active = lane < nK;
  const int tl = active ? t : T - 1;
  const bool last = (t == T - 1);
  const double kap2 = 2.0 * P.kappa;
  if (lane == 0) ctl[0] = D.status[b] >= 0 ? 0 : (D.skip[b] ? 1 : 2);
  __syncthreads();
  {
    const int st = ctl[0];
    if (st == 0) return;
    if (st == 1) {  // sits this launch out (k_step_free's wrapper)
      if (lane == 0) {
        D.skip[b] = 0;
        atomicAdd(D.n_running, 1);
      }
      return;
    }
  }
  sm[0][lane] = active ? SEL(D.merit, slot)[(size_t)tl * Bp + b] : 0.0;
  if constexpr (GUARD) {
    sm[1][lane] = active ? SEL(GB.psi, slot)[(size_t)tl * Bp + b] : 0.0;
    sm[2][lane] = active ? SEL(D.cv, slot)[(size_t)tl * Bp + b] : 0.0;
  }
  __syncthreads();
  if (lane == 0) {
    atomicAdd(D.work, 1ULL);
    double f = D.fconst[b], fpsi = 0.0, meas = 0.0;
    for (int l = 0; l < nK; ++l) {  // in knot order, like the serial sweep
      f += sm[0][l];
      if constexpr (GUARD) {
        fpsi += sm[1][l];
        meas = fmax(meas, sm[2][l]);
      }
    }
    LMState lm{D.mu[b], D.nun[b]};
    int cur = 1 - slot;
    ctl[0] = free_accept<N, GUARD>(P, D, GB, b, slot, f, fpsi, meas, cur, lm);
    ctl[1] = cur;
    ctld = lm.mu;
  }
  __syncthreads();
  if (ctl[0] == 0) return;
  if (ctl[0] == 2) {  // line search: the rejected step again, shorter
    if (active) {
#pragma unroll
      for (int a = 0; a < N; ++a) D.zstep[IDX(t, N, a)] *= OH_LS_SHRINK;
    }
    if (lane == 0 && free_line_search<GUARD>(P, D, GB, b)) atomicAdd(D.n_running, 1);
    return;
  }
  const int cur = ctl[1];
  double mu = ctld;
  auto block_sum = [&](double v, const int slot_r, const bool is_max) {  // all lanes get the result
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double o = __shfl_xor(v, m);
      v = is_max ? fmax(v, o) : v + o;
    }
    if ((lane & 63) == 0) red[slot_r][lane >> 6] = v;
    __syncthreads();
    double out = red[slot_r][0];
    if (NT > 64) out = is_max ? fmax(out, red[slot_r][1]) : out + red[slot_r][1];
    return out;
  };
  const double* __restrict__ Drc = SEL(D.Dr, cur);
  const double* __restrict__ gtc = SEL(D.gt, cur);
  const double* __restrict__ Ec = cur ? D.E[1] : D.E[0];  // velocity rows: the coupling vectors (VEL)
  double stat = 0.0;
  if (active) {
#pragma unroll
    for (int a = 0; a < N; ++a) stat = fmax(stat, fabs(gtc[GTX(tl, a)]));
  }
  stat = block_sum(stat, 0, true);
  // Registers are the budget (256 + 256 per lane at one wavefront per SIMD): the diagonal block is kept as its lower triangle (every update of
  // it is symmetric: Lw A_-^{-1} Lw^T and U A_+^{-1} U^T, because U_i = Lw_{i+s}^T), the new U overwrites the old row by row, and the stage data
  // is read again for a damped retry instead of being kept.  (Tried: reading the new Lw off the neighbour's new U through LDS instead of
  // computing it -- 343 fewer multiply-adds per level, but the compiler spills more around the extra exchange: 144 against 111 us per launch.)
  double r[N];
  bool factored = false;
  for (int attempt = 0; attempt < 40; ++attempt) {
    double A[NP], Lw[N * N], U[N * N];
#pragma unroll
    for (int i = 0; i < NP; ++i) A[i] = active ? Drc[DRX(tl, i)] : 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      A[tri(i, i)] = active ? A[tri(i, i)] + (last ? kap2 : 2.0 * kap2) + mu : 1.0;
      r[i] = active ? -gtc[GTX(tl, i)] : 0.0;
      // coupling to the next / previous knot: -2 kappa I, with velocity rows -diag(2 kappa + w) of the interval in between (D.E)
      double eu = kap2, el = kap2;
      if constexpr (VEL) {
        if (active && !last) eu = Ec[IDX(tl, N, i)];
        if (active && lane > 0) el = Ec[IDX(tl - 1, N, i)];
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        U[i * N + j] = (i == j && active && !last) ? -eu : 0.0;
        Lw[i * N + j] = (i == j && active && lane > 0) ? -el : 0.0;
      }
    }
    bool ok = true;
    for (int sft = 1; sft < nK; sft <<= 1) {
      double Lc[NP], rd[N];
#pragma unroll
      for (int i = 0; i < NP; ++i) Lc[i] = A[i];
      ok = chol_rcp<N>(Lc, rd, 1e-12) && ok;
      const int lm_ = lane - sft, lp_ = lane + sft;
      const bool hm = lm_ >= 0, hp = lp_ < NT;
      const int im = hm ? lm_ : lane, ip = hp ? lp_ : lane;  // (Lw / U of a lane without that neighbour are zero: the clamped reads add nothing)
      // pass 1: Y^L = A^{-1} Lw and y = A^{-1} r, parked for the neighbours
#pragma unroll
      for (int c = 0; c <= N; ++c) {
        double col[N];
#pragma unroll
        for (int i = 0; i < N; ++i) col[i] = c < N ? Lw[i * N + c] : r[i];
        fsub_rcp<N>(Lc, rd, col);
        bsub_rcp<N>(Lc, rd, col);
#pragma unroll
        for (int i = 0; i < N; ++i) sm[c * N + i][lane] = col[i];
      }
      __syncthreads();
      double Ln[N * N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double racc = r[i];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          double lacc = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) lacc -= Lw[i * N + k] * sm[j * N + k][im];  // -Lw Y^L_{-}
          Ln[i * N + j] = hm ? lacc : 0.0;
          if (j <= i) {
            double aacc = A[tri(i, j)];
#pragma unroll
            for (int k = 0; k < N; ++k) aacc -= U[i * N + k] * sm[j * N + k][ip];  // A - U Y^L_{+}
            A[tri(i, j)] = aacc;
          }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) racc -= Lw[i * N + k] * sm[O_R + k][im] + U[i * N + k] * sm[O_R + k][ip];
        r[i] = racc;
      }
      __syncthreads();
      // pass 2: Y^U = A^{-1} U (Lc is still the factor of the block as it stood before pass 1)
#pragma unroll
      for (int c = 0; c < N; ++c) {
        double col[N];
#pragma unroll
        for (int i = 0; i < N; ++i) col[i] = U[i * N + c];
        fsub_rcp<N>(Lc, rd, col);
        bsub_rcp<N>(Lc, rd, col);
#pragma unroll
        for (int i = 0; i < N; ++i) sm[c * N + i][lane] = col[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double urow[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          double uacc = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) uacc -= U[i * N + k] * sm[j * N + k][ip];  // -U Y^U_{+}
          urow[j] = hp ? uacc : 0.0;
          if (j <= i) {
            double aacc = A[tri(i, j)];
#pragma unroll
            for (int k = 0; k < N; ++k) aacc -= Lw[i * N + k] * sm[j * N + k][im];  // A - Lw Y^U_{-}
            A[tri(i, j)] = aacc;
          }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
          U[i * N + j] = urow[j];
          Lw[i * N + j] = Ln[i * N + j];
        }
      }
      __syncthreads();
    }
    {
      double rd[N];
      ok = chol_rcp<N>(A, rd, 1e-12) && ok;
      fsub_rcp<N>(A, rd, r);
      bsub_rcp<N>(A, rd, r);  // r is the step of this knot now
    }
    if (__syncthreads_and(ok || !active)) {
      factored = true;
      break;
    }
    mu = fmax(4.0 * mu, 1e-2);
  }
  if (!factored) stat = __builtin_nan("");  // (see step_instance_free)
  if (lane == 0) ctl[0] = free_decide<N, GUARD>(P, D, GB, b, stat, mu, D.iters[b]);
  __syncthreads();
  const int dec = ctl[0];
  if (dec == 0) return;
  if (dec == 1 && !P.al_fuse) {  // outer iteration without a step (al_fuse = 0; see step_instance_free)
    if (active) {
#pragma unroll
      for (int a = 0; a < N; ++a) D.zstep[IDX(t, N, a)] = 0.0;
    }
    if (lane == 0) atomicAdd(D.n_running, 1);
    return;
  }
  double gd = 0.0, z2 = 0.0;
  if (active) {
#pragma unroll
    for (int a = 0; a < N; ++a) {
      D.zstep[IDX(t, N, a)] = r[a];
      gd += gtc[GTX(t, a)] * r[a];
      z2 += r[a] * r[a];
    }
  }
  gd = block_sum(gd, 1, false);
  z2 = block_sum(z2, 2, false);
  if (lane == 0) {
    D.pred[b] = -0.5 * gd + 0.5 * mu * z2;
    if constexpr (GUARD) {
      GB.ls_gd[b] = gd;
      GB.ls_q[b] = gd + mu * z2;
    }
    D.mu[b] = mu;
    if (dec != 1) D.iters[b] += 1;  // (a multiplier update was counted by free_decide)
    atomicAdd(D.n_running, 1);
  }
}

// K3 of a latency-bound launch, round 4: one block of two wavefronts per instance, the block-tridiagonal system solved by a twisted factorisation
// ("burn at both ends").  Wavefront 0 eliminates the knots 0, 1, .. m-1 downwards, wavefront 1 the knots nK-1, .. m+1 upwards at the same time;
// knot m then sees both neighbours eliminated, its solve gives z_m, and the two halves substitute back outwards in parallel: T / 2 dependent knots each
// way instead of T.  A knot costs one in-place Gauss-Jordan inversion of its 7 x 7 Schur complement with the lane layout of k_tq_step (lane 8 r + c
// owns entry (r, c), column 7 the right-hand side; pivot row / column / pivot by shuffles: no LDS, no barrier):
//     S_k = A_k - E X_{k-1} E,  y_k = r_k - E w_{k-1},  [X_k | w_k] = S_k^{-1} [I | y_k],    E = -diag(e): entry-wise on the lanes that hold the entries,
// X_k and w_k go to LDS for the substitution  z_k = w_k + X_k (e . z_{k+1}).  All pivots are positive iff the matrix is positive definite (they are the
// pivots of a Cholesky factorisation in the twisted order), so the damping loop raises mu exactly when the serial sweep would.
// k_step_free_pcr does ceil(log2 T) levels of 15 triangular solve pairs and four 7 x 7 x 7 products on one lane per knot: 111-115 us per launch at
// T = 100 however few instances there are; this one ~50.  Ratio test and outer-loop decisions are the serial kernel's (free_accept / free_decide on
// lane 0); the decision "converged / outer update: no step" is taken before the solve (it depends on the reduced gradient only), which spares the
// solve on a quarter of the launches of a guarded handle.
template <int N, bool GUARD, bool VEL = false>
OH_DEV void step_free_bb_block(const FigParams& P, const FigBuffers& D, const GuardBuffers& GB, const int KH) {
  static_assert(N == 7, "lane layout: 8 rows x 8 columns");
  constexpr int NP = N * (N + 1) / 2;
  extern __shared__ double xw[];  // [2][KH][64] X | w of the knots, then [2][KH][8] their right-hand sides
  __shared__ double sm[3][128];
  __shared__ double red[3][2];
  __shared__ int ctl[2];
  __shared__ double ctld;
  __shared__ double zmid[8];
  const int per = (D.B + 7) / 8;  // blocks are dealt round-robin to the 8 XCDs: XCD x takes the instances [x per, (x + 1) per)
  const int b = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);
  const int lane = threadIdx.x;
  const int Bp = D.Bp;
  const int T = P.T;
  const int nK = T - P.t0;
  const double kap2 = 2.0 * P.kappa;
  // The bookkeeping of an instance (free_accept / free_decide on lane 0) walks through two dozen per-instance scalars, one dependent load after the
  // other, a memory latency each: the lanes of the second wavefront touch them all at once first, so that lane 0 finds them in the CU's cache.
  {
    double wd = 0.0;
    int wi = 0;
    const int w = lane - 64;
    const double* const dptr[12] = {D.mu, D.nun, D.f_cur, D.pred, D.feas, D.fconst, D.stat, GUARD ? GB.omega : D.mu, GUARD ? GB.rho : D.mu, GUARD ? GB.rho_next : D.mu,
                                    GUARD ? GB.meas_prev : D.mu, GUARD ? GB.ls_gd : D.mu};
    const int* const iptr[8] = {D.first, D.status, D.cur, D.iters, D.skip, GUARD ? GB.outer : D.first, GUARD ? GB.ls_count : D.first, GUARD ? GB.n_outer : D.first};
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (w == k) wd = __builtin_nontemporal_load(dptr[k] + b);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (w == 12 + k) wi = __builtin_nontemporal_load(iptr[k] + b);
    if (wd == -1.2345e300 || wi == -2047483647) sm[2][lane] = 1.0;  // (keeps the loads; never true)
  }
  if (lane == 0) ctl[0] = D.status[b] >= 0 ? 0 : (D.skip[b] ? 1 : 2);
  __syncthreads();
  {
    const int st = ctl[0];
    if (st == 0) return;
    if (st == 1) {
      if (lane == 0) {
        D.skip[b] = 0;
        atomicAdd(D.n_running, 1);
      }
      return;
    }
  }
  {
    const bool act = lane < nK;
    const int tl = act ? P.t0 + lane : T - 1;
    sm[0][lane] = act ? SEL(D.merit, slot)[(size_t)tl * Bp + b] : 0.0;
    if constexpr (GUARD) {
      sm[1][lane] = act ? SEL(GB.psi, slot)[(size_t)tl * Bp + b] : 0.0;
      sm[2][lane] = act ? SEL(D.cv, slot)[(size_t)tl * Bp + b] : 0.0;
    }
  }
  __syncthreads();
  if (lane == 0) {
    atomicAdd(D.work, 1ULL);
    double f = D.fconst[b], fpsi = 0.0, meas = 0.0;
    for (int l = 0; l < nK; ++l) {  // in knot order, like the serial sweep
      f += sm[0][l];
      if constexpr (GUARD) {
        fpsi += sm[1][l];
        meas = fmax(meas, sm[2][l]);
      }
    }
    LMState lm{D.mu[b], D.nun[b]};
    int cur = 1 - slot;
    ctl[0] = free_accept<N, GUARD>(P, D, GB, b, slot, f, fpsi, meas, cur, lm);
    ctl[1] = cur;
    ctld = lm.mu;
  }
  __syncthreads();
  if (ctl[0] == 0) return;
  if (ctl[0] == 2) {  // line search: the rejected step again, shorter
    if (lane < nK) {
      const int t = P.t0 + lane;
#pragma unroll
      for (int a = 0; a < N; ++a) D.zstep[IDX(t, N, a)] *= OH_LS_SHRINK;
    }
    if (lane == 0 && free_line_search<GUARD>(P, D, GB, b)) atomicAdd(D.n_running, 1);
    return;
  }
  const int cur = ctl[1];
  double mu = ctld;
  auto block_sum = [&](double v, const int slot_r, const bool is_max) {  // all lanes get the result
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double o = __shfl_xor(v, m);
      v = is_max ? fmax(v, o) : v + o;
    }
    if ((lane & 63) == 0) red[slot_r][lane >> 6] = v;
    __syncthreads();
    return is_max ? fmax(red[slot_r][0], red[slot_r][1]) : red[slot_r][0] + red[slot_r][1];
  };
  const double* __restrict__ Drc = SEL(D.Dr, cur);
  const double* __restrict__ gtc = SEL(D.gt, cur);
  const double* __restrict__ Ec = cur ? D.E[1] : D.E[0];  // velocity rows: the coupling vectors (VEL)
  double stat = 0.0;
  if (lane < nK) {
    const int t = P.t0 + lane;
#pragma unroll
    for (int a = 0; a < N; ++a) stat = fmax(stat, fabs(gtc[GTX(t, a)]));
  }
  stat = block_sum(stat, 0, true);
  if (lane == 0) ctl[0] = free_decide<N, GUARD>(P, D, GB, b, stat, mu, D.iters[b]);
  __syncthreads();
  {
    const int dec = ctl[0];
    if (dec == 0) return;
    if (dec == 1 && !P.al_fuse) {  // outer iteration without a step (al_fuse = 0; see step_instance_free)
      if (lane < nK) {
        const int t = P.t0 + lane;
#pragma unroll
        for (int a = 0; a < N; ++a) D.zstep[IDX(t, N, a)] = 0.0;
      }
      if (lane == 0) atomicAdd(D.n_running, 1);
      return;
    }
  }
  const bool upd = ctl[0] == 1;  // multiplier update with the step taken along: free_decide has counted the launch already
  // ---- the solve ----
  const int wv = lane >> 6, l = lane & 63, r = l >> 3, c = l & 7;
  const bool mat = r < N && c < N, vec = r < N && c == N;
  const int m = nK / 2;                      // meeting knot
  const int cnt = wv == 0 ? m : nK - m;      // knots of this wavefront (wavefront 1 ends on the meeting knot)
  const int k0 = wv == 0 ? 0 : nK - 1, dir = wv == 0 ? 1 : -1;
  double* xs = xw + (size_t)wv * KH * 64;
  const int rr = r < N ? r : 0, cc = c < N ? c : 0;
  const int pko = rr >= cc ? tri(rr, cc) : tri(cc, rr);
  // coupling coefficient of the interval below knot index k (between k and k + 1), for row r and column c of this lane
  auto coup = [&](const int k, double& er, double& ec) {
    er = ec = kap2;
    if constexpr (VEL) {
      er = Ec[IDX(P.t0 + k, N, rr)];
      ec = Ec[IDX(P.t0 + k, N, cc)];
    }
  };
  // one load instruction per knot for all lanes (the pointer is chosen per lane; idle lanes read along on entry 0), addressed by a per-lane base and
  // stride computed once: the index expressions of 50 unrolled fetches were 5000 instructions of 64-bit arithmetic
  const double* fbase = vec ? gtc + GTX(P.t0 + k0, rr) : Drc + DRX(P.t0 + k0, pko);
  const long long fstride = (long long)dir * (vec ? (long long)(GTX(P.t0 + 1, rr) - GTX(P.t0, rr)) : (long long)(DRX(P.t0 + 1, pko) - DRX(P.t0, pko)));
  const double dsel = (mat && r == c) ? 1.0 : 0.0, vsgn = vec ? -1.0 : 1.0, live = (mat || vec) ? 1.0 : 0.0;
  bool factored = false;
  double* gsh = xw + (size_t)2 * KH * 64 + (size_t)wv * KH * 8;  // [KH][8]: -g of the wavefront's knots (the substitution needs them again)
  for (int attempt = 0; attempt < 40; ++attempt) {
    bool ok = true;
    // the wavefront's entries of all its knots in one go (a lane's loads of consecutive knots are independent: one memory latency, not one per knot)
    {  // every load of the wavefront in flight before the first store (a loop of load-store pairs pays a memory latency per knot)
      double v[64];
      const double* fp = fbase;
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        v[u] = (u < cnt) ? *fp : 0.0;
        fp += fstride;
      }
#pragma unroll
      for (int u = 0; u < 64; ++u)
        if (u < cnt) {
          const bool lastk = (k0 + dir * u == nK - 1);
          const double a = live * fma(dsel, lastk ? kap2 : 2.0 * kap2, vsgn * v[u]);
          xs[(size_t)u * 64 + l] = a;
          if (vec) gsh[u * 8 + r] = a;
        }
    }
    double xprev = 0.0;  // matrix lanes: X of the knot before; vector lanes: w
    for (int i = 0; i < cnt; ++i) {
      const int k = k0 + dir * i;
      double a = xs[(size_t)i * 64 + l];
      if (mat && r == c) a += mu;
      if (i > 0) {
        double er, ec;
        coup(wv == 0 ? k - 1 : k, er, ec);
        a = mat ? fma(-(er * ec), xprev, a) : (vec ? fma(er, xprev, a) : a);
      }
      if (wv == 1 && i == cnt - 1) {
        __syncthreads();  // wavefront 0 has parked X and w of knot m - 1
        if (m > 0) {
          double er, ec;
          coup(m - 1, er, ec);
          const double x0 = xw[(size_t)(m - 1) * 64 + l];
          a = mat ? fma(-(er * ec), x0, a) : (vec ? fma(er, x0, a) : a);
        }
      }
      // in-place Gauss-Jordan inversion, right-hand side carried along in column 7; pivots in 2 x 2 blocks (the dependent chain of a knot is
      // shuffle -> reciprocal -> update: 4 of them per knot instead of 7).  Block J = {j, j + 1}, Q = P^{-1}:
      //   a[r][c] -= a[r][J] Q a[J][c];  rows of J: Q a[J][c];  columns of J: -a[r][J] Q;  the block itself: Q
#pragma unroll
      for (int j = 0; j + 1 < N; j += 2) {
        const double p00 = readlane_f64(a, 9 * j), p01 = readlane_f64(a, 9 * j + 1), p11 = readlane_f64(a, 9 * j + 9);
        const double u0 = __shfl(a, 8 * r + j), u1 = __shfl(a, 8 * r + j + 1);
        const double v0 = __shfl(a, 8 * j + c), v1 = __shfl(a, 8 * j + 8 + c);
        const double det = fma(p00, p11, -(p01 * p01));
        if (!(p00 > 1e-12) || !(det > 1e-14 * p00 * p11) || !(det < 1e300)) ok = false;
        double d = __builtin_amdgcn_rcp(det);
        d = d * fma(-det, d, 2.0);
        d = d * fma(-det, d, 2.0);
        const double q00 = p11 * d, q01 = -(p01 * d), q11 = p00 * d;
        const double w0 = fma(q00, v0, q01 * v1), w1 = fma(q01, v0, q11 * v1);  // rows of Q a[J][c]
        const double gen = a - fma(u0, w0, u1 * w1);
        const double c0 = -fma(u0, q00, u1 * q01), c1 = -fma(u0, q01, u1 * q11);  // columns of -a[r][J] Q
        const bool rJ = (r == j) || (r == j + 1), cJ = (c == j) || (c == j + 1);
        const double blk = (r == j) ? ((c == j) ? q00 : q01) : ((c == j) ? q01 : q11);
        const double rowv = (r == j) ? w0 : w1, colv = (c == j) ? c0 : c1;
        a = rJ ? (cJ ? blk : rowv) : (cJ ? colv : gen);
      }
      {  // the last pivot alone
        constexpr int j = N - 1;
        const double piv = readlane_f64(a, 9 * j);
        const double cj = __shfl(a, 8 * r + j), rj = __shfl(a, 8 * j + c);
        if (!(piv > 1e-12) || !(piv < 1e300)) ok = false;
        double d = __builtin_amdgcn_rcp(piv);
        d = d * fma(-piv, d, 2.0);
        d = d * fma(-piv, d, 2.0);
        const double gen = fma(-(cj * rj), d, a), sc = a * d;
        a = (r == j) ? ((c == j) ? d : sc) : ((c == j) ? -sc : gen);
      }
      xs[(size_t)i * 64 + l] = a;
      xprev = a;
    }
    if (wv == 0) __syncthreads();  // (the barrier wavefront 1 waits at before the meeting knot)
    if (__syncthreads_and(ok)) {
      factored = true;
      break;
    }
    mu = fmax(4.0 * mu, 1e-2);
  }
  if (!factored) {
    if (lane == 0) {
      D.stat[b] = __builtin_nan("");
      D.status[b] = OH_STATUS_NUMERICAL;
      D.mu[b] = mu;
    }
    return;
  }
  // z_m sits on the vector lanes of wavefront 1 (the last knot it inverted)
  if (wv == 1 && vec) zmid[r] = xs[(size_t)(cnt - 1) * 64 + l];
  __syncthreads();
  double gd = 0.0, z2 = 0.0;
  {
    double zc = zmid[cc];  // z of the knot solved before, component c
    if (wv == 1 && vec) {
      const int t = P.t0 + m;
      const double z = zmid[r];
      D.zstep[IDX(t, N, r)] = z;
      gd = -gsh[(cnt - 1) * 8 + r] * z;
      z2 = z * z;
    }
    const int first = wv == 0 ? m - 1 : cnt - 2;  // index into xs
    for (int i = first; i >= 0; --i) {
      const int k = k0 + dir * i;
      double er, ec;
      coup(wv == 0 ? k : k - 1, er, ec);
      const double x = xs[(size_t)i * 64 + l];
      double part = mat ? x * (ec * zc) : 0.0;
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      const double z = x + part;  // meaningful on the vector lanes: w_r + sum_c X[r][c] e_c z_c
      zc = __shfl(z, 8 * cc + 7);
      if (vec) {
        const int t = P.t0 + k;
        D.zstep[IDX(t, N, r)] = z;
        gd = fma(-gsh[i * 8 + r], z, gd);
        z2 = fma(z, z, z2);
      }
    }
  }
  gd = block_sum(gd, 1, false);
  z2 = block_sum(z2, 2, false);
  if (lane == 0) {
    D.pred[b] = -0.5 * gd + 0.5 * mu * z2;
    if constexpr (GUARD) {
      GB.ls_gd[b] = gd;
      GB.ls_q[b] = gd + mu * z2;
    }
    D.mu[b] = mu;
    if (!upd) D.iters[b] += 1;  // (a multiplier update was counted by free_decide)
    atomicAdd(D.n_running, 1);
  }
}

template <int N, bool GUARD, bool VEL = false>
__global__ __launch_bounds__(128) void k_step_free_bb(FigParams P, FigBuffers D, GuardBuffers GB, const int slot_batch, const int KH) {
  step_free_bb_block<N, GUARD, VEL>(P, D, GB, KH);
}

// The whole solve of an instance in ONE launch (round 4; handles with limit / sphere rows, no velocity rows, coupling folded into the evaluation): the
// block of two wavefronts that sweeps also evaluates -- thread t the knot t -- and loops until its instance has a status.  Same device functions, same
// memory, same iterates as the launch pair k_eval_guarded / k_step_free_bb; what goes away is the launch gaps, the host's looks at the running count
// (every 8th iteration: up to 7 idle launches at the end of a latency-bound solve) and the waiting for the slowest instance of the batch.
template <int N>
__global__ __launch_bounds__(128) void k_free_persist(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int KH, const int max_rounds) {
  __shared__ int go;
  const int per = (D.B + 7) / 8;
  const int b = (blockIdx.x % 8) * per + blockIdx.x / 8;  // (as step_free_bb_block deals the instances)
  if (b >= D.B) return;
  const int lane = threadIdx.x;
  const int nK = P.T - P.t0;
  for (int round = 0; round < max_rounds; ++round) {
    if (lane < nK) eval_guarded_knot<N>(P, D, GP, GB, b, P.t0 + lane);
    __threadfence_block();
    __syncthreads();
    step_free_bb_block<N, true, false>(P, D, GB, KH);
    __threadfence_block();
    __syncthreads();
    if (lane == 0) go = (__builtin_nontemporal_load(D.status + b) < 0) ? 1 : 0;
    __syncthreads();
    if (!go) break;
  }
}

// K3 of a launch of a few hundred instances with at most 64 free knots (dual_arm.py as shipped: T = 50), round 3: the same block cyclic
// reduction with a knot's ROWS spread over eight lanes.  k_step_free_pcr gives a knot one lane: 15 triangular solve pairs and four 7 x 7 x 7 products per level on
// one lane, 512 registers with ~86 doubles spilled, 87 us per launch at T = 50 however few instances there are.  Here
// lane c < N of a knot owns row c of its three blocks (A, Lw, U) and r_c; lane N solves A y = r.  Per level: (1) rows to LDS, (2) every lane
// factorises the knot's A for itself and solves for its column of A^{-1} Lw and A^{-1} U (lane N: A^{-1} r), which replace Lw / U in LDS,
// (3) every lane forms its rows of the reduced blocks from its own rows (registers) and the two neighbours' solved blocks (LDS; the eight
// lanes of a knot read the same words: broadcasts).  ~400 multiply-adds per lane and level instead of ~2100, 61 us per launch at T = 50.  Same system, same
// ratio test and outer-loop decisions (free_accept / free_decide on lane 0), the damping raised exactly when a pivot of any knot fails.
template <int N, bool GUARD, int KN, bool VEL = false>
__global__ __launch_bounds__(8 * KN) void k_step_free_cp(FigParams P, FigBuffers D, GuardBuffers GB, const int slot_batch) {
  static_assert(N <= 7, "eight lanes per knot");
  constexpr int NP = N * (N + 1) / 2;
  constexpr int NW = (8 * KN) / 64;  // wavefronts
  __shared__ double TA[NP][KN];      // packed lower triangle of the knots' diagonal blocks
  __shared__ double TL[N * N][KN];   // Lw row-major, then A^{-1} Lw
  __shared__ double TU[N * N][KN];   // U row-major, then A^{-1} U
  __shared__ double TY[N][KN];       // r, then A^{-1} r
  __shared__ double red[3][NW];
  __shared__ int ctl[2];
  __shared__ double ctld;
  const int per = (D.B + 7) / 8;  // blocks are dealt round-robin to the 8 XCDs (see k_step_free_pcr)
  const int b = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (b >= D.B) return;
  const int slot = OH_FREE_SLOT(D, b);  // (uniform over the block: one instance)
  const int tid = threadIdx.x;
  const int kn = tid >> 3, c = tid & 7;  // knot (lane of the reduction), row within the knot (c == N: the right-hand side solver)
  const int Bp = D.Bp;
  const int T = P.T;
  const int nK = T - P.t0;
  const int t = P.t0 + kn;
  const bool active = kn < nK;
  const bool row = c < N;
  const int tl = active ? t : T - 1;
  const int cr = row ? c : 0;
  const bool last = (t == T - 1);
  const double kap2 = 2.0 * P.kappa;
  if (tid == 0) ctl[0] = D.status[b] >= 0 ? 0 : (D.skip[b] ? 1 : 2);
  __syncthreads();
  {
    const int st = ctl[0];
    if (st == 0) return;
    if (st == 1) {
      if (tid == 0) {
        D.skip[b] = 0;
        atomicAdd(D.n_running, 1);
      }
      return;
    }
  }
  if (c == 0) {
    TY[0][kn] = active ? SEL(D.merit, slot)[(size_t)tl * Bp + b] : 0.0;
    if constexpr (GUARD) {
      TY[1][kn] = active ? SEL(GB.psi, slot)[(size_t)tl * Bp + b] : 0.0;
      TY[2][kn] = active ? SEL(D.cv, slot)[(size_t)tl * Bp + b] : 0.0;
    }
  }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(D.work, 1ULL);
    double f = D.fconst[b], fpsi = 0.0, meas = 0.0;
    for (int l = 0; l < nK; ++l) {  // in knot order, like the serial sweep
      f += TY[0][l];
      if constexpr (GUARD) {
        fpsi += TY[1][l];
        meas = fmax(meas, TY[2][l]);
      }
    }
    LMState lm{D.mu[b], D.nun[b]};
    int cur = 1 - slot;
    ctl[0] = free_accept<N, GUARD>(P, D, GB, b, slot, f, fpsi, meas, cur, lm);
    ctl[1] = cur;
    ctld = lm.mu;
  }
  __syncthreads();
  if (ctl[0] == 0) return;
  if (ctl[0] == 2) {  // line search: the rejected step again, shorter
    if (active && row) D.zstep[IDX(t, N, c)] *= OH_LS_SHRINK;
    if (tid == 0 && free_line_search<GUARD>(P, D, GB, b)) atomicAdd(D.n_running, 1);
    return;
  }
  const int cur = ctl[1];
  double mu = ctld;
  auto block_sum = [&](double v, const int slot_r, const bool is_max) {  // all lanes get the result
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double o = __shfl_xor(v, m);
      v = is_max ? fmax(v, o) : v + o;
    }
    __syncthreads();  // (the previous reduction's readers are through)
    if ((tid & 63) == 0) red[slot_r][tid >> 6] = v;
    __syncthreads();
    double out = red[slot_r][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) out = is_max ? fmax(out, red[slot_r][w]) : out + red[slot_r][w];
    return out;
  };
  const double* __restrict__ Drc = cur ? D.Dr[1] : D.Dr[0];
  const double* __restrict__ gtc = cur ? D.gt[1] : D.gt[0];
  const double* __restrict__ Ec = cur ? D.E[1] : D.E[0];
  const double gmine = (active && row) ? gtc[GTX(tl, cr)] : 0.0;
  const double stat = block_sum(fabs(gmine), 0, true);
  double z = 0.0;  // the step component (t, c) at the end
  bool factored = false;
  for (int attempt = 0; attempt < 40; ++attempt) {
    double Ar[N], Lr[N], Ur[N], rc;  // row c of the three blocks, r_c
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int hi = cr > j ? cr : j, lo = cr > j ? j : cr;
      Ar[j] = (active && row) ? Drc[DRX(tl, tri(hi, lo))] : (j == cr ? 1.0 : 0.0);
      Lr[j] = 0.0;
      Ur[j] = 0.0;
    }
    rc = -gmine;
    if (active && row) {
      double eu = kap2, el = kap2;
      if constexpr (VEL) {
        if (!last) eu = Ec[IDX(tl, N, cr)];
        if (kn > 0) el = Ec[IDX(tl - 1, N, cr)];
      }
      Ar[cr] += (last ? kap2 : 2.0 * kap2) + mu;
      if (!last) Ur[cr] = -eu;
      if (kn > 0) Lr[cr] = -el;
    }
    bool ok = true;
    for (int sft = 1; sft < nK; sft <<= 1) {
      // (1) rows to LDS
      if (row) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          if (j <= c) TA[tri(c, j)][kn] = Ar[j];
          TL[c * N + j][kn] = Lr[j];
          TU[c * N + j][kn] = Ur[j];
        }
        TY[c][kn] = rc;
      }
      __syncthreads();
      // (2) factor, solve for this lane's column
      double Lc[NP], rd[N], colL[N], colU[N];
#pragma unroll
      for (int i = 0; i < NP; ++i) Lc[i] = TA[i][kn];
      ok = chol_rcp<N>(Lc, rd, 1e-12) && ok;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        colL[i] = row ? TL[i * N + cr][kn] : TY[i][kn];
        colU[i] = row ? TU[i * N + cr][kn] : 0.0;
      }
      fsub_rcp<N>(Lc, rd, colL);
      bsub_rcp<N>(Lc, rd, colL);
      if (row) {
        fsub_rcp<N>(Lc, rd, colU);
        bsub_rcp<N>(Lc, rd, colU);
      }
      __syncthreads();  // every lane has read the rows
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (row) {
          TL[i * N + c][kn] = colL[i];
          TU[i * N + c][kn] = colU[i];
        } else {
          TY[i][kn] = colL[i];
        }
      }
      __syncthreads();
      // (3) this lane's rows of the reduced blocks
      const int km = kn - sft, kp = kn + sft;
      const bool hm = km >= 0, hp = kp < KN;
      const int im = hm ? km : kn, ip = hp ? kp : kn;  // (rows of a lane without that neighbour are zero: the clamped reads add nothing)
      if (row) {
        double An[N], Ln[N], Un[N];
        double rn = rc;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          double aacc = Ar[j], lacc = 0.0, uacc = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) {
            lacc -= Lr[k] * TL[k * N + j][im];                              // -Lw (A^{-1} Lw)_-
            uacc -= Ur[k] * TU[k * N + j][ip];                              // -U (A^{-1} U)_+
            aacc -= Lr[k] * TU[k * N + j][im] + Ur[k] * TL[k * N + j][ip];  // A - Lw (A^{-1} U)_- - U (A^{-1} Lw)_+
          }
          An[j] = aacc;
          Ln[j] = hm ? lacc : 0.0;
          Un[j] = hp ? uacc : 0.0;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) rn -= Lr[k] * TY[k][im] + Ur[k] * TY[k][ip];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          Ar[j] = An[j];
          Lr[j] = Ln[j];
          Ur[j] = Un[j];
        }
        rc = rn;
      }
      __syncthreads();
    }
    // the knots are decoupled: A z = r
    if (row) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (j <= c) TA[tri(c, j)][kn] = Ar[j];
      TY[c][kn] = rc;
    }
    __syncthreads();
    {
      double Lc[NP], rd[N], y[N];
#pragma unroll
      for (int i = 0; i < NP; ++i) Lc[i] = TA[i][kn];
      ok = chol_rcp<N>(Lc, rd, 1e-12) && ok;
#pragma unroll
      for (int i = 0; i < N; ++i) y[i] = TY[i][kn];
      fsub_rcp<N>(Lc, rd, y);
      bsub_rcp<N>(Lc, rd, y);
      z = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i == c) z = y[i];
    }
    if (__syncthreads_and(ok || !active)) {
      factored = true;
      break;
    }
    mu = fmax(4.0 * mu, 1e-2);
  }
  double stat_out = stat;
  if (!factored) stat_out = __builtin_nan("");  // (see step_instance_free)
  if (tid == 0) ctl[0] = free_decide<N, GUARD>(P, D, GB, b, stat_out, mu, D.iters[b]);
  __syncthreads();
  const int dec = ctl[0];
  if (dec == 0) return;
  if (dec == 1 && !P.al_fuse) {  // outer iteration without a step (al_fuse = 0; see step_instance_free)
    if (active && row) D.zstep[IDX(t, N, c)] = 0.0;
    if (tid == 0) atomicAdd(D.n_running, 1);
    return;
  }
  double gd = 0.0, z2 = 0.0;
  if (active && row) {
    D.zstep[IDX(t, N, c)] = z;
    gd = gmine * z;
    z2 = z * z;
  }
  gd = block_sum(gd, 1, false);
  z2 = block_sum(z2, 2, false);
  if (tid == 0) {
    D.pred[b] = -0.5 * gd + 0.5 * mu * z2;
    if constexpr (GUARD) {
      GB.ls_gd[b] = gd;
      GB.ls_q[b] = gd + mu * z2;
    }
    D.mu[b] = mu;
    if (dec != 1) D.iters[b] += 1;  // (a multiplier update was counted by free_decide)
    atomicAdd(D.n_running, 1);
  }
}

// Chain lengths (round 5): the one-thread-per-unit kernels of this family are instantiated for 2 ... 8 actuated joints (planar_3dof, the tester
// robots, 8-joint arms); the block-per-instance sweeps (cyclic reduction, twisted factorisation, the persistent kernel) stay with 6 and 7, whose
// lane layouts they are written for -- other chain lengths take the serial sweep at every batch size.
#define OH_FREE_DISPATCH_N(n, call) \
  switch (n) {                       \
    case 2: call(2); break;          \
    case 3: call(3); break;          \
    case 4: call(4); break;          \
    case 5: call(5); break;          \
    case 6: call(6); break;          \
    case 7: call(7); break;          \
    case 8: call(8); break;          \
    default: return false;           \
  }
bool oh_launch_eval_free(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot) {
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
#define C(NN) hipLaunchKernelGGL(k_eval_free<NN>, g, b, 0, s, P, D, slot)
  OH_FREE_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_couple_free(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot) {
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
#define C(NN) hipLaunchKernelGGL(k_couple_free<NN>, g, b, 0, s, P, D, slot)
  OH_FREE_DISPATCH_N(n, C)
#undef C
  return true;
}
// pcr: one block per instance (k_step_free_pcr, or k_step_free_cp with eight lanes per knot while the launch has at most free_cp_max instances;
// the knots must fit 128 lanes)
template <int N, bool GUARD, bool VEL = false>
static void launch_step_free_pcr(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardBuffers& GB, int slot) {
  const int free_cp_max = oh_launch_opts().free_cp_max;  // (per call: the handle's option "free_cp_max"; a local: the parts of a split solve launch from threads of their own)
  const dim3 g(8 * ((D.B + 7) / 8));
  if constexpr (N == 7) {
    if (oh_launch_opts().free_bb != 0) {  // option "free_bb" = 0: the cyclic-reduction kernels of rounds 2-3 (A/B, tests)
      const int nK = P.T - P.t0, KH = nK - nK / 2;
      hipLaunchKernelGGL((k_step_free_bb<N, GUARD, VEL>), g, dim3(128), sizeof(double) * 2 * (size_t)KH * 72, s, P, D, GB, slot, KH);
      return;
    }
  }
  // (only up to 64 knots: 128 knots x 8 lanes are 1024 threads, which leaves 128 registers per lane -- the kernel then spills and takes 147 us
  //  against k_step_free_pcr's 111 at T = 100; at T = 50 it is 61 against 87 us)
  if (D.B <= free_cp_max && P.T - P.t0 <= 64) {
    hipLaunchKernelGGL((k_step_free_cp<N, GUARD, 64, VEL>), g, dim3(512), 0, s, P, D, GB, slot);
    return;
  }
  if (P.T - P.t0 <= 64) hipLaunchKernelGGL((k_step_free_pcr<N, GUARD, 64, VEL>), g, dim3(64), 0, s, P, D, GB, slot);
  else hipLaunchKernelGGL((k_step_free_pcr<N, GUARD, 128, VEL>), g, dim3(128), 0, s, P, D, GB, slot);
}
bool oh_launch_step_free(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot, bool pcr) {
  const dim3 g((D.B + 63) / 64), b(64);
  const GuardBuffers none{};
  if (pcr && P.T - P.t0 <= 128 && (n == 7 || n == 6)) {
    if (n == 7) launch_step_free_pcr<7, false>(s, P, D, none, slot);
    else launch_step_free_pcr<6, false>(s, P, D, none, slot);
    return true;
  }
#define C(NN) hipLaunchKernelGGL((k_step_free<NN, false>), g, b, 0, s, P, D, none, slot)
  OH_FREE_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_setup_guards(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, const double* p) {
  hipLaunchKernelGGL(k_setup_guards, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, GP, GB, p, n);
  return true;
}
bool oh_launch_free_persist(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB) {
  if (n != 7) return false;
  const int nK = P.T - P.t0, KH = nK - nK / 2;
  if (nK > 128) return false;
  hipLaunchKernelGGL(k_free_persist<7>, dim3(8 * ((D.B + 7) / 8)), dim3(128), sizeof(double) * 2 * (size_t)KH * 72, s, P, D, GP, GB, KH, 2 * P.max_iter + 8);
  return true;
}
bool oh_launch_eval_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
#define C(NN) hipLaunchKernelGGL(k_eval_guarded<NN>, g, b, 0, s, P, D, GP, GB, slot)
  OH_FREE_DISPATCH_N(n, C)
#undef C
  return true;
}
template <int N>
static void launch_step_guarded_t(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot, bool pcr) {
  const dim3 g((D.B + 63) / 64), b(64);
  if constexpr (N == 6 || N == 7) {  // (the block-per-instance sweeps are written for these lane layouts)
    if (pcr && P.T - P.t0 <= 128) {
      if (GP.vel) launch_step_free_pcr<N, true, true>(s, P, D, GB, slot);
      else launch_step_free_pcr<N, true>(s, P, D, GB, slot);
      return;
    }
  }
  if (GP.vel) hipLaunchKernelGGL((k_step_free<N, true, true>), g, b, 0, s, P, D, GB, slot);
  else hipLaunchKernelGGL((k_step_free<N, true>), g, b, 0, s, P, D, GB, slot);
}
bool oh_launch_step_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot, bool pcr) {
#define C(NN) launch_step_guarded_t<NN>(s, P, D, GP, GB, slot, pcr)
  OH_FREE_DISPATCH_N(n, C)
#undef C
  return true;
}
// position-tracking family with joint-velocity rows: multiplier refresh of those rows (outer updates only), then the coupling
bool oh_launch_couple_free_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
#define C(NN)                                                                 \
  hipLaunchKernelGGL(k_vel_update_free<NN>, g, b, 0, s, P, D, GP, GB, slot); \
  hipLaunchKernelGGL(k_couple_free_vel<NN>, g, b, 0, s, P, D, GP, GB, slot)
  OH_FREE_DISPATCH_N(n, C)
#undef C
  return true;
}

// ---- guard state of a batch that is compacted while it drains (oh_api.hip) ----------------------------------------------------------
// The multipliers, the obstacle parameters and the outer-loop scalars of an instance move with it (k_compact_* move the knots); the
// multipliers of an instance that has finished are written out at its original index first, where oh_get_multipliers reads them.
__global__ __launch_bounds__(256) void k_guard_emit(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int NV, const int only_done) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  if (only_done && D.status[b] < 0) return;
  const int o = D.orig[b];
  for (int i = 0; i < GP.NC; ++i) GB.lam_out[((size_t)t * GP.NC + i) * Bp + o] = GB.lam[((size_t)t * GP.NC + i) * Bp + b];
  for (int i = 0; i < NV; ++i) GB.lamv_out[((size_t)t * NV + i) * Bp + o] = GB.lamv[((size_t)t * NV + i) * Bp + b];
}
__global__ __launch_bounds__(256) void k_guard_gather(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int NV) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int nb = D.newidx[b];
  if (nb < 0) return;
  const int NR = GP.NC + NV;
  for (int i = 0; i < GP.NC; ++i) GB.scr[((size_t)t * NR + i) * Bp + nb] = GB.lam[((size_t)t * GP.NC + i) * Bp + b];
  for (int i = 0; i < NV; ++i) GB.scr[((size_t)t * NR + GP.NC + i) * Bp + nb] = GB.lamv[((size_t)t * NV + i) * Bp + b];
  if (!P.lock && P.al_fuse && GB.outer[b] && t >= P.t0) {
    // position-tracking family, multiplier update pending with its step taken along (al_fuse): the point the next evaluation refreshes the multipliers
    // at -- and accepts as it is -- is the accepted knot plus that step; the restart lays THAT down (k_compact_gather, launched before this kernel, has
    // put the accepted knots into the scratch rows of D.Z[1])
    const int N = P.nx / (2 * P.T - 1);
    double* tq = D.Z[1];
    for (int j = 0; j < N; ++j) tq[((size_t)t * N + j) * Bp + nb] += D.zstep[((size_t)t * N + j) * Bp + b];
  }
  if (t == 0) {
    double* sc = GB.scr + (size_t)P.T * NR * Bp;
    const int npar = GP.n_links + 4 * GP.n_obs;
    for (int i = 0; i < npar; ++i) sc[(size_t)i * Bp + nb] = GB.par[(size_t)i * Bp + b];
    sc += (size_t)npar * Bp;
    sc[(size_t)0 * Bp + nb] = GB.rho[b];
    sc[(size_t)1 * Bp + nb] = GB.rho_next[b];
    sc[(size_t)2 * Bp + nb] = GB.omega[b];
    sc[(size_t)3 * Bp + nb] = GB.meas_prev[b];
    sc[(size_t)4 * Bp + nb] = (double)GB.outer[b];
    sc[(size_t)5 * Bp + nb] = (double)GB.n_outer[b];
  }
}
__global__ __launch_bounds__(256) void k_guard_scatter(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int NV, const int Bnew) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= Bnew) return;
  const int NR = GP.NC + NV;
  for (int i = 0; i < GP.NC; ++i) GB.lam[((size_t)t * GP.NC + i) * Bp + b] = GB.scr[((size_t)t * NR + i) * Bp + b];
  for (int i = 0; i < NV; ++i) GB.lamv[((size_t)t * NV + i) * Bp + b] = GB.scr[((size_t)t * NR + GP.NC + i) * Bp + b];
  if (t == 0) {
    const double* sc = GB.scr + (size_t)P.T * NR * Bp;
    const int npar = GP.n_links + 4 * GP.n_obs;
    for (int i = 0; i < npar; ++i) GB.par[(size_t)i * Bp + b] = sc[(size_t)i * Bp + b];
    sc += (size_t)npar * Bp;
    GB.rho[b] = sc[(size_t)0 * Bp + b];
    GB.rho_next[b] = sc[(size_t)1 * Bp + b];
    GB.omega[b] = sc[(size_t)2 * Bp + b];
    GB.meas_prev[b] = sc[(size_t)3 * Bp + b];
    GB.outer[b] = (int)sc[(size_t)4 * Bp + b];
    GB.n_outer[b] = (int)sc[(size_t)5 * Bp + b];
    GB.ls_count[b] = 0;  // the restart re-derives the full step: a line search in progress starts over (the counter at this index was another instance's)
  }
}
void oh_launch_guard_emit(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int NV, int only_done) {
  hipLaunchKernelGGL(k_guard_emit, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D, GP, GB, NV, only_done);
}
void oh_launch_guard_compact(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int NV, int phase, int Bnew) {
  if (phase == 0) hipLaunchKernelGGL(k_guard_gather, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D, GP, GB, NV);
  else hipLaunchKernelGGL(k_guard_scatter, dim3((Bnew + 255) / 256, P.T), dim3(256), 0, s, P, D, GP, GB, NV, Bnew);
}

// ---- rows no iterate can change (round 5) ---------------------------------------------------------------------------------------------
// The knots t < t0 are pinned by linear equality rows (q_0 = qc, and q_1 = qc when dq_0 is fixed too), so the inequality rows the builder
// writes for them -- limits (builder.py:471-509), sphere clearances (:366-417), velocity limits on dq_0 = 0 -- are functions of the
// parameters alone.  The kernels leave them out of the iteration; an instance in which one of them is negative has an empty feasible set.
// The reference reports that as did_solve() == False (solver.py:407-412: IPOPT's return_status is not a success) and raises under
// error_on_fail (:133-134).  One thread per instance of the ORIGINAL batch, after the last k_finalize: status <- OH_STATUS_INFEASIBLE,
// kkt[1] <- the worst such row (the literal form's feasibility residual includes it).
__global__ __launch_bounds__(64) void k_guard_infeasible(FigParams P, const oh_chain* __restrict__ ch, GuardParams GP, const double* __restrict__ pin,
                                                         const int B, const int n, double* __restrict__ kkt, int* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* __restrict__ pr = pin + (size_t)b * P.np;
  double worst = 0.0;  // most negative constant row
  if (GP.limits)
    for (int j = 0; j < n; ++j) worst = fmin(worst, fmin(pr[j] - GP.lo[j], GP.up[j] - pr[j]));
  if (GP.vel && P.t0 >= 2)  // dq_0 = 0
    for (int j = 0; j < n; ++j) worst = fmin(worst, fmin(0.0 - GP.vlo[j], GP.vup[j] - 0.0));
  if (GP.n_links > 0) {
    double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0}, p[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < n; ++k) {
      double tv[3], zk[3];
      mv3(R, ch->p0[k], tv);
      p[0] += tv[0]; p[1] += tv[1]; p[2] += tv[2];
      if (!ch->r0ident[k]) {
        double Rn[9];
        mm3(R, ch->R0[k], Rn);
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
      }
      const double qk = pr[ch->qidx[k]];
      if (ch->jtype[k] == 0) {
        double sn, cs;
        sincos_joint(qk, &sn, &cs);
        const int code = ch->axcode[k];
        if (code != 0) rot_principal_right(R, code, sn, cs, zk);
        else rot_axis_right(R, ch->axis[k], sn, cs, zk);
      } else {
        mv3(R, ch->axis[k], zk);
        p[0] += zk[0] * qk; p[1] += zk[1] * qk; p[2] += zk[2] * qk;
      }
      for (int l = 0; l < GP.n_links; ++l) {
        if (GP.link_joint[l] != k) continue;
        double c[3];
        mv3(R, GP.link_off[l], c);
        c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
        const double rl = pr[n + l];
        for (int o = 0; o < GP.n_obs; ++o) {
          const double* ob = pr + n + GP.n_links + 4 * o;
          const double d[3] = {c[0] - ob[0], c[1] - ob[1], c[2] - ob[2]};
          const double rr = rl + ob[3];
          worst = fmin(worst, dot3(d, d) - rr * rr);
        }
      }
    }
  }
  if (worst < -P.tol_feas) {
    if (status) status[b] = OH_STATUS_INFEASIBLE;
    if (kkt) kkt[3 * (size_t)b + 1] = fmax(kkt[3 * (size_t)b + 1], -worst);
  }
}
void oh_launch_guard_infeasible(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const double* p, int B, double* kkt, int* status) {
  if (!kkt && !status) return;
  hipLaunchKernelGGL(k_guard_infeasible, dim3((B + 63) / 64), dim3(64), 0, s, P, D.chain, GP, p, B, n, kkt, status);
}
