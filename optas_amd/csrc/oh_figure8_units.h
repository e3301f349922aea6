// Unit functions of the figure-eight family: what ONE lane does for its (instance, knot) or its instance in each phase of an iteration
// (setup_unit, eval_unit, couple_unit, step_instance, knot_multipliers, finalize_unit).  The __global__ kernels in oh_kernels.hip are thin
// wrappers that map blockIdx / threadIdx to (b, t) and call these.
//
// Platform hooks the including translation unit provides BEFORE this header (oh_platform_gfx950.h for the product):
//   OH_DEV                          function qualifiers (default in oh_device.h: __device__ __forceinline__)
//   RowBuf, rowbuf, rb_ld, rb_st    row addressing of the stage arrays in the Riccati sweep
//   oh_count(unsigned long long*)   event counter increment
//   oh_fence(double)                the value has arrived in a register here: pins a batch of loads ahead of a branch (no-op on the host)
#pragma once
#include "oh_figure8.h"

#define IDX(t, K, k) (((size_t)(t) * (K) + (k)) * Bp + b)
// knot t of an array with K rows per knot; row k of that knot
#define KNOT(arr, t, K) rowbuf((arr) + (size_t)(t) * (K) * (size_t)Bp)
#define RB(k) ((unsigned)(k) * rowB)


// ---------------------------------------------------------------------------------------------
// Figure-eight family.  N = ndof (chain covers all joints in order), NZ = N-3 (orientation locked).
// ---------------------------------------------------------------------------------------------

// setup, one lane per (instance, knot): every lane lays down its knot of the seed (slot 0) with q_0 = q_1 = qc imposed; the lane of knot 0
// also computes the references from qc and resets the solver state.  (Round 1 had one lane per instance walk all T knots: 40 us of a
// 0.38 ms single-instance solve.)
template <int N>
OH_DEV void setup_unit(const FigParams& P, const FigBuffers& D, const double* __restrict__ x0, const double* __restrict__ pin, const int b, const int tt) {
  const int Bp = D.Bp;
  if (b >= D.B) {
    if (b < Bp && tt == 0) D.status[b] = OH_STATUS_CONVERGED;  // padding lanes never run
    return;
  }
  const oh_chain* ch = D.chain;
  double qc[N];
#pragma unroll
  for (int j = 0; j < N; ++j) qc[j] = pin[(size_t)b * P.np + j];
  // knots: slot 0 holds the seed with q_0 = q_1 = qc imposed (linear rows eliminated)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double v = (tt < P.t0) ? qc[j] : x0[(size_t)b * P.nx + (size_t)tt * N + j];
    D.q[0][IDX(tt, N, j)] = v;
    D.q[1][IDX(tt, N, j)] = (tt < P.t0) ? qc[j] : 0.0;
  }
  // p = [qc of the optimised joints (N); lead angle of qc; lead angle of every knot (T)]
  if (ch->has_lead) D.lead[(size_t)tt * Bp + b] = pin[(size_t)b * P.np + N + 1 + tt];
  if (tt != 0) return;
  double R[9], p[3], z[N][3], pj[N][3];
  if (ch->has_lead) {
    double Rb[9], pb[3];
    lead_base(ch, pin[(size_t)b * P.np + N], Rb, pb);
    fk_chain<N, true>(ch, qc, R, p, z, pj, Rb, pb);
  } else {
    fk_chain<N>(ch, qc, R, p, z, pj);
  }
  double e[3], t[3], Re[9];
  mv3(R, ch->p_tool, t);
  e[0] = p[0] + t[0]; e[1] = p[1] + t[1]; e[2] = p[2] + t[2];
  mm3(R, ch->R_tool, Re);
#pragma unroll
  for (int i = 0; i < 3; ++i) D.ref[(size_t)i * Bp + b] = e[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) D.ref[(size_t)(3 + i) * Bp + b] = Re[i];
  // constant cost of the fixed knots t=0,1 (q_0 = q_1 = qc): w * ||Rc local_t||^2
  double fconst = 0.0;
  for (int tf = 0; tf < P.t0 && tf < P.T; ++tf) {
    double l[3] = {P.local_path[3 * tf], P.local_path[3 * tf + 1], P.local_path[3 * tf + 2]};
    fconst += P.w_path * dot3(l, l);  // Rc orthonormal
  }
  D.fconst[b] = fconst;
  D.cur[b] = 1;  // trial slot of launch 0 is slot 0
  D.first[b] = 1;
  D.skip[b] = 0;
  D.polish[b] = 0;
  D.stale[b] = 0;
  D.orig[b] = b;
  D.status[b] = -1;  // running
  D.iters[b] = 0;
  D.f_cur[b] = 0.0;
  D.pred[b] = 0.0;
  D.mu[b] = P.mu0;
  D.nun[b] = 2.0;
  D.stat[b] = 0.0;
  D.feas[b] = 0.0;
}

// Where the evaluation reads the kinematic constants.  The kernels specialised at run time (oh_jit.hip) define this as the address of a
// constexpr copy of the handle's chain, so that every ch->... below folds into the instruction stream: no scalar loads to wait for, no
// branches on the joint hints, no multiplications by the zeros and ones of a URDF.
#ifndef OH_CHAIN
#define OH_CHAIN(D) ((D).chain)
#endif
#ifndef OH_RETRACT_PREFETCH
#define OH_RETRACT_PREFETCH 0
#endif
#ifndef OH_EVALB_PREFETCH_G
#define OH_EVALB_PREFETCH_G 1
#endif
// ZC: 1 = the neighbours' knots are requested with the lane's other inputs, ahead of the early-exit branch, and condensed at once into the
// coupling term of G (N doubles) and ||q_t - q_{t-1}||^2; 0 = they are fetched where g is final (a dependent round trip in mid-kernel:
// k_eval 45.7 against 42.7 ms per bench step, A/B on one box)
#ifndef OH_ZC_EARLY
#define OH_ZC_EARLY 1
#endif
// Householder vectors of knot t from their packed stage array ([t][3N - 3][Bp], written by eval_unit)
template <int N>
OH_DEV void load_householder(const double* __restrict__ Vs, const int Bp, const int b, const int t, double (&V)[3][N]) {
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int k = 0; k < N; ++k) V[m][k] = (k < m) ? 0.0 : Vs[IDX(t, HV_ROWS(N), HV_OFF(N, m) + k - m)];
}

// K2: one lane per (instance b, free knot t): trial knot, retraction onto R(q_t)=Rc, FK chain + Jacobians,
// tracking cost / gradient / Hessian block, null-space basis of the orientation rows, reduced block
// (eval_knot in oh_figure8.h).
// ZC (EVAL_ONLY only; "coupling folded in", round 3): the lane also forms what k_couple used to add in a launch of its own and that needs no
// null-space basis of a neighbour -- the Lagrangian gradient G_t = g_t + 2 kappa (2 q_t - q_{t-1} - q_{t+1}) and the merit share phi_t + kappa
// ||q_t - q_{t-1}||^2, from the neighbours' retracted knots (in HBM since k_retract) -- and stores G where it stored g, the merit where it
// stored phi: same bytes out, 2N doubles more in, and the sweep (step_instance_zc) rebuilds E_t and gt_t from V and G on the fly.
// SPH = false (GUARD only): the handle has no sphere rows -- their walk over the links is compiled out (86 registers of a kernel at the limit)
template <int N, bool GUARD = false, bool LEAD = false, int MODE = EVAL_FUSED, bool ZC = false, bool SPH = true>
OH_DEV void eval_unit(const FigParams& P, const FigBuffers& D, const int slot, const int b, const int t, const GuardParams* GPp = nullptr,
                      const GuardBuffers* GBp = nullptr) {
  static_assert(!ZC || (MODE == EVAL_ONLY && !GUARD && !LEAD), "the folded coupling belongs to the plain batched evaluation");
  constexpr int NZ = N - 3;
  constexpr int NP = NZ * (NZ + 1) / 2;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  // Uniform slots: every running instance writes this launch's trial into `slot` and keeps its accepted
  // point in `cur` = 1 - slot, so all lanes of a wavefront touch the same arrays (full 512-B lines).  An
  // instance whose previous trial was rejected has its accepted point in `slot`: it sits this launch out
  // (skip flag) and is back in phase at the next one -- cheaper than moving its stage data.
  const int cur = 1 - slot;
  // ONE memory round trip before the arithmetic starts: everything the lane will need is requested before any of it is looked at
  // (status -> first -> knot data as dependent loads cost three round trips of ~1-2 us each at the start of a wave that lives ~12 us;
  // the SQ counters showed a third of every k_evalb wave's lifetime in s_waitcnt).  Finished / skipping lanes fetch for nothing; the
  // batch is compacted when a tenth of it has finished.
  const int status_b = D.status[b], skip_b = D.skip[b], first_b = D.first[b];
  // polish == 2 (set by step_instance_zc, see there): the sweep of the accepted point did not factorise at its damping; this launch lays the accepted
  // point down as the trial AS IT IS (no step, no retraction), the evaluation rebuilds its stage data bit for bit and the next sweep runs at the raised damping
  const bool asis = (MODE == EVAL_RETRACT_ONLY && !GUARD) ? D.polish[b] == 2 : false;
  const double stat_b = D.stat[b], pred_b = D.pred[b];
  constexpr bool EARLY = MODE == EVAL_ONLY || OH_RETRACT_PREFETCH;  // the generic retraction kernel sits at the register limit: it fetches
                                                                    // its knot data after the branch, as before
  double Rc[9], pc[3];
  if constexpr (EARLY) {
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = D.ref[(size_t)i * Bp + b];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = D.ref[(size_t)(3 + i) * Bp + b];
  }
  double q[N], e_tgt[3] = {0.0, 0.0, 0.0};
  double sm_zc = 0.0;  // ||q_t - q_{t-1}||^2 (ZC)
  double zs[NZ], Vc[3][N], mdlc[3 + 3 * NZ];
  double Gpre[N];  // Lagrangian gradient of the accepted point: wanted deep inside the evaluation (exact-curvature branch), requested here
  constexpr bool PRE_G = OH_EVALB_PREFETCH_G && MODE == EVAL_ONLY && !GUARD;
  if constexpr (MODE == EVAL_ONLY) {  // the retracted trial knot is already in the slot (k_retract)
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = D.q[slot][IDX(t, N, j)];
    if constexpr (PRE_G) {
      if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) {
#pragma unroll
        for (int k = 0; k < N; ++k) Gpre[k] = D.Gfull[cur][IDX(t, N, k)];
      }
    }
  } else if constexpr (OH_RETRACT_PREFETCH) {  // the accepted knot, the reduced step and the model of the step (a first evaluation replaces q below: rare)
#pragma unroll
    for (int a = 0; a < NZ; ++a) zs[a] = D.zstep[IDX(t, NZ, a)];
    load_householder<N>(D.Z[cur], Bp, b, t, Vc);
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = D.q[cur][IDX(t, N, j)];
#pragma unroll
    for (int i = 0; i < 3 + 3 * NZ; ++i) mdlc[i] = D.mdl[cur][IDX(t, MDL_ROWS(N), i)];
  }
  double cpl_zc[N];  // ZC, OH_ZC_EARLY: 2 kappa ((q_t - q_{t-1}) - (q_{t+1} - q_t))
  if constexpr (ZC && OH_ZC_EARLY) {
    const double* __restrict__ qs = D.q[slot];
    const bool lastk = (t == P.T - 1);
    double qm[N], qp[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      qm[k] = qs[IDX(t - 1, N, k)];
      qp[k] = lastk ? 0.0 : qs[IDX(t + 1, N, k)];
    }
    oh_fence(q[N - 1]);
    oh_fence(Rc[8]);
    oh_fence(qp[N - 1]);
    const double kap2 = 2.0 * P.kappa;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double dm = q[k] - qm[k];
      sm_zc += dm * dm;
      double c = kap2 * dm;
      if (!lastk) c -= kap2 * (qp[k] - q[k]);
      cpl_zc[k] = c;
    }
  } else if constexpr (EARLY) {
    oh_fence(q[N - 1]);
    oh_fence(Rc[8]);
  }
  if (status_b >= 0 || skip_b) return;
  const bool first = first_b != 0;
  // trial knot: the seed on the first evaluation, otherwise q_cur + Z_cur z (roll-out of the step k_step solved for)
  if constexpr (MODE != EVAL_ONLY) {
    if (first) {
#pragma unroll
      for (int j = 0; j < N; ++j) q[j] = D.q[slot][IDX(t, N, j)];
    } else {
      if constexpr (!OH_RETRACT_PREFETCH) {  // the generic kernels sit at the register limit: their knot data is fetched after the branch
#pragma unroll
        for (int a = 0; a < NZ; ++a) zs[a] = D.zstep[IDX(t, NZ, a)];
        load_householder<N>(D.Z[cur], Bp, b, t, Vc);
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = D.q[cur][IDX(t, N, j)];
#pragma unroll
        for (int i = 0; i < 3 + 3 * NZ; ++i) mdlc[i] = D.mdl[cur][IDX(t, MDL_ROWS(N), i)];
      }
      double Zc[N][NZ];
      z_from_householder<N>(Vc, Zc);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        double v = q[j];
#pragma unroll
        for (int a = 0; a < NZ; ++a) v += Zc[j][a] * zs[a];
        q[j] = asis ? q[j] : v;
      }
      // where the linear model puts the end effector after this step: e_cur + (Jp Z)_cur z
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double v = mdlc[m];
#pragma unroll
        for (int a = 0; a < NZ; ++a) v += mdlc[3 + m * NZ + a] * zs[a];
        e_tgt[m] = v;
      }
    }
  }
  if constexpr (!EARLY) {
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = D.ref[(size_t)i * Bp + b];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = D.ref[(size_t)(3 + i) * Bp + b];
  }
  double Gprev[N];
  // exact curvature: always (OH_HESSIAN_EXACT) or once the accepted point is nearly stationary (OH_HESSIAN_HYBRID)
  const bool fresh = first_b == 1;  // a seed: nothing is known about the point (first_b == 2: a restart after a compaction, its gradients came along)
  bool exact = (P.hessian == OH_HESSIAN_EXACT) || (P.hessian == OH_HESSIAN_HYBRID && !fresh && stat_b <= P.hyb_switch);
  if constexpr (GUARD) {
    // the exact block carries no curvature of the sphere rows (-s d2g, s ~ w_path): with them the exact model is worse than
    // Gauss-Newton (the oracle run crawls), so sphere-guarded problems stay on Gauss-Newton
    if (SPH && GPp->n_links > 0) exact = false;
  }
  const bool have_G = exact && !fresh;
#pragma unroll
  for (int k = 0; k < N; ++k) Gprev[k] = 0.0;  // fetched by the hook below, inside the exact-curvature branch

  double phi, cv, g[N], Dr[NP], Z[N][NZ];
  struct Hooks {
    double* __restrict__ qo;
    double* __restrict__ go;
    double* __restrict__ vo;
    const double* __restrict__ Gc;
    int Bp, b, t;
    OH_DEV void q_final(const double (&qv)[N]) const {
      if constexpr (!GUARD && MODE != EVAL_ONLY) {
#pragma unroll
        for (int j = 0; j < N; ++j) qo[IDX(t, N, j)] = qv[j];
      }
    }
    OH_DEV void g_final(const double (&gv)[N]) const {
      if constexpr (ZC && OH_ZC_EARLY) {
#pragma unroll
        for (int k = 0; k < N; ++k) Go[IDX(t, N, k)] = gv[k] + cpl[k];
      } else if constexpr (ZC) {
        // couple_knot's G and merit share, operation for operation (qs: this launch's knots, q_{t-1} of a fixed knot included)
        double sm = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const double q0 = qr[k];
          const double dm = q0 - qs[IDX(t - 1, N, k)];
          sm += dm * dm;
          double Gk = gv[k] + kap2 * dm;
          if (!last) Gk -= kap2 * (qs[IDX(t + 1, N, k)] - q0);
          Go[IDX(t, N, k)] = Gk;
        }
        *smo = sm;
      } else if constexpr (!GUARD) {  // the guard rows still add to g
#pragma unroll
        for (int k = 0; k < N; ++k) go[IDX(t, N, k)] = gv[k];
      }
    }
    OH_DEV void v_final(const double (&Vv)[3][N]) const {
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int k = m; k < N; ++k) vo[IDX(t, HV_ROWS(N), HV_OFF(N, m) + k - m)] = Vv[m][k];
    }
    const double* Gp;  // prefetched copy (registers), or null
    OH_DEV void load_G(const double (&)[N], double (&G)[N]) const {
#pragma unroll
      for (int k = 0; k < N; ++k) G[k] = PRE_G ? Gp[k] : Gc[IDX(t, N, k)];
    }
    // ZC
    const double* __restrict__ qs;
    const double* qr;  // this knot, in registers
    double* __restrict__ Go;
    double* smo;
    double kap2;
    bool last;
    const double* cpl;
  };
  const Hooks hooks{D.q[slot], D.g[slot], D.Z[slot], D.Gfull[cur], Bp, b, t, Gpre, D.q[slot], q, D.Gfull[slot], &sm_zc, 2.0 * P.kappa, t == P.T - 1, cpl_zc};
  double e_new[3], JZ_new[3][NZ];
  // A restart after a compaction (first_b == 2) evaluates the accepted point AS IT IS: its knots are the retracted
  // knots the instance accepted, and retracting them again (to the floor tolerance, as a seed would be) moved them by ~1e-10 -- enough to send an
  // instance between two basins down another path than the same instance takes alone (round 3's "not batch-invariant", HISTORY).  Without the
  // retraction the stage data of the restart are those of the accepted point bit for bit and the interrupted step is re-derived exactly.
  const double tol_r = (first_b == 2 || asis) ? 1e300 : retract_tol(P, !first, pred_b, stat_b);
  if constexpr (LEAD)
    eval_knot<N, true, Hooks, MODE>(OH_CHAIN(D), P, t, q, pc, Rc, exact, have_G, Gprev, phi, cv, g, Dr, Z, !first, e_tgt, tol_r, e_new, JZ_new,
                                    D.lead[(size_t)t * Bp + b], hooks);
  else eval_knot<N, false, Hooks, MODE>(OH_CHAIN(D), P, t, q, pc, Rc, exact, have_G, Gprev, phi, cv, g, Dr, Z, !first, e_tgt, tol_r, e_new, JZ_new, 0.0, hooks);
  if constexpr (MODE == EVAL_RETRACT_ONLY) return;  // q is in the slot; everything else is k_evalb's
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    D.mdl[slot][IDX(t, MDL_ROWS(N), m)] = e_new[m];
#pragma unroll
    for (int a = 0; a < NZ; ++a) D.mdl[slot][IDX(t, MDL_ROWS(N), 3 + m * NZ + a)] = JZ_new[m][a];
  }
  if constexpr (GUARD) {
    // inequality rows through the same augmented Lagrangian as the position-tracking family (oh_free.hip), added after the
    // retraction: joint limits q - lo >= 0, up - q >= 0 (enforce_model_limits, builder.py:471-509) have gradients +-e_j, so W gains
    // a diagonal d_j and the reduced block Z^T diag(d) Z; a sphere row (builder.py:366-417) adds rho (Z^T dg)(Z^T dg)^T.
    const GuardParams& GP = *GPp;
    const GuardBuffers& GB = *GBp;
    const bool upd = GB.outer[b] != 0;
    const double rho_old = GB.rho[b];
    const double rho = upd ? GB.rho_next[b] : rho_old;
    const double irho = 1.0 / rho, i2rho = 1.0 / (2.0 * rho);
    double psi = 0.0, meas = 0.0, dd[N];
    const int nl = GP.limits ? 2 * N : 0;
#pragma unroll
    for (int j = 0; j < N; ++j) dd[j] = 0.0;
    if (GP.limits) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const double gval = side ? GP.up[j] - q[j] : q[j] - GP.lo[j];
        double* lam_ptr = GB.lam + IDX(t, GP.NC, side * N + j);
        double lam = *lam_ptr;
        if (upd) {
          lam = fmax(0.0, lam - rho_old * gval);
          *lam_ptr = lam;
        }
        const double sv = lam - rho * gval;
        meas = fmax(meas, fabs(fmin(gval, lam * irho)));
        if (sv > 0.0) {
          psi += (sv * sv - lam * lam) * i2rho;
          g[j] += side ? sv : -sv;
          dd[j] += rho;
        } else {
          psi -= lam * lam * i2rho;
        }
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
    if (SPH && GP.n_links > 0) {
      sphere_rows_walk<N>(D.chain, GP, GB.par, (size_t)Bp, b, q, [&](const int l, const int o, const double gval, const double (&dg)[N]) {
        double* lam_ptr = GB.lam + IDX(t, GP.NC, nl + l * GP.n_obs + o);
        double lam = *lam_ptr;
        if (upd) {
          lam = fmax(0.0, lam - rho_old * gval);
          *lam_ptr = lam;
        }
        const double sv = lam - rho * gval;
        meas = fmax(meas, fabs(fmin(gval, lam * irho)));
        if (sv > 0.0) {
          psi += (sv * sv - lam * lam) * i2rho;
          double v[NZ];
#pragma unroll
          for (int a = 0; a < NZ; ++a) v[a] = 0.0;
#pragma unroll
          for (int j = 0; j < N; ++j) {
            g[j] -= sv * dg[j];
#pragma unroll
            for (int a = 0; a < NZ; ++a) v[a] += Z[j][a] * dg[j];
          }
#pragma unroll
          for (int a = 0; a < NZ; ++a)
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2) Dr[tri(a, c2)] += rho * v[a] * v[c2];
        } else {
          psi -= lam * lam * i2rho;
        }
      });
    }
    phi += psi;
    GB.psi[slot][(size_t)t * Bp + b] = psi;
    GB.mcv[slot][(size_t)t * Bp + b] = meas;
  }

  if constexpr (GUARD) {
#pragma unroll
    for (int j = 0; j < N; ++j) D.q[slot][IDX(t, N, j)] = q[j];
#pragma unroll
    for (int k = 0; k < N; ++k) D.g[slot][IDX(t, N, k)] = g[k];
  }
  if constexpr (ZC) D.merit[slot][(size_t)t * Bp + b] = phi + P.kappa * sm_zc;
  else D.phi[slot][(size_t)t * Bp + b] = phi;
  D.cv[slot][(size_t)t * Bp + b] = cv;
#pragma unroll
  for (int i = 0; i < NP; ++i) D.Dr[slot][IDX(t, NP, i)] = Dr[i];
}

// K2b: one lane per (instance b, free knot t), after k_eval: everything of the reduced block-tridiagonal
// system that needs the neighbouring knots but not the recursion (couple_knot in oh_figure8.h).
// (velocity_rows: oh_figure8.h)
// multiplier refresh of the velocity rows at an outer update (before k_couple of the same iteration reads them; a launch of its own so that
// no lane reads a neighbour's row block while it is being rewritten): lam <- max(0, lam - rho_old g) at the re-evaluated accepted point
template <int N>
OH_DEV void vel_update_unit(const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, const int slot, const int b, const int t) {
  const int Bp = D.Bp;
  if (b >= D.B) return;
  if (D.status[b] >= 0 || D.skip[b] || !GB.outer[b]) return;
  const double rho = GB.rho[b] * GP.vscale, idt = 1.0 / P.dt;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double v = (D.q[slot][IDX(t, N, k)] - D.q[slot][IDX(t - 1, N, k)]) * idt;
    double* l_lo = GB.lamv + IDX(t, 2 * N, k);
    double* l_up = GB.lamv + IDX(t, 2 * N, N + k);
    *l_lo = fmax(0.0, *l_lo - rho * (v - GP.vlo[k]));
    *l_up = fmax(0.0, *l_up - rho * (GP.vup[k] - v));
  }
}

template <int N, bool VEL = false>
OH_DEV void couple_unit(const FigParams& P, const FigBuffers& D, const int slot, const int b, const int t, const GuardParams* GPp = nullptr,
                        const GuardBuffers* GBp = nullptr) {
  constexpr int NZ = N - 3;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  if (D.status[b] >= 0 || D.skip[b]) return;
  const double* __restrict__ qs = D.q[slot];
  const double* __restrict__ Zs = D.Z[slot];
  const bool last = (t == P.T - 1);
  double qm[N], q0[N], qp[N], g[N], Zt[N][NZ], Zn[N][NZ];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    qm[k] = qs[IDX(t - 1, N, k)];
    q0[k] = qs[IDX(t, N, k)];
    qp[k] = last ? 0.0 : qs[IDX(t + 1, N, k)];
    g[k] = D.g[slot][IDX(t, N, k)];
  }
  {
    double Vt[3][N];
    load_householder<N>(Zs, Bp, b, t, Vt);
    z_from_householder<N>(Vt, Zt);
    if (!last) {
      load_householder<N>(Zs, Bp, b, t + 1, Vt);
      z_from_householder<N>(Vt, Zn);
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int a = 0; a < NZ; ++a) Zn[k][a] = 0.0;
    }
  }
  double G[N], gt[NZ], E[NZ * NZ], merit;
  double wn[N];  // Gauss-Newton weight of the velocity rows of interval (t, t+1)
  if constexpr (VEL) {
    // intervals (t-1, t) and (t, t+1): their augmented-Lagrangian gradient goes into g before the projection, the value of (t-1, t) is booked
    // on knot t like kappa ||q_t - q_{t-1}||^2, the weights join 2 kappa in E_t and add Z_t^T diag(w_prev + w_next) Z_t to the diagonal block
    const GuardParams& GP = *GPp;
    const GuardBuffers& GB = *GBp;
    const double rho = (GB.outer[b] ? GB.rho_next[b] : GB.rho[b]) * GP.vscale;
    double lam[2 * N], sp[N], wp[N], sn[N], psi_p, meas_p, psi_n, meas_n;
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
    bool any = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
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
          for (int k = 0; k < N; ++k) acc += (wp[k] + wn[k]) * Zt[k][a] * Zt[k][c2];
          D.Dr[slot][IDX(t, NZ * (NZ + 1) / 2, tri(a, c2))] += acc;
        }
    }
    D.phi[slot][(size_t)t * Bp + b] += psi_p;  // couple_knot folds phi into the merit
    GB.psi[slot][(size_t)t * Bp + b] += psi_p;
    GB.mcv[slot][(size_t)t * Bp + b] = fmax(GB.mcv[slot][(size_t)t * Bp + b], meas_p);
  }
  couple_knot<N>(P.kappa, last, qm, q0, qp, g, Zt, Zn, D.phi[slot][(size_t)t * Bp + b], G, gt, E, merit);
  if constexpr (VEL) {
    if (!last) {
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int a = 0; a < NZ; ++a)
#pragma unroll
          for (int c2 = 0; c2 < NZ; ++c2) E[a * NZ + c2] -= wn[k] * Zt[k][a] * Zn[k][c2];
    }
  }
  if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) {
#pragma unroll
    for (int k = 0; k < N; ++k) D.Gfull[slot][IDX(t, N, k)] = G[k];
  }
#pragma unroll
  for (int a = 0; a < NZ; ++a) D.gt[slot][IDX(t, NZ, a)] = gt[a];
  if (!last) {
#pragma unroll
    for (int i = 0; i < NZ * NZ; ++i) D.E[slot][IDX(t, NZ * NZ, i)] = E[i];
  }
  D.merit[slot][(size_t)t * Bp + b] = merit;
}

#ifndef OH_STEP_PREFETCH_BACK
#define OH_STEP_PREFETCH_BACK 1
#endif
#ifndef OH_STEP_PREFETCH_FWD
#define OH_STEP_PREFETCH_FWD 1
#endif
// The acceptance phase of K3 (shared by step_instance and step_instance_zc): Levenberg-Marquardt ratio test of the trial slot against the
// accepted point, restart / polish / line-search bookkeeping.  Returns 0: the instance has finished (status set), 1: it goes on without a
// sweep (restart, polish, shorter trial along the rejected step), 2: sweep on the slot `cur` with damping lm.mu.
// slot of a two-slot array by selection: an index that is not a constant makes the compiler keep a private copy of the whole argument
// struct (k_step_lg: 440 B of scratch per lane, one wavefront per SIMD instead of two, 300 instead of 90 us per latency-bound launch)
template <class T_>
OH_DEV T_* slot_of(T_* const (&a)[2], const int s) { return s ? a[1] : a[0]; }
struct StepHead { int go; int cur; LMState lm; };
template <int N, bool GUARD = false>
OH_DEV StepHead step_head(const FigParams& P, const FigBuffers& D, const int b, const int ts, const GuardBuffers* GBp, int cur, LMState lm, const int iters) {
  constexpr int NZ = N - 3;
  const int Bp = D.Bp;
  const int T = P.T;
  const unsigned lb = (unsigned)b * 8u;      // this lane's byte offset inside a row
  const unsigned rowB = (unsigned)Bp * 8u;  // bytes per row
  bool polish_request = false;
  bool line_search = false;

  // ---- phase A: merit of the trial slot ---------------------------------------------------------
  {
    double f = D.fconst[b];
    double feas = 0.0, fpsi = 0.0, meas = 0.0;
    const double* const merit_ts = slot_of(D.merit, ts);
    const double* const cv_ts = slot_of(D.cv, ts);
    const double* psi_ts = nullptr;
    const double* mcv_ts = nullptr;
    if constexpr (GUARD) {
      psi_ts = slot_of(GBp->psi, ts);
      mcv_ts = slot_of(GBp->mcv, ts);
    }
    for (int t = P.t0; t < T; ++t) {
      f += rb_ld(KNOT(merit_ts, t, 1), 0, lb);
      feas = fmax(feas, rb_ld(KNOT(cv_ts, t, 1), 0, lb));
      if constexpr (GUARD) {
        fpsi += rb_ld(KNOT(psi_ts, t, 1), 0, lb);
        meas = fmax(meas, rb_ld(KNOT(mcv_ts, t, 1), 0, lb));
      }
    }
    bool accept;
    if (D.first[b]) {
      if (!(f == f) || !(fabs(f) < 1e300)) {  // non-finite seed / parameters: report, do not iterate (fmax would hide the NaN)
        D.status[b] = OH_STATUS_NUMERICAL;
        D.cur[b] = ts;
        D.f_cur[b] = f;
        D.stat[b] = f;
        return StepHead{0, cur, lm};
      }
      accept = true;
      D.first[b] = 0;
      if constexpr (GUARD) {
        // restart after a batch compaction with a multiplier update pending: this evaluation has refreshed the multipliers
        if (GBp->outer[b]) {
          GBp->outer[b] = 0;
          GBp->rho[b] = GBp->rho_next[b];
        }
      }
    } else if (GUARD && GBp->outer[b]) {
      // re-evaluation of the accepted point after a multiplier update: the merit function itself changed
      accept = true;
      GBp->outer[b] = 0;
      GBp->rho[b] = GBp->rho_next[b];
    } else if (D.polish[b]) {
      accept = true;  // the accepted point itself, re-retracted to the floor tolerance
      D.polish[b] = 0;
    } else {
      const LMState lm_before = lm;
      accept = lm_accept(P, f, feas, D.f_cur[b], D.pred[b], D.stat[b], lm, D.feas[b]);
      // A rejected trial against an accepted point that was retracted loosely (retract_tol): its objective is off by (multiplier) x
      // violation, and steps that predict less than that can never be accepted.  Before blaming the model, re-evaluate the accepted
      // point at the floor tolerance: zero step, accepted unconditionally at the next k_step.
      polish_request = !accept && !D.stale[b] && D.feas[b] > 10.0 * retract_tol(P, false, D.pred[b], 0.0);
      if (polish_request) lm = lm_before;
      if constexpr (GUARD) {
        // handles with inequality rows: a shorter step along the same direction before the damping is raised (OH_LS_MAX, oh_types.h)
        if (!accept && !polish_request && GBp->ls_count[b] < OH_LS_MAX && iters < P.max_iter / 2) {  // (first half of the budget only, see free_accept)
          lm = lm_before;
          line_search = true;
        }
      }
      D.nun[b] = lm.nun;
    }
    if (!accept && D.stale[b]) {
      // the accepted point's stage data did not survive the last compaction (k_carry_*): restart from its knots, which wait in the
      // next trial slot; the rejection has updated the LM state, the step is re-derived (and counted) at the restart
      D.stale[b] = 0;
      D.first[b] = 1;
      D.polish[b] = 0;
      D.mu[b] = lm.mu;
      D.nun[b] = lm.nun;
      D.cur[b] = cur;
      oh_count(D.work + 1);
      return StepHead{1, cur, lm};
    }
    if (accept) {
      D.stale[b] = 0;
      cur = ts;
      D.f_cur[b] = f;
      D.feas[b] = feas;
      if constexpr (GUARD) {
        D.fpsi[b] = fpsi;
        GBp->meas[b] = meas;
        GBp->ls_count[b] = 0;
      }
    }
    D.cur[b] = cur;
    if (!accept) {  // the accepted point sits where the next trial would go: sit the next launch out
      D.skip[b] = 1;
      oh_count(D.work + 1);
    }
  }
  if constexpr (GUARD) {
    if (line_search) {  // the rejected step again, shorter: no sweep, the damping untouched
      if (iters >= P.max_iter) {
        D.status[b] = OH_STATUS_MAX_ITER;
        return StepHead{0, cur, lm};
      }
      for (int t = P.t0; t < T; ++t) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) rb_st(KNOT(D.zstep, t, NZ), RB(a), lb, OH_LS_SHRINK * rb_ld(KNOT(D.zstep, t, NZ), RB(a), lb));
      }
      const int k = GBp->ls_count[b] + 1;
      GBp->ls_count[b] = k;
      double sk = 1.0;
      for (int i = 0; i < k; ++i) sk *= OH_LS_SHRINK;
      D.pred[b] = -sk * GBp->ls_gd[b] + 0.5 * sk * sk * GBp->ls_q[b];
      D.iters[b] = iters + 1;
      return StepHead{1, cur, lm};
    }
  }
  if (polish_request) {
    for (int t = P.t0; t < T; ++t) {
#pragma unroll
      for (int a = 0; a < NZ; ++a) rb_st(KNOT(D.zstep, t, NZ), RB(a), lb, 0.0);
    }
    D.pred[b] = 0.0;
    D.polish[b] = 1;
    D.iters[b] = iters + 1;
    if (iters < P.max_iter + 40) return StepHead{1, cur, lm};
    D.status[b] = OH_STATUS_MAX_ITER;
    return StepHead{0, cur, lm};
  }
  return StepHead{2, cur, lm};
}

// K3: one lane per instance: accept/reject the trial point (Levenberg-Marquardt ratio test on the
// objective; iterates are feasible by retraction), then the backward Riccati sweep over the reduced
// block-tridiagonal system (blocks prepared by k_eval/k_couple, next knot's blocks prefetched while the
// current knot factorises) and the forward recursion for the reduced step z_t.  Returns "still running".
template <int N, bool GUARD = false>
OH_DEV bool step_instance(const FigParams& P, const FigBuffers& D, const int b, const int ts, const GuardBuffers* GBp = nullptr) {
  constexpr int NZ = N - 3;
  constexpr int NP = NZ * (NZ + 1) / 2;
  const int Bp = D.Bp;
  const int T = P.T;
  const unsigned lb = (unsigned)b * 8u;      // this lane's byte offset inside a row
  const unsigned rowB = (unsigned)Bp * 8u;  // bytes per row
  const double kap2 = 2.0 * P.kappa;
  int cur = 1 - ts;  // uniform-slot invariant (see k_eval): the accepted point is in the other slot
  LMState lm{D.mu[b], D.nun[b]};
  const int iters = D.iters[b];
  {
    const StepHead hd = step_head<N, GUARD>(P, D, b, ts, GBp, cur, lm, iters);
    if (hd.go != 2) return hd.go == 1;
    cur = hd.cur;
    lm = hd.lm;
  }
  double mu = lm.mu;

  // ---- phase B: backward sweep on the current slot ------------------------------------------------
  // the accepted slot differs from lane to lane (a rejected trial leaves it where it was): the two slots of an array are adjacent in
  // the pool, so the slot goes into the lane offset and the row pointers stay uniform
  const double* __restrict__ Ec = D.E[0];
  const double* __restrict__ Drc = D.Dr[0];
  const double* __restrict__ gtc = D.gt[0];
  const unsigned oE = lb + (cur ? (unsigned)((const char*)D.E[1] - (const char*)D.E[0]) : 0u);
  const unsigned oD = lb + (cur ? (unsigned)((const char*)D.Dr[1] - (const char*)D.Dr[0]) : 0u);
  const unsigned oG = lb + (cur ? (unsigned)((const char*)D.gt[1] - (const char*)D.gt[0]) : 0u);
  double stat = 0.0;
  double S[NP], rd[NZ], rn[NZ];
  bool factored = false;
  double quad = 0.0;  // plain handles: g^T M^{-1} g, by-product of the backward substitutions (see step_instance_zc)
  for (int attempt = 0; attempt < 40; ++attempt) {
    bool ok = true;
    stat = 0.0;
    quad = 0.0;
    // last knot: S = Dr + (kap2 + mu) I, r = gt
    {
      const int t = T - 1;
#pragma unroll
      for (int i = 0; i < NP; ++i) S[i] = rb_ld(KNOT(Drc, t, NP), RB(i), oD);
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        S[tri(a, a)] += kap2 + mu;
        rn[a] = rb_ld(KNOT(gtc, t, NZ), RB(a), oG);
        stat = fmax(stat, fabs(rn[a]));
      }
    }
    // The sweep is a dependent chain per lane: a knot's blocks have to be in registers when the previous knot's factor is done, and one
    // knot of arithmetic (~0.4 us) hides a fraction of a memory round trip under load (~2 us).  PFB knots are kept in flight: buffer j
    // is refilled with knot t - PFB the moment knot t has been copied out of it (static indices: the knot loop is unrolled PFB-fold).
    constexpr int PFB = OH_STEP_PREFETCH_BACK;
    double nE[PFB][NZ * NZ], nH[PFB][NP], ng[PFB][NZ];
    auto fetch = [&](const int j, const int t) {
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) nE[j][i] = rb_ld(KNOT(Ec, t, NZ * NZ), RB(i), oE);
#pragma unroll
      for (int i = 0; i < NP; ++i) nH[j][i] = rb_ld(KNOT(Drc, t, NP), RB(i), oD);
#pragma unroll
      for (int a = 0; a < NZ; ++a) ng[j][a] = rb_ld(KNOT(gtc, t, NZ), RB(a), oG);
    };
#pragma unroll
    for (int j = 0; j < PFB; ++j)
      if (T - 2 - j >= P.t0) fetch(j, T - 2 - j);
    for (int tb = T - 2; tb >= P.t0; tb -= PFB) {
#pragma unroll
      for (int j = 0; j < PFB; ++j) {
        const int t = tb - j;
        if (t >= P.t0) {
          double E[NZ * NZ], Ht[NP], gt[NZ];
#pragma unroll
          for (int i = 0; i < NZ * NZ; ++i) E[i] = nE[j][i];
#pragma unroll
          for (int i = 0; i < NP; ++i) Ht[i] = nH[j][i];
#pragma unroll
          for (int a = 0; a < NZ; ++a) gt[a] = ng[j][a];
          if (t - PFB >= P.t0) fetch(j, t - PFB);  // issued before the dependent arithmetic of this knot
#pragma unroll
          for (int a = 0; a < NZ; ++a) {
            stat = fmax(stat, fabs(gt[a]));
            Ht[tri(a, a)] += 2.0 * kap2 + mu;
          }
          double Kmat[NZ * NZ], kv[NZ];
          ok = riccati_back<NZ>(S, rd, rn, E, Ht, gt, Kmat, kv, GUARD ? nullptr : &quad) && ok;
#pragma unroll
          for (int a = 0; a < NZ; ++a) rb_st(KNOT(D.kvec, t + 1, NZ), RB(a), lb, kv[a]);
#pragma unroll
          for (int i = 0; i < NZ * NZ; ++i) rb_st(KNOT(D.Kmat, t + 1, NZ * NZ), RB(i), lb, Kmat[i]);
        }
      }
    }
    ok = chol_rcp<NZ>(S, rd, 1e-12) && ok;
    if (ok) {
      factored = true;
      break;
    }
    mu = fmax(4.0 * mu, 1e-2);
  }
  D.stat[b] = stat;
  if (!(stat == stat) || !factored) {  // NaN in the reduced gradient, or no damping (up to 4^40) made the reduced Hessian factorisable
    D.status[b] = OH_STATUS_NUMERICAL;
    D.mu[b] = mu;
    return false;
  }
  const double feas_cur = D.feas[b];
  if constexpr (GUARD) {
    const GuardBuffers& GB = *GBp;
    if (stat <= GB.omega[b]) {
      const double meas = GB.meas[b];
      if (stat <= P.tol && feas_cur <= P.tol_feas && meas <= P.tol_feas) {
        D.status[b] = OH_STATUS_CONVERGED;
        D.mu[b] = mu;
        return false;
      }
      if (iters >= P.max_iter) {
        D.status[b] = OH_STATUS_MAX_ITER;
        D.mu[b] = mu;
        return false;
      }
      // outer iteration (see step_instance_free): refresh the multipliers at the accepted point, tighten the inner tolerance
      const double rho = GB.rho[b];
      GB.rho_next[b] = (meas > 0.25 * GB.meas_prev[b]) ? fmin(10.0 * rho, 1e8) : rho;
      GB.meas_prev[b] = meas;
      GB.omega[b] = fmax(P.tol, fmin(GB.omega[b], 0.1 * meas));
      GB.outer[b] = 1;
      GB.n_outer[b] += 1;
      for (int t = P.t0; t < T; ++t) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) rb_st(KNOT(D.zstep, t, NZ), RB(a), lb, 0.0);
      }
      D.pred[b] = 0.0;
      D.mu[b] = mu;
      D.iters[b] = iters + 1;
      return true;
    }
  } else {
    if (stat <= P.tol && feas_cur <= P.tol_feas) {
      D.status[b] = OH_STATUS_CONVERGED;
      D.mu[b] = mu;
      return false;
    }
  }
  if (iters >= P.max_iter) {
    D.status[b] = OH_STATUS_MAX_ITER;
    D.mu[b] = mu;
    return false;
  }

  // ---- forward recursion: z_2 = -S_2^{-1} r_2, z_{t+1} = -(kvec + Kmat z_t) ----------------------------
  {
    double zz[NZ];
#pragma unroll
    for (int a = 0; a < NZ; ++a) zz[a] = -rn[a];
    fsub_rcp<NZ>(S, rd, zz);
    if constexpr (!GUARD) {
#pragma unroll
      for (int a = 0; a < NZ; ++a) quad = fma(zz[a], zz[a], quad);  // the first free knot's share
    }
    bsub_rcp<NZ>(S, rd, zz);
    // plain handles: the directional derivative g.z = -g^T M^{-1} g comes out of the backward sweep, the reduced gradients are not read a second time
    double gd = GUARD ? 0.0 : -quad, z2 = 0.0;
    // Over-relaxation in the Gauss-Newton phase of the hybrid scheme: the tracking residual does not vanish (f* ~ 8), Gauss-Newton
    // over-estimates the curvature along the valley and its full steps, although accepted with mu = 0, crawl (10 of the 17 steps of a
    // typical instance go by between the first step and the switch to exact curvature).  Taking alpha z instead (alpha = 1.5 from the
    // fourth step on, ratio test against the model's own prediction for alpha z) cuts the mean step count by 13 % on the numpy port
    // (96 instances: 16.4 -> 14.2); larger alpha or an earlier start cost more rejections than they save.
    const double alpha = (!GUARD && P.hessian == OH_HESSIAN_HYBRID && stat > P.hyb_switch && iters >= P.relax_from) ? P.relax : 1.0;
    // gains of the next PFF knots in flight (they do not depend on z: without the explicit buffers every knot waits a full round trip)
    constexpr int PFF = OH_STEP_PREFETCH_FWD;
    double fK[PFF][NZ * NZ], fk[PFF][NZ], fg[PFF][NZ];
    auto fetchf = [&](const int j, const int t) {
      if (t > P.t0) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) fk[j][a] = rb_ld(KNOT(D.kvec, t, NZ), RB(a), lb);
#pragma unroll
        for (int i = 0; i < NZ * NZ; ++i) fK[j][i] = rb_ld(KNOT(D.Kmat, t, NZ * NZ), RB(i), lb);
      }
      if constexpr (GUARD) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) fg[j][a] = rb_ld(KNOT(gtc, t, NZ), RB(a), oG);
      }
    };
#pragma unroll
    for (int j = 0; j < PFF; ++j)
      if (P.t0 + j < T) fetchf(j, P.t0 + j);
    for (int tb = P.t0; tb < T; tb += PFF) {
#pragma unroll
      for (int j = 0; j < PFF; ++j) {
        const int t = tb + j;
        if (t < T) {
          double gtt[NZ];
#pragma unroll
          for (int a = 0; a < NZ; ++a) gtt[a] = GUARD ? fg[j][a] : 0.0;
          if (t > P.t0) {
            double zn[NZ];
#pragma unroll
            for (int a = 0; a < NZ; ++a) {
              double sacc = fk[j][a];
#pragma unroll
              for (int c2 = 0; c2 < NZ; ++c2) sacc += fK[j][a * NZ + c2] * zz[c2];
              zn[a] = -sacc;
            }
#pragma unroll
            for (int a = 0; a < NZ; ++a) zz[a] = zn[a];
          }
          if (t + PFF < T) fetchf(j, t + PFF);
#pragma unroll
          for (int a = 0; a < NZ; ++a) {
            rb_st(KNOT(D.zstep, t, NZ), RB(a), lb, alpha * zz[a]);
            if constexpr (GUARD) gd += gtt[a] * zz[a];
            z2 += zz[a] * zz[a];
          }
        }
      }
    }
    // decrease the model predicts for alpha z, with (H + mu I) z = -g:  -alpha g.z - alpha^2/2 z^T H z = -alpha gd + alpha^2/2 (gd + mu z2)
    D.pred[b] = -alpha * gd + 0.5 * alpha * alpha * (gd + mu * z2);
    if constexpr (GUARD) {  // for the line search along this step, should it be rejected (alpha = 1 on these handles)
      GBp->ls_gd[b] = alpha * gd;
      GBp->ls_q[b] = alpha * alpha * (gd + mu * z2);
    }
  }
  D.mu[b] = mu;
  D.iters[b] = iters + 1;
  return true;
}


// K3 with the coupling folded in (round 3; plain orientation-locked handles).  k_couple wrote E_t = -2 kappa Z_t^T Z_{t+1}, gt_t = Z_t^T G_t
// and the merit share of every knot (21 doubles per unit) for this kernel to read back one launch later, and read 47 doubles per unit to do
// so.  Here the sweep rebuilds Z_t from its Householder vectors (the 18 doubles k_couple read too), keeps Z_{t+1} in a lane-private column
// of LDS (zl: [N * NZ][64], one 8-byte word per lane and row -> conflict-free) and forms E_t and gt_t with couple_knot's own loop, in its
// operation order: the iterates are those of the three-kernel path.  G_t and the merit share come from the evaluation (eval_unit<.., ZC>).
// Per knot the backward pass reads V 18, G 7, Dr 10 (35 doubles; E 16, Dr 10, gt 4 before).  End of round 5: the reduced gradients gt_t no longer travel to
// the forward pass and back (8 of the sweep's 89 doubles per unit): all the forward pass did with them was the directional derivative g.z of the step, and that is
// -g^T M^{-1} g = -sum_t |L_t^{-1} r_t|^2, a by-product of the backward substitutions (riccati_back's quad).
template <int N>
OH_DEV bool step_instance_zc(const FigParams& P, const FigBuffers& D, const int b, const int ts, double* __restrict__ zl) {
  constexpr int NZ = N - 3;
  constexpr int NP = NZ * (NZ + 1) / 2;
  constexpr int NV = HV_ROWS(N);
  const int Bp = D.Bp;
  const int T = P.T;
  const unsigned lb = (unsigned)b * 8u;
  const unsigned rowB = (unsigned)Bp * 8u;
  const double kap2 = 2.0 * P.kappa;
  int cur = 1 - ts;
  LMState lm{D.mu[b], D.nun[b]};
  const int iters = D.iters[b];
  {
    const StepHead hd = step_head<N, false>(P, D, b, ts, nullptr, cur, lm, iters);
    if (hd.go != 2) return hd.go == 1;
    cur = hd.cur;
    lm = hd.lm;
  }
  double mu = lm.mu;
  const double* __restrict__ Vc = D.Z[0];
  // (Gfull[] trades places with the compaction's spare array: the lower of the two slots is the base, see ensure_capacity)
  const double* __restrict__ Gc = D.Gfull[0] < D.Gfull[1] ? D.Gfull[0] : D.Gfull[1];
  const double* __restrict__ Drc = D.Dr[0];
  const unsigned oV = lb + (cur ? (unsigned)((const char*)D.Z[1] - (const char*)D.Z[0]) : 0u);
  const unsigned oGf = lb + (unsigned)((const char*)D.Gfull[cur] - (const char*)Gc);
  const unsigned oD = lb + (cur ? (unsigned)((const char*)D.Dr[1] - (const char*)D.Dr[0]) : 0u);
  double stat = 0.0;
  double S[NP], rd[NZ], rn[NZ];
  bool factored = false;
  // a knot's inputs are requested PF knots ahead of their use (the kernel runs one wavefront per SIMD: registers to spare, and nothing
  // but its own loads in flight to cover the memory latency with)
#ifndef OH_STEP_ZC_PF
#define OH_STEP_ZC_PF 2  // (round 6, once the retries and the deferring lane's copy were out of the kernel: 2 against 1 is -2 % over a solve, -12 % on launches of <= 20 000 instances; 3 the same)
#endif
  constexpr int PF = OH_STEP_ZC_PF;
  double rV[PF][NV], rG[PF][N], rH[PF][NP];  // ring: knot t sits in slot (T - 1 - t) % PF (static indices: the knot loop is unrolled PF-fold)
  double nV[NV], nG[N], nH[NP];
  auto fetch = [&](const int j, const int t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) rV[j][i] = rb_ld(KNOT(Vc, t, NV), RB(i), oV);
#pragma unroll
    for (int k = 0; k < N; ++k) rG[j][k] = rb_ld(KNOT(Gc, t, N), RB(k), oGf);
#pragma unroll
    for (int i = 0; i < NP; ++i) rH[j][i] = rb_ld(KNOT(Drc, t, NP), RB(i), oD);
  };
  auto take = [&](const int j) {
#pragma unroll
    for (int i = 0; i < NV; ++i) nV[i] = rV[j][i];
#pragma unroll
    for (int k = 0; k < N; ++k) nG[k] = rG[j][k];
#pragma unroll
    for (int i = 0; i < NP; ++i) nH[i] = rH[j][i];
  };
  // Z of the knot in the buffers (z_from_householder on the packed vectors), its reduced gradient, and -- against the Z of knot t + 1 parked
  // in LDS -- E_t; Z_t then takes that place
  auto knot_blocks = [&](const bool last, double (&E)[NZ * NZ], double (&gt)[NZ]) {
    double V[3][N], Zt[N][NZ];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int k = 0; k < N; ++k) V[m][k] = (k < m) ? 0.0 : nV[HV_OFF(N, m) + k - m];
    z_from_householder<N>(V, Zt);
#pragma unroll
    for (int a = 0; a < NZ; ++a) {
      gt[a] = 0.0;
#pragma unroll
      for (int c2 = 0; c2 < NZ; ++c2) E[a * NZ + c2] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double Gk = nG[k];
      double Zn[NZ];
      if (!last) {
#pragma unroll
        for (int c2 = 0; c2 < NZ; ++c2) Zn[c2] = zl[(k * NZ + c2) * 64];
      }
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        gt[a] += Zt[k][a] * Gk;
        if (!last) {
#pragma unroll
          for (int c2 = 0; c2 < NZ; ++c2) E[a * NZ + c2] -= kap2 * Zt[k][a] * Zn[c2];
        }
      }
#pragma unroll
      for (int a = 0; a < NZ; ++a) zl[(k * NZ + a) * 64] = Zt[k][a];
    }
  };
  double quad = 0.0;  // g^T M^{-1} g of the system being solved (see above)
  // A reduced Hessian that does not factorise at the instance's damping (exact curvature, indefinite away from the solution: 0.5 % of all sweeps) used
  // to be swept again with 4 x the damping inside this loop -- by the whole wavefront: 64 lanes wait a second backward pass for one of them, and with
  // 1-3 % of the instances failing in iterations 9-16 that was every second wavefront (round 6: the per-iteration histogram of the host port).  Now
  // the lane defers (OH_STEP_ZC_DEFER): it raises its damping exactly as the loop would have, asks for its accepted point to be laid down and
  // evaluated again as it is (D.polish = 2: k_retract copies the knots, k_evalb_zc rebuilds the stage data bit for bit -- the older Lagrangian gradient
  // its multiplier estimate came from is put back where the evaluation reads it), and sweeps at the raised damping one launch later.  Same iterates,
  // same step counts (the extra launch is not counted); the price is one evaluation of the instance instead of one sweep of its wavefront.
#ifndef OH_STEP_ZC_DEFER
#define OH_STEP_ZC_DEFER 1
#endif
  for (int attempt = 0; attempt < (OH_STEP_ZC_DEFER ? 1 : 40); ++attempt) {
    bool ok = true;
    stat = 0.0;
    quad = 0.0;
    // knot T - 1 - i travels in ring slot i % PF
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (T - 1 - j >= P.t0) fetch(j, T - 1 - j);
    {
      double E[NZ * NZ], gt[NZ];
      take(0);
      if (T - 1 - PF >= P.t0) fetch(0, T - 1 - PF);
      knot_blocks(true, E, gt);
#pragma unroll
      for (int i = 0; i < NP; ++i) S[i] = nH[i];
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        S[tri(a, a)] += kap2 + mu;
        rn[a] = gt[a];
        stat = fmax(stat, fabs(gt[a]));
      }
    }
    for (int tb = T - 2; tb >= P.t0; tb -= PF) {
#pragma unroll
      for (int jj = 0; jj < PF; ++jj) {
        const int t = tb - jj;
        const int j = (jj + 1) % PF;  // slot of knot t: (T - 1 - t) % PF with T - 1 - tb = 1 (mod PF) at every pass
        if (t >= P.t0) {
          double E[NZ * NZ], Ht[NP], gt[NZ];
          take(j);
          if (t - PF >= P.t0) fetch(j, t - PF);  // issued before the dependent arithmetic of this knot
          knot_blocks(false, E, gt);
#pragma unroll
          for (int i = 0; i < NP; ++i) Ht[i] = nH[i];
#pragma unroll
          for (int a = 0; a < NZ; ++a) {
            stat = fmax(stat, fabs(gt[a]));
            Ht[tri(a, a)] += 2.0 * kap2 + mu;
          }
          double Kmat[NZ * NZ], kv[NZ];
          ok = riccati_back<NZ>(S, rd, rn, E, Ht, gt, Kmat, kv, &quad) && ok;
#pragma unroll
          for (int a = 0; a < NZ; ++a) rb_st(KNOT(D.kvec, t + 1, NZ), RB(a), lb, kv[a]);
#pragma unroll
          for (int i = 0; i < NZ * NZ; ++i) rb_st(KNOT(D.Kmat, t + 1, NZ * NZ), RB(i), lb, Kmat[i]);
        }
      }
    }
    ok = chol_rcp<NZ>(S, rd, 1e-12) && ok;
    if (ok) {
      factored = true;
      break;
    }
    mu = fmax(4.0 * mu, 1e-2);
  }
  if (OH_STEP_ZC_DEFER && !factored && stat == stat && mu < 1e24) {  // (4^40 was the loop's limit)
    D.mu[b] = mu;
    D.polish[b] = 2;
    // (D.stat keeps the value the evaluation of this point saw: its hybrid-curvature decision is taken again, alike)
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON && cur == ts) {
      // accepted in this launch: the gradient its evaluation took the multiplier estimate from is still in the other slot; this point's own is rebuilt by the
      // evaluation that follows.  The copy (T x N doubles) is k_defer_copy's, right after this kernel, a thread per (instance, knot) of the list: done here, by the
      // one lane of the instance, it held the wavefront up for as long as the retry it replaced (k_step_zc 9 % slower over a solve)
      D.defer_list[(size_t)ts * Bp + oh_take_ticket(D.n_defer + ts)] = b;
    }
    return true;
  }
  D.stat[b] = stat;
  if (!(stat == stat) || !factored) {
    D.status[b] = OH_STATUS_NUMERICAL;
    D.mu[b] = mu;
    return false;
  }
  if (stat <= P.tol && D.feas[b] <= P.tol_feas) {
    D.status[b] = OH_STATUS_CONVERGED;
    D.mu[b] = mu;
    return false;
  }
  if (iters >= P.max_iter) {
    D.status[b] = OH_STATUS_MAX_ITER;
    D.mu[b] = mu;
    return false;
  }
  // ---- forward recursion (as step_instance): z_2 = -S_2^{-1} r_2, z_{t+1} = -(kvec + Kmat z_t) ----
  {
    double zz[NZ];
#pragma unroll
    for (int a = 0; a < NZ; ++a) zz[a] = -rn[a];
    fsub_rcp<NZ>(S, rd, zz);
#pragma unroll
    for (int a = 0; a < NZ; ++a) quad = fma(zz[a], zz[a], quad);  // the first free knot's share
    bsub_rcp<NZ>(S, rd, zz);
    const double gd = -quad;
    double z2 = 0.0;
    const double alpha = (P.hessian == OH_HESSIAN_HYBRID && stat > P.hyb_switch && iters >= P.relax_from) ? P.relax : 1.0;
    double fK[NZ * NZ], fk[NZ];
    auto fetchf = [&](const int t) {
      if (t > P.t0) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) fk[a] = rb_ld(KNOT(D.kvec, t, NZ), RB(a), lb);
#pragma unroll
        for (int i = 0; i < NZ * NZ; ++i) fK[i] = rb_ld(KNOT(D.Kmat, t, NZ * NZ), RB(i), lb);
      }
    };
    fetchf(P.t0);
    for (int t = P.t0; t < T; ++t) {
      if (t > P.t0) {
        double zn[NZ];
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
          double sacc = fk[a];
#pragma unroll
          for (int c2 = 0; c2 < NZ; ++c2) sacc += fK[a * NZ + c2] * zz[c2];
          zn[a] = -sacc;
        }
#pragma unroll
        for (int a = 0; a < NZ; ++a) zz[a] = zn[a];
      }
      if (t + 1 < T) fetchf(t + 1);
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        rb_st(KNOT(D.zstep, t, NZ), RB(a), lb, alpha * zz[a]);
        z2 += zz[a] * zz[a];
      }
    }
    D.pred[b] = -alpha * gd + 0.5 * alpha * alpha * (gd + mu * z2);
  }
  D.mu[b] = mu;
  D.iters[b] = iters + 1;
  return true;
}


// Least-squares multipliers of knot t at the point in slot `cur`, mapped to the reference's rows
// h = quat_c - quat(q_t) (figure_eight_plan.py:105-107): stationarity reads G_t + Jc^T mu = 0 with
// Jc = Jw on the manifold and dh/dq = -1/2 Ec Jw (Ec o = (o,0)(x)quat_c), hence nu = -2 Ec mu
// satisfies G_t + (dh/dq)^T nu = 0.
template <int N>
OH_DEV void knot_multipliers(const FigParams& P, const FigBuffers& D, const int b, const int t, const int cur, double* out) {
  const int Bp = D.Bp;
  if (t < P.t0) {  // knots fixed by the linear rows: the quaternion rows are redundant there, multiplier 0
    out[0] = out[1] = out[2] = out[3] = 0.0;
    return;
  }
  const oh_chain* ch = OH_CHAIN(D);  // (the chain-specialised module compiles this walk too: oh_spec_finalize)
  const double* __restrict__ qs = D.q[cur];
  const double kap2 = 2.0 * P.kappa;
  double q[N], G[N];
  const bool last = (t == P.T - 1);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    q[k] = qs[IDX(t, N, k)];
    if (P.zc) {  // the evaluation stored the Lagrangian gradient itself (eval_unit<.., ZC>)
      G[k] = D.Gfull[cur][IDX(t, N, k)];
    } else {
      G[k] = D.g[cur][IDX(t, N, k)] + kap2 * (q[k] - qs[IDX(t - 1, N, k)]);
      if (!last) G[k] -= kap2 * (qs[IDX(t + 1, N, k)] - q[k]);
    }
  }
  double R[9], p[3], z[N][3], pj[N][3];
  if (ch->has_lead) {
    double Rb[9], pb[3];
    lead_base(ch, D.lead[(size_t)t * Bp + b], Rb, pb);
    fk_chain<N, true>(ch, q, R, p, z, pj, Rb, pb);
  } else {
    fk_chain<N>(ch, q, R, p, z, pj);
  }
  double S[6] = {1e-14, 0, 1e-14, 0, 0, 1e-14};
  double rhs[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (ch->jtype[k] == 0) {
      S[0] += z[k][0] * z[k][0];
      S[1] += z[k][1] * z[k][0];
      S[2] += z[k][1] * z[k][1];
      S[3] += z[k][2] * z[k][0];
      S[4] += z[k][2] * z[k][1];
      S[5] += z[k][2] * z[k][2];
      rhs[0] -= z[k][0] * G[k]; rhs[1] -= z[k][1] * G[k]; rhs[2] -= z[k][2] * G[k];
    }
  }
  chol_packed<3>(S, 0.0);
  fsub<3>(S, rhs);
  bsub<3>(S, rhs);  // mu
  // quat_c: rebuild with the reference's chain product at qc = q_0
  double quat[4] = {0, 0, 0, 1};
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double qn[4];
    qmul(quat, ch->quat0[k], qn);
    if (ch->jtype[k] == 0) {
      double sh, chh;
      sincos_joint(0.5 * qs[IDX(0, N, k)], &sh, &chh);
      const double qa[4] = {sh * ch->axis[k][0], sh * ch->axis[k][1], sh * ch->axis[k][2], chh};
      qmul(qn, qa, quat);
    } else {
      quat[0] = qn[0]; quat[1] = qn[1]; quat[2] = qn[2]; quat[3] = qn[3];
    }
  }
  double qcq[4];
  qmul(quat, ch->quat_tool, qcq);
  const double o[4] = {rhs[0], rhs[1], rhs[2], 0.0};
  double nu[4];
  qmul(o, qcq, nu);
  out[0] = -2.0 * nu[0]; out[1] = -2.0 * nu[1]; out[2] = -2.0 * nu[2]; out[3] = -2.0 * nu[3];
}

// Solution out in the reference layout x = [vec(Q); vec(dQ)] (sx_container.py:83-89), plus f, kkt,
// iterations, status and the h-row multipliers, written at the instance's ORIGINAL index (instances are
// compacted while the batch drains).  only_done: emit just the instances that have finished.
template <int N>
OH_DEV void finalize_unit(const FigParams& P, const FigBuffers& D, int only_done, double* __restrict__ x, double* __restrict__ f,
                          double* __restrict__ kkt, int* __restrict__ iters, int* __restrict__ status, const int b, const int t) {
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int st = D.status[b];
  if (only_done && st < 0) return;
  const size_t ob = (size_t)D.orig[b];
  const int cur = D.cur[b];
  const double* __restrict__ qs = D.q[cur];
  if (x) {
    double* xb = x + ob * P.nx;
    double q0[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      q0[j] = qs[IDX(t, N, j)];
      xb[(size_t)t * N + j] = q0[j];
    }
    if (t < P.T - 1) {
      const double inv_dt = 1.0 / P.dt;
#pragma unroll
      for (int j = 0; j < N; ++j) xb[(size_t)P.T * N + (size_t)t * N + j] = (qs[IDX(t + 1, N, j)] - q0[j]) * inv_dt;
    }
  }
  if constexpr (N > 3) {  // (the orientation rows exist from four joints on)
    if (D.lam_h) knot_multipliers<N>(P, D, b, t, cur, D.lam_h + (ob * P.T + t) * 4);
  }
  if (t == 0) {
    if (f) f[ob] = D.f_cur[b] - (D.fpsi ? D.fpsi[b] : 0.0);
    if (kkt) {
      kkt[3 * ob + 0] = D.stat[b];
      kkt[3 * ob + 1] = D.feas[b];
      kkt[3 * ob + 2] = D.fpsi ? D.feas[b] : 0.0;  // with inequality rows feas = |min(g, lam/rho)|_inf covers both
    }
    if (iters) iters[ob] = D.iters[b];
    if (status) status[ob] = (st < 0) ? OH_STATUS_MAX_ITER : st;
  }
}


