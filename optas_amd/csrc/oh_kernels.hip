// HIP kernels of liboptas_hip (gfx950).  See DESIGN.md for the data layout and per-kernel rooflines.
//
// All per-(instance, knot) arrays are structure-of-arrays with the instance index fastest:
//   a[(t*K + k)*Bp + b]        (Bp = B rounded up to 64)
// so that the 64 lanes of a wavefront, which always hold 64 consecutive instances b at one knot t,
// read and write full 512-byte lines.
#include "oh_figure8.h"

#define IDX(t, K, k) (((size_t)(t) * (K) + (k)) * Bp + b)
// Row addressing for the hot loops: a stage array is [row = t*K + k][Bp].  Through a buffer resource the row offset travels in an SGPR
// and the lane adds ONE 32-bit byte offset shared by every stream ("buffer_load_dwordx2 v, v_off, s[rsrc], s_row offen"); with flat
// global pointers the compiler keeps a 64-bit VGPR address per stream alive (k_step: 50 of them, 55 registers spilled inside its
// serial sweep).  A RowBuf is rebased per knot (scalar ALU), so the SGPR offset k*Bp*8 always fits 32 bits.
#if defined(__HIP_DEVICE_COMPILE__)
struct RowBuf {
  __amdgpu_buffer_rsrc_t r;
};
OH_DEV RowBuf rowbuf(const double* knot_base) {
  return RowBuf{__builtin_amdgcn_make_buffer_rsrc((void*)knot_base, 0, 0xFFFFFFFF, 0x00020000)};  // raw buffer, gfx9 data format word
}
OH_DEV double rb_ld(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  const v2i v = __builtin_amdgcn_raw_buffer_load_b64(rb.r, lane_bytes, row_bytes, 0);
  return __builtin_bit_cast(double, v);
}
OH_DEV void rb_st(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes, const double x) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, x), rb.r, lane_bytes, row_bytes, 0);
}
#else  // host build of oracle/cpu_port
struct RowBuf {
  char* p;
};
OH_DEV RowBuf rowbuf(const double* knot_base) { return RowBuf{(char*)knot_base}; }
OH_DEV double rb_ld(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes) { return *(const double*)(rb.p + row_bytes + lane_bytes); }
OH_DEV void rb_st(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes, const double x) { *(double*)(rb.p + row_bytes + lane_bytes) = x; }
#endif
// knot t of an array with K rows per knot; row k of that knot
#define KNOT(arr, t, K) rowbuf((arr) + (size_t)(t) * (K) * (size_t)Bp)
#define RB(k) ((unsigned)(k) * rowB)

// ---------------------------------------------------------------------------------------------
// K1: batched FK + geometric Jacobian (+ reference-signed quaternion), arbitrary chain.
//   replaces get_global_link_{position,quaternion,geometric_jacobian}_function(link, n=N)
//   (optas/models.py:935-947,1090-1106,1199-1281).  One lane per unit.
//   SOA=true : q[ndof][N], pose[7][N], J[6*ndof][N]      (coalesced; solver-internal / roofline)
//   SOA=false: q[N][ndof], pose[N][7], J[N][6][ndof]     (reference layout at the ABI)
// ---------------------------------------------------------------------------------------------
#ifndef OH_HOST_PORT
template <bool SOA>
__global__ __launch_bounds__(256) void k_fk_jac(const oh_chain* __restrict__ ch, int n, const double* __restrict__ q,
                                                double* __restrict__ pose, double* __restrict__ J) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const int nc = ch->n_chain;
  const int ndof = ch->ndof;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
  double quat[4] = {0, 0, 0, 1};
  double z[OH_MAX_CHAIN][3];
  double pj[OH_MAX_CHAIN][3];
#pragma unroll
  for (int k = 0; k < OH_MAX_CHAIN; ++k) {
    if (k < nc) {
      const int qi = ch->qidx[k];
      const double qk = SOA ? q[(size_t)qi * n + u] : q[(size_t)u * ndof + qi];
      double t[3];
      mv3(R, ch->p0[k], t);
      p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
      if (!ch->r0ident[k]) {
        double Rn[9];
        mm3(R, ch->R0[k], Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
      }
      double qn[4];
      qmul(quat, ch->quat0[k], qn);  // == fromrpy(rpy) * quat in the reference's reversed product
      pj[k][0] = p[0]; pj[k][1] = p[1]; pj[k][2] = p[2];
      if (ch->jtype[k] == 0) {
        double sh, chh;
        sincos_joint(0.5 * qk, &sh, &chh);  // half angle: quaternion (spatialmath.py:372-375) ...
        const double s = 2.0 * sh * chh, c = 1.0 - 2.0 * sh * sh;  // ... and full angle for Rodrigues
        if (ch->axcode[k] != 0) rot_principal_right(R, ch->axcode[k], s, c, z[k]);
        else rot_axis_right(R, ch->axis[k], s, c, z[k]);
        const double qa[4] = {sh * ch->axis[k][0], sh * ch->axis[k][1], sh * ch->axis[k][2], chh};
        qmul(qn, qa, quat);
      } else {
        mv3(R, ch->axis[k], z[k]);
        p[0] += z[k][0] * qk; p[1] += z[k][1] * qk; p[2] += z[k][2] * qk;
        quat[0] = qn[0]; quat[1] = qn[1]; quat[2] = qn[2]; quat[3] = qn[3];
      }
    }
  }
  double e[3], t[3];
  mv3(R, ch->p_tool, t);
  e[0] = p[0] + t[0]; e[1] = p[1] + t[1]; e[2] = p[2] + t[2];
  if (pose) {
    double qe[4];
    qmul(quat, ch->quat_tool, qe);
    const double o[7] = {e[0], e[1], e[2], qe[0], qe[1], qe[2], qe[3]};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (SOA) pose[(size_t)i * n + u] = o[i];
      else pose[(size_t)u * 7 + i] = o[i];
    }
  }
  if (J) {
    // columns of joints that are not on the chain are zero (models.py:1251-1254)
    if (!SOA) {
      for (int i = 0; i < 6 * ndof; ++i) J[(size_t)u * 6 * ndof + i] = 0.0;
    } else if (nc != ndof) {
      for (int i = 0; i < 6 * ndof; ++i) J[(size_t)i * n + u] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < OH_MAX_CHAIN; ++k) {
      if (k < nc) {
        const int col = ch->qidx[k];
        double col6[6];
        if (ch->jtype[k] == 0) {
          const double d[3] = {e[0] - pj[k][0], e[1] - pj[k][1], e[2] - pj[k][2]};
          cross3(z[k], d, col6);  // models.py:1236-1239
          col6[3] = z[k][0]; col6[4] = z[k][1]; col6[5] = z[k][2];
        } else {
          col6[0] = z[k][0]; col6[1] = z[k][1]; col6[2] = z[k][2];  // models.py:1245-1246
          col6[3] = col6[4] = col6[5] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          if (SOA) J[((size_t)r * ndof + col) * n + u] = col6[r];
          else J[(size_t)u * 6 * ndof + r * ndof + col] = col6[r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K5: batched recursive Newton-Euler inverse dynamics, one lane per sample, NB bodies (the last one rigidly
// attached).  Statement by statement RobotModel.rnea (optas/models.py:1819-1880); AoS [N][NB-1] at the ABI.
// ---------------------------------------------------------------------------------------------
OH_DEV void mTv3(const double* A, const double* v, double* o) {  // o = A^T v
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
template <int NB>
__global__ __launch_bounds__(256) void k_rnea(const oh_dynamics* __restrict__ dy, int n, const double* __restrict__ q,
                                              const double* __restrict__ qd, const double* __restrict__ qdd, double* __restrict__ tau) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  constexpr int ND = NB - 1;
  double f[NB][3], nn[NB][3], sj[NB], cj[NB];
  double om[3] = {0, 0, 0}, omD[3] = {0, 0, 0}, vD[3] = {dy->vd0[0], dy->vd0[1], dy->vd0[2]};
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    // iRp = (R0_i Rot(axis_i, q_i))^T ; for the last body no joint rotation (models.py:1820-1832)
    double Rp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = dy->R0[i][k];
    double qdi = 0.0, qddi = 0.0;
    if (i != NB - 1) {
      double s, c, zc[3];
      sincos_joint(q[(size_t)u * ND + i], &s, &c);
      sj[i] = s; cj[i] = c;
      rot_axis_right(Rp, dy->axis[i], s, c, zc);
      qdi = qd[(size_t)u * ND + i];
      qddi = qdd[(size_t)u * ND + i];
    } else {
      sj[i] = 0.0; cj[i] = 1.0;
    }
    double a[3], omp[3], omDp[3];
    mTv3(Rp, dy->axis[i], a);  // iaxisi
    mTv3(Rp, om, omp);
    mTv3(Rp, omD, omDp);
    double omi[3], omDi[3];
    if (i != NB - 1) {
      const double aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
      double cr[3];
      cross3(omp, aq, cr);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        omi[k] = omp[k] + aq[k];
        omDi[k] = omDp[k] + cr[k] + a[k] * qddi;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) { omi[k] = omp[k]; omDi[k] = omDp[k]; }
    }
    // vDi = iRp (vD + omD x r + om x (om x r)),  r = joint origin
    double t1[3], t2[3], t3[3], acc[3], vDi[3];
    cross3(omD, dy->xyz[i], t1);
    cross3(om, dy->xyz[i], t2);
    cross3(om, t2, t3);
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = vD[k] + t1[k] + t3[k];
    mTv3(Rp, acc, vDi);
    // fi = m (vDi + omDi x c + omi x (omi x c)) ; ni = I omDi + omi x (I omi)
    cross3(omDi, dy->com[i], t1);
    cross3(omi, dy->com[i], t2);
    cross3(omi, t2, t3);
#pragma unroll
    for (int k = 0; k < 3; ++k) f[i][k] = dy->mass[i] * (vDi[k] + t1[k] + t3[k]);
    double Io[3], IoD[3];
    mv3(dy->inertia[i], omi, Io);
    mv3(dy->inertia[i], omDi, IoD);
    cross3(omi, Io, t1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      nn[i][k] = IoD[k] + t1[k];
      om[k] = omi[k]; omD[k] = omDi[k]; vD[k] = vDi[k];
    }
  }
  // backward (models.py:1858-1880); reference lists fs/ns carry a leading zero entry: fs[i] == f[i-1]
  double ifi[3] = {f[NB - 1][0], f[NB - 1][1], f[NB - 1][2]};
  double ini[3], t1[3];
  cross3(dy->com[NB - 1], f[NB - 1], t1);
#pragma unroll
  for (int k = 0; k < 3; ++k) ini[k] = nn[NB - 1][k] + t1[k];
#pragma unroll
  for (int i = NB - 1; i >= 1; --i) {
    double pRi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) pRi[k] = dy->R0[i][k];
    if (i < NB - 1) { double zc[3]; rot_axis_right(pRi, dy->axis[i], sj[i], cj[i], zc); }
    double a1[3], a2[3], a3[3], a4[3];
    mv3(pRi, ini, a1);
    cross3(dy->com[i - 1], f[i - 1], a2);
    mv3(pRi, ifi, a3);
    cross3(dy->xyz[i], a3, a4);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ini[k] = nn[i - 1][k] + a1[k] + a2[k] + a4[k];
      ifi[k] = a3[k] + f[i - 1][k];
    }
    double pR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) pR[k] = dy->R0[i - 1][k];
    double zc[3];
    rot_axis_right(pR, dy->axis[i - 1], sj[i - 1], cj[i - 1], zc);
    double ax[3];
    mTv3(pR, dy->axis[i - 1], ax);  // pRi^T axis
    tau[(size_t)u * ND + (i - 1)] = dot3(ini, ax);
  }
}

#endif  // OH_HOST_PORT

// ---------------------------------------------------------------------------------------------
// Figure-eight family.  N = ndof (chain covers all joints in order), NZ = N-3 (orientation locked).
// ---------------------------------------------------------------------------------------------

// per-instance setup: references from qc, fixed knots, seed -> slot 0, solver state.
template <int N>
OH_DEV void setup_unit(const FigParams& P, const FigBuffers& D, const double* __restrict__ x0, const double* __restrict__ pin, const int b) {
  const int Bp = D.Bp;
  if (b >= D.B) {
    if (b < Bp) D.status[b] = OH_STATUS_CONVERGED;  // padding lanes never run
    return;
  }
  const oh_chain* ch = D.chain;
  double qc[N];
#pragma unroll
  for (int j = 0; j < N; ++j) qc[j] = pin[(size_t)b * P.np + j];
  double R[9], p[3], z[N][3], pj[N][3];
  if (ch->has_lead) {
    // p = [qc of the optimised joints (N); lead angle of qc; lead angle of every knot (T)]
    double Rb[9], pb[3];
    lead_base(ch, pin[(size_t)b * P.np + N], Rb, pb);
    fk_chain<N, true>(ch, qc, R, p, z, pj, Rb, pb);
    for (int tt = 0; tt < P.T; ++tt) D.lead[(size_t)tt * Bp + b] = pin[(size_t)b * P.np + N + 1 + tt];
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
  for (int tt = 0; tt < P.t0 && tt < P.T; ++tt) {
    double l[3] = {P.local_path[3 * tt], P.local_path[3 * tt + 1], P.local_path[3 * tt + 2]};
    fconst += P.w_path * dot3(l, l);  // Rc orthonormal
  }
  D.fconst[b] = fconst;
  // knots: slot 0 holds the seed with q_0 = q_1 = qc imposed (linear rows eliminated)
  for (int tt = 0; tt < P.T; ++tt) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double v = (tt < P.t0) ? qc[j] : x0[(size_t)b * P.nx + (size_t)tt * N + j];
      D.q[0][IDX(tt, N, j)] = v;
      D.q[1][IDX(tt, N, j)] = (tt < P.t0) ? qc[j] : 0.0;
    }
  }
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
#ifndef OH_HOST_PORT
template <int N>
__global__ __launch_bounds__(64) void k_setup(FigParams P, FigBuffers D, const double* __restrict__ x0, const double* __restrict__ pin) {
  setup_unit<N>(P, D, x0, pin, blockIdx.x * blockDim.x + threadIdx.x);
}
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
template <int N, bool GUARD = false, bool LEAD = false, int MODE = EVAL_FUSED>
OH_DEV void eval_unit(const FigParams& P, const FigBuffers& D, const int slot, const int b, const int t, const GuardParams* GPp = nullptr,
                      const GuardBuffers* GBp = nullptr) {
  constexpr int NZ = N - 3;
  constexpr int NP = NZ * (NZ + 1) / 2;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  if (D.status[b] >= 0 || D.skip[b]) return;
  // Uniform slots: every running instance writes this launch's trial into `slot` and keeps its accepted
  // point in `cur` = 1 - slot, so all lanes of a wavefront touch the same arrays (full 512-B lines).  An
  // instance whose previous trial was rejected has its accepted point in `slot`: it sits this launch out
  // (skip flag) and is back in phase at the next one -- cheaper than moving its stage data.
  const int cur = 1 - slot;
  const bool first = D.first[b] != 0;
  // trial knot: the seed on the first evaluation, otherwise q_cur + Z_cur z (roll-out of the step k_step solved for)
  double q[N], e_tgt[3] = {0.0, 0.0, 0.0};
  if (first || MODE == EVAL_ONLY) {  // EVAL_ONLY: the retracted trial knot is already in the slot (k_retract)
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = D.q[slot][IDX(t, N, j)];
  } else {
    double zs[NZ];
#pragma unroll
    for (int a = 0; a < NZ; ++a) zs[a] = D.zstep[IDX(t, NZ, a)];
    double Vc[3][N], Zc[N][NZ];
    load_householder<N>(D.Z[cur], Bp, b, t, Vc);
    z_from_householder<N>(Vc, Zc);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double v = D.q[cur][IDX(t, N, j)];
#pragma unroll
      for (int a = 0; a < NZ; ++a) v += Zc[j][a] * zs[a];
      q[j] = v;
    }
    // where the linear model puts the end effector after this step: e_cur + (Jp Z)_cur z
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double v = D.mdl[cur][IDX(t, MDL_ROWS(N), m)];
#pragma unroll
      for (int a = 0; a < NZ; ++a) v += D.mdl[cur][IDX(t, MDL_ROWS(N), 3 + m * NZ + a)] * zs[a];
      e_tgt[m] = v;
    }
  }
  double Rc[9], pc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = D.ref[(size_t)i * Bp + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = D.ref[(size_t)(3 + i) * Bp + b];
  double Gprev[N];
  // exact curvature: always (OH_HESSIAN_EXACT) or once the accepted point is nearly stationary (OH_HESSIAN_HYBRID)
  bool exact = (P.hessian == OH_HESSIAN_EXACT) || (P.hessian == OH_HESSIAN_HYBRID && !first && D.stat[b] <= P.hyb_switch);
  if constexpr (GUARD) {
    // the exact block carries no curvature of the sphere rows (-s d2g, s ~ w_path): with them the exact model is worse than
    // Gauss-Newton (the oracle run crawls), so sphere-guarded problems stay on Gauss-Newton
    if (GPp->n_links > 0) exact = false;
  }
  const bool have_G = exact && !first;
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
      if constexpr (!GUARD) {  // the guard rows still add to g
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
    OH_DEV void load_G(const double (&)[N], double (&G)[N]) const {
#pragma unroll
      for (int k = 0; k < N; ++k) G[k] = Gc[IDX(t, N, k)];
    }
  };
  const Hooks hooks{D.q[slot], D.g[slot], D.Z[slot], D.Gfull[cur], Bp, b, t};
  double e_new[3], JZ_new[3][NZ];
  const double tol_r = retract_tol(P, !first, D.pred[b], D.stat[b]);
  if constexpr (LEAD)
    eval_knot<N, true, Hooks, MODE>(D.chain, P, t, q, pc, Rc, exact, have_G, Gprev, phi, cv, g, Dr, Z, !first, e_tgt, tol_r, e_new, JZ_new,
                                    D.lead[(size_t)t * Bp + b], hooks);
  else eval_knot<N, false, Hooks, MODE>(D.chain, P, t, q, pc, Rc, exact, have_G, Gprev, phi, cv, g, Dr, Z, !first, e_tgt, tol_r, e_new, JZ_new, 0.0, hooks);
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
        meas = fmax(meas, fabs(fmin(gval, lam / rho)));
        if (sv > 0.0) {
          psi += (sv * sv - lam * lam) / (2.0 * rho);
          g[j] += side ? sv : -sv;
          dd[j] += rho;
        } else {
          psi -= lam * lam / (2.0 * rho);
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
    if (GP.n_links > 0) {
      sphere_rows_walk<N>(D.chain, GP, GB.par, (size_t)Bp, b, q, [&](const int l, const int o, const double gval, const double (&dg)[N]) {
        double* lam_ptr = GB.lam + IDX(t, GP.NC, nl + l * GP.n_obs + o);
        double lam = *lam_ptr;
        if (upd) {
          lam = fmax(0.0, lam - rho_old * gval);
          *lam_ptr = lam;
        }
        const double sv = lam - rho * gval;
        meas = fmax(meas, fabs(fmin(gval, lam / rho)));
        if (sv > 0.0) {
          psi += (sv * sv - lam * lam) / (2.0 * rho);
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
          psi -= lam * lam / (2.0 * rho);
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
  D.phi[slot][(size_t)t * Bp + b] = phi;
  D.cv[slot][(size_t)t * Bp + b] = cv;
#pragma unroll
  for (int i = 0; i < NP; ++i) D.Dr[slot][IDX(t, NP, i)] = Dr[i];
}
#ifndef OH_HOST_PORT
// Blocks per CU of k_eval: with the six-row retraction the kernel needs ~360 live registers in its loop; at 2 waves/SIMD (256) it
// spills 99 of them and runs 10 % slower than at 1 wave/SIMD with none (A/B on one box: 64.1 vs 57.8 ms per bench step).
#ifndef OH_EVAL_WAVES
#define OH_EVAL_WAVES 1
#endif
template <int N>
__global__ __launch_bounds__(256, OH_EVAL_WAVES) void k_eval(FigParams P, FigBuffers D, const int slot) {
  eval_unit<N>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}
// The same knot in two launches, each at two waves per SIMD (see EVAL_RETRACT_ONLY / EVAL_ONLY in oh_figure8.h): k_retract leaves the
// retracted trial knot in the slot, k_evalb evaluates it.  One kinematics pass more than fused, both kernels without the register
// overflow of the fused loop.
template <int N>
__global__ __launch_bounds__(256, 2) void k_retract(FigParams P, FigBuffers D, const int slot) {
  eval_unit<N, false, false, EVAL_RETRACT_ONLY>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}
template <int N>
__global__ __launch_bounds__(256, 2) void k_evalb(FigParams P, FigBuffers D, const int slot) {
  eval_unit<N, false, false, EVAL_ONLY>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}
// chains with a parameterised lead joint (RobotModel(param_joints=[first joint]), figure_eight_plan_6dof.py): same evaluation
// from the frame that follows the lead joint at the knot's parameter angle
template <int N>
__global__ __launch_bounds__(256, OH_EVAL_WAVES) void k_eval_lead(FigParams P, FigBuffers D, const int slot) {
  eval_unit<N, false, true>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}
bool oh_launch_eval_lead(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot) {
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
  if (n == 6) hipLaunchKernelGGL(k_eval_lead<6>, g, b, 0, s, P, D, slot);
  else return false;
  return true;
}
#endif

// K2b: one lane per (instance b, free knot t), after k_eval: everything of the reduced block-tridiagonal
// system that needs the neighbouring knots but not the recursion (couple_knot in oh_figure8.h).
template <int N>
OH_DEV void couple_unit(const FigParams& P, const FigBuffers& D, const int slot, const int b, const int t) {
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
  couple_knot<N>(P.kappa, last, qm, q0, qp, g, Zt, Zn, D.phi[slot][(size_t)t * Bp + b], G, gt, E, merit);
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
#ifndef OH_HOST_PORT
// XCD-aware 1-D grid: workgroup w runs on XCD w % 8 (observed dispatch order), and knot t of an instance block re-reads what knot
// t+1 of the same block reads (Z_{t+1}, q_{t+1}).  Consecutive workgroups of one XCD therefore walk the knots of ONE instance block:
// w -> (chunk, r), XCD = r % 8 owns instance block chunk*8 + XCD, knot = r / 8, so the neighbour data is still in that XCD's 4 MB L2
// (in knot-major order the reuse distance was the whole batch: 11 MB per XCD at B = 131 072).
template <int N>
__global__ __launch_bounds__(256) void k_couple(FigParams P, FigBuffers D, const int slot) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *D.n_running = 0;  // k_step, next in the stream, counts the instances that go on
  const int Tn = P.T - P.t0;
  const int w = blockIdx.x;
  const int chunk = w / (8 * Tn), r = w - chunk * 8 * Tn;
  const int bx = chunk * 8 + (r & 7);
  couple_unit<N>(P, D, slot, bx * blockDim.x + threadIdx.x, (r >> 3) + P.t0);
}
#endif

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
  bool polish_request = false;

  // ---- phase A: merit of the trial slot ---------------------------------------------------------
  {
    double f = D.fconst[b];
    double feas = 0.0, fpsi = 0.0, meas = 0.0;
    for (int t = P.t0; t < T; ++t) {
      f += rb_ld(KNOT(D.merit[ts], t, 1), 0, lb);
      feas = fmax(feas, rb_ld(KNOT(D.cv[ts], t, 1), 0, lb));
      if constexpr (GUARD) {
        fpsi += rb_ld(KNOT(GBp->psi[ts], t, 1), 0, lb);
        meas = fmax(meas, rb_ld(KNOT(GBp->mcv[ts], t, 1), 0, lb));
      }
    }
    bool accept;
    if (D.first[b]) {
      if (!(f == f) || !(fabs(f) < 1e300)) {  // non-finite seed / parameters: report, do not iterate (fmax would hide the NaN)
        D.status[b] = OH_STATUS_NUMERICAL;
        D.cur[b] = ts;
        D.f_cur[b] = f;
        D.stat[b] = f;
        return false;
      }
      accept = true;
      D.first[b] = 0;
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
      accept = lm_accept(P, f, feas, D.f_cur[b], D.pred[b], D.stat[b], lm);
      // A rejected trial against an accepted point that was retracted loosely (retract_tol): its objective is off by (multiplier) x
      // violation, and steps that predict less than that can never be accepted.  Before blaming the model, re-evaluate the accepted
      // point at the floor tolerance: zero step, accepted unconditionally at the next k_step.
      polish_request = !accept && !D.stale[b] && D.feas[b] > 10.0 * P.tol_retract;
      if (polish_request) lm = lm_before;
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
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(D.work + 1, 1ULL);
#else
      D.work[1] += 1ULL;
#endif
      return true;
    }
    if (accept) {
      D.stale[b] = 0;
      cur = ts;
      D.f_cur[b] = f;
      D.feas[b] = feas;
      if constexpr (GUARD) {
        D.fpsi[b] = fpsi;
        GBp->meas[b] = meas;
      }
    }
    D.cur[b] = cur;
    if (!accept) {  // the accepted point sits where the next trial would go: sit the next launch out
      D.skip[b] = 1;
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(D.work + 1, 1ULL);
#else
      D.work[1] += 1ULL;
#endif
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
    return iters < P.max_iter + 40 ? true : (D.status[b] = OH_STATUS_MAX_ITER, false);
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
  for (int attempt = 0; attempt < 40; ++attempt) {
    bool ok = true;
    stat = 0.0;
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
    // prefetch registers for knot T-2
    double nE[NZ * NZ], nH[NP], ng[NZ];
    if (T - 2 >= P.t0) {
      const int t = T - 2;
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) nE[i] = rb_ld(KNOT(Ec, t, NZ * NZ), RB(i), oE);
#pragma unroll
      for (int i = 0; i < NP; ++i) nH[i] = rb_ld(KNOT(Drc, t, NP), RB(i), oD);
#pragma unroll
      for (int a = 0; a < NZ; ++a) ng[a] = rb_ld(KNOT(gtc, t, NZ), RB(a), oG);
    }
    for (int t = T - 2; t >= P.t0; --t) {
      double E[NZ * NZ], Ht[NP], gt[NZ];
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) E[i] = nE[i];
#pragma unroll
      for (int i = 0; i < NP; ++i) Ht[i] = nH[i];
#pragma unroll
      for (int a = 0; a < NZ; ++a) gt[a] = ng[a];
      if (t > P.t0) {  // issue the next knot's loads before the dependent arithmetic of this one
        const int tn = t - 1;
#pragma unroll
        for (int i = 0; i < NZ * NZ; ++i) nE[i] = rb_ld(KNOT(Ec, tn, NZ * NZ), RB(i), oE);
#pragma unroll
        for (int i = 0; i < NP; ++i) nH[i] = rb_ld(KNOT(Drc, tn, NP), RB(i), oD);
#pragma unroll
        for (int a = 0; a < NZ; ++a) ng[a] = rb_ld(KNOT(gtc, tn, NZ), RB(a), oG);
      }
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        stat = fmax(stat, fabs(gt[a]));
        Ht[tri(a, a)] += 2.0 * kap2 + mu;
      }
      double Kmat[NZ * NZ], kv[NZ];
      ok = riccati_back<NZ>(S, rd, rn, E, Ht, gt, Kmat, kv) && ok;
#pragma unroll
      for (int a = 0; a < NZ; ++a) rb_st(KNOT(D.kvec, t + 1, NZ), RB(a), lb, kv[a]);
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) rb_st(KNOT(D.Kmat, t + 1, NZ * NZ), RB(i), lb, Kmat[i]);
    }
    ok = chol_rcp<NZ>(S, rd, 1e-12) && ok;
    if (ok) break;
    mu = fmax(4.0 * mu, 1e-2);
  }
  D.stat[b] = stat;
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
  if (!(stat == stat)) {
    D.status[b] = OH_STATUS_NUMERICAL;
    return false;
  }

  // ---- forward recursion: z_2 = -S_2^{-1} r_2, z_{t+1} = -(kvec + Kmat z_t) ----------------------------
  {
    double zz[NZ];
#pragma unroll
    for (int a = 0; a < NZ; ++a) zz[a] = -rn[a];
    fsub_rcp<NZ>(S, rd, zz);
    bsub_rcp<NZ>(S, rd, zz);
    double gd = 0.0, z2 = 0.0;
    for (int t = P.t0; t < T; ++t) {
      if (t > P.t0) {
        double zn[NZ];
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
          double sacc = rb_ld(KNOT(D.kvec, t, NZ), RB(a), lb);
#pragma unroll
          for (int c2 = 0; c2 < NZ; ++c2) sacc += rb_ld(KNOT(D.Kmat, t, NZ * NZ), RB(a * NZ + c2), lb) * zz[c2];
          zn[a] = -sacc;
        }
#pragma unroll
        for (int a = 0; a < NZ; ++a) zz[a] = zn[a];
      }
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        rb_st(KNOT(D.zstep, t, NZ), RB(a), lb, zz[a]);
        gd += rb_ld(KNOT(gtc, t, NZ), RB(a), oG) * zz[a];
        z2 += zz[a] * zz[a];
      }
    }
    D.pred[b] = -0.5 * gd + 0.5 * mu * z2;
  }
  D.mu[b] = mu;
  D.iters[b] = iters + 1;
  return true;
}

#ifndef OH_HOST_PORT
template <int N>
__global__ __launch_bounds__(256) void k_eval_lg(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot) {
  eval_unit<N, true>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0, &GP, &GB);
}
template <int N>
__global__ __launch_bounds__(64) void k_step_lg(FigParams P, FigBuffers D, GuardBuffers GB, const int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool alive = (b < D.B) && (D.status[b] < 0);
  const bool skipping = alive && D.skip[b];
  const bool running = alive && !skipping;
  {
    const unsigned long long m = __ballot(running);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(D.work, (unsigned long long)__popcll(m));
  }
  bool still = skipping;
  if (skipping) D.skip[b] = 0;
  if (running) still = step_instance<N, true>(P, D, b, slot, &GB);
  const unsigned long long m2 = __ballot(still);
  if ((threadIdx.x & 63) == 0 && m2) atomicAdd(D.n_running, __popcll(m2));
}
bool oh_launch_eval_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
  if (n == 7) hipLaunchKernelGGL(k_eval_lg<7>, g, b, 0, s, P, D, GP, GB, slot);
  else if (n == 6) hipLaunchKernelGGL(k_eval_lg<6>, g, b, 0, s, P, D, GP, GB, slot);
  else return false;
  return true;
}
bool oh_launch_step_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
  const dim3 g((D.B + 63) / 64), b(64);
  if (n == 7) hipLaunchKernelGGL(k_step_lg<7>, g, b, 0, s, P, D, GB, slot);
  else if (n == 6) hipLaunchKernelGGL(k_step_lg<6>, g, b, 0, s, P, D, GB, slot);
  else return false;
  return true;
}
#endif  // OH_HOST_PORT

#ifndef OH_HOST_PORT
template <int N>
__global__ __launch_bounds__(64, 2) void k_step(FigParams P, FigBuffers D, const int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool alive = (b < D.B) && (D.status[b] < 0);
  const bool skipping = alive && D.skip[b];
  const bool running = alive && !skipping;
  {
    const unsigned long long m = __ballot(running);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(D.work, (unsigned long long)__popcll(m));
  }
  bool still = skipping;
  if (skipping) D.skip[b] = 0;
  if (running) still = step_instance<N>(P, D, b, slot);
  const unsigned long long m2 = __ballot(still);
  if ((threadIdx.x & 63) == 0 && m2) atomicAdd(D.n_running, __popcll(m2));
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

template <int N>
__global__ __launch_bounds__(64) void k_tail(FigParams P, FigBuffers D, const int slot) {
  constexpr int NZ = N - 3;
  constexpr int NP = NZ * (NZ + 1) / 2;
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
  const oh_chain* ch = D.chain;

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

  // accepted point (per lane = per knot)
  double q_c[N], Z_c[N][NZ], Dr_c[NP], E_c[NZ * NZ], gt_c[NZ], g_c[N], G_c[N], e_c[3] = {0.0, 0.0, 0.0}, JZ_c[3][NZ];
  double e_tgt[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int a = 0; a < NZ; ++a) JZ_c[m][a] = 0.0;
  double f_cur = 0.0, feas_cur = 0.0, pred = 0.0, stat = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) { q_c[k] = qt[k]; g_c[k] = 0.0; G_c[k] = 0.0; }
#pragma unroll
  for (int a = 0; a < NZ; ++a) gt_c[a] = 0.0;
  double Kmine[NZ * NZ], kmine[NZ], zmine[NZ];

  for (;;) {
    ++n_launch_equiv;
    // ---- evaluate the trial knots (k_eval) -----------------------------------------------------------------
    double phi = 0.0, cv = 0.0, g[N], Dr[NP], Z[N][NZ];
    const bool exact = (P.hessian == OH_HESSIAN_EXACT) || (P.hessian == OH_HESSIAN_HYBRID && !first && stat <= P.hyb_switch);
    const bool have_G = exact && !first;
    double e_new[3] = {0.0, 0.0, 0.0}, JZ_new[3][NZ];
    if (active) eval_knot<N>(ch, P, t, qt, pc, Rc, exact, have_G, G_c, phi, cv, g, Dr, Z, !first, e_tgt, retract_tol(P, !first, pred, stat), e_new, JZ_new);
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
    if (active) couple_knot<N>(P.kappa, last, qm, qt, qp, g, Z, Zn, phi, G, gt, E, merit);
    // ---- phase A: merit in knot order, ratio test (wave-uniform) ---------------------------------------------
    double f = fconst, feas = 0.0;
    for (int l = 0; l < nK; ++l) {
      f += bcast(merit, l);
      feas = fmax(feas, bcast(cv, l));
    }
    bool accept;
    if (first) {
      if (!(f == f) || !(fabs(f) < 1e300)) {  // non-finite seed / parameters
        status = OH_STATUS_NUMERICAL;
        f_cur = f;
        stat = f;
        break;
      }
      accept = true;
      first = false;
    } else if (polish) {
      accept = true;
      polish = false;
    } else {
      const LMState lm_before = lm;
      accept = lm_accept(P, f, feas, f_cur, pred, stat, lm);
      if (!accept) ++n_reject;
      if (!accept && feas_cur > 10.0 * P.tol_retract) {  // see step_instance: re-retract the accepted point before blaming the model
        lm = lm_before;
        polish = true;
        pred = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) qt[j] = q_c[j];
#pragma unroll
        for (int m = 0; m < 3; ++m) e_tgt[m] = e_c[m];
        ++iters;
        if (iters >= P.max_iter + 40) { status = OH_STATUS_MAX_ITER; break; }
        continue;
      }
    }
    if (accept) {
      f_cur = f;
      feas_cur = feas;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        q_c[k] = qt[k];
        g_c[k] = g[k];
        G_c[k] = G[k];
#pragma unroll
        for (int a = 0; a < NZ; ++a) Z_c[k][a] = Z[k][a];
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) Dr_c[i] = Dr[i];
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) E_c[i] = E[i];
#pragma unroll
      for (int a = 0; a < NZ; ++a) gt_c[a] = gt[a];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        e_c[m] = e_new[m];
#pragma unroll
        for (int a = 0; a < NZ; ++a) JZ_c[m][a] = JZ_new[m][a];
      }
    }
    double mu = lm.mu;
    // ---- phase B: backward Riccati sweep, knot l's blocks broadcast to the whole wave --------------------------
    double S[NP], rd[NZ], rn[NZ];
    for (int attempt = 0; attempt < 40; ++attempt) {
      bool ok = true;
      stat = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) S[i] = bcast(Dr_c[i], nK - 1);
#pragma unroll
      for (int a = 0; a < NZ; ++a) {
        S[tri(a, a)] += kap2 + mu;
        rn[a] = bcast(gt_c[a], nK - 1);
        stat = fmax(stat, fabs(rn[a]));
      }
      for (int l = nK - 2; l >= 0; --l) {
        double El[NZ * NZ], Ht[NP], gl[NZ];
#pragma unroll
        for (int i = 0; i < NZ * NZ; ++i) El[i] = bcast(E_c[i], l);
#pragma unroll
        for (int i = 0; i < NP; ++i) Ht[i] = bcast(Dr_c[i], l);
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
          gl[a] = bcast(gt_c[a], l);
          stat = fmax(stat, fabs(gl[a]));
          Ht[tri(a, a)] += 2.0 * kap2 + mu;
        }
        double Kmat[NZ * NZ], kv[NZ];
        ok = riccati_back<NZ>(S, rd, rn, El, Ht, gl, Kmat, kv) && ok;
        if (lane == l + 1) {
#pragma unroll
          for (int i = 0; i < NZ * NZ; ++i) Kmine[i] = Kmat[i];
#pragma unroll
          for (int a = 0; a < NZ; ++a) kmine[a] = kv[a];
        }
      }
      ok = chol_rcp<NZ>(S, rd, 1e-12) && ok;
      if (ok) break;
      mu = fmax(4.0 * mu, 1e-2);
    }
    lm.mu = mu;
    if (stat <= P.tol && feas_cur <= P.tol_feas) { status = OH_STATUS_CONVERGED; break; }
    if (iters >= P.max_iter) { status = OH_STATUS_MAX_ITER; break; }
    if (!(stat == stat)) { status = OH_STATUS_NUMERICAL; break; }
    // ---- forward recursion (wave-uniform), each lane keeps its knot's z -------------------------------------------
    {
      double zz[NZ];
#pragma unroll
      for (int a = 0; a < NZ; ++a) zz[a] = -rn[a];
      fsub_rcp<NZ>(S, rd, zz);
      bsub_rcp<NZ>(S, rd, zz);
      double gd = 0.0, z2 = 0.0;
      for (int l = 0; l < nK; ++l) {
        if (l > 0) {
          double zn[NZ];
#pragma unroll
          for (int a = 0; a < NZ; ++a) {
            double sacc = bcast(kmine[a], l);
#pragma unroll
            for (int c2 = 0; c2 < NZ; ++c2) sacc += bcast(Kmine[a * NZ + c2], l) * zz[c2];
            zn[a] = -sacc;
          }
#pragma unroll
          for (int a = 0; a < NZ; ++a) zz[a] = zn[a];
        }
        if (lane == l) {
#pragma unroll
          for (int a = 0; a < NZ; ++a) zmine[a] = zz[a];
        }
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
          gd += bcast(gt_c[a], l) * zz[a];
          z2 += zz[a] * zz[a];
        }
      }
      pred = -0.5 * gd + 0.5 * mu * z2;
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
      double v = e_c[m];
#pragma unroll
      for (int a = 0; a < NZ; ++a) v += JZ_c[m][a] * zmine[a];
      e_tgt[m] = v;
    }
    ++iters;
  }

  // ---- hand the result to k_finalize ---------------------------------------------------------------------------
  if (active) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      D.q[slot][IDX(t, N, j)] = q_c[j];
      D.g[slot][IDX(t, N, j)] = g_c[j];
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
  }
}

#endif  // OH_HOST_PORT

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
  const oh_chain* ch = D.chain;
  const double* __restrict__ qs = D.q[cur];
  const double kap2 = 2.0 * P.kappa;
  double q[N], G[N];
  const bool last = (t == P.T - 1);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    q[k] = qs[IDX(t, N, k)];
    G[k] = D.g[cur][IDX(t, N, k)] + kap2 * (q[k] - qs[IDX(t - 1, N, k)]);
    if (!last) G[k] -= kap2 * (qs[IDX(t + 1, N, k)] - q[k]);
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
  if (D.lam_h) knot_multipliers<N>(P, D, b, t, cur, D.lam_h + (ob * P.T + t) * 4);
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
#ifndef OH_HOST_PORT
template <int N>
__global__ __launch_bounds__(256) void k_finalize(FigParams P, FigBuffers D, int only_done, double* __restrict__ x, double* __restrict__ f,
                                                  double* __restrict__ kkt, int* __restrict__ iters, int* __restrict__ status) {
  finalize_unit<N>(P, D, only_done, x, f, kkt, iters, status, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
#endif

#ifndef OH_HOST_PORT
// ---- batch compaction: drop finished instances so that the tail of slow instances keeps full waves ----
// newidx[b] = new position of b among the running instances (or -1), deterministic.  With sort != 0 the survivors are ordered by how
// far they still are from a stationary point (binary exponent of the reduced gradient, 8 classes, order kept within a class):
// instances that will finish at about the same time share wavefronts, so whole waves retire between compactions instead of riding
// along with a few live lanes, and the lanes of a wave run similar numbers of retraction passes.  Keeping the order within a class
// also keeps the gather/scatter of the compaction coalesced (neighbours stay neighbours).
// Three launches: per-block class counts -> offsets of every (class, block) -> positions.  (One 1024-thread block did the whole
// batch at first: 200 us per compaction at B = 262 144.)
constexpr int SCAN_NB = 8;       // classes
constexpr int SCAN_TPB = 256;    // threads per block
constexpr int SCAN_MAXBLK = 1024;
OH_DEV int scan_class(const FigBuffers& D, const int b, const int sort) {
  if (!sort) return 0;
  int e = 0;
  frexp(D.stat[b], &e);
  const int k = (e + 24) / 4;
  return k < 0 ? 0 : (k > SCAN_NB - 1 ? SCAN_NB - 1 : k);
}
// counts of this thread's contiguous run of instances, then an inclusive scan over the block's threads in LDS
OH_DEV void scan_block(const FigBuffers& D, const int sort, const int chunk, int (&c)[SCAN_NB], int (*cnt)[SCAN_TPB], int& lo, int& hi) {
  const int tid = threadIdx.x;
  const int per = (chunk + SCAN_TPB - 1) / SCAN_TPB;
  const int blo = blockIdx.x * chunk;
  lo = blo + tid * per;
  hi = min(min(D.B, blo + chunk), lo + per);
#pragma unroll
  for (int k = 0; k < SCAN_NB; ++k) c[k] = 0;
  for (int b = lo; b < hi; ++b)
    if (D.status[b] < 0) {
      const int kb = scan_class(D, b, sort);
#pragma unroll
      for (int k = 0; k < SCAN_NB; ++k) c[k] += (k == kb);
    }
#pragma unroll
  for (int k = 0; k < SCAN_NB; ++k) cnt[k][tid] = c[k];
  __syncthreads();
  for (int off = 1; off < SCAN_TPB; off <<= 1) {
    int v[SCAN_NB];
#pragma unroll
    for (int k = 0; k < SCAN_NB; ++k) v[k] = (tid >= off) ? cnt[k][tid - off] : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_NB; ++k) cnt[k][tid] += v[k];
    __syncthreads();
  }
}
__global__ __launch_bounds__(SCAN_TPB) void k_scan_count(FigBuffers D, const int sort, const int chunk, int* __restrict__ blk) {
  __shared__ int cnt[SCAN_NB][SCAN_TPB];
  int c[SCAN_NB], lo, hi;
  scan_block(D, sort, chunk, c, cnt, lo, hi);
  if (threadIdx.x < SCAN_NB) blk[threadIdx.x * SCAN_MAXBLK + blockIdx.x] = cnt[threadIdx.x][SCAN_TPB - 1];
}
// blk[k][j] (counts) -> exclusive offsets in class-major, block-minor order; *n_new = number of survivors
__global__ __launch_bounds__(SCAN_MAXBLK) void k_scan_offsets(FigBuffers D, const int nblk, int* __restrict__ blk) {
  __shared__ int sc[SCAN_MAXBLK];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int k = 0; k < SCAN_NB; ++k) {
    const int mine = (tid < nblk) ? blk[k * SCAN_MAXBLK + tid] : 0;
    sc[tid] = mine;
    __syncthreads();
    for (int off = 1; off < SCAN_MAXBLK; off <<= 1) {
      const int v = (tid >= off) ? sc[tid - off] : 0;
      __syncthreads();
      sc[tid] += v;
      __syncthreads();
    }
    if (tid < nblk) blk[k * SCAN_MAXBLK + tid] = carry + sc[tid] - mine;
    __syncthreads();
    if (tid == 0) carry += sc[SCAN_MAXBLK - 1];
    __syncthreads();
  }
  if (tid == 0) *D.n_new = carry;
}
__global__ __launch_bounds__(SCAN_TPB) void k_scan_assign(FigBuffers D, const int sort, const int chunk, const int* __restrict__ blk) {
  __shared__ int cnt[SCAN_NB][SCAN_TPB];
  int c[SCAN_NB], lo, hi;
  scan_block(D, sort, chunk, c, cnt, lo, hi);
  int pos[SCAN_NB];
#pragma unroll
  for (int k = 0; k < SCAN_NB; ++k) pos[k] = blk[k * SCAN_MAXBLK + blockIdx.x] + cnt[k][threadIdx.x] - c[k];
  for (int b = lo; b < hi; ++b) {
    if (D.status[b] < 0) {
      const int kb = scan_class(D, b, sort);
      int p = 0;
#pragma unroll
      for (int k = 0; k < SCAN_NB; ++k)
        if (k == kb) p = pos[k]++;
      D.newidx[b] = p;
    } else {
      D.newidx[b] = -1;
    }
  }
}

// gather the persistent state of running instances (accepted knots + a few scalars) into scratch ...
template <int N>
__global__ __launch_bounds__(256) void k_compact_gather(FigParams P, FigBuffers D) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int nb = D.newidx[b];
  if (nb < 0) return;
  const double* __restrict__ qs = D.q[D.cur[b]];
  double* __restrict__ tq = D.Z[1];   // scratch: stage data is rebuilt after compaction
  double* __restrict__ ts = D.Dr[1];
#pragma unroll
  for (int j = 0; j < N; ++j) tq[((size_t)t * N + j) * Bp + nb] = qs[IDX(t, N, j)];
  if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) {
#pragma unroll
    for (int j = 0; j < N; ++j) D.g[1][((size_t)t * N + j) * Bp + nb] = D.Gfull[D.cur[b]][IDX(t, N, j)];
  }
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) ts[(size_t)i * Bp + nb] = D.ref[(size_t)i * Bp + b];
    ts[(size_t)12 * Bp + nb] = D.fconst[b];
    ts[(size_t)13 * Bp + nb] = D.mu[b];
    ts[(size_t)14 * Bp + nb] = (double)D.iters[b];
    ts[(size_t)15 * Bp + nb] = (double)D.orig[b];
    ts[(size_t)16 * Bp + nb] = D.nun[b];
  }
}
// ... and lay it down densely; the state machine restarts at "evaluate this point" (first = 1), which
// re-derives the pending step bit-identically, so an instance's iterates do not depend on the batch.
template <int N>
__global__ __launch_bounds__(256) void k_compact_scatter(FigParams P, FigBuffers D, int Bnew, const int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= Bnew) return;
  const double* __restrict__ tq = D.Z[1];
  const double* __restrict__ ts = D.Dr[1];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double v = tq[IDX(t, N, j)];
    D.q[slot][IDX(t, N, j)] = v;
    if (t < P.t0) D.q[1 - slot][IDX(t, N, j)] = v;
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) D.Gfull[1 - slot][IDX(t, N, j)] = D.g[1][IDX(t, N, j)];
  }
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) D.ref[(size_t)i * Bp + b] = ts[(size_t)i * Bp + b];
    D.fconst[b] = ts[(size_t)12 * Bp + b];
    D.mu[b] = ts[(size_t)13 * Bp + b];
    D.nun[b] = ts[(size_t)16 * Bp + b];
    const int it = (int)ts[(size_t)14 * Bp + b];
    D.iters[b] = it > 0 ? it - 1 : 0;  // the pending step is recomputed and counted again
    D.orig[b] = (int)ts[(size_t)15 * Bp + b];
    D.cur[b] = 1 - slot;
    D.first[b] = 1;
    D.skip[b] = 0;
    D.polish[b] = 0;
    D.stale[b] = 0;
    D.status[b] = -1;
  }
}

// ---- compaction that carries the pending trial along (between k_retract and k_evalb of an iteration) -------------------------------
// The restart above costs the survivors one evaluation (their pending step is re-derived).  Here the retracted trial knots move with
// the instance, so the iteration simply continues on the dense batch: per knot the trial q, the accepted q (the fall-back point) and
// the Lagrangian gradient of the accepted point, per instance the scalars of the ratio test and the LM state.  What does not move is
// the rest of the accepted point's stage data (V, Dr, E, gt, model): it is only needed again if this very trial is rejected (1-2 %);
// such an instance is flagged `stale` and, if rejected, restarts from its accepted q like after a plain compaction.
// Instances that sit the launch out (skip) or are at a restart point (first) have no trial: q[slot] is their accepted point, they restart.
template <int N>
__global__ __launch_bounds__(256) void k_carry_gather(FigParams P, FigBuffers D, const int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= D.B) return;
  const int nb = D.newidx[b];
  if (nb < 0) return;
  const bool restart = D.skip[b] != 0 || D.first[b] != 0;
  const int cur = restart ? slot : 1 - slot;
  double* __restrict__ t_trial = D.Z[0];                                // scratch rows [t][0..N)
  double* __restrict__ t_cur = D.Z[0] + (size_t)P.T * N * Bp;           // scratch rows (needs 2 N <= 3 N - 3)
  double* __restrict__ t_G = D.Z[1];
  double* __restrict__ ts = D.Dr[1];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    t_trial[((size_t)t * N + j) * Bp + nb] = D.q[slot][IDX(t, N, j)];
    t_cur[((size_t)t * N + j) * Bp + nb] = D.q[cur][IDX(t, N, j)];
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) t_G[((size_t)t * N + j) * Bp + nb] = D.Gfull[cur][IDX(t, N, j)];
  }
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) ts[(size_t)i * Bp + nb] = D.ref[(size_t)i * Bp + b];
    ts[(size_t)12 * Bp + nb] = D.fconst[b];
    ts[(size_t)13 * Bp + nb] = D.mu[b];
    ts[(size_t)14 * Bp + nb] = (double)(D.iters[b] - ((D.skip[b] != 0 && D.first[b] == 0) ? 1 : 0));  // a skipping instance re-derives its step
    ts[(size_t)15 * Bp + nb] = (double)D.orig[b];
    ts[(size_t)16 * Bp + nb] = D.nun[b];
    ts[(size_t)17 * Bp + nb] = D.f_cur[b];
    ts[(size_t)18 * Bp + nb] = D.pred[b];
    ts[(size_t)19 * Bp + nb] = D.stat[b];
    ts[(size_t)20 * Bp + nb] = D.feas[b];
    ts[(size_t)21 * Bp + nb] = (double)((restart ? 1 : 0) + 2 * (D.polish[b] != 0 ? 1 : 0));
  }
}
template <int N>
__global__ __launch_bounds__(256) void k_carry_scatter(FigParams P, FigBuffers D, int Bnew, const int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  const int Bp = D.Bp;
  if (b >= Bnew) return;
  const double* __restrict__ t_trial = D.Z[0];
  const double* __restrict__ t_cur = D.Z[0] + (size_t)P.T * N * Bp;
  const double* __restrict__ t_G = D.Z[1];
  const double* __restrict__ ts = D.Dr[1];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    D.q[slot][IDX(t, N, j)] = t_trial[IDX(t, N, j)];
    D.q[1 - slot][IDX(t, N, j)] = t_cur[IDX(t, N, j)];
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) D.Gfull[1 - slot][IDX(t, N, j)] = t_G[IDX(t, N, j)];
  }
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) D.ref[(size_t)i * Bp + b] = ts[(size_t)i * Bp + b];
    const int flags = (int)ts[(size_t)21 * Bp + b];
    const bool restart = (flags & 1) != 0;
    D.fconst[b] = ts[(size_t)12 * Bp + b];
    D.mu[b] = ts[(size_t)13 * Bp + b];
    const int it = (int)ts[(size_t)14 * Bp + b];
    D.iters[b] = it > 0 ? it : 0;
    D.orig[b] = (int)ts[(size_t)15 * Bp + b];
    D.nun[b] = ts[(size_t)16 * Bp + b];  // restart lanes too: consecutive rejections keep escalating as they would without compaction
    D.f_cur[b] = ts[(size_t)17 * Bp + b];
    D.pred[b] = ts[(size_t)18 * Bp + b];
    D.stat[b] = ts[(size_t)19 * Bp + b];
    D.feas[b] = ts[(size_t)20 * Bp + b];
    D.cur[b] = 1 - slot;
    D.first[b] = restart ? 1 : 0;
    D.skip[b] = 0;
    D.polish[b] = (!restart && (flags & 2)) ? 1 : 0;
    D.stale[b] = restart ? 0 : 1;
    D.status[b] = -1;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from oh_api.hip)
// ---------------------------------------------------------------------------------------------
bool oh_launch_rnea(hipStream_t s, const oh_dynamics* d_dyn, int nbodies, int n, const double* q, const double* qd, const double* qdd,
                    double* tau) {
  const dim3 g((n + 255) / 256), b(256);
  switch (nbodies) {
    case 2: hipLaunchKernelGGL(k_rnea<2>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 3: hipLaunchKernelGGL(k_rnea<3>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 4: hipLaunchKernelGGL(k_rnea<4>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 5: hipLaunchKernelGGL(k_rnea<5>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 6: hipLaunchKernelGGL(k_rnea<6>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 7: hipLaunchKernelGGL(k_rnea<7>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 8: hipLaunchKernelGGL(k_rnea<8>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    case 9: hipLaunchKernelGGL(k_rnea<9>, g, b, 0, s, d_dyn, n, q, qd, qdd, tau); break;
    default: return false;
  }
  return true;
}
void oh_launch_fk_jac(hipStream_t s, bool soa, const oh_chain* d_chain, int n, const double* q, double* pose, double* J) {
  const int threads = 256;
  const int blocks = (n + threads - 1) / threads;
  if (soa) hipLaunchKernelGGL(k_fk_jac<true>, dim3(blocks), dim3(threads), 0, s, d_chain, n, q, pose, J);
  else hipLaunchKernelGGL(k_fk_jac<false>, dim3(blocks), dim3(threads), 0, s, d_chain, n, q, pose, J);
}

template <int N>
static void launch_setup_t(hipStream_t s, const FigParams& P, const FigBuffers& D, const double* x0, const double* p) {
  hipLaunchKernelGGL(k_setup<N>, dim3(D.Bp / 64), dim3(64), 0, s, P, D, x0, p);
}
template <int N>
static void launch_eval_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot, int part) {
  // part 0: the whole evaluation; 1: k_retract only; 2: k_evalb only (the compaction that carries the trial along sits between them)
#if defined(OH_EVAL_FUSED)
  if (part != 2) hipLaunchKernelGGL(k_eval<N>, dim3((D.B + 255) / 256, P.T - P.t0), dim3(256), 0, s, P, D, slot);
#else
  if (part != 2) hipLaunchKernelGGL(k_retract<N>, dim3((D.B + 255) / 256, P.T - P.t0), dim3(256), 0, s, P, D, slot);
  if (part != 1) hipLaunchKernelGGL(k_evalb<N>, dim3((D.B + 255) / 256, P.T - P.t0), dim3(256), 0, s, P, D, slot);
#endif
}
template <int N>
static void launch_carry_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
  if (phase == 0) hipLaunchKernelGGL(k_carry_gather<N>, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D, slot);
  else hipLaunchKernelGGL(k_carry_scatter<N>, dim3((Bnew + 255) / 256, P.T), dim3(256), 0, s, P, D, Bnew, slot);
}
template <int N>
static void launch_couple_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot) {
  const int nbx8 = (((D.B + 255) / 256) + 7) / 8 * 8;  // instance blocks, padded to whole groups of 8 XCDs (couple_unit drops b >= B)
  hipLaunchKernelGGL(k_couple<N>, dim3(nbx8 * (P.T - P.t0)), dim3(256), 0, s, P, D, slot);
}
template <int N>
static void launch_step_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot) {
  hipLaunchKernelGGL(k_step<N>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, slot);
}
template <int N>
static void launch_tail_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot) {
  hipLaunchKernelGGL(k_tail<N>, dim3(D.B), dim3(64), 0, s, P, D, slot);
}
template <int N>
static void launch_finalize_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int only_done, double* x, double* f, double* kkt,
                              int* iters, int* status) {
  hipLaunchKernelGGL(k_finalize<N>, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D, only_done, x, f, kkt, iters, status);
}
template <int N>
static void launch_compact_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
  if (phase == 0) hipLaunchKernelGGL(k_compact_gather<N>, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D);
  else hipLaunchKernelGGL(k_compact_scatter<N>, dim3((Bnew + 255) / 256, P.T), dim3(256), 0, s, P, D, Bnew, slot);
}

#define OH_DISPATCH_N(n, call)         \
  switch (n) {                         \
    case 6: call(6); break;            \
    case 7: call(7); break;            \
    default: return false;             \
  }

bool oh_launch_setup(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const double* x0, const double* p) {
#define C(NN) launch_setup_t<NN>(s, P, D, x0, p)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_eval(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot, int part) {
#define C(NN) launch_eval_t<NN>(s, P, D, slot, part)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_eval_is_split() {  // false in the -DOH_EVAL_FUSED ablation build: the compaction between the two launches needs them
#if defined(OH_EVAL_FUSED)
  return false;
#else
  return true;
#endif
}
bool oh_launch_carry(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
#define C(NN) launch_carry_t<NN>(s, P, D, phase, Bnew, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_couple(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot) {
#define C(NN) launch_couple_t<NN>(s, P, D, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_step(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot) {
#define C(NN) launch_step_t<NN>(s, P, D, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_tail(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot) {
#define C(NN) launch_tail_t<NN>(s, P, D, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_finalize(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int only_done, double* x, double* f, double* kkt,
                        int* iters, int* status) {
#define C(NN) launch_finalize_t<NN>(s, P, D, only_done, x, f, kkt, iters, status)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
void oh_launch_scan_running(hipStream_t s, const FigBuffers& D, int sort) {
  int nblk = (D.B + 1023) / 1024;
  if (nblk > SCAN_MAXBLK) nblk = SCAN_MAXBLK;
  if (nblk < 1) nblk = 1;
  const int chunk = (D.B + nblk - 1) / nblk;
  hipLaunchKernelGGL(k_scan_count, dim3(nblk), dim3(SCAN_TPB), 0, s, D, sort, chunk, D.scan_blk);
  hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(SCAN_MAXBLK), 0, s, D, nblk, D.scan_blk);
  hipLaunchKernelGGL(k_scan_assign, dim3(nblk), dim3(SCAN_TPB), 0, s, D, sort, chunk, D.scan_blk);
}
bool oh_launch_compact(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
#define C(NN) launch_compact_t<NN>(s, P, D, phase, Bnew, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}

#endif  // OH_HOST_PORT
