// Device-side math for liboptas_hip: small fixed-size linear algebra kept entirely in VGPRs
// (all loops are compile-time unrolled so that no array is ever dynamically indexed -> no scratch),
// and the serial-chain forward kinematics that RobotModel.get_global_link_transform
// (optas/models.py:826-868) expresses as a CasADi SX graph.
#pragma once
#include <hip/hip_runtime.h>

#include "optas_hip.h"

// Function qualifiers and the reciprocal square root of the Cholesky pivots: the two things a translation unit may set before including
// this header (oracle/cpu_port builds the same device functions for the host cores with its own definitions; the product never does).
#ifndef OH_DEV
#define OH_DEV __device__ __forceinline__
#endif
#ifndef OH_RSQRT
#define OH_RSQRT(x) rsqrt(x)
#endif

// value of lane `lane` (uniform, compile-time after unrolling) on every lane: two v_readlane_b32 into scalar registers instead of two ds_bpermute_b32
// through the LDS crossbar (the pivots of the shuffle-based Gauss-Jordan sweeps sit on the critical path of every elimination step)
__device__ __forceinline__ double readlane_f64(const double v, const int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// 3-vectors / 3x3 row-major matrices
// ---------------------------------------------------------------------------------------------
OH_DEV void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
OH_DEV double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// C = A * B (3x3)
OH_DEV void mm3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// C = A * B^T
OH_DEV void mmT3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
// o = A * v
OH_DEV void mv3(const double* A, const double* v, double* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// Hamilton product a (x) b, xyzw storage.  The reference's Quaternion.__mul__ (spatialmath.py:298-312)
// is the reversed product: ref(a*b) == hamilton(b, a).
OH_DEV void qmul(const double* a, const double* b, double* o) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

// R <- R * Rot(a, theta) with Rot = c I + s [a]x + (1-c) a a^T (== Rodrigues, spatialmath.py:89-99),
// done row-wise:  r_i <- c r_i + s (r_i x a) + (1-c) (r_i . a) a.   zc = R a (unchanged by the spin).
OH_DEV void rot_axis_right(double* R, const double* a, double s, double c, double* zc) {
  const double omc = 1.0 - c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double* r = R + 3 * i;
    double x[3];
    cross3(r, a, x);
    const double d = dot3(r, a);
    zc[i] = d;
    const double k = omc * d;
    r[0] = c * r[0] + s * x[0] + k * a[0];
    r[1] = c * r[1] + s * x[1] + k * a[1];
    r[2] = c * r[2] + s * x[2] + k * a[2];
  }
}

// R <- R * Rot(+-e_m, theta) for a principal axis (code = +-(m+1)): only two columns of R mix,
// Rot(e_m): col_a' = c col_a + s col_b, col_b' = -s col_a + c col_b with (a,b) = (m+1, m+2) mod 3.
// Written out per axis with literal indices so that R stays in registers.
#define OH_ROT_COLS(A, B, M)                              \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) {         \
    const double ca = R[3 * i + A], cb = R[3 * i + B];    \
    R[3 * i + A] = c * ca + s * cb;                       \
    R[3 * i + B] = c * cb - s * ca;                       \
    zc[i] = sg * R[3 * i + M];                            \
  }
OH_DEV void rot_principal_right(double* R, const int code, double s, const double c, double* zc) {
  const double sg = (code < 0) ? -1.0 : 1.0;
  s *= sg;
  const int m = (code < 0 ? -code : code);
  if (m == 1) { OH_ROT_COLS(1, 2, 0) }
  else if (m == 2) { OH_ROT_COLS(2, 0, 1) }
  else { OH_ROT_COLS(0, 1, 2) }
}
#undef OH_ROT_COLS

// ---------------------------------------------------------------------------------------------
// sin/cos for joint angles (|x| up to a few thousand radians): two-term Cody-Waite reduction by pi/2
// and the fdlibm kernel polynomials on [-pi/4, pi/4]; < 1 ulp, ~30 FMA-class instructions instead of the
// generic library routine with its large-argument path.  7 of these per FK evaluation.
// ---------------------------------------------------------------------------------------------
OH_DEV void sincos_joint(const double x, double* s, double* c) {
  const double kf = rint(x * 6.36619772367581382433e-01);  // 2/pi
  double r = fma(-kf, 1.57079632673412561417e+00, x);        // pio2_1  (33 bits)
  r = fma(-kf, 6.07710050650619224932e-11, r);               // pio2_1t
  const double z = r * r;
  // sin kernel
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  const double v = z * r;
  const double sr = fma(v, fma(z, ps, -1.66666666666666324348e-01), r);
  // cos kernel
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w1 = 1.0 - hz;
  const double cr = w1 + (((1.0 - w1) - hz) + z * (z * pc));
  const int k = (int)kf;
  const bool swap = k & 1;
  const double ss = swap ? cr : sr;
  const double cc = swap ? sr : cr;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}

// ---------------------------------------------------------------------------------------------
// Forward kinematics of a folded serial chain whose k-th actuated joint reads q[k] (solver path:
// chain covers all model joints in order).  Outputs: R,p = frame after the last joint (before the
// tool transform), z[k] = world joint axis, pj[k] = world joint origin.
// ---------------------------------------------------------------------------------------------
// frame that follows the parameterised lead joint (oh_chain.has_lead) at angle theta: [lead_R0 lead_p0] Rot(lead_axis, theta)
OH_DEV void lead_base(const oh_chain* __restrict__ ch, const double theta, double (&Rb)[9], double (&pb)[3]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) Rb[i] = ch->lead_R0[i];
  pb[0] = ch->lead_p0[0]; pb[1] = ch->lead_p0[1]; pb[2] = ch->lead_p0[2];
  double s, c, zc[3];
  sincos_joint(theta, &s, &c);
  rot_axis_right(Rb, ch->lead_axis, s, c, zc);
}

// BASE: start from the frame (Rb, pb) instead of the root frame (chains with a parameterised lead joint)
template <int N, bool BASE = false>
OH_DEV void fk_chain(const oh_chain* __restrict__ ch, const double (&q)[N], double (&R)[9], double (&p)[3],
                     double (&z)[N][3], double (&pj)[N][3], const double* __restrict__ Rb = nullptr, const double* __restrict__ pb = nullptr) {
  if constexpr (BASE) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rb[i];
    p[0] = pb[0]; p[1] = pb[1]; p[2] = pb[2];
  } else {
    R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
    R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
    R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
    p[0] = p[1] = p[2] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double t[3];
    mv3(R, ch->p0[k], t);
    p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
    if (!ch->r0ident[k]) {
      double Rn[9];
      mm3(R, ch->R0[k], Rn);
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    pj[k][0] = p[0]; pj[k][1] = p[1]; pj[k][2] = p[2];
    if (ch->jtype[k] == 0) {
      double s, c;
      sincos_joint(q[k], &s, &c);
      const int code = ch->axcode[k];
      if (code != 0) rot_principal_right(R, code, s, c, z[k]);
      else rot_axis_right(R, ch->axis[k], s, c, z[k]);
    } else {
      mv3(R, ch->axis[k], z[k]);
      p[0] += z[k][0] * q[k]; p[1] += z[k][1] * q[k]; p[2] += z[k][2] * q[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Dense symmetric positive definite helpers, packed lower storage idx(i,j) = i(i+1)/2 + j, j<=i.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// In-place Cholesky of packed lower S (M x M).  Returns false if a pivot is <= piv_min.
template <int M>
OH_DEV bool chol_packed(double (&S)[M * (M + 1) / 2], double piv_min) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    double d = S[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= S[tri(j, k)] * S[tri(j, k)];
    if (!(d > piv_min)) { ok = false; d = 1.0; }
    const double l = sqrt(d);
    const double inv = 1.0 / l;
    S[tri(j, j)] = l;
#pragma unroll
    for (int i = j + 1; i < M; ++i) {
      double v = S[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= S[tri(i, k)] * S[tri(j, k)];
      S[tri(i, j)] = v * inv;
    }
  }
  return ok;
}
// x <- L^{-1} x
template <int M>
OH_DEV void fsub(const double (&L)[M * (M + 1) / 2], double (&x)[M]) {
#pragma unroll
  for (int i = 0; i < M; ++i) {
    double v = x[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[tri(i, k)] * x[k];
    x[i] = v / L[tri(i, i)];
  }
}
// x <- L^{-T} x
template <int M>
OH_DEV void bsub(const double (&L)[M * (M + 1) / 2], double (&x)[M]) {
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    double v = x[i];
#pragma unroll
    for (int k = i + 1; k < M; ++k) v -= L[tri(k, i)] * x[k];
    x[i] = v / L[tri(i, i)];
  }
}

// Cholesky with reciprocal pivots (divisions are off the critical path of the Riccati chain).
template <int M>
OH_DEV bool chol_rcp(double (&S)[M * (M + 1) / 2], double (&rd)[M], double piv_min) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    double d = S[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= S[tri(j, k)] * S[tri(j, k)];
    if (!(d > piv_min)) { ok = false; d = 1.0; }
    const double inv = OH_RSQRT(d);
    rd[j] = inv;
    S[tri(j, j)] = d * inv;
#pragma unroll
    for (int i = j + 1; i < M; ++i) {
      double v = S[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= S[tri(i, k)] * S[tri(j, k)];
      S[tri(i, j)] = v * inv;
    }
  }
  return ok;
}
template <int M>
OH_DEV void fsub_rcp(const double (&L)[M * (M + 1) / 2], const double (&rd)[M], double (&x)[M]) {
#pragma unroll
  for (int i = 0; i < M; ++i) {
    double v = x[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[tri(i, k)] * x[k];
    x[i] = v * rd[i];
  }
}
template <int M>
OH_DEV void bsub_rcp(const double (&L)[M * (M + 1) / 2], const double (&rd)[M], double (&x)[M]) {
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    double v = x[i];
#pragma unroll
    for (int k = i + 1; k < M; ++k) v -= L[tri(k, i)] * x[k];
    x[i] = v * rd[i];
  }
}
