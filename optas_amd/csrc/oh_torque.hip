// OH_PROBLEM_TORQUE_MPC -- BASELINE configs[4] / SURVEY 8(a) H5, App. B.5: joint-space torque MPC with the inverse dynamics
// RobotModel.rnea (optas/models.py:1731-1884) as equality rows,
//     x = [vec(Q); vec(dQ); vec(ddQ); vec(TAU)]  (robot time_derivs [0,1,2] + TaskModel "tau", derivs_align; builder.py:90-99)
//     a = [qc - q_0; dqc - dq_0; Euler rows of integrate_model_states(.., 1, dt) and (.., 2, dt)]      (builder.py:419-469,525-539)
//     h = TAU - rnea(Q, dQ, ddQ)                                                                        (builder.py:354)
//     k = [TAU - lo; up - TAU]                                enforce_model_limits("tau")               (builder.py:471-509)
//     f = w_path sum ||p_link(q_t) - goal_t||^2 + w_vel sum ||dQ||^2 + w_tau sum ||TAU||^2.
// Lowering: the linear rows and h are eliminated exactly (TAU_t = rnea(q_t, dq_t, ddq_t); q, dq rolled out from u_t = ddq_t with
// x_{t+1} = A x_t + B u_t, A = [[I, dt I], [0, I]], B = [0; dt I]), the effort rows enter through the Powell-Hestenes-Rockafellar
// augmented Lagrangian, and each iteration is a Levenberg-Marquardt step of the Gauss-Newton model from a Riccati sweep over the
// stages z_t = (q_t, dq_t | u_t).  numpy restatement of this state machine: oracle/torque.py:solve_torque_lm.
//
// Kernels:
//   k_tq_eval   one lane per (instance, knot, tangent direction), 3 units x 21 directions per wavefront.  Every lane runs the reference's
//               Newton-Euler recursion on dual numbers seeded with its direction (d tau / d z_d is the tangent output: the derivative is the
//               derivative of the literal recursion by construction), the chain walk for p_link and its Jacobian column, then the lanes of a
//               unit exchange their columns through LDS and write the Gauss-Newton stage block H_t (packed lower 21 x 21) and gradient.
//   k_tq_step   16 lanes per instance (4 instances per wavefront): ratio test, costate recursion for the stationarity measure,
//               outer augmented-Lagrangian logic, Riccati sweep with the value matrix P (14 x 14) in LDS and one column per lane,
//               forward rollout of the trial point.
#include "oh_device.h"
#include "oh_kernels.h"

namespace {

// ---- dual numbers -----------------------------------------------------------------------------------------------------------------
struct Dual {
  double v, d;
};
OH_DEV Dual operator+(const Dual a, const Dual b) { return {a.v + b.v, a.d + b.d}; }
OH_DEV Dual operator-(const Dual a, const Dual b) { return {a.v - b.v, a.d - b.d}; }
OH_DEV Dual operator*(const Dual a, const Dual b) { return {a.v * b.v, fma(a.v, b.d, a.d * b.v)}; }
OH_DEV Dual operator+(const Dual a, const double b) { return {a.v + b, a.d}; }
OH_DEV Dual operator+(const double a, const Dual b) { return {a + b.v, b.d}; }
OH_DEV Dual operator-(const Dual a, const double b) { return {a.v - b, a.d}; }
OH_DEV Dual operator-(const double a, const Dual b) { return {a - b.v, -b.d}; }
OH_DEV Dual operator*(const Dual a, const double b) { return {a.v * b, a.d * b}; }
OH_DEV Dual operator*(const double a, const Dual b) { return {a * b.v, a * b.d}; }
OH_DEV Dual operator-(const Dual a) { return {-a.v, -a.d}; }

OH_DEV void sincosT(const double x, double* s, double* c) { sincos_joint(x, s, c); }
OH_DEV void sincosT(const Dual x, Dual* s, Dual* c) {
  double sv, cv;
  sincos_joint(x.v, &sv, &cv);
  *s = {sv, cv * x.d};
  *c = {cv, -sv * x.d};
}

// ---- one primal, three tangents (round 3) --------------------------------------------------------------------------------------------
// k_tq_eval used to run the recursion once per tangent direction (21 lanes per unit, each carrying the primal: 63 primal-equivalents).  A
// lane now owns one JOINT j and carries the tangents with respect to q_j, dq_j and ddq_j next to one primal.  What depends on the joint
// angles alone -- sines, cosines, the joint rotations, the joint axes in their body frames -- has a single tangent (DualR); the velocities,
// accelerations and wrenches have all three (Dual3); the products between the two classes never form the two tangents that are zero by
// construction.  Per unit: 7 lanes x (1 primal + ~4 tangent-equivalents) instead of 21 x 3.
struct DualR {
  double v, d;  // d / d q_j
};
struct Dual3 {
  double v, d0, d1, d2;  // d / d q_j, d / d dq_j, d / d ddq_j
};
OH_DEV DualR operator+(const DualR a, const DualR b) { return {a.v + b.v, a.d + b.d}; }
OH_DEV DualR operator-(const DualR a, const DualR b) { return {a.v - b.v, a.d - b.d}; }
OH_DEV DualR operator*(const DualR a, const DualR b) { return {a.v * b.v, fma(a.v, b.d, a.d * b.v)}; }
OH_DEV DualR operator+(const DualR a, const double b) { return {a.v + b, a.d}; }
OH_DEV DualR operator+(const double a, const DualR b) { return {a + b.v, b.d}; }
OH_DEV DualR operator-(const DualR a, const double b) { return {a.v - b, a.d}; }
OH_DEV DualR operator-(const double a, const DualR b) { return {a - b.v, -b.d}; }
OH_DEV DualR operator*(const DualR a, const double b) { return {a.v * b, a.d * b}; }
OH_DEV DualR operator*(const double a, const DualR b) { return {a * b.v, a * b.d}; }
OH_DEV DualR operator-(const DualR a) { return {-a.v, -a.d}; }
OH_DEV Dual3 operator+(const Dual3 a, const Dual3 b) { return {a.v + b.v, a.d0 + b.d0, a.d1 + b.d1, a.d2 + b.d2}; }
OH_DEV Dual3 operator-(const Dual3 a, const Dual3 b) { return {a.v - b.v, a.d0 - b.d0, a.d1 - b.d1, a.d2 - b.d2}; }
OH_DEV Dual3 operator*(const Dual3 a, const Dual3 b) {
  return {a.v * b.v, fma(a.v, b.d0, a.d0 * b.v), fma(a.v, b.d1, a.d1 * b.v), fma(a.v, b.d2, a.d2 * b.v)};
}
OH_DEV Dual3 operator*(const DualR a, const Dual3 b) { return {a.v * b.v, fma(a.v, b.d0, a.d * b.v), a.v * b.d1, a.v * b.d2}; }
OH_DEV Dual3 operator*(const Dual3 a, const DualR b) { return b * a; }
OH_DEV Dual3 operator+(const Dual3 a, const DualR b) { return {a.v + b.v, a.d0 + b.d, a.d1, a.d2}; }
OH_DEV Dual3 operator+(const DualR a, const Dual3 b) { return b + a; }
OH_DEV Dual3 operator-(const Dual3 a, const DualR b) { return {a.v - b.v, a.d0 - b.d, a.d1, a.d2}; }
OH_DEV Dual3 operator-(const DualR a, const Dual3 b) { return {a.v - b.v, a.d - b.d0, -b.d1, -b.d2}; }
OH_DEV Dual3 operator+(const Dual3 a, const double b) { return {a.v + b, a.d0, a.d1, a.d2}; }
OH_DEV Dual3 operator+(const double a, const Dual3 b) { return {a + b.v, b.d0, b.d1, b.d2}; }
OH_DEV Dual3 operator-(const Dual3 a, const double b) { return {a.v - b, a.d0, a.d1, a.d2}; }
OH_DEV Dual3 operator-(const double a, const Dual3 b) { return {a - b.v, -b.d0, -b.d1, -b.d2}; }
OH_DEV Dual3 operator*(const Dual3 a, const double b) { return {a.v * b, a.d0 * b, a.d1 * b, a.d2 * b}; }
OH_DEV Dual3 operator*(const double a, const Dual3 b) { return {a * b.v, a * b.d0, a * b.d1, a * b.d2}; }
OH_DEV Dual3 operator-(const Dual3 a) { return {-a.v, -a.d0, -a.d1, -a.d2}; }
OH_DEV void sincosT(const DualR x, DualR* s, DualR* c) {
  double sv, cv;
  sincos_joint(x.v, &sv, &cv);
  *s = {sv, cv * x.d};
  *c = {cv, -sv * x.d};
}

// scalar class of what depends on the joint angles alone, given the class of the velocities / accelerations / wrenches
template <class S>
struct RotOf {
  using T = S;
};
template <>
struct RotOf<Dual3> {
  using T = DualR;
};

template <class A, class B>
struct Prom {
  using T = Dual;
};
template <>
struct Prom<double, double> {
  using T = double;
};
template <> struct Prom<DualR, DualR> { using T = DualR; };
template <> struct Prom<DualR, double> { using T = DualR; };
template <> struct Prom<double, DualR> { using T = DualR; };
template <> struct Prom<Dual3, Dual3> { using T = Dual3; };
template <> struct Prom<Dual3, double> { using T = Dual3; };
template <> struct Prom<double, Dual3> { using T = Dual3; };
template <> struct Prom<Dual3, DualR> { using T = Dual3; };
template <> struct Prom<DualR, Dual3> { using T = Dual3; };
template <class A, class B>
OH_DEV void crossT(const A* a, const B* b, typename Prom<A, B>::T* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// o = M v, o = M^T v (row-major 3x3)
template <class A, class B>
OH_DEV void mvT(const A* M, const B* v, typename Prom<A, B>::T* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2];
}
template <class A, class B>
OH_DEV void mTvT(const A* M, const B* v, typename Prom<A, B>::T* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = M[i] * v[0] + M[3 + i] * v[1] + M[6 + i] * v[2];
}
// R = R0 Rot(a, theta), Rot = c I + s [a]x + (1 - c) a a^T (spatialmath.py:89-99), row-wise as in rot_axis_right
template <class S>
OH_DEV void joint_rotation(const double* R0, const double* a, const S s, const S c, S* R) {
  const S omc = 1.0 - c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double* r = R0 + 3 * i;
    double x[3];
    cross3(r, a, x);
    const double d = dot3(r, a);
    const S k = omc * d;
    R[3 * i + 0] = c * r[0] + s * x[0] + k * a[0];
    R[3 * i + 1] = c * r[1] + s * x[1] + k * a[1];
    R[3 * i + 2] = c * r[2] + s * x[2] + k * a[2];
  }
}

// RobotModel.rnea (models.py:1819-1880) on scalars S (double or Dual): NB bodies, the last one on a fixed joint.
// The loops over the bodies are kept rolled (the per-body wrenches f, nn and sin/cos live in lane-private memory, indexed by the
// loop counter): unrolled, the dual-number recursion needs ~1500 live registers and the compiler spills two thirds of them.
template <int NB, class S, class SR = typename RotOf<S>::T>
OH_DEV void rnea_forward_body(const oh_dynamics* __restrict__ dy, const int i, const bool moving, const SR qi, const S qdi, const S qddi, S (&om)[3],
                              S (&omD)[3], S (&vD)[3], S* __restrict__ fi, S* __restrict__ ni, SR& sji, SR& cji) {
  S omi[3], omDi[3], vDi[3];
  S t1[3], t2[3], t3[3], acc[3];
  crossT(omD, dy->xyz[i], t1);
  crossT(om, dy->xyz[i], t2);
  crossT(om, t2, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = vD[k] + t1[k] + t3[k];
  if (moving) {
    SR Rp[9];
    sincosT(qi, &sji, &cji);
    joint_rotation(dy->R0[i], dy->axis[i], sji, cji, Rp);
    SR a[3];
    S omp[3], omDp[3];
    mTvT(Rp, dy->axis[i], a);  // iaxisi
    mTvT(Rp, om, omp);
    mTvT(Rp, omD, omDp);
    S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
    S cr[3];
    crossT(omp, aq, cr);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      omi[k] = omp[k] + aq[k];
      omDi[k] = omDp[k] + cr[k] + a[k] * qddi;
    }
    mTvT(Rp, acc, vDi);
  } else {
    mTvT(dy->R0[i], om, omi);
    mTvT(dy->R0[i], omD, omDi);
    mTvT(dy->R0[i], acc, vDi);
  }
  crossT(omDi, dy->com[i], t1);
  crossT(omi, dy->com[i], t2);
  crossT(omi, t2, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) fi[k] = dy->mass[i] * (vDi[k] + t1[k] + t3[k]);
  S Io[3], IoD[3];
  mvT(dy->inertia[i], omi, Io);
  mvT(dy->inertia[i], omDi, IoD);
  crossT(omi, Io, t1);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ni[k] = IoD[k] + t1[k];
    om[k] = omi[k];
    omD[k] = omDi[k];
    vD[k] = vDi[k];
  }
}

template <int NB, class S, class SR = typename RotOf<S>::T>
OH_DEV void rnea_lit(const oh_dynamics* __restrict__ dy, const SR (&q)[NB - 1], const S (&qd)[NB - 1], const S (&qdd)[NB - 1], S (&tau)[NB - 1]) {
  S f[NB][3], nn[NB][3];
  SR sj[NB], cj[NB];
  S om[3], omD[3], vD[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    om[k] = S{};
    omD[k] = S{};
    vD[k] = S{} + dy->vd0[k];
  }
#pragma unroll 1
  for (int i = 0; i < NB - 1; ++i) rnea_forward_body<NB, S>(dy, i, true, q[i], qd[i], qdd[i], om, omD, vD, f[i], nn[i], sj[i], cj[i]);
  rnea_forward_body<NB, S>(dy, NB - 1, false, SR{}, S{}, S{}, om, omD, vD, f[NB - 1], nn[NB - 1], sj[NB - 1], cj[NB - 1]);
  // backward (models.py:1858-1880); the reference's fs/ns lists carry a leading zero entry: fs[i] == f[i-1]
  S ifi[3] = {f[NB - 1][0], f[NB - 1][1], f[NB - 1][2]};
  S ini[3], t1[3];
  crossT(dy->com[NB - 1], f[NB - 1], t1);
#pragma unroll
  for (int k = 0; k < 3; ++k) ini[k] = nn[NB - 1][k] + t1[k];
#pragma unroll 1
  for (int i = NB - 1; i >= 1; --i) {
    S a1[3], a2[3], a3[3], a4[3];
    if (i < NB - 1) {
      SR pRi[9];
      joint_rotation(dy->R0[i], dy->axis[i], sj[i], cj[i], pRi);
      mvT(pRi, ini, a1);
      mvT(pRi, ifi, a3);
    } else {
      mvT(dy->R0[i], ini, a1);
      mvT(dy->R0[i], ifi, a3);
    }
    crossT(dy->com[i - 1], f[i - 1], a2);
    crossT(dy->xyz[i], a3, a4);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ini[k] = nn[i - 1][k] + a1[k] + a2[k] + a4[k];
      ifi[k] = a3[k] + f[i - 1][k];
    }
    SR pR[9], ax[3];
    joint_rotation(dy->R0[i - 1], dy->axis[i - 1], sj[i - 1], cj[i - 1], pR);
    mTvT(pR, dy->axis[i - 1], ax);  // pRi^T axis
    tau[i - 1] = ini[0] * ax[0] + ini[1] * ax[1] + ini[2] * ax[2];
  }
}

// d tau / d (q, qd, qdd) of RobotModel.rnea (what the reference obtains with casadi.jacobian of the same graph, optimization.py:8-24): one lane per
// (sample, direction), the literal recursion on dual numbers.  q, qd, qdd [n][N] -> J [n][N][3 N] row-major.
template <int N>
__global__ __launch_bounds__(64) void k_rnea_jac(const oh_dynamics* __restrict__ dy, const int n, const double* __restrict__ q, const double* __restrict__ qd,
                                                 const double* __restrict__ qdd, double* __restrict__ J) {
  constexpr int NZ = 3 * N, UPW = 64 / NZ;
  const int lane = threadIdx.x;
  const int ul = lane / NZ, d = lane - ul * NZ;
  const long long u = (long long)blockIdx.x * UPW + ul;
  if (ul >= UPW || u >= n) return;
  Dual a[N], b[N], c[N], tau[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    a[j] = {q[u * N + j], d == j ? 1.0 : 0.0};
    b[j] = {qd[u * N + j], d == N + j ? 1.0 : 0.0};
    c[j] = {qdd[u * N + j], d == 2 * N + j ? 1.0 : 0.0};
  }
  rnea_lit<N + 1, Dual>(dy, a, b, c, tau);
#pragma unroll
  for (int i = 0; i < N; ++i) J[((size_t)u * N + i) * NZ + d] = tau[i].d;
}

OH_DEV size_t xs_off(const TqBuffers& D, const int T, const int slot, const int b, const int t) { return (((size_t)slot * D.B + b) * T + t) * TQ_XS; }
OH_DEV size_t st_off(const TqBuffers& D, const int T, const int slot, const int b, const int t) { return (((size_t)slot * D.B + b) * T + t) * TQ_SD; }

// ---- setup: reference layouts -> device records ---------------------------------------------------------------------------------
// x0 [B][4 N T] = [vec(Q); vec(dQ); vec(ddQ); vec(TAU)] (only ddQ is a free variable: q, dq are rolled out, tau follows from the dynamics rows);
// p [B][2 N + 3 T] = [qc; dqc; vec(goal 3 x T)].
template <int N>
__global__ __launch_bounds__(64) void k_tq_setup(TqParams P, TqBuffers D, const double* __restrict__ x0, const double* __restrict__ p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.B) return;
  const int T = P.T;
  const double* xb = x0 + (size_t)b * P.nx;
  const double* pb = p + (size_t)b * P.np;
  double q[N], dq[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    q[j] = pb[j];
    dq[j] = pb[N + j];
  }
  for (int t = 0; t < T; ++t) {
    double* r = D.xs + xs_off(D, T, 0, b, t);
    double* r1 = D.xs + xs_off(D, T, 1, b, t);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double u = xb[2 * N * T + N * t + j];
      r[j] = q[j];
      r[8 + j] = dq[j];
      r[16 + j] = u;
      r1[j] = q[j];
      r1[8 + j] = dq[j];
      r1[16 + j] = u;
      q[j] = q[j] + P.dt * dq[j];
      dq[j] = dq[j] + P.dt * u;
    }
    r[7] = r[15] = r[23] = r1[7] = r1[15] = r1[23] = 0.0;
    double* gl = D.goal + ((size_t)b * T + t) * 4;
    gl[0] = pb[2 * N + 3 * t];
    gl[1] = pb[2 * N + 3 * t + 1];
    gl[2] = pb[2 * N + 3 * t + 2];
    gl[3] = 0.0;
    double* lm = D.lam + ((size_t)b * T + t) * TQ_LAM;
#pragma unroll
    for (int j = 0; j < TQ_LAM; ++j) lm[j] = 0.0;
  }
  D.f_cur[b] = 0.0;
  D.f_true[b] = 0.0;
  D.pred[b] = 0.0;
  D.mu[b] = P.mu0;
  D.nun[b] = 2.0;
  D.rho[b] = P.rho0;
  D.rho_next[b] = P.rho0;
  D.omega[b] = fmax(P.tol, 1e-2);
  D.meas_prev[b] = 1e300;
  D.hcnt[b] = 0;
  D.aa[b] = 0;
  D.meas[b] = 0.0;
  D.stat[b] = 0.0;
  D.cur[b] = 1;  // the seed sits in slot 0 = the first "trial"
  D.first[b] = 1;
  D.outer[b] = 0;
  D.status[b] = -1;
  D.iters[b] = 0;
  D.rejected[b] = 0;
  D.n_outer[b] = 0;
  D.list[b] = b;
}

__global__ __launch_bounds__(256) void k_tq_list(TqBuffers D) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.B) return;
  if (D.status[b] < 0) D.list[atomicAdd(D.n_list, 1)] = b;
}

// ---- evaluation -----------------------------------------------------------------------------------------------------------------------
#ifndef OH_TQ_EVAL_WAVES
#define OH_TQ_EVAL_WAVES 2
#endif
// Joint-velocity rows on the velocity states (oh_torque_desc.dq_lo / dq_up; enforce_model_limits(name, time_deriv=1), builder.py:471-509), round 3:
// stage-local rows of the state, through the same augmented Lagrangian and outer loop as the effort rows.  Every lane of a unit runs this
// (cheap, and all of them need psi / meas); `writer` stores the refreshed multipliers at an outer update.  cv[j] joins the gradient component
// of dq_j, dv[j] its diagonal entry of the Gauss-Newton block.
template <int N>
OH_DEV void tq_velocity_rows(const TqParams& P, double* __restrict__ lm, const double (&dqv)[N], const double rho, const double rho_old, const bool outer,
                             const bool writer, double& psi, double& meas, double& viol, double& cmpl, double (&cv)[N], double (&dv)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double g_lo = dqv[i] - P.dq_lo[i], g_up = P.dq_up[i] - dqv[i];
    double l_lo = lm[16 + i], l_up = lm[16 + N + i];
    if (outer) {
      l_lo = fmax(0.0, l_lo - rho_old * g_lo);
      l_up = fmax(0.0, l_up - rho_old * g_up);
    }
    const double s_lo = fmax(0.0, l_lo - rho * g_lo), s_up = fmax(0.0, l_up - rho * g_up);
    psi += (s_lo * s_lo - l_lo * l_lo) / (2.0 * rho) + (s_up * s_up - l_up * l_up) / (2.0 * rho);
    meas = fmax(meas, fmax(fabs(fmin(g_lo, l_lo / rho)), fabs(fmin(g_up, l_up / rho))));
    viol = fmax(viol, fmax(-g_lo, -g_up));
    cmpl = fmax(cmpl, fmax(fabs(s_lo * g_lo), fabs(s_up * g_up)));
    cv[i] = s_up - s_lo;
    dv[i] = rho * ((s_lo > 0.0 ? 1.0 : 0.0) + (s_up > 0.0 ? 1.0 : 0.0));
    if (outer && writer) {
      lm[16 + i] = l_lo;
      lm[16 + N + i] = l_up;
    }
  }
}

template <int N, bool VEL = false>
__global__ __launch_bounds__(64, OH_TQ_EVAL_WAVES) void k_tq_eval(TqParams P, TqBuffers D) {
  constexpr int NZ = 3 * N;  // 21 tangent directions: q, dq, ddq
  constexpr int UPW = 64 / NZ;  // units per wavefront (3)
  __shared__ double tile[UPW][N + 3][NZ + 1];
  const int T = P.T;
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane == 0) *D.n_running = 0;  // k_tq_step, next in the stream, counts the instances that go on
  int ul = lane / NZ, d = lane - ul * NZ;
  if (ul >= UPW) {  // lane 63 has no unit: it rides along and parks its LDS writes in the padding column
    ul = UPW - 1;
    d = NZ;
  }
  const long long n_units = (long long)D.n_run * T;
  long long unit = (long long)blockIdx.x * UPW + ul;
  bool active = d < NZ && unit < n_units;
  if (unit >= n_units) unit = n_units - 1;
  const int li = (int)(unit / T), t = (int)(unit - (long long)li * T);
  const int b = D.list[li];
  if (D.status[b] >= 0) active = false;
  if (!__any(active)) return;  // one wavefront per block: every instance of this wavefront has finished since the list was built
  const int ts = 1 - D.cur[b];
  const double* xr = D.xs + xs_off(D, T, ts, b, t);
  const bool outer = D.outer[b] != 0;
  const double rho_old = D.rho[b];
  const double rho = outer ? D.rho_next[b] : rho_old;

  // inverse dynamics on dual numbers seeded with direction d
  Dual q[N], qd[N], qdd[N], tau[N];
  double qv[N], dqv[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    qv[j] = xr[j];
    dqv[j] = xr[8 + j];
    q[j] = {qv[j], d == j ? 1.0 : 0.0};
    qd[j] = {dqv[j], d == N + j ? 1.0 : 0.0};
    qdd[j] = {xr[16 + j], d == 2 * N + j ? 1.0 : 0.0};
  }
  rnea_lit<N + 1, Dual>(D.dyn, q, qd, qdd, tau);

  // effort rows through the augmented Lagrangian: psi(g, lam, rho) = (max(0, lam - rho g)^2 - lam^2) / (2 rho)
  double* lm = D.lam + ((size_t)b * T + t) * TQ_LAM;
  double cw[N], dw[N];
  double psi = 0.0, meas = 0.0, viol = 0.0, cmpl = 0.0, tau2 = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double tv = tau[i].v;
    const double g_lo = tv - P.tau_lo[i], g_up = P.tau_up[i] - tv;
    double l_lo = lm[i], l_up = lm[N + i];
    if (outer) {
      l_lo = fmax(0.0, l_lo - rho_old * g_lo);
      l_up = fmax(0.0, l_up - rho_old * g_up);
      if (active && d == 0) {
        lm[i] = l_lo;
        lm[N + i] = l_up;
      }
    }
    const double s_lo = fmax(0.0, l_lo - rho * g_lo), s_up = fmax(0.0, l_up - rho * g_up);
    psi += (s_lo * s_lo - l_lo * l_lo) / (2.0 * rho) + (s_up * s_up - l_up * l_up) / (2.0 * rho);
    meas = fmax(meas, fmax(fabs(fmin(g_lo, l_lo / rho)), fabs(fmin(g_up, l_up / rho))));
    viol = fmax(viol, fmax(-g_lo, -g_up));
    cmpl = fmax(cmpl, fmax(fabs(s_lo * g_lo), fabs(s_up * g_up)));
    cw[i] = 2.0 * P.w_tau * tv - s_lo + s_up;
    dw[i] = 2.0 * P.w_tau + rho * ((s_lo > 0.0 ? 1.0 : 0.0) + (s_up > 0.0 ? 1.0 : 0.0));
    tau2 += tv * tv;
  }
  double cv[N], dv[N];
#pragma unroll
  for (int i = 0; i < N; ++i) cv[i] = dv[i] = 0.0;
  if constexpr (VEL) tq_velocity_rows<N>(P, lm, dqv, rho, rho_old, outer, active && d == 0, psi, meas, viol, cmpl, cv, dv);

  // link position and column d of its Jacobian (models.py:826-868, 1211-1264)
  double R[9], pp[3], z[N][3], pj[N][3];
  fk_chain<N>(D.chain, qv, R, pp, z, pj);
  double e[3], tv3[3];
  mv3(R, D.chain->p_tool, tv3);
  const double* gl = D.goal + ((size_t)b * T + t) * 4;
  double r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e[k] = pp[k] + tv3[k];
    r[k] = e[k] - gl[k];
  }
  double zd[3] = {0.0, 0.0, 0.0}, pd[3] = {0.0, 0.0, 0.0};
  int jt_d = 0;
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k == d) {
      zd[0] = z[k][0]; zd[1] = z[k][1]; zd[2] = z[k][2];
      pd[0] = pj[k][0]; pd[1] = pj[k][1]; pd[2] = pj[k][2];
      jt_d = D.chain->jtype[k];
    }
  double jp[3] = {0.0, 0.0, 0.0};
  if (d < N) {
    if (jt_d == 0) {
      const double dd[3] = {e[0] - pd[0], e[1] - pd[1], e[2] - pd[2]};
      cross3(zd, dd, jp);
    } else {
      jp[0] = zd[0]; jp[1] = zd[1]; jp[2] = zd[2];
    }
  }

  // gradient component d of the stage cost
  double gd = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) gd = fma(cw[i], tau[i].d, gd);
  gd += 2.0 * P.w_path * dot3(jp, r);
  double dqd = 0.0, cvd = 0.0, dvd = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (d == N + j) {
      dqd = dqv[j];
      cvd = cv[j];
      dvd = dv[j];
    }
  gd += 2.0 * P.w_vel * dqd;
  gd += cvd;  // (its own statement: the sum above keeps the rounding it had before the velocity rows existed)

  // exchange the columns through LDS and form column d of the Gauss-Newton block (rows d..NZ-1)
#pragma unroll
  for (int i = 0; i < N; ++i) tile[ul][i][d] = tau[i].d;
#pragma unroll
  for (int k = 0; k < 3; ++k) tile[ul][N + k][d] = jp[k];
  __syncthreads();
  double* sr = D.st + st_off(D, T, ts, b, t);
  if (active) {
    for (int rr = d; rr < NZ; ++rr) {
      double hv = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) hv = fma(dw[i] * tile[ul][i][rr], tau[i].d, hv);
      double hp = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) hp = fma(tile[ul][N + k][rr], jp[k], hp);
      hv = fma(2.0 * P.w_path, hp, hv);
      if (rr == d && d >= N && d < 2 * N) {
        hv += 2.0 * P.w_vel;
        hv += dvd;
      }
      sr[rr * (rr + 1) / 2 + d] = hv;
    }
    sr[231 + d] = gd;
    if (d == 0) {
      double dq2 = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) dq2 = fma(dqv[j], dqv[j], dq2);
      const double phi_true = P.w_path * dot3(r, r) + P.w_vel * dq2 + P.w_tau * tau2;
      sr[252] = phi_true + psi;
      sr[253] = phi_true;
      sr[254] = meas;
      sr[255] = viol;
      sr[263] = cmpl;
    }
    if (d < N) {
      double tvd = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i == d) tvd = tau[i].v;
      sr[256 + d] = tvd;
    }
  }
}

// The same evaluation with one lane per (instance, knot, JOINT): 9 units x 7 joints per wavefront.  Lane j runs the recursion once on
// (DualR, Dual3) scalars seeded with q_j, dq_j and ddq_j, so it ends up with columns j, N + j and 2 N + j of d tau / d z; the link position
// adds column j of its Jacobian.  The columns meet in LDS as before and every lane writes ITS THREE columns of the packed stage block.
// Same arithmetic per entry as k_tq_eval (sums over the 7 torque rows in the same order), so the two agree to rounding; bound: f64 FMA.
#ifndef OH_TQ_EVAL3_WAVES
#define OH_TQ_EVAL3_WAVES 1
#endif
template <int N, bool VEL = false>
__global__ __launch_bounds__(64, OH_TQ_EVAL3_WAVES) void k_tq_eval3(TqParams P, TqBuffers D) {
  constexpr int NZ = 3 * N;
  constexpr int UPW = 64 / N;  // units per wavefront (9)
  __shared__ double tile[UPW][N + 3][NZ + 1];
  const int T = P.T;
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane == 0) *D.n_running = 0;  // k_tq_step, next in the stream, counts the instances that go on
  int ul = lane / N, j = lane - ul * N;
  const bool lane_ok = ul < UPW;
  if (!lane_ok) {  // lane 63 has no unit: it rides along on the last unit and keeps its hands off the tile
    ul = UPW - 1;
    j = N - 1;
  }
  const long long n_units = (long long)D.n_run * T;
  long long unit = (long long)blockIdx.x * UPW + ul;
  bool active = lane_ok && unit < n_units;
  if (unit >= n_units) unit = n_units - 1;
  const int li = (int)(unit / T), t = (int)(unit - (long long)li * T);
  const int b = D.list[li];
  if (D.status[b] >= 0) active = false;
  if (!__any(active)) return;
  const int ts = 1 - D.cur[b];
  const double* xr = D.xs + xs_off(D, T, ts, b, t);
  const bool outer = D.outer[b] != 0;
  const double rho_old = D.rho[b];
  const double rho = outer ? D.rho_next[b] : rho_old;

  DualR q[N];
  Dual3 qd[N], qdd[N], tau[N];
  double qv[N], dqv[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double one = (k == j) ? 1.0 : 0.0;
    qv[k] = xr[k];
    dqv[k] = xr[8 + k];
    q[k] = {qv[k], one};
    qd[k] = {dqv[k], 0.0, one, 0.0};
    qdd[k] = {xr[16 + k], 0.0, 0.0, one};
  }
  rnea_lit<N + 1, Dual3>(D.dyn, q, qd, qdd, tau);

  // effort rows through the augmented Lagrangian (every lane of the unit computes them: they are cheap and everyone needs cw, dw)
  double* lm = D.lam + ((size_t)b * T + t) * TQ_LAM;
  double cw[N], dw[N];
  double psi = 0.0, meas = 0.0, viol = 0.0, cmpl = 0.0, tau2 = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double tv = tau[i].v;
    const double g_lo = tv - P.tau_lo[i], g_up = P.tau_up[i] - tv;
    double l_lo = lm[i], l_up = lm[N + i];
    if (outer) {
      l_lo = fmax(0.0, l_lo - rho_old * g_lo);
      l_up = fmax(0.0, l_up - rho_old * g_up);
    }
    const double s_lo = fmax(0.0, l_lo - rho * g_lo), s_up = fmax(0.0, l_up - rho * g_up);
    psi += (s_lo * s_lo - l_lo * l_lo) / (2.0 * rho) + (s_up * s_up - l_up * l_up) / (2.0 * rho);
    meas = fmax(meas, fmax(fabs(fmin(g_lo, l_lo / rho)), fabs(fmin(g_up, l_up / rho))));
    viol = fmax(viol, fmax(-g_lo, -g_up));
    cmpl = fmax(cmpl, fmax(fabs(s_lo * g_lo), fabs(s_up * g_up)));
    cw[i] = 2.0 * P.w_tau * tv - s_lo + s_up;
    dw[i] = 2.0 * P.w_tau + rho * ((s_lo > 0.0 ? 1.0 : 0.0) + (s_up > 0.0 ? 1.0 : 0.0));
    tau2 += tv * tv;
    if (outer && active && j == 0) {  // (after every lane of the wavefront has read the old values: lanes run in lock step, the reads above precede this store)
      lm[i] = l_lo;
      lm[N + i] = l_up;
    }
  }
  double cv[N], dv[N];
#pragma unroll
  for (int i = 0; i < N; ++i) cv[i] = dv[i] = 0.0;
  if constexpr (VEL) tq_velocity_rows<N>(P, lm, dqv, rho, rho_old, outer, active && j == 0, psi, meas, viol, cmpl, cv, dv);

  // link position and column j of its Jacobian (models.py:826-868, 1211-1264)
  double R[9], pp[3], z[N][3], pj[N][3];
  fk_chain<N>(D.chain, qv, R, pp, z, pj);
  double e[3], tv3[3];
  mv3(R, D.chain->p_tool, tv3);
  const double* gl = D.goal + ((size_t)b * T + t) * 4;
  double r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e[k] = pp[k] + tv3[k];
    r[k] = e[k] - gl[k];
  }
  double zd[3] = {0.0, 0.0, 0.0}, pd[3] = {0.0, 0.0, 0.0};
  int jt_d = 0;
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k == j) {
      zd[0] = z[k][0]; zd[1] = z[k][1]; zd[2] = z[k][2];
      pd[0] = pj[k][0]; pd[1] = pj[k][1]; pd[2] = pj[k][2];
      jt_d = D.chain->jtype[k];
    }
  double jp[3];
  if (jt_d == 0) {
    const double dd[3] = {e[0] - pd[0], e[1] - pd[1], e[2] - pd[2]};
    cross3(zd, dd, jp);
  } else {
    jp[0] = zd[0]; jp[1] = zd[1]; jp[2] = zd[2];
  }

  // gradient components j, N + j, 2 N + j of the stage cost
  double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    g0 = fma(cw[i], tau[i].d0, g0);
    g1 = fma(cw[i], tau[i].d1, g1);
    g2 = fma(cw[i], tau[i].d2, g2);
  }
  g0 += 2.0 * P.w_path * dot3(jp, r);
  double dqj = 0.0, cvj = 0.0, dvj = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k == j) {
      dqj = dqv[k];
      cvj = cv[k];
      dvj = dv[k];
    }
  g1 += 2.0 * P.w_vel * dqj;
  g1 += cvj;

  // exchange the columns through LDS; the link position has no dq / ddq columns
  if (lane_ok) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      tile[ul][i][j] = tau[i].d0;
      tile[ul][i][N + j] = tau[i].d1;
      tile[ul][i][2 * N + j] = tau[i].d2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      tile[ul][N + k][j] = jp[k];
      tile[ul][N + k][N + j] = 0.0;
      tile[ul][N + k][2 * N + j] = 0.0;
    }
  }
  __syncthreads();
  double* sr = D.st + st_off(D, T, ts, b, t);
  if (active) {
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
      const int d = c3 * N + j;
      double col[N];
#pragma unroll
      for (int i = 0; i < N; ++i) col[i] = c3 == 0 ? tau[i].d0 : (c3 == 1 ? tau[i].d1 : tau[i].d2);
      for (int rr = d; rr < NZ; ++rr) {
        double hv = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) hv = fma(dw[i] * tile[ul][i][rr], col[i], hv);
        if (c3 == 0) {
          double hp = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) hp = fma(tile[ul][N + k][rr], jp[k], hp);
          hv = fma(2.0 * P.w_path, hp, hv);
        }
        if (rr == d && c3 == 1) {
          hv += 2.0 * P.w_vel;
          hv += dvj;
        }
        sr[rr * (rr + 1) / 2 + d] = hv;
      }
    }
    sr[231 + j] = g0;
    sr[231 + N + j] = g1;
    sr[231 + 2 * N + j] = g2;
    if (j == 0) {
      double dq2 = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) dq2 = fma(dqv[k], dqv[k], dq2);
      const double phi_true = P.w_path * dot3(r, r) + P.w_vel * dq2 + P.w_tau * tau2;
      sr[252] = phi_true + psi;
      sr[253] = phi_true;
      sr[254] = meas;
      sr[255] = viol;
      sr[263] = cmpl;
    }
    double tvd = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i == j) tvd = tau[i].v;
    sr[256 + j] = tvd;
  }
}

// ---- step ------------------------------------------------------------------------------------------------------------------------------
OH_DEV double group_sum(double v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 16);
  return v;
}
OH_DEV double group_max(double v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 16));
  return v;
}

#ifndef OH_TQ_STEP_WAVES
#define OH_TQ_STEP_WAVES 1
#endif
template <int N>
__global__ __launch_bounds__(64, OH_TQ_STEP_WAVES) void k_tq_step(TqParams P, TqBuffers D) {
  constexpr int NX = 2 * N, NZ = 3 * N, NH = NZ * (NZ + 1) / 2, NU = N * (N + 1) / 2;
  constexpr int PS = NX + 1;  // row stride of P in LDS
  constexpr int OFF_P = 256, OFF_PV = OFF_P + NX * PS, OFF_QUX = OFF_PV + 16, OFF_DX = OFF_QUX + N * 16, OFF_DU = OFF_DX + 16, LDS_N = OFF_DU + 8;
  static_assert(NH + NZ <= 256, "stage block does not fit the LDS window");
  __shared__ double sm[4][LDS_N];
  const int T = P.T;
  const int gi = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int b_raw = blockIdx.x * 4 + gi;
  const bool valid = b_raw < D.n_run;
  const int b = D.list[valid ? b_raw : D.n_run - 1];
  double* S = sm[gi];
  double* Hs = S;
  double* Ps = S + OFF_P;
  double* pv = S + OFF_PV;
  double* Quxs = S + OFF_QUX;
  double* dxs = S + OFF_DX;
  double* dus = S + OFF_DU;
  const double dt = P.dt;

  bool run = valid && D.status[b] < 0;
  if (!__any(run)) return;
  int cur = D.cur[b];
  const int ts = 1 - cur;
  // merit of the trial point
  double fsum = 0.0, ftrue = 0.0, meas_t = 0.0;
  for (int t = c; t < T; t += 16) {
    const double* sr = D.st + st_off(D, T, ts, b, t);
    fsum += sr[252];
    ftrue += sr[253];
    meas_t = fmax(meas_t, sr[254]);
  }
  fsum = group_sum(fsum);
  ftrue = group_sum(ftrue);
  meas_t = group_max(meas_t);

  const bool first = D.first[b] != 0, outer = D.outer[b] != 0;
  double f_cur = D.f_cur[b], mu = D.mu[b], nun = D.nun[b], rho = D.rho[b], omega = D.omega[b], meas_prev = D.meas_prev[b];
  double meas_cur = D.meas[b], f_true = D.f_true[b];
  const double rho_next_in = D.rho_next[b];
  const double pred = D.pred[b];
  int iters = D.iters[b], rejected = D.rejected[b], n_outer = D.n_outer[b];
  // Anderson acceleration (oracle/torque.py:solve_torque_lm, anderson_mix): the tracking residual does not vanish, Gauss-Newton converges
  // linearly (rate ~0.8), and its steps are the residuals of a fixed-point iteration.  Once the reduced gradient is below aa_from every
  // other trial is the point the last aa_m + 1 (control sequence, step) pairs extrapolate to; it is accepted if it lowers the merit at
  // all, otherwise the history is dropped and the Levenberg-Marquardt step follows.
  int hcnt = D.hcnt[b];
  const bool aa_trial = D.aa[b] != 0;
  bool aa_was = false;
  bool accept;
  if (first || outer) {
    accept = true;
    if (outer) {
      rho = rho_next_in;
      n_outer += 1;
      hcnt = 0;  // the merit function changes
    }
  } else if (aa_trial) {
    accept = isfinite(fsum) && fsum < f_cur;
    aa_was = true;
    if (!accept) {
      hcnt = 0;
      rejected += 1;
    }
  } else {
    const double ratio = (f_cur - fsum) / fmax(pred, 1e-300);
    accept = isfinite(fsum) && (ratio > 1e-4 || (pred <= 1e-15 * fabs(f_cur) && fsum <= f_cur + 1e-14 * fabs(f_cur)));
    if (accept) {
      const double w = 2.0 * ratio - 1.0;
      mu *= fmax(1.0 / 3.0, 1.0 - w * w * w);
      if (mu < 1e-7) mu = 0.0;
      nun = 2.0;
    } else {
      mu = fmax(mu * nun, 1e-3);
      nun *= 2.0;
      rejected += 1;
    }
  }
  if (accept) {
    cur = ts;
    f_cur = fsum;
    f_true = ftrue;
    meas_cur = meas_t;
  }
  const int nts = 1 - cur;  // slot of the next trial

  // stationarity of the accepted point: gradient of the rolled-out objective w.r.t. u_t by the costate recursion
  // (the serial loops of this kernel walk the T knots in dependent chains: whatever a knot needs from memory is requested one knot
  // ahead, otherwise every knot costs a memory round trip -- at small batches the kernel IS those round trips: 230-350 us per launch)
  double lamc = 0.0, stat = 0.0;
  double gx_n, gu_n;
  {
    const double* sr = D.st + st_off(D, T, cur, b, T - 1);
    gx_n = c < NX ? sr[231 + c] : 0.0;
    gu_n = (c >= N && c < NX) ? sr[231 + NX + (c - N)] : 0.0;
  }
  for (int t = T - 1; t >= 0; --t) {
    const double gx = gx_n, gu = gu_n;
    if (t > 0) {
      const double* sr = D.st + st_off(D, T, cur, b, t - 1);
      gx_n = c < NX ? sr[231 + c] : 0.0;
      gu_n = (c >= N && c < NX) ? sr[231 + NX + (c - N)] : 0.0;
    }
    if (c >= N && c < NX) stat = fmax(stat, fabs(fma(dt, lamc, gu)));
    const double lq = __shfl(lamc, c >= N ? c - N : c, 16);
    lamc = c < N ? gx + lamc : gx + fma(dt, lq, lamc);
  }
  stat = group_max(stat);
  if (!(stat == stat)) stat = 1e300;

  // outer logic (oracle/torque.py:solve_torque_lm)
  int status = -1;
  bool do_outer = false, do_step = false;
  double rho_next = rho;
  if (!isfinite(f_cur)) {
    status = OH_STATUS_NUMERICAL;
  } else if (stat <= omega) {
    if (stat <= P.tol && meas_cur <= P.tol_feas) status = OH_STATUS_CONVERGED;
    else if (iters >= P.max_iter) status = OH_STATUS_MAX_ITER;
    else {
      rho_next = meas_cur > 0.25 * meas_prev ? fmin(rho * 10.0, 1e8) : rho;
      meas_prev = meas_cur;
      omega = fmax(P.tol, fmin(omega, 0.1 * meas_cur));
      do_outer = true;
      iters += 1;
    }
  } else if (iters >= P.max_iter) {
    status = OH_STATUS_MAX_ITER;
  } else {
    do_step = true;
    iters += 1;
  }
  if (!run) { do_outer = do_step = false; }

  // Riccati sweep: lane c < NX owns column c of P / Qxx / Qux; every lane factorises Quu (N x N) for itself
  double qk = 0.0;
  bool chol_ok = true;
  if (c < NX) {
#pragma unroll
    for (int r = 0; r < NX; ++r) Ps[r * PS + c] = 0.0;
    pv[c] = 0.0;
  }
  __syncthreads();
  constexpr int NREC = (NH + NZ + 15) / 16;  // values of a stage record per lane
  double hn[NREC];
  if (do_step) {
    const double* sr = D.st + st_off(D, T, cur, b, T - 1);
#pragma unroll
    for (int k = 0; k < NREC; ++k) hn[k] = (c + 16 * k < NH + NZ) ? sr[c + 16 * k] : 0.0;
  }
  for (int t = T - 1; t >= 0; --t) {
    if (do_step) {
#pragma unroll
      for (int k = 0; k < NREC; ++k)
        if (c + 16 * k < NH + NZ) Hs[c + 16 * k] = hn[k];
      if (t > 0) {
        const double* sr = D.st + st_off(D, T, cur, b, t - 1);
#pragma unroll
        for (int k = 0; k < NREC; ++k) hn[k] = (c + 16 * k < NH + NZ) ? sr[c + 16 * k] : 0.0;
      }
    }
    __syncthreads();
    double qxx[NX], qux[N], kc[N], Quu[NU], rd[N], kk[N], qu[N];
    double qx = 0.0;
    if (do_step) {
      const int cc = c < NX ? c : 0;
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
        double v = Hs[hi * (hi + 1) / 2 + lo] + (r == cc ? mu : 0.0);
        // (A^T P A)[r][cc]
        double a = Ps[r * PS + cc];
        if (r >= N) a = fma(dt, Ps[(r - N) * PS + cc], a);
        if (cc >= N) {
          a = fma(dt, Ps[r * PS + cc - N], a);
          if (r >= N) a = fma(dt * dt, Ps[(r - N) * PS + cc - N], a);
        }
        qxx[r] = v + a;
      }
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double a = dt * Ps[(N + k) * PS + cc];
        if (cc >= N) a = fma(dt * dt, Ps[(N + k) * PS + cc - N], a);
        qux[k] = Hs[(NX + k) * (NX + k + 1) / 2 + cc] + a;
        qu[k] = Hs[NH + NX + k] + dt * pv[N + k];
#pragma unroll
        for (int l = 0; l <= k; ++l) Quu[tri(k, l)] = Hs[(NX + k) * (NX + k + 1) / 2 + NX + l] + dt * dt * Ps[(N + k) * PS + N + l];
      }
      qx = Hs[NH + cc] + pv[cc] + (cc >= N ? dt * pv[cc - N] : 0.0);
      if (!chol_rcp<N>(Quu, rd, 0.0)) chol_ok = false;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        kc[k] = qux[k];
        kk[k] = qu[k];
      }
      fsub_rcp<N>(Quu, rd, kc);
      bsub_rcp<N>(Quu, rd, kc);
      fsub_rcp<N>(Quu, rd, kk);
      bsub_rcp<N>(Quu, rd, kk);
#pragma unroll
      for (int k = 0; k < N; ++k) qk = fma(qu[k], kk[k], qk);
      if (c < NX) {
#pragma unroll
        for (int k = 0; k < N; ++k) Quxs[k * 16 + c] = qux[k];
      }
    }
    __syncthreads();  // every lane has read the old P
    if (do_step && c < NX) {
      double* gn = D.gains + ((size_t)b * T + t) * TQ_GN;
#pragma unroll
      for (int k = 0; k < N; ++k) gn[c * N + k] = kc[k];
      if (c == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) gn[NX * N + k] = kk[k];
      }
      // P <- Qxx - Qux^T K, lower part of the column mirrored so that P stays exactly symmetric
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        if (r >= c) {
          double v = qxx[r];
#pragma unroll
          for (int k = 0; k < N; ++k) v = fma(-Quxs[k * 16 + r], kc[k], v);
          Ps[r * PS + c] = v;
          Ps[c * PS + r] = v;
        }
      }
      double v = qx;
#pragma unroll
      for (int k = 0; k < N; ++k) v = fma(-qux[k], kk[k], v);
      pv[c] = v;
    }
    __syncthreads();
  }
  if (do_step && !chol_ok) {
    status = OH_STATUS_NUMERICAL;
    do_step = false;
  }

  // forward rollout of the next trial point (or the copy of the accepted point for an outer update)
  double ndx = 0.0, ndu = 0.0;
  const bool use_hist = do_step && P.aa_m > 0 && stat < P.aa_from;
  {
    const bool fw = do_step || do_outer;
    double xt = 0.0;  // lane c < NX: component c of the trial state
    {
      const double* x0r = D.xs + xs_off(D, T, cur, b, 0);
      if (c < NX) xt = x0r[c < N ? c : 8 + (c - N)];
    }
    // state, control and gains of the next knot are in flight while this one is rolled out
    double xcur_n = 0.0, ucur_n = 0.0, gk_n[NX + 1];
    auto fetch_knot = [&](const int t) {
      const double* xc = D.xs + xs_off(D, T, cur, b, t);
      xcur_n = c < NX ? xc[c < N ? c : 8 + (c - N)] : 0.0;
      if (c < N) {
        ucur_n = xc[16 + c];
        if (do_step) {
          const double* gn = D.gains + ((size_t)b * T + t) * TQ_GN;
#pragma unroll
          for (int j = 0; j < NX; ++j) gk_n[j] = gn[j * N + c];
          gk_n[NX] = gn[NX * N + c];
        }
      }
    };
    fetch_knot(0);
    for (int t = 0; t < T; ++t) {  // every group runs the loop (uniform barriers); only groups with fw touch memory
      double* xn = D.xs + xs_off(D, T, nts, b, t);
      const double xcur = xcur_n, ucur = ucur_n;
      double gk[NX + 1];
#pragma unroll
      for (int j = 0; j <= NX; ++j) gk[j] = gk_n[j];
      if (t + 1 < T) fetch_knot(t + 1);
      const double dx = do_step ? xt - xcur : 0.0;
      if (c < NX) dxs[c] = dx;
      __syncthreads();
      if (c < N) {
        double du = 0.0;
        if (do_step) {
          du = -gk[NX];
#pragma unroll
          for (int j = 0; j < NX; ++j) du = fma(-gk[j], dxs[j], du);
        }
        const double un = ucur + du;
        if (fw) xn[16 + c] = un;
        dus[c] = un;
        ndu = fma(du, du, ndu);
        if (use_hist) {
          double* hr = D.hist + (((size_t)b * 4 + (hcnt & 3)) * T + t) * TQ_HS;
          hr[c] = ucur;
          hr[8 + c] = du;
        }
      }
      if (c < NX) {
        if (do_outer) xt = xcur;
        if (fw) xn[c < N ? c : 8 + (c - N)] = xt;
        ndx = fma(dx, dx, ndx);
      }
      __syncthreads();
      // x_{t+1} = A x_t + B u_t on the trial values themselves: the Euler rows hold to the rounding of one operation
      const double xo = __shfl(xt, c < N ? c + N : c, 16);
      if (c < N) xt = fma(dt, xo, xt);
      else if (c < NX) xt = fma(dt, dus[c - N], xt);
    }
    ndx = group_sum(ndx);
    ndu = group_sum(ndu);
  }
  // ---- Anderson extrapolation: replaces the trial just written -------------------------------------------------------------------------
  bool do_aa = false;
  if (use_hist) hcnt += 1;
  if (P.aa_m > 0) {  // (uniform: every group of the block walks the barriers below)
    const int h = hcnt < P.aa_m + 1 ? hcnt : P.aa_m + 1;  // entries in use, oldest first: ring slots (hcnt - h + j) & 3
    const bool want = use_hist && h >= 2 && !aa_was;
    double gam[3] = {0.0, 0.0, 0.0};
    if (want) {
      // Gram matrix of the step differences and its right-hand side: 9 sums over the T x N entries, N lanes of the group at a time
      double G[6] = {0, 0, 0, 0, 0, 0}, r[3] = {0, 0, 0};
      if (c < N) {
        for (int t = 0; t < T; ++t) {
          double F[4] = {0, 0, 0, 0};
          for (int j = 0; j < h; ++j) F[j] = D.hist[(((size_t)b * 4 + ((hcnt - h + j) & 3)) * T + t) * TQ_HS + 8 + c];
          double dF[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) dF[j] = (j + 1 < h) ? F[j + 1] - F[j] : 0.0;
          const double Fl = F[h - 1];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            r[i] = fma(dF[i], Fl, r[i]);
#pragma unroll
            for (int j = 0; j <= i; ++j) G[tri(i, j)] = fma(dF[i], dF[j], G[tri(i, j)]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) G[i] = group_sum(G[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) r[i] = group_sum(r[i]);
      const double dmax = fmax(G[0], fmax(G[2], G[5]));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (i + 1 < h) G[tri(i, i)] += 1e-10 * fmax(dmax, 1e-300);
        else G[tri(i, i)] = 1.0;  // unused row: gamma_i = 0
      }
      double rd3[3];
      if (chol_rcp<3>(G, rd3, 0.0)) {
        fsub_rcp<3>(G, rd3, r);
        bsub_rcp<3>(G, rd3, r);
        gam[0] = r[0]; gam[1] = r[1]; gam[2] = r[2];
        do_aa = isfinite(gam[0]) && isfinite(gam[1]) && isfinite(gam[2]);
      }
    }
    if (__syncthreads_or(do_aa ? 1 : 0)) {
      // open-loop rollout of the extrapolated control sequence from the fixed initial state
      double xt = 0.0;
      {
        const double* x0r = D.xs + xs_off(D, T, cur, b, 0);
        if (c < NX) xt = x0r[c < N ? c : 8 + (c - N)];
      }
      for (int t = 0; t < T; ++t) {
        double* xn = D.xs + xs_off(D, T, nts, b, t);
        if (c < N) {
          double ua = 0.0;
          if (do_aa) {
            double X[4] = {0, 0, 0, 0}, F[4] = {0, 0, 0, 0};
            for (int j = 0; j < h; ++j) {
              const double* hr = D.hist + (((size_t)b * 4 + ((hcnt - h + j) & 3)) * T + t) * TQ_HS;
              X[j] = hr[c];
              F[j] = hr[8 + c];
            }
            ua = X[h - 1] + F[h - 1];
#pragma unroll
            for (int j = 0; j < 3; ++j)
              if (j + 1 < h) ua = fma(-gam[j], (X[j + 1] - X[j]) + (F[j + 1] - F[j]), ua);
            xn[16 + c] = ua;
          }
          dus[c] = ua;
        }
        if (do_aa && c < NX) xn[c < N ? c : 8 + (c - N)] = xt;
        __syncthreads();
        const double xo = __shfl(xt, c < N ? c + N : c, 16);
        if (c < N) xt = fma(dt, xo, xt);
        else if (c < NX) xt = fma(dt, dus[c - N], xt);
        __syncthreads();
      }
    }
  }
  if (run && c == 0) {
    D.hcnt[b] = hcnt;
    D.aa[b] = do_aa ? 1 : 0;
    D.cur[b] = cur;
    D.first[b] = 0;
    D.outer[b] = do_outer ? 1 : 0;
    D.f_cur[b] = f_cur;
    D.f_true[b] = f_true;
    D.meas[b] = meas_cur;
    D.mu[b] = mu;
    D.nun[b] = nun;
    D.rho[b] = rho;
    D.rho_next[b] = rho_next;
    D.omega[b] = omega;
    D.meas_prev[b] = meas_prev;
    D.stat[b] = stat;
    D.pred[b] = 0.5 * qk + 0.5 * mu * ndx;
    D.iters[b] = iters;
    D.rejected[b] = rejected;
    D.n_outer[b] = n_outer;
    D.status[b] = status;
    if (status < 0) atomicAdd(D.n_running, 1);
  }
}

// ---- results in the reference layout ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_tq_finalize(TqParams P, TqBuffers D, double* __restrict__ x, double* __restrict__ f, double* __restrict__ kkt,
                                                    int* __restrict__ iters, int* __restrict__ status, double* __restrict__ mult) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.B) return;
  const int T = P.T, cur = D.cur[b];
  const double rho = D.rho[b];
  double viol = 0.0, cmpl = 0.0;
  for (int t = 0; t < T; ++t) {
    const double* xr = D.xs + xs_off(D, T, cur, b, t);
    const double* sr = D.st + st_off(D, T, cur, b, t);
    const double* lm = D.lam + ((size_t)b * T + t) * TQ_LAM;
    viol = fmax(viol, sr[255]);
    cmpl = fmax(cmpl, sr[263]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (x) {
        double* xb = x + (size_t)b * P.nx;
        xb[N * t + j] = xr[j];
        xb[N * T + N * t + j] = xr[8 + j];
        xb[2 * N * T + N * t + j] = xr[16 + j];
        xb[3 * N * T + N * t + j] = sr[256 + j];
      }
      if (mult) {
        const double tv = sr[256 + j];
        const int NR = P.vel ? 4 * N : 2 * N;  // rows per knot: effort rows, then (with velocity limits) [dq - dq_lo; dq_up - dq]
        mult[((size_t)b * T + t) * NR + j] = fmax(0.0, lm[j] - rho * (tv - P.tau_lo[j]));
        mult[((size_t)b * T + t) * NR + N + j] = fmax(0.0, lm[N + j] - rho * (P.tau_up[j] - tv));
        if (P.vel) {
          mult[((size_t)b * T + t) * NR + 2 * N + j] = fmax(0.0, lm[16 + j] - rho * (xr[8 + j] - P.dq_lo[j]));
          mult[((size_t)b * T + t) * NR + 3 * N + j] = fmax(0.0, lm[16 + N + j] - rho * (P.dq_up[j] - xr[8 + j]));
        }
      }
    }
  }
  if (f) f[b] = D.f_true[b];
  if (kkt) {
    kkt[3 * b] = D.stat[b];
    kkt[3 * b + 1] = viol;
    kkt[3 * b + 2] = cmpl;
  }
  if (iters) iters[b] = D.iters[b];
  if (status) status[b] = D.status[b] < 0 ? OH_STATUS_MAX_ITER : D.status[b];
}

}  // namespace

bool oh_launch_rnea_jac(hipStream_t s, const oh_dynamics* d_dyn, int nbodies, int n, const double* q, const double* qd, const double* qdd, double* J) {
#define OH_RJ(NN)                                                                                                                         \
  case NN + 1:                                                                                                                            \
    hipLaunchKernelGGL(k_rnea_jac<NN>, dim3((unsigned)((n + (64 / (3 * NN)) - 1) / (64 / (3 * NN)))), dim3(64), 0, s, d_dyn, n, q, qd, qdd, J); \
    return true;
  switch (nbodies) {
    OH_RJ(1) OH_RJ(2) OH_RJ(3) OH_RJ(4) OH_RJ(5) OH_RJ(6) OH_RJ(7) OH_RJ(8)
    default: return false;
  }
#undef OH_RJ
}
bool oh_launch_tq_setup(hipStream_t s, const TqParams& P, const TqBuffers& D, const double* x0, const double* p) {
  if (P.N != 7) return false;
  hipLaunchKernelGGL(k_tq_setup<7>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, x0, p);
  return true;
}
void oh_launch_tq_list(hipStream_t s, const TqBuffers& D) { hipLaunchKernelGGL(k_tq_list, dim3((D.B + 255) / 256), dim3(256), 0, s, D); }
bool oh_launch_tq_eval(hipStream_t s, const TqParams& P, const TqBuffers& D) {
  if (P.N != 7) return false;
  const long long units = (long long)D.n_run * P.T;
  static const int per_joint = [] { const char* e = getenv("OH_TQ_EVAL3"); return e ? atoi(e) : 1; }();  // 0: one lane per tangent direction (round 2)
  // (The round-2 kernel, one lane per tangent direction, has the shorter dependent chain on a latency-bound launch -- one instance: 170 against
  // 190 us per evaluation -- but the two kernels round differently, and choosing by launch size would make an instance's iterates depend on how
  // fast the rest of its batch drains: one kernel for every launch.  1024 instances 369 -> 260 us, 8192 instances 2.73 -> 1.77 ms per launch.)
  // (the variants with joint-velocity rows are instantiations of their own: as a run-time branch the rows cost k_tq_eval3 21 % at 8192 instances
  //  -- 250 registers at two wavefronts per SIMD with spills instead of 268 at one)
  if (per_joint && P.vel) hipLaunchKernelGGL((k_tq_eval3<7, true>), dim3((unsigned)((units + 8) / 9)), dim3(64), 0, s, P, D);
  else if (per_joint) hipLaunchKernelGGL((k_tq_eval3<7>), dim3((unsigned)((units + 8) / 9)), dim3(64), 0, s, P, D);
  else if (P.vel) hipLaunchKernelGGL((k_tq_eval<7, true>), dim3((unsigned)((units + 2) / 3)), dim3(64), 0, s, P, D);
  else hipLaunchKernelGGL((k_tq_eval<7>), dim3((unsigned)((units + 2) / 3)), dim3(64), 0, s, P, D);
  return true;
}
bool oh_launch_tq_step(hipStream_t s, const TqParams& P, const TqBuffers& D) {
  if (P.N != 7) return false;
  hipLaunchKernelGGL(k_tq_step<7>, dim3((D.n_run + 3) / 4), dim3(64), 0, s, P, D);
  return true;
}
bool oh_launch_tq_finalize(hipStream_t s, const TqParams& P, const TqBuffers& D, double* x, double* f, double* kkt, int* iters, int* status, double* mult) {
  if (P.N != 7) return false;
  hipLaunchKernelGGL(k_tq_finalize<7>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, x, f, kkt, iters, status, mult);
  return true;
}

namespace {
template <class K>
bool kernel_info(K kernel, int block, OhKernelInfo* out) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) != hipSuccess) return false;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess) nb = 0;
  *out = OhKernelInfo{a.numRegs, (int)a.localSizeBytes, (int)a.sharedSizeBytes, block, nb};
  return true;
}
}  // namespace
bool oh_kernel_info_torque(const char* name, OhKernelInfo* out) {
  const std::string n(name);
  if (n == "k_tq_eval") return kernel_info(k_tq_eval3<7>, 64, out);
  if (n == "k_tq_eval_directions") return kernel_info(k_tq_eval<7>, 64, out);
  if (n == "k_tq_step") return kernel_info(k_tq_step<7>, 64, out);
  return false;
}
