// OH_PROBLEM_TORQUE_MPC -- BASELINE configs[4] / SURVEY 8(a) H5, App. B.5: joint-space torque MPC with the inverse dynamics
// RobotModel.rnea (optas/models.py:1731-1884) as equality rows,
//     x = [vec(Q); vec(dQ); vec(ddQ); vec(TAU)]  (robot time_derivs [0,1,2] + TaskModel "tau", derivs_align; builder.py:90-99)
//     a = [qc - q_0; dqc - dq_0; Euler rows of integrate_model_states(.., 1, dt) and (.., 2, dt)]      (builder.py:419-469,525-539)
//     h = TAU - rnea(Q, dQ, ddQ)                                                                        (builder.py:354)
//     k = [TAU - lo; up - TAU]                                enforce_model_limits("tau")               (builder.py:471-509)
//     f = w_path sum ||p_link(q_t) - goal_t||^2 + w_vel sum ||dQ||^2 + w_tau sum ||TAU||^2.
// Lowering: the linear rows and h are eliminated exactly (TAU_t = rnea(q_t, dq_t, ddq_t); q, dq rolled out from u_t = ddq_t with
// x_{t+1} = A x_t + B u_t, A = [[I, dt I], [0, I]], B = [0; dt I]).  The inequality rows enter through a primal-dual interior point (round 4; the
// reference's own algorithm class, solver.py:355-398 -> IPOPT): log barrier with multipliers of their own, relaxed below delta = theta mu_b, barrier
// parameter lowered by the monotone rule of Waechter & Biegler (2006) without re-evaluation; steps are Levenberg-Marquardt / Newton steps from a Riccati
// sweep over the stages z_t = (q_t, dq_t | u_t), scaled by the fraction-to-the-boundary rule on the linearised rows.  Near the solution the stage blocks
// hold the exact Hessian of the Lagrangian (k_tq_curv).  numpy restatement of this state machine: oracle/torque_ipm.py:solve_torque_ipm; the
// augmented-Lagrangian machine of rounds 1-3 survives as oracle/torque.py:solve_torque_lm, the independent second solver of the same problem.
//
// Kernels:
//   k_tq_eval3  one lane per (instance, knot, joint), 9 units x 7 joints per wavefront: the reference's Newton-Euler recursion on dual numbers seeded
//               with q_j, dq_j, ddq_j (the derivative is the derivative of the literal recursion by construction), the chain walk for p_link, then
//               the lanes of a unit exchange their columns through LDS and write the stage block H_t (packed lower 21 x 21), the gradients of cost
//               and barrier, and d tau / d z.
//   k_tq_curv   same mapping: second derivatives of the inverse dynamics and of the link position, added to H_t for instances in the Newton phase.
//   k_tq_step   16 lanes per instance (4 instances per wavefront): ratio test, costate recursion for the stationarity measure, barrier update,
//               Riccati sweep with the value matrix P (14 x 14) in LDS and one column per lane, fraction to the boundary, rollout of the trial.
#include <type_traits>

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

// ---- two tangents (round 4): second derivatives of the inverse dynamics ----------------------------------------------------------------------
// rnea_ctau_grad below is the hand-written adjoint of the recursion; run on (DualR, Dual2) scalars seeded with q_j and dq_j it returns rows q_j and
// dq_j of  sum_i c_i d^2 tau_i / d(q, dq, ddq)^2  (the torques are linear in ddq, so the rows of ddq_j are the transposed columns of those).
struct Dual2 {
  double v, d0, d1;  // d / d q_j, d / d dq_j
};
OH_DEV Dual2 operator+(const Dual2 a, const Dual2 b) { return {a.v + b.v, a.d0 + b.d0, a.d1 + b.d1}; }
OH_DEV Dual2 operator-(const Dual2 a, const Dual2 b) { return {a.v - b.v, a.d0 - b.d0, a.d1 - b.d1}; }
OH_DEV Dual2 operator*(const Dual2 a, const Dual2 b) { return {a.v * b.v, fma(a.v, b.d0, a.d0 * b.v), fma(a.v, b.d1, a.d1 * b.v)}; }
OH_DEV Dual2 operator*(const DualR a, const Dual2 b) { return {a.v * b.v, fma(a.v, b.d0, a.d * b.v), a.v * b.d1}; }
OH_DEV Dual2 operator*(const Dual2 a, const DualR b) { return b * a; }
OH_DEV Dual2 operator+(const Dual2 a, const DualR b) { return {a.v + b.v, a.d0 + b.d, a.d1}; }
OH_DEV Dual2 operator+(const DualR a, const Dual2 b) { return b + a; }
OH_DEV Dual2 operator-(const Dual2 a, const DualR b) { return {a.v - b.v, a.d0 - b.d, a.d1}; }
OH_DEV Dual2 operator-(const DualR a, const Dual2 b) { return {a.v - b.v, a.d - b.d0, -b.d1}; }
OH_DEV Dual2 operator+(const Dual2 a, const double b) { return {a.v + b, a.d0, a.d1}; }
OH_DEV Dual2 operator+(const double a, const Dual2 b) { return {a + b.v, b.d0, b.d1}; }
OH_DEV Dual2 operator-(const Dual2 a, const double b) { return {a.v - b, a.d0, a.d1}; }
OH_DEV Dual2 operator-(const double a, const Dual2 b) { return {a - b.v, -b.d0, -b.d1}; }
OH_DEV Dual2 operator*(const Dual2 a, const double b) { return {a.v * b, a.d0 * b, a.d1 * b}; }
OH_DEV Dual2 operator*(const double a, const Dual2 b) { return {a * b.v, a * b.d0, a * b.d1}; }
OH_DEV Dual2 operator-(const Dual2 a) { return {-a.v, -a.d0, -a.d1}; }
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
template <>
struct RotOf<Dual2> {
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
template <> struct Prom<Dual2, Dual2> { using T = Dual2; };
template <> struct Prom<Dual2, double> { using T = Dual2; };
template <> struct Prom<double, Dual2> { using T = Dual2; };
template <> struct Prom<Dual2, DualR> { using T = Dual2; };
template <> struct Prom<DualR, Dual2> { using T = Dual2; };
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

// ---- the same torques by virtual work, outward pass only (round 4) -------------------------------------------------------------------------------
// tau_k = sum_{b >= k} f_b . v_b^(k) + (n_b + com_b x f_b) . w_b^(k): the inertial wrench of body b (models.py:1819-1856, the outward pass of the
// reference) paired with the twist (w^(k), v^(k)) a unit rate of joint k alone gives the frame of body b -- what the reference's inward pass
// (models.py:1858-1880) sums by handing wrenches to the parents, summed the other way round.  The twists travel outward with the recursion itself,
// so nothing has to wait for the last body: no per-body arrays.  rnea_lit keeps 8 bodies x (f, n, sin, cos) of dual numbers in lane-private memory
// (1.8 KB per lane, written once and read once: 36 KB per unit of k_tq_eval3, which made that kernel HBM-bound on its own scratch at 1.6 % of the
// bytes being useful); here the state is 7 twists in registers.  The twists depend on the joint angles alone (SR).  Equal to rnea_lit up to rounding.
template <int NB, bool MOVING, class S, class SR>
OH_DEV void vw_body(const oh_dynamics* __restrict__ dy, const int i, const SR qi, const S qdi, const S qddi, S (&om)[3], S (&omD)[3], S (&vD)[3],
                    SR (&wk)[NB - 1][3], SR (&vk)[NB - 1][3], S (&tau)[NB - 1]) {
  constexpr int NJ = NB - 1;
  using RT = typename std::conditional<MOVING, SR, double>::type;
  S t1[3], t2[3], t3[3], acc[3];
  crossT(omD, dy->xyz[i], t1);
  crossT(om, dy->xyz[i], t2);
  crossT(om, t2, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = vD[k] + t1[k] + t3[k];
  RT Rp[9];
  SR a[3];
  if constexpr (MOVING) {
    SR sj, cj;
    sincosT(qi, &sj, &cj);
    joint_rotation(dy->R0[i], dy->axis[i], sj, cj, Rp);
    mTvT(Rp, dy->axis[i], a);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = dy->R0[i][k];
  }
  // twists first: they need the parent's values of nothing else, and the registers of (om, omD, vD) of the parent die right after
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    if (k < i) {
      SR x[3], y[3], wn[3], vn[3];
      crossT(wk[k], dy->xyz[i], x);
#pragma unroll
      for (int c = 0; c < 3; ++c) y[c] = vk[k][c] + x[c];
      mTvT(Rp, y, vn);
      mTvT(Rp, wk[k], wn);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        wk[k][c] = wn[c];
        vk[k][c] = vn[c];
      }
    } else if (MOVING && k == i) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        wk[k][c] = a[c];
        vk[k][c] = SR{};
      }
    }
  }
  S omi[3], omDi[3], vDi[3];
  {
    S omp[3], omDp[3];
    mTvT(Rp, om, omp);
    mTvT(Rp, omD, omDp);
    if constexpr (MOVING) {
      S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
      S cr[3];
      crossT(omp, aq, cr);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        omi[k] = omp[k] + aq[k];
        omDi[k] = omDp[k] + cr[k] + a[k] * qddi;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        omi[k] = omp[k];
        omDi[k] = omDp[k];
      }
    }
    mTvT(Rp, acc, vDi);
  }
  S f[3], m[3];
  crossT(omDi, dy->com[i], t1);
  crossT(omi, dy->com[i], t2);
  crossT(omi, t2, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) f[k] = dy->mass[i] * (vDi[k] + t1[k] + t3[k]);
  {
    S Io[3], IoD[3];
    mvT(dy->inertia[i], omi, Io);
    mvT(dy->inertia[i], omDi, IoD);
    crossT(omi, Io, t1);
    crossT(dy->com[i], f, t2);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      m[k] = IoD[k] + t1[k] + t2[k];
      om[k] = omi[k];
      omD[k] = omDi[k];
      vD[k] = vDi[k];
    }
  }
#pragma unroll
  for (int k = 0; k < NJ; ++k)
    if (k <= i) tau[k] = tau[k] + (dotT(f, vk[k]) + dotT(m, wk[k]));
}

// qs: the unit's (q | dq | ddq) at offsets 0, 8, 16 (LDS or global: indexed by the loop counter); lane j seeds joint j.  Seed: S / SR from (value, is-seed).
template <class S>
struct VwSeed;
template <>
struct VwSeed<Dual3> {
  static OH_DEV DualR q(double v, double one) { return {v, one}; }
  static OH_DEV Dual3 qd(double v, double one) { return {v, 0.0, one, 0.0}; }
  static OH_DEV Dual3 qdd(double v, double one) { return {v, 0.0, 0.0, one}; }
};
template <int NB, class S, class SR = typename RotOf<S>::T>
OH_DEV void rnea_vw3(const oh_dynamics* __restrict__ dy, const double* qs, const int j, S (&tau)[NB - 1]) {
  constexpr int NJ = NB - 1;
  S om[3], omD[3], vD[3];
  SR wk[NJ][3], vk[NJ][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    om[k] = S{};
    omD[k] = S{};
    vD[k] = S{} + dy->vd0[k];
  }
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    tau[k] = S{};
#pragma unroll
    for (int c = 0; c < 3; ++c) wk[k][c] = vk[k][c] = SR{};
  }
#pragma unroll 1
  for (int i = 0; i < NJ; ++i) {
    const double one = (i == j) ? 1.0 : 0.0;
    vw_body<NB, true, S, SR>(dy, i, VwSeed<S>::q(qs[i], one), VwSeed<S>::qd(qs[8 + i], one), VwSeed<S>::qdd(qs[16 + i], one), om, omD, vD, wk, vk, tau);
  }
  vw_body<NB, false, S, SR>(dy, NJ, SR{}, S{}, S{}, om, omD, vD, wk, vk, tau);
}

// ---- d tau / d (q, dq, ddq) in closed form (round 4; numpy: oracle/torque.py:rnea_jacobian_spatial) ----------------------------------------------
// The dual-number recursions above cost a unit 7 lanes x (1 primal + 3 tangents) of the whole chain: ~120 k instructions per unit, 0.45 of the batch's
// device time.  In world coordinates (spatial vectors about the world origin, Featherstone 2008) the same derivative has a closed form.  With
//     S_l = (z_l, o_l x z_l)            joint screw,  v_b = sum_{l<=b} S_l dq_l,  a_b = a_0 + sum_{l<=b} (S_l ddq_l + v_l x S_l dq_l),
//     W_b = I_b a_b + v_b x* I_b v_b,   tau_k = S_k . sum_{b>=k} W_b          (what the reference's two passes compute, models.py:1819-1880)
// and  dS_l/dq_m = S_m x S_l (m < l),  dI_b/dq_m = S_m x* I_b - I_b S_m x (m <= b)  the product rule collapses (Jacobi identity) to
//     dW_b/dddq_j = I_b S_j,    dW_b/ddq_j = 2 (B_b S_j + I_b Sd_j),    dW_b/dq_j = S_j x* W_b + I_b Sdd_j + 2 B_b Sd_j          (b >= j)
//     Sd_j = v_j x S_j,  Sdd_j = a_j x S_j + v_j x Sd_j,  2 B_b x = I_b (x x v_b) + x x* I_b v_b + v_b x* I_b x = (Xi_b w_x, -2 p_b x w_x)
// (2 B_b sees only the angular part of x: Xi_b 3 x 3, p_b the linear momentum; Carpentier & Mansard 2018 and Singh, Russell & Wensing 2022 arrive at the
// same terms).  Summed over the subtree (composites I^C, Xi^C, p^C, F^C of body m = max(k, j)):
//     d tau_k / d(q_j, dq_j, ddq_j) = S_k . u(max(k, j)),   u_ddq = I^C S_j,  u_dq = 2 (B^C S_j + I^C Sd_j),  u_q = I^C Sdd_j + 2 B^C Sd_j (+ S_j x* F^C_j if k <= j).
// Lane j of a unit: the serial world-frame chain (cheap, every lane), the world inertia / Xi / wrench of body j (the fixed last body rides on lane N-1),
// exchange through LDS, then the inward composite sums and column j.  ~3 k instructions per lane.  Valid when the reference's recursion is the
// dynamics of a rigid-body chain: unit axes that the joint-origin rotation leaves in place (R0^T axis = axis: the angular velocity the reference adds,
// iRp @ axis, is then the axis Rot(axis, q) turns about; models.py:1821-1823).  oh_create_torque checks it; other tables take the dual-number path.
template <int N>
struct IdsWs {
  // A unit's LDS in k_tq_eval3, 277 doubles for N = 7 (nine units: 19.9 KB, so that two blocks share a SIMD's quarter of the CU's 160 KB):
  static constexpr int TW = 28;                  // pitch of the per-body slots: 28 = m, h (3), A (6), Xi (9), p (3), W (6) whatever the chain length; once phase 3 has consumed body
                                                 // m its slot takes row m of d tau / dz (3 N entries) and, behind it, lane m's row coefficients (RW)
  static constexpr int BD = 0;
  static constexpr int RW = 3 * N + 1;           // offset inside a slot: cf, cb, dw, bar, nrel, viol (6 doubles; the slot has 28 - 22 = 6 to spare)
  static constexpr int S = N * TW;               // joint screws, 6 each (phases 1-3); then, together with QS, the three rows of d p_link / dz (JP)
  static constexpr int QS = S + 6 * N;           // (q | dq | ddq) at 0, 8, 16: read by the loop counter in phase 1 and by the chain walk
  static constexpr int JP = S;                   // pitch 3 N + 1
  static constexpr int RW2 = QS + 24;            // cmpl[N], fsum[N]
  static constexpr int SIZE = (RW2 + 2 * N) | 1;  // odd: the units of a wavefront land in different banks
  static_assert(3 * (3 * N + 1) <= 6 * N + 24, "the rows of d p_link / dz take the place of the screws and of (q | dq | ddq)");
  static_assert(3 * N + 1 + 6 <= TW, "row m of d tau / dz and the six row coefficients behind it fit the slot of body m");
};
OH_DEV void mcross6(const double* x, const double* y, double* o) {  // motion x motion
  double t[3];
  cross3(x, y, o);
  cross3(x, y + 3, o + 3);
  cross3(x + 3, y, t);
  o[3] += t[0]; o[4] += t[1]; o[5] += t[2];
}
OH_DEV void fcross6(const double* x, const double* f, double* o) {  // motion x* force
  double t[3];
  cross3(x, f, o);
  cross3(x + 3, f + 3, t);
  o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
  cross3(x, f + 3, o + 3);
}
// (n, f) = I (w, v) for a rigid-body inertia about the world origin: n = A w + h x v, f = m v - h x w;  A = (xx xy xz yy yz zz)
OH_DEV void inert6(const double m, const double* h, const double* A, const double* x, double* o) {
  double t[3];
  cross3(h, x + 3, t);
  o[0] = A[0] * x[0] + A[1] * x[1] + A[2] * x[2] + t[0];
  o[1] = A[1] * x[0] + A[3] * x[1] + A[4] * x[2] + t[1];
  o[2] = A[2] * x[0] + A[4] * x[1] + A[5] * x[2] + t[2];
  cross3(h, x, t);
  o[3] = m * x[3] - t[0];
  o[4] = m * x[4] - t[1];
  o[5] = m * x[5] - t[2];
}
OH_DEV double dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }

// world inertia, momentum coupling and wrench of body b from its (R, o, v, a), added to acc[28]
OH_DEV void ids_body(const oh_dynamics* __restrict__ dy, const int b, const double* __restrict__ p1, double* acc) {
  const double* R = p1;
  const double* o = p1 + 9;
  const double* v = p1 + 12;
  const double* a = p1 + 18;
  double c[3], T[9], Ic[9];
  mv3(R, dy->com[b], c);
  c[0] += o[0]; c[1] += o[1]; c[2] += o[2];
  mm3(R, dy->inertia[b], T);
  mmT3(T, R, Ic);
  const double m = dy->mass[b];
  const double h[3] = {m * c[0], m * c[1], m * c[2]};
  const double c2 = dot3(c, c);
  double A[6];
  A[0] = Ic[0] + m * (c2 - c[0] * c[0]);
  A[1] = 0.5 * (Ic[1] + Ic[3]) - m * c[0] * c[1];
  A[2] = 0.5 * (Ic[2] + Ic[6]) - m * c[0] * c[2];
  A[3] = Ic[4] + m * (c2 - c[1] * c[1]);
  A[4] = 0.5 * (Ic[5] + Ic[7]) - m * c[1] * c[2];
  A[5] = Ic[8] + m * (c2 - c[2] * c[2]);
  double Pm[6], W[6], t6[6];
  inert6(m, h, A, v, Pm);
  inert6(m, h, A, a, W);
  fcross6(v, Pm, t6);
#pragma unroll
  for (int k = 0; k < 6; ++k) W[k] += t6[k];
  // Xi = [w]x A + ([w]x A)^T - (h vl^T + vl h^T - 2 (vl . h) 1) - [n_P]x
  const double* w = v;
  const double* vl = v + 3;
  const double Af[9] = {A[0], A[1], A[2], A[1], A[3], A[4], A[2], A[4], A[5]};
  double OA[9];  // columns w x A[:, k]
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double col[3] = {Af[k], Af[3 + k], Af[6 + k]};
    double x[3];
    cross3(w, col, x);
    OA[k] = x[0]; OA[3 + k] = x[1]; OA[6 + k] = x[2];
  }
  const double vh = dot3(vl, h);
  double Xi[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) Xi[3 * r + k] = OA[3 * r + k] + OA[3 * k + r] - h[r] * vl[k] - vl[r] * h[k] + (r == k ? 2.0 * vh : 0.0);
  Xi[1] += Pm[2]; Xi[2] -= Pm[1];
  Xi[3] -= Pm[2]; Xi[5] += Pm[0];
  Xi[6] += Pm[1]; Xi[7] -= Pm[0];
  acc[0] += m;
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[1 + k] += h[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[4 + k] += A[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[10 + k] += Xi[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[19 + k] += Pm[3 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[22 + k] += W[k];
}

// ws: the unit's LDS workspace (IdsWs<N>::SIZE doubles), qs: (q | dq | ddq) at 0, 8, 16; every lane of the unit calls (block of one wavefront).
// Lane j leaves column j, N + j, 2 N + j of d tau / d (q, dq, ddq) in rows 0 .. N-1 of the tile at ws[0] and returns tau_j.
template <int N>
OH_DEV double rnea_idsva(const oh_dynamics* __restrict__ dy, double* __restrict__ ws, const int j, const bool writer) {
  using L = IdsWs<N>;
  const double* qs = ws + L::QS;
  double Sj[6], Sdj[6], Sddj[6];
  double own[24];  // (R, o, v, a) of body j, picked up on the way (every lane walks the whole chain)
  {
    double Rw[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0}, ow[3] = {0.0, 0.0, 0.0};
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, a[6] = {0.0, 0.0, 0.0, dy->vd0[0], dy->vd0[1], dy->vd0[2]};
#pragma unroll
    for (int k = 0; k < 6; ++k) Sj[k] = Sdj[k] = Sddj[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 24; ++k) own[k] = 0.0;
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
      double o[3], Ri[9];
      mv3(Rw, dy->xyz[i], o);
      o[0] += ow[0]; o[1] += ow[1]; o[2] += ow[2];
      double S[6], Sd[6], Sdd[6], t6[6], Rp[9], sj, cj;
      mv3(Rw, dy->axis[i], S);
      cross3(o, S, S + 3);
      sincos_joint(qs[i], &sj, &cj);
      joint_rotation(dy->R0[i], dy->axis[i], sj, cj, Rp);
      mm3(Rw, Rp, Ri);
      mcross6(v, S, Sd);
      const double qd = qs[8 + i], qdd = qs[16 + i];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        v[k] = fma(S[k], qd, v[k]);
        a[k] = fma(Sd[k], qd, fma(S[k], qdd, a[k]));
      }
      mcross6(a, S, Sdd);
      mcross6(v, Sd, t6);
      const bool mine = i == j;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        Sdd[k] += t6[k];
        Sj[k] = mine ? S[k] : Sj[k];
        Sdj[k] = mine ? Sd[k] : Sdj[k];
        Sddj[k] = mine ? Sdd[k] : Sddj[k];
        own[12 + k] = mine ? v[k] : own[12 + k];
        own[18 + k] = mine ? a[k] : own[18 + k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) own[k] = mine ? Ri[k] : own[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) own[9 + k] = mine ? o[k] : own[9 + k];
      if (writer && mine) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ws[L::S + 6 * i + k] = S[k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) Rw[k] = Ri[k];
      ow[0] = o[0]; ow[1] = o[1]; ow[2] = o[2];
    }
  }
  {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    ids_body(dy, j, own, acc);
    {  // the fixed last body moves with body N - 1: its frame follows from that body's (lane N - 1 keeps the result)
      double last[24], t3[3];
      mm3(own, dy->R0[N], last);
      mv3(own, dy->xyz[N], t3);
      last[9] = own[9] + t3[0]; last[10] = own[10] + t3[1]; last[11] = own[11] + t3[2];
#pragma unroll
      for (int k = 12; k < 24; ++k) last[k] = own[k];
      double acc2[28];
#pragma unroll
      for (int k = 0; k < 28; ++k) acc2[k] = 0.0;
      ids_body(dy, N, last, acc2);
#pragma unroll
      for (int k = 0; k < 28; ++k) acc[k] += (j == N - 1) ? acc2[k] : 0.0;
    }
    if (writer) {
#pragma unroll
      for (int k = 0; k < 28; ++k) ws[L::BD + L::TW * j + k] = acc[k];
    }
  }
  __syncthreads();
  double C[28];
  double u0s[6], u1s[6], u2s[6];
  double tau_j = 0.0;
#pragma unroll
  for (int k = 0; k < 28; ++k) C[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) u0s[k] = u1s[k] = u2s[k] = 0.0;
#pragma unroll 1
  for (int m = N - 1; m >= 0; --m) {
    const double* bd = ws + L::BD + L::TW * m;
    double Sm[6];
#pragma unroll
    for (int k = 0; k < 28; ++k) C[k] += bd[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Sm[k] = ws[L::S + 6 * m + k];
    const double* hC = C + 1;
    const double* AC = C + 4;
    const double* XC = C + 10;
    const double* pC = C + 19;
    const double* FC = C + 22;
    double u0[6], u1[6], u2[6], t6[6], x[3];
    inert6(C[0], hC, AC, Sj, u2);
    inert6(C[0], hC, AC, Sdj, t6);
    mv3(XC, Sj, u1);
    cross3(pC, Sj, x);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u1[k] = fma(2.0, t6[k], u1[k]);
      u1[3 + k] = 2.0 * (t6[3 + k] - x[k]);
    }
    inert6(C[0], hC, AC, Sddj, u0);
    mv3(XC, Sdj, t6);
    cross3(pC, Sdj, x);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u0[k] += t6[k];
      u0[3 + k] -= 2.0 * x[k];
    }
    if (m == j) {
      fcross6(Sj, FC, t6);
      tau_j = dot6(Sm, FC);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        u0s[k] = u0[k] + t6[k];
        u1s[k] = u1[k];
        u2s[k] = u2[k];
      }
    }
    const bool below = m > j;  // row below the diagonal: the composites of body m; else what column j froze at its own body
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      u0[k] = below ? u0[k] : u0s[k];
      u1[k] = below ? u1[k] : u1s[k];
      u2[k] = below ? u2[k] : u2s[k];
    }
    const double e0 = dot6(Sm, u0), e1 = dot6(Sm, u1), e2 = dot6(Sm, u2);
    __syncthreads();  // every lane of the unit has taken body m out of its slot: the slot becomes row m of d tau / dz
    if (writer) {
      ws[m * L::TW + j] = e0;
      ws[m * L::TW + N + j] = e1;
      ws[m * L::TW + 2 * N + j] = e2;
    }
  }
  return tau_j;
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


// ---- gradient of c^T tau (round 4; numpy: oracle/torque.py:rnea_ctau_gradient) ------------------------------------------------------------------
// Virtual work: c^T rnea(q, qd, qdd) = sum_b f_b . v_b(c) + n_b . w_b(c), the inertial wrench of body b (models.py:1819-1856, the outward pass of the
// reference) paired with the twist the joint rates c would give it.  Both come out of one outward recursion, so the gradient with respect to
// (q, qd, qdd) is one inward adjoint recursion: body i hands the adjoints of its (om, omD, vD) and of the virtual (wc, vo) to its parent.  A joint
// angle enters only through R_i^T = Rot(axis, q_i)^T R0^T, and d(R_i^T v)/dq_i = -axis x (R_i^T v), which is what `sw` collects.
// Scalars: S for what depends on (q, qd, qdd), SR for what depends on the joint angles alone (rotations, axes, the virtual twists); qdd and c carry
// no tangent (nothing is differentiated twice with respect to them: tau is linear in qdd, c is a multiplier).
template <class A, class B>
OH_DEV typename Prom<A, B>::T dotT(const A* a, const B* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

template <int NB, class S, class SR = typename RotOf<S>::T>
OH_DEV void rnea_ctau_grad(const oh_dynamics* __restrict__ dy, const SR (&q)[NB - 1], const S (&qd)[NB - 1], const double (&qdd)[NB - 1], const double (&c)[NB - 1],
                           S (&gq)[NB - 1], S (&gqd)[NB - 1], S (&gqdd)[NB - 1]) {
  // outward pass: the state every body leaves to its child (lane-private memory, the loops stay rolled as in rnea_lit)
  S om_[NB][3], omD_[NB][3], vD_[NB][3];
  SR wc_[NB][3], vo_[NB][3];
  SR sj[NB], cj[NB];
  {
    S om[3], omD[3], vD[3];
    SR wc[3], vo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      om[k] = S{};
      omD[k] = S{};
      vD[k] = S{} + dy->vd0[k];
      wc[k] = SR{};
      vo[k] = SR{};
    }
#pragma unroll 1
    for (int i = 0; i < NB; ++i) {
      const bool moving = i < NB - 1;
      S t1[3], t2[3], t3[3], acc[3];
      SR w[3], tw[3];
      crossT(omD, dy->xyz[i], t1);
      crossT(om, dy->xyz[i], t2);
      crossT(om, t2, t3);
      crossT(wc, dy->xyz[i], tw);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        acc[k] = vD[k] + t1[k] + t3[k];
        w[k] = vo[k] + tw[k];
      }
      SR Rp[9];
      if (moving) {
        sincosT(q[i], &sj[i], &cj[i]);
        joint_rotation(dy->R0[i], dy->axis[i], sj[i], cj[i], Rp);
      } else {
        sj[i] = SR{};
        cj[i] = SR{};
#pragma unroll
        for (int k = 0; k < 9; ++k) Rp[k] = SR{} + dy->R0[i][k];
      }
      S omp[3], omDp[3];
      SR wcp[3];
      mTvT(Rp, om, omp);
      mTvT(Rp, omD, omDp);
      mTvT(Rp, wc, wcp);
      if (moving) {
        SR a[3];
        mTvT(Rp, dy->axis[i], a);
        const S qdi = qd[i];
        S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
        S cr[3];
        crossT(omp, aq, cr);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          om[k] = omp[k] + aq[k];
          omD[k] = omDp[k] + cr[k] + a[k] * qdd[i];
          wc[k] = wcp[k] + a[k] * c[i];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          om[k] = omp[k];
          omD[k] = omDp[k];
          wc[k] = wcp[k];
        }
      }
      S vDi[3];
      SR voi[3];
      mTvT(Rp, acc, vDi);
      mTvT(Rp, w, voi);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vD[k] = vDi[k];
        vo[k] = voi[k];
        om_[i][k] = om[k];
        omD_[i][k] = omD[k];
        vD_[i][k] = vD[k];
        wc_[i][k] = wc[k];
        vo_[i][k] = vo[k];
      }
    }
  }
  // inward pass
  S b_om[3], b_omD[3], b_vD[3], b_wc[3], b_vo[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b_om[k] = b_omD[k] = b_vD[k] = b_wc[k] = b_vo[k] = S{};
#pragma unroll 1
  for (int i = NB - 1; i >= 0; --i) {
    const bool moving = i < NB - 1;
    // what the parent left (the base: at rest, accelerating against gravity)
    S om_p[3], omD_p[3], vD_p[3];
    SR wc_p[3], vo_p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (i > 0) {
        om_p[k] = om_[i - 1][k];
        omD_p[k] = omD_[i - 1][k];
        vD_p[k] = vD_[i - 1][k];
        wc_p[k] = wc_[i - 1][k];
        vo_p[k] = vo_[i - 1][k];
      } else {
        om_p[k] = S{};
        omD_p[k] = S{};
        vD_p[k] = S{} + dy->vd0[k];
        wc_p[k] = SR{};
        vo_p[k] = SR{};
      }
    }
    SR Rp[9];
    if (moving) {
      joint_rotation(dy->R0[i], dy->axis[i], sj[i], cj[i], Rp);
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rp[k] = SR{} + dy->R0[i][k];
    }
    const double* cm = dy->com[i];
    const double* r = dy->xyz[i];
    const double m = dy->mass[i];
    S omi[3], omDi[3], vDi[3];
    SR wci[3], voi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      omi[k] = om_[i][k];
      omDi[k] = omD_[i][k];
      vDi[k] = vD_[i][k];
      wci[k] = wc_[i][k];
      voi[k] = vo_[i][k];
    }
    // local term f_i . vc_i + n_i . wc_i
    {
      S t1[3], t2[3], t3[3], fi[3], Io[3], IoD[3], ni[3];
      crossT(omDi, cm, t1);
      crossT(omi, cm, t2);
      crossT(omi, t2, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) fi[k] = m * (vDi[k] + t1[k] + t3[k]);
      mvT(dy->inertia[i], omi, Io);
      mvT(dy->inertia[i], omDi, IoD);
      crossT(omi, Io, t1);
#pragma unroll
      for (int k = 0; k < 3; ++k) ni[k] = IoD[k] + t1[k];
      SR tw[3], vci[3];
      crossT(wci, cm, tw);
#pragma unroll
      for (int k = 0; k < 3; ++k) vci[k] = voi[k] + tw[k];
      S cf[3];
      crossT(cm, fi, cf);
      SR cv[3], Itw[3];
      crossT(cm, vci, cv);
      mTvT(dy->inertia[i], wci, Itw);
      const S oc = dotT(omi, cm), ov = dotT(omi, vci);
      const SR cvv = dotT(cm, vci);
      S wxo[3], Itwo[3], Ixw[3];
      crossT(wci, omi, wxo);
      mTvT(dy->inertia[i], wxo, Itwo);
      crossT(Io, wci, Ixw);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b_vo[k] = b_vo[k] + fi[k];
        b_wc[k] = b_wc[k] + cf[k] + ni[k];
        b_vD[k] = b_vD[k] + m * vci[k];
        b_omD[k] = b_omD[k] + m * cv[k] + Itw[k];
        b_om[k] = b_om[k] + m * (vci[k] * oc + cm[k] * ov - 2.0 * (omi[k] * cvv)) + Itwo[k] + Ixw[k];
      }
    }
    // through the step of body i
    S b_omp[3];
    if (moving) {
      SR a[3];
      S omp[3], omDp[3];
      SR wcp[3];
      mTvT(Rp, dy->axis[i], a);
      mTvT(Rp, om_p, omp);
      mTvT(Rp, omD_p, omDp);
      mTvT(Rp, wc_p, wcp);
      const S qdi = qd[i];
      S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
      S x1[3], x2[3], b_aq[3], b_a[3];
      crossT(aq, b_omD, x1);
      crossT(b_omD, omp, x2);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b_omp[k] = b_om[k] + x1[k];
        b_aq[k] = b_om[k] + x2[k];
        b_a[k] = b_aq[k] * qdi + b_omD[k] * qdd[i] + b_wc[k] * c[i];
      }
      gqd[i] = dotT(b_aq, a);
      gqdd[i] = dotT(b_omD, a);
      S s1[3], s2[3], s3[3], s4[3], s5[3], s6[3];
      crossT(omp, b_omp, s1);
      crossT(omDp, b_omD, s2);
      crossT(wcp, b_wc, s3);
      crossT(a, b_a, s4);
      crossT(vDi, b_vD, s5);
      crossT(voi, b_vo, s6);
      S sw[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) sw[k] = s1[k] + s2[k] + s3[k] + s4[k] + s5[k] + s6[k];
      gq[i] = -dotT(sw, dy->axis[i]);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) b_omp[k] = b_om[k];
    }
    if (i > 0) {
      S b_acc[3], b_w[3], Ro[3], RoD[3], Rw[3], x1[3], x2[3];
      mvT(Rp, b_vD, b_acc);
      mvT(Rp, b_vo, b_w);
      mvT(Rp, b_omp, Ro);
      mvT(Rp, b_omD, RoD);
      mvT(Rp, b_wc, Rw);
      crossT(r, b_acc, x1);
      crossT(r, b_w, x2);
      const S opr = dotT(om_p, r), opb = dotT(om_p, b_acc), rb = dotT(r, b_acc);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b_om[k] = Ro[k] + b_acc[k] * opr + r[k] * opb - 2.0 * (om_p[k] * rb);
        b_omD[k] = RoD[k] + x1[k];
        b_vD[k] = b_acc[k];
        b_wc[k] = Rw[k] + x2[k];
        b_vo[k] = b_w[k];
      }
    }
  }
}

// The same adjoint without per-body arrays (round 4).  rnea_ctau_grad keeps what every body left to its child -- 8 x (om, omD, vD, wc, vo, sin, cos) of
// dual numbers, 2.7 KB per lane, plus the lane-indexed inputs and outputs: 3.7 KB of scratch per lane at one wavefront per SIMD, which is what
// k_tq_curv spent its time on.  The outward recursion is invertible: from the state of body i and its joint, the state of the parent follows
// (om_p = Rp (om_i - a dq_i), ...), so the inward pass rebuilds each parent on the way and stores nothing (~300 more instructions per body, no memory).
// Inputs come from LDS by the loop counter: zs = the unit's (q | dq | ddq | c) at 0, 8, 16, 24; lane j seeds joint j.  sink(i, gq_i, gqd_i, gqdd_i).
template <class S>
struct CtSeed;
template <>
struct CtSeed<Dual2> {
  static OH_DEV DualR q(double v, double one) { return {v, one}; }
  static OH_DEV Dual2 qd(double v, double one) { return {v, 0.0, one}; }
};
template <>
struct CtSeed<double> {
  static OH_DEV double q(double v, double) { return v; }
  static OH_DEV double qd(double v, double) { return v; }
};
template <int NB, class S, class SR, class Sink>
OH_DEV void rnea_ctau_grad_inv(const oh_dynamics* __restrict__ dy, const double* zs, const int j, Sink&& sink) {
  S om[3], omD[3], vD[3];
  SR wc[3], vo[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    om[k] = S{};
    omD[k] = S{};
    vD[k] = S{} + dy->vd0[k];
    wc[k] = SR{};
    vo[k] = SR{};
  }
  // outward: only the state of the last body survives
#pragma unroll 1
  for (int i = 0; i < NB; ++i) {
    const bool moving = i < NB - 1;
    S t1[3], t2[3], t3[3], acc[3];
    SR w[3], tw[3];
    crossT(omD, dy->xyz[i], t1);
    crossT(om, dy->xyz[i], t2);
    crossT(om, t2, t3);
    crossT(wc, dy->xyz[i], tw);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      acc[k] = vD[k] + t1[k] + t3[k];
      w[k] = vo[k] + tw[k];
    }
    SR Rp[9];
    if (moving) {
      SR sj, cj;
      sincosT(CtSeed<S>::q(zs[i], i == j ? 1.0 : 0.0), &sj, &cj);
      joint_rotation(dy->R0[i], dy->axis[i], sj, cj, Rp);
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rp[k] = SR{} + dy->R0[i][k];
    }
    S omp[3], omDp[3];
    SR wcp[3];
    mTvT(Rp, om, omp);
    mTvT(Rp, omD, omDp);
    mTvT(Rp, wc, wcp);
    if (moving) {
      SR a[3];
      mTvT(Rp, dy->axis[i], a);
      const S qdi = CtSeed<S>::qd(zs[8 + i], i == j ? 1.0 : 0.0);
      const double qddi = zs[16 + i], ci = zs[24 + i];
      S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
      S cr[3];
      crossT(omp, aq, cr);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        om[k] = omp[k] + aq[k];
        omD[k] = omDp[k] + cr[k] + a[k] * qddi;
        wc[k] = wcp[k] + a[k] * ci;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        om[k] = omp[k];
        omD[k] = omDp[k];
        wc[k] = wcp[k];
      }
    }
    S vDi[3];
    SR voi[3];
    mTvT(Rp, acc, vDi);
    mTvT(Rp, w, voi);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vD[k] = vDi[k];
      vo[k] = voi[k];
    }
  }
  // inward: (om, omD, vD, wc, vo) is the state of body i; its parent is rebuilt from it
  S b_om[3], b_omD[3], b_vD[3], b_wc[3], b_vo[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b_om[k] = b_omD[k] = b_vD[k] = b_wc[k] = b_vo[k] = S{};
#pragma unroll 1
  for (int i = NB - 1; i >= 0; --i) {
    const bool moving = i < NB - 1;
    SR Rp[9], a[3];
    S qdi = S{};
    double qddi = 0.0, ci = 0.0;
    if (moving) {
      SR sj, cj;
      sincosT(CtSeed<S>::q(zs[i], i == j ? 1.0 : 0.0), &sj, &cj);
      joint_rotation(dy->R0[i], dy->axis[i], sj, cj, Rp);
      mTvT(Rp, dy->axis[i], a);
      qdi = CtSeed<S>::qd(zs[8 + i], i == j ? 1.0 : 0.0);
      qddi = zs[16 + i];
      ci = zs[24 + i];
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rp[k] = SR{} + dy->R0[i][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) a[k] = SR{};
    }
    const double* cm = dy->com[i];
    const double* r = dy->xyz[i];
    const double m = dy->mass[i];
    // the parent's state (the base: at rest, accelerating against gravity), and what it looked like in the frame of body i
    S om_p[3], omD_p[3], vD_p[3], omp[3], omDp[3];
    SR wc_p[3], vo_p[3], wcp[3];
    {
      S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
#pragma unroll
      for (int k = 0; k < 3; ++k) omp[k] = om[k] - aq[k];
      S cr[3];
      crossT(omp, aq, cr);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        omDp[k] = omD[k] - cr[k] - a[k] * qddi;
        wcp[k] = wc[k] - a[k] * ci;
      }
      if (i > 0) {
        S accp[3], t1[3], t2[3], t3[3];
        SR wp[3], tw[3];
        mvT(Rp, omp, om_p);
        mvT(Rp, omDp, omD_p);
        mvT(Rp, wcp, wc_p);
        mvT(Rp, vD, accp);
        mvT(Rp, vo, wp);
        crossT(omD_p, r, t1);
        crossT(om_p, r, t2);
        crossT(om_p, t2, t3);
        crossT(wc_p, r, tw);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          vD_p[k] = accp[k] - t1[k] - t3[k];
          vo_p[k] = wp[k] - tw[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          om_p[k] = S{};
          omD_p[k] = S{};
          vD_p[k] = S{} + dy->vd0[k];
          wc_p[k] = SR{};
          vo_p[k] = SR{};
          omp[k] = S{};    // exactly, not up to the rounding of the inversion
          omDp[k] = S{};
          wcp[k] = SR{};
        }
      }
    }
    // local term f_i . vc_i + n_i . wc_i
    {
      S t1[3], t2[3], t3[3], fi[3], Io[3], IoD[3], ni[3];
      crossT(omD, cm, t1);
      crossT(om, cm, t2);
      crossT(om, t2, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) fi[k] = m * (vD[k] + t1[k] + t3[k]);
      mvT(dy->inertia[i], om, Io);
      mvT(dy->inertia[i], omD, IoD);
      crossT(om, Io, t1);
#pragma unroll
      for (int k = 0; k < 3; ++k) ni[k] = IoD[k] + t1[k];
      SR tw[3], vci[3];
      crossT(wc, cm, tw);
#pragma unroll
      for (int k = 0; k < 3; ++k) vci[k] = vo[k] + tw[k];
      S cf[3];
      crossT(cm, fi, cf);
      SR cv[3], Itw[3];
      crossT(cm, vci, cv);
      mTvT(dy->inertia[i], wc, Itw);
      const S oc = dotT(om, cm), ov = dotT(om, vci);
      const SR cvv = dotT(cm, vci);
      S wxo[3], Itwo[3], Ixw[3];
      crossT(wc, om, wxo);
      mTvT(dy->inertia[i], wxo, Itwo);
      crossT(Io, wc, Ixw);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b_vo[k] = b_vo[k] + fi[k];
        b_wc[k] = b_wc[k] + cf[k] + ni[k];
        b_vD[k] = b_vD[k] + m * vci[k];
        b_omD[k] = b_omD[k] + m * cv[k] + Itw[k];
        b_om[k] = b_om[k] + m * (vci[k] * oc + cm[k] * ov - 2.0 * (om[k] * cvv)) + Itwo[k] + Ixw[k];
      }
    }
    // through the step of body i
    S b_omp[3];
    if (moving) {
      S aq[3] = {a[0] * qdi, a[1] * qdi, a[2] * qdi};
      S x1[3], x2[3], b_aq[3], b_a[3];
      crossT(aq, b_omD, x1);
      crossT(b_omD, omp, x2);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b_omp[k] = b_om[k] + x1[k];
        b_aq[k] = b_om[k] + x2[k];
        b_a[k] = b_aq[k] * qdi + b_omD[k] * qddi + b_wc[k] * ci;
      }
      const S gqd_i = dotT(b_aq, a);
      const S gqdd_i = dotT(b_omD, a);
      S s1[3], s2[3], s3[3], s4[3], s5[3], s6[3];
      crossT(omp, b_omp, s1);
      crossT(omDp, b_omD, s2);
      crossT(wcp, b_wc, s3);
      crossT(a, b_a, s4);
      crossT(vD, b_vD, s5);
      crossT(vo, b_vo, s6);
      S sw[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) sw[k] = s1[k] + s2[k] + s3[k] + s4[k] + s5[k] + s6[k];
      sink(i, -dotT(sw, dy->axis[i]), gqd_i, gqdd_i);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) b_omp[k] = b_om[k];
    }
    if (i > 0) {
      S b_acc[3], b_w[3], Ro[3], RoD[3], Rw[3], x1[3], x2[3];
      mvT(Rp, b_vD, b_acc);
      mvT(Rp, b_vo, b_w);
      mvT(Rp, b_omp, Ro);
      mvT(Rp, b_omD, RoD);
      mvT(Rp, b_wc, Rw);
      crossT(r, b_acc, x1);
      crossT(r, b_w, x2);
      const S opr = dotT(om_p, r), opb = dotT(om_p, b_acc), rb = dotT(r, b_acc);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b_om[k] = Ro[k] + b_acc[k] * opr + r[k] * opb - 2.0 * (om_p[k] * rb);
        b_omD[k] = RoD[k] + x1[k];
        b_vD[k] = b_acc[k];
        b_wc[k] = Rw[k] + x2[k];
        b_vo[k] = b_w[k];
        om[k] = om_p[k];
        omD[k] = omD_p[k];
        vD[k] = vD_p[k];
        wc[k] = wc_p[k];
        vo[k] = vo_p[k];
      }
    }
  }
}

// sum_i c_i d^2 tau_i / d (q, qd, qdd)^2 (what the reference obtains as ddh by AD of the CasADi graph, optimization.py:8-24): one lane per
// (sample, joint); q, qd, qdd, c [n][N] -> H [n][3 N][3 N] row-major.  Lane j writes rows j and N + j and, by symmetry, column j of the ddq rows.
template <int N>
__global__ __launch_bounds__(64) void k_rnea_hess(const oh_dynamics* __restrict__ dy, const int n, const double* __restrict__ q, const double* __restrict__ qd,
                                                  const double* __restrict__ qdd, const double* __restrict__ c, double* __restrict__ H) {
  constexpr int NZ = 3 * N, UPW = 64 / N;
  __shared__ double zs_l[UPW][32];
  const int lane = threadIdx.x;
  int ul = lane / N, j = lane - ul * N;
  const bool lane_ok = ul < UPW;
  if (!lane_ok) {
    ul = UPW - 1;
    j = N - 1;
  }
  long long u = (long long)blockIdx.x * UPW + ul;
  const bool active = lane_ok && u < n;
  if (u >= n) u = n - 1;
  if (lane_ok) {
    zs_l[ul][j] = q[u * N + j];
    zs_l[ul][8 + j] = qd[u * N + j];
    zs_l[ul][16 + j] = qdd[u * N + j];
    zs_l[ul][24 + j] = c[u * N + j];
  }
  __syncthreads();
  double* Hu = H + (size_t)u * NZ * NZ;
  rnea_ctau_grad_inv<N + 1, Dual2, DualR>(dy, zs_l[ul], j, [&](const int k, const Dual2 gq, const Dual2 gqd, const Dual2 gqdd) {
    if (!active) return;
    Hu[j * NZ + k] = gq.d0;
    Hu[j * NZ + N + k] = gqd.d0;
    Hu[j * NZ + 2 * N + k] = gqdd.d0;
    Hu[(N + j) * NZ + k] = gq.d1;
    Hu[(N + j) * NZ + N + k] = gqd.d1;
    Hu[(N + j) * NZ + 2 * N + k] = 0.0;
    Hu[(2 * N + k) * NZ + j] = gqdd.d0;
    Hu[(2 * N + k) * NZ + N + j] = 0.0;
    Hu[(2 * N + k) * NZ + 2 * N + j] = 0.0;
  });
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
  }
  D.f_cur[b] = 0.0;
  D.f_true[b] = 0.0;
  D.bsum[b] = 0.0;
  D.mu[b] = P.mu0;
  D.nun[b] = 4.0;
  D.mub[b] = P.mu_b0;
  D.stat[b] = 0.0;
  D.alpha[b] = 1.0;
  D.qk[b] = 0.0;
  D.ndx[b] = 0.0;
  D.viol[b] = 0.0;
  D.cur[b] = 1;  // the seed sits in slot 0 = the first "trial"
  D.first[b] = 1;
  D.curv[b] = 0;
  D.curv_age[b] = -1;
  D.status[b] = -1;
  D.iters[b] = 0;
  D.rejected[b] = 0;
  D.n_barrier[b] = 0;
  D.nrel[b] = 0;
  D.n_back[b] = 0;
  D.stall[b] = 0;
  D.list[b] = b;
}

__global__ __launch_bounds__(256) void k_tq_list(TqBuffers D) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.B) return;
  if (D.status[b] < 0) D.list[atomicAdd(D.n_list, 1)] = b;
}

// ---- evaluation -----------------------------------------------------------------------------------------------------------------------
// One inequality row s >= 0 under the relaxed log barrier (numpy: oracle/torque_ipm.py:evalp).  In: the slack at the trial point, slack and
// multiplier at the accepted point (none at the seed).  Out: the new multiplier (linearised complementarity  s dlam + lam ds = mu_b - lam s  with
// the slack change the step really produced, kept above 0.5 % of the old one), bco = -psi'(s) / mu_b, sig = the row's weight in the stage block,
// and the row's barrier value per unit mu_b.  Below delta = theta mu_b the logarithm is continued by its second-order Taylor polynomial.
struct TqRow {
  double lam, bco, sig, bar;
  bool relaxed;
};
OH_DEV TqRow tq_row(const double s, const double s_old, const double lam_old, const bool have_old, const double mub, const double delta, const double lam_floor) {
  TqRow r;
  r.relaxed = s < delta;
  const double sc = r.relaxed ? delta : s;
  if (r.relaxed) {
    r.bco = (2.0 * delta - s) / (delta * delta);
    r.lam = mub * r.bco;
    r.sig = mub / (delta * delta);
    const double e = (s - delta) / delta;
    r.bar = -log(delta) - e + 0.5 * e * e;
  } else {
    double lam = mub / sc;
    if (have_old) {
      lam = (mub - lam_old * (s - s_old)) / fmax(s_old, delta);
      lam = fmax(lam, lam_floor * lam_old);
      lam = fmin(fmax(lam, mub / (1e10 * sc)), 1e10 * mub / sc);
    }
    r.lam = lam;
    r.bco = 1.0 / sc;
    r.sig = lam / sc;
    r.bar = -log(sc);
  }
  return r;
}

// One lane per (instance, knot, JOINT): 9 units x 7 joints per wavefront.  Lane j runs the reference's Newton-Euler recursion once on (DualR, Dual3)
// scalars seeded with q_j, dq_j and ddq_j (d tau / d z is the tangent output: the derivative of the literal recursion by construction), so it ends up
// with columns j, N + j and 2 N + j of d tau / d z; the chain walk for p_link adds column j of its Jacobian.  The columns meet in LDS and every lane
// writes ITS THREE columns of the packed stage block  J^T diag(2 w_tau + Sigma) J + 2 w_p Jp^T Jp + ...,  of the two gradients (cost, barrier per unit
// mu_b) and of d tau / d z itself.  Bound: f64 FMA issue (268 registers, one wavefront per SIMD).
template <int N, bool VEL = false, bool IDS = true>
__global__ __launch_bounds__(64, IDS ? 2 : 1) void k_tq_eval3(TqParams P, TqBuffers D) {
  constexpr int NZ = 3 * N;
  constexpr int UPW = 64 / N;  // units per wavefront (9)
  // A unit's LDS.  Closed-form path: IdsWs<N> (277 doubles: two blocks per SIMD).  Dual-number path: tile [N + 3][NZ + 1], row coefficients, (q|dq|ddq).
  using L = IdsWs<N>;
  constexpr int TW = IDS ? L::TW : NZ + 1;                  // pitch of rows 0 .. N-1 of the tile (d tau / dz)
  constexpr int JP = IDS ? L::JP : N * (NZ + 1);            // rows of d p_link / dz, pitch NZ + 1
  constexpr int QS = IDS ? L::QS : (N + 3) * (NZ + 1) + 8 * N;
  constexpr int WS = IDS ? L::SIZE : ((QS + 24) | 1);
  __shared__ double lds_raw[UPW * WS];
  // row coefficients of joint m: which = 0 cf, 1 cb, 2 dw, 3 bar, 4 nrel, 5 viol, 6 cmpl, 7 fsum
  auto rw = [](double* tl, const int which, const int m) -> double& {
    if constexpr (IDS) return which < 6 ? tl[m * L::TW + L::RW + which] : tl[L::RW2 + (which - 6) * N + m];
    else return tl[(N + 3) * (NZ + 1) + which * N + m];
  };
  const int T = P.T;
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane == 0) *D.n_running = 0;  // k_tq_step, next in the stream, counts the instances that go on
  int ul = lane / N, j = lane - ul * N;
  const bool lane_ok = ul < UPW;
  if (!lane_ok) {  // lane 63 has no unit: it rides along on the last unit and keeps its hands off LDS
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
  const int cur = D.cur[b], ts = 1 - cur;
  const double* xr = D.xs + xs_off(D, T, ts, b, t);
  const bool have_old = D.first[b] == 0;
  const double mub = D.mub[b], delta = P.theta * mub;

  // 1. torques and their derivative: lane j leaves its three columns in the unit's tile and keeps tau_j
  double* tl = lds_raw + ul * WS;
  double* qs_u = tl + QS;
  const double qj = xr[j], dqj = xr[8 + j];
  if (lane_ok) {  // the recursions read (q | dq | ddq) of their unit by a loop counter: LDS, not a lane-private array
    qs_u[j] = qj;
    qs_u[8 + j] = dqj;
    qs_u[16 + j] = xr[16 + j];
  }
  __syncthreads();
  double qv[N];
#pragma unroll
  for (int k = 0; k < N; ++k) qv[k] = qs_u[k];
  double tvj;
  if constexpr (IDS) {
    tvj = rnea_idsva<N>(D.dyn, tl, j, lane_ok);
  } else {
    Dual3 tau[N];
    rnea_vw3<N + 1, Dual3>(D.dyn, qs_u, j, tau);
    tvj = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      tvj = (i == j) ? tau[i].v : tvj;
      if (lane_ok) {
        tl[i * TW + j] = tau[i].d0;
        tl[i * TW + N + j] = tau[i].d1;
        tl[i * TW + 2 * N + j] = tau[i].d2;
      }
    }
  }

  // 2. inequality rows under the barrier: lane j takes the rows of joint j (a row costs two divisions and a logarithm: ~350 instructions -- with every
  // lane computing all 14 of them, as until round 4, they were a third of the kernel); the unit shares cf, cb, dw and the sums through LDS
  const double* lm_old = D.lam + (((size_t)cur * D.B + b) * T + t) * TQ_LAM;
  double* lm_new = D.lam + (((size_t)ts * D.B + b) * T + t) * TQ_LAM;
  const double* sr_old = D.st + st_off(D, T, cur, b, t);
  const double* xr_old = D.xs + xs_off(D, T, cur, b, t);
  double cbv = 0.0, dvj = 0.0;
  {
    double tlo = 0.0, tup = 0.0, vlo = 0.0, vup = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {  // (P lives in scalar registers: no indexing by the lane)
      tlo = (i == j) ? P.tau_lo[i] : tlo;
      tup = (i == j) ? P.tau_up[i] : tup;
      if constexpr (VEL) {
        vlo = (i == j) ? P.dq_lo[i] : vlo;
        vup = (i == j) ? P.dq_up[i] : vup;
      }
    }
    const double tv_old = have_old ? sr_old[256 + j] : 0.0;
    const TqRow lo = tq_row(tvj - tlo, tv_old - tlo, have_old ? lm_old[j] : 0.0, have_old, mub, delta, 1.0 - P.tau_ftb);
    const TqRow up = tq_row(tup - tvj, tup - tv_old, have_old ? lm_old[N + j] : 0.0, have_old, mub, delta, 1.0 - P.tau_ftb);
    double bar_j = lo.bar + up.bar;
    double nrel_j = (lo.relaxed ? 1.0 : 0.0) + (up.relaxed ? 1.0 : 0.0);
    double viol_j = fmax(tlo - tvj, tvj - tup);
    double cmpl_j = fmax(lo.lam * (tvj - tlo), up.lam * (tup - tvj));
    if (active) {
      lm_new[j] = lo.lam;
      lm_new[N + j] = up.lam;
    }
    // joint-velocity rows on the velocity states (enforce_model_limits(name, time_deriv=1), builder.py:471-509): stage-local, linear in the state
    if constexpr (VEL) {
      const double dv_old = have_old ? xr_old[8 + j] : 0.0;
      const TqRow l2 = tq_row(dqj - vlo, dv_old - vlo, have_old ? lm_old[16 + j] : 0.0, have_old, mub, delta, 1.0 - P.tau_ftb);
      const TqRow u2 = tq_row(vup - dqj, vup - dv_old, have_old ? lm_old[16 + N + j] : 0.0, have_old, mub, delta, 1.0 - P.tau_ftb);
      bar_j += l2.bar + u2.bar;
      nrel_j += (l2.relaxed ? 1.0 : 0.0) + (u2.relaxed ? 1.0 : 0.0);
      viol_j = fmax(viol_j, fmax(vlo - dqj, dqj - vup));
      cmpl_j = fmax(cmpl_j, fmax(l2.lam * (dqj - vlo), u2.lam * (vup - dqj)));
      cbv = u2.bco - l2.bco;
      dvj = l2.sig + u2.sig;
      if (active) {
        lm_new[16 + j] = l2.lam;
        lm_new[16 + N + j] = u2.lam;
      }
    }
    __syncthreads();  // (the last row of d tau / dz is in its slot: the slots' spare words are free)
    if (lane_ok) {
      rw(tl, 0, j) = 2.0 * P.w_tau * tvj;              // cf
      rw(tl, 1, j) = up.bco - lo.bco;                  // cb
      rw(tl, 2, j) = 2.0 * P.w_tau + lo.sig + up.sig;  // dw
      rw(tl, 3, j) = bar_j;
      rw(tl, 4, j) = nrel_j;
      rw(tl, 5, j) = viol_j;
      rw(tl, 6, j) = cmpl_j;
      rw(tl, 7, j) = fma(P.w_tau * tvj, tvj, P.w_vel * dqj * dqj);
    }
  }

  // 3. link position and column j of its Jacobian (models.py:826-868, 1211-1264); its rows take the place of the screws and of (q | dq | ddq)
  double jp[3], r[3];
  {
    double R[9], pp[3], z[N][3], pj[N][3];
    fk_chain<N>(D.chain, qv, R, pp, z, pj);
    double e[3], tv3[3];
    mv3(R, D.chain->p_tool, tv3);
    const double* gl = D.goal + ((size_t)b * T + t) * 4;
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
    if (jt_d == 0) {
      const double dd[3] = {e[0] - pd[0], e[1] - pd[1], e[2] - pd[2]};
      cross3(zd, dd, jp);
    } else {
      jp[0] = zd[0]; jp[1] = zd[1]; jp[2] = zd[2];
    }
  }
  if (lane_ok) {  // the link position has no dq / ddq columns
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      tl[JP + k * (NZ + 1) + j] = jp[k];
      tl[JP + k * (NZ + 1) + N + j] = 0.0;
      tl[JP + k * (NZ + 1) + 2 * N + j] = 0.0;
    }
  }
  __syncthreads();

  // 4. components j, N + j, 2 N + j of the gradient of the cost and of the barrier (per unit mu_b), the lane's three columns of the stage block
  double dw[N], J0[N], J1[N], J2[N];
  double g0 = 0.0, g1 = 0.0, g2 = 0.0, h0 = 0.0, h1 = 0.0, h2 = 0.0;
  double bar = 0.0, viol = 0.0, cmpl = 0.0, fsum = 0.0;
  int nrel = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double cf = rw(tl, 0, i), cb = rw(tl, 1, i);
    dw[i] = rw(tl, 2, i);
    bar += rw(tl, 3, i);
    nrel += (int)rw(tl, 4, i);
    viol = fmax(viol, rw(tl, 5, i));
    cmpl = fmax(cmpl, rw(tl, 6, i));
    fsum += rw(tl, 7, i);
    J0[i] = tl[i * TW + j];
    J1[i] = tl[i * TW + N + j];
    J2[i] = tl[i * TW + 2 * N + j];
    g0 = fma(cf, J0[i], g0);
    g1 = fma(cf, J1[i], g1);
    g2 = fma(cf, J2[i], g2);
    h0 = fma(cb, J0[i], h0);
    h1 = fma(cb, J1[i], h1);
    h2 = fma(cb, J2[i], h2);
  }
  g0 += 2.0 * P.w_path * dot3(jp, r);
  g1 += 2.0 * P.w_vel * dqj;
  h1 += cbv;
  double* sr = D.st + st_off(D, T, ts, b, t);
  // exact curvature, stored term (D.curv = 2: k_tq_curv computed it at an earlier evaluation of this instance and skips this one): added as the block is written
  const double* hc = D.hc + ((size_t)b * T + t) * TQ_HC;
  const bool add_hc = D.curv[b] == 2;
  if (active) {
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
      const int d = c3 * N + j;
      double col[N];
#pragma unroll
      for (int i = 0; i < N; ++i) col[i] = dw[i] * (c3 == 0 ? J0[i] : (c3 == 1 ? J1[i] : J2[i]));
      for (int rr = d; rr < NZ; ++rr) {
        double hv = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) hv = fma(tl[i * TW + rr], col[i], hv);
        if (c3 == 0) {
          double hp = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) hp = fma(tl[JP + k * (NZ + 1) + rr], jp[k], hp);
          hv = fma(2.0 * P.w_path, hp, hv);
        }
        if (rr == d && c3 == 1) {
          hv += 2.0 * P.w_vel;
          hv += dvj;
        }
        if (add_hc) hv += hc[rr * (rr + 1) / 2 + d];
        sr[rr * (rr + 1) / 2 + d] = hv;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) sr[TQ_SD_J + i * NZ + d] = c3 == 0 ? J0[i] : (c3 == 1 ? J1[i] : J2[i]);
    }
    sr[231 + j] = g0;
    sr[231 + N + j] = g1;
    sr[231 + 2 * N + j] = g2;
    sr[TQ_SD_GB + j] = h0;
    sr[TQ_SD_GB + N + j] = h1;
    sr[TQ_SD_GB + 2 * N + j] = h2;
    if (j == 0) {
      sr[252] = P.w_path * dot3(r, r) + fsum;
      sr[253] = bar;
      sr[254] = (double)nrel;
      sr[255] = viol;
      sr[263] = cmpl;
    }
    sr[256 + j] = tvj;
  }
}

// Exact curvature of the Lagrangian for the instances the step kernel has switched to Newton steps (D.curv): adds
//     sum_i cH_i d^2 tau_i / dz^2,  cH = 2 w_tau tau - lam_lo + lam_up   (the multiplier of the dynamics row TAU_i - rnea_i = 0 at a stationary point)
//   + 2 w_p sum_k r_k d^2 p_k / dq^2,  d^2 p / dq_a dq_b = z_b x (z_a x (e - o_a)) for b <= a
// to the stage block k_tq_eval3 has just written.  One lane per (instance, knot, joint) again: lane j runs the hand-written adjoint of the recursion on
// (DualR, Dual2) scalars seeded with q_j and dq_j, which yields rows q_j and dq_j of the first term; every packed entry is owned by exactly one lane.
// Round 5: the term also goes to a record of its own (D.hc), and an instance computes it afresh only at every
// (curv_lag + 1)-th evaluation (D.curv: 1 compute here, 2 k_tq_eval3 adds the stored term as it writes the block and this kernel skips the instance).  Near the solution the term moves little between steps: with a lag of 3 the
// port takes 24.70 instead of 24.64 steps on 256 instances and computes the term 3.2 times per solve instead of 11.3 (oracle/torque_ipm.py, HISTORY).
#ifndef OH_TQ_CURV_WAVES
#define OH_TQ_CURV_WAVES 1
#endif
template <int N>
__global__ __launch_bounds__(64, OH_TQ_CURV_WAVES) void k_tq_curv(TqParams P, TqBuffers D) {
  constexpr int UPW = 64 / N;
  __shared__ double zs_l[UPW][32];
  const int T = P.T;
  const int lane = threadIdx.x;
  int ul = lane / N, j = lane - ul * N;
  const bool lane_ok = ul < UPW;
  if (!lane_ok) {
    ul = UPW - 1;
    j = N - 1;
  }
  const long long n_units = (long long)D.n_run * T;
  long long unit = (long long)blockIdx.x * UPW + ul;
  bool active = lane_ok && unit < n_units;
  if (unit >= n_units) unit = n_units - 1;
  const int li = (int)(unit / T), t = (int)(unit - (long long)li * T);
  const int b = D.list[li];
  if (D.status[b] >= 0 || D.curv[b] != 1) active = false;  // (2: k_tq_eval3 has added the stored term already)
  if (!__any(active)) return;
  const int ts = 1 - D.cur[b];
  const double* xr = D.xs + xs_off(D, T, ts, b, t);
  double* sr = D.st + st_off(D, T, ts, b, t);
  double* hc = D.hc + ((size_t)b * T + t) * TQ_HC;
  const double* lm = D.lam + (((size_t)ts * D.B + b) * T + t) * TQ_LAM;
  const bool comp = active;
  const bool direct = P.curv_lag <= 0;
  {
  if (lane_ok) {
    zs_l[ul][j] = xr[j];
    zs_l[ul][8 + j] = xr[8 + j];
    zs_l[ul][16 + j] = xr[16 + j];
    zs_l[ul][24 + j] = 2.0 * P.w_tau * sr[256 + j] - lm[j] + lm[N + j];
  }
  __syncthreads();
  // curvature of the tracking term
  double inner[3] = {0.0, 0.0, 0.0}, r[3], z[N][3];
  {
    double qv[N], R[9], pp[3], pj[N][3], e[3], tv3[3];
#pragma unroll
    for (int k = 0; k < N; ++k) qv[k] = zs_l[ul][k];
    fk_chain<N>(D.chain, qv, R, pp, z, pj);
    mv3(R, D.chain->p_tool, tv3);
    const double* gl = D.goal + ((size_t)b * T + t) * 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      e[k] = pp[k] + tv3[k];
      r[k] = e[k] - gl[k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (k == j) {
        if (D.chain->jtype[k] == 0) {
          const double dd[3] = {e[0] - pj[k][0], e[1] - pj[k][1], e[2] - pj[k][2]};
          cross3(z[k], dd, inner);
        } else {
          inner[0] = z[k][0]; inner[1] = z[k][1]; inner[2] = z[k][2];
        }
      }
  }
  if (comp) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (k <= j) {  // rows q_j, columns up to the diagonal (this lane's entries: the store initialises them, the adjoint below adds to them)
        double x[3] = {0.0, 0.0, 0.0};
        if (D.chain->jtype[k] == 0) cross3(z[k], inner, x);
        const double v = 2.0 * P.w_path * dot3(r, x);
        sr[j * (j + 1) / 2 + k] += v;
        if (!direct) hc[j * (j + 1) / 2 + k] = v;
      }
  }
  rnea_ctau_grad_inv<N + 1, Dual2, DualR>(D.dyn, zs_l[ul], j, [&](const int k, const Dual2 gq, const Dual2 gqd, const Dual2 gqdd) {
    if (!comp) return;
    // onto the stage block ...
    if (k <= j) {
      sr[j * (j + 1) / 2 + k] += gq.d0;
      sr[(N + j) * (N + j + 1) / 2 + N + k] += gqd.d1;
    }
    sr[(N + j) * (N + j + 1) / 2 + k] += gq.d1;
    sr[(2 * N + k) * (2 * N + k + 1) / 2 + j] += gqdd.d0;
    if (direct) return;  // curv_lag = 0: nothing is kept
    // ... and into the record the next evaluations of this instance add instead (k_tq_eval3)
    if (k <= j) {  // rows q_j and dq_j, columns up to the diagonal
      hc[j * (j + 1) / 2 + k] += gq.d0;
      hc[(N + j) * (N + j + 1) / 2 + N + k] = gqd.d1;
    }
    hc[(N + j) * (N + j + 1) / 2 + k] = gq.d1;             // (dq_j, q_k)
    hc[(2 * N + k) * (2 * N + k + 1) / 2 + j] = gqdd.d0;   // (ddq_k, q_j)
  });
  }
}

// ---- step ------------------------------------------------------------------------------------------------------------------------------
OH_DEV double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
OH_DEV double wave_max(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
  return v;
}
OH_DEV double wave_min(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
  return v;
}
OH_DEV double row_sum8(double v) {  // sum over the 8 lanes of a row (lane = 8 r + c)
  v += __shfl_xor(v, 1, 8);
  v += __shfl_xor(v, 2, 8);
  v += __shfl_xor(v, 4, 8);
  return v;
}

// One wavefront per instance (round 4; rounds 1-3: 16 lanes per instance around a value matrix in LDS, every lane factorising Q_uu for itself, three
// block barriers per knot -- 105 us per sweep however small the batch).  Lane 8 r + c owns entry (r, c) of every N x N block of the stage matrix
//     M = H_t + [A^T P A, A^T P B; B^T P A, B^T P B] + mu I_x   over (q, dq | u),
// column c = 7 holds the vectors.  With A = [[I, dt I], [0, I]], B = [0; dt I] every block of M is an entry-wise combination of the four blocks of P,
// so forming M needs nothing from another lane; the N pivots of the u rows are then eliminated by Gauss-Jordan steps whose operands travel by lane
// shuffles (row r of the pivot column, column c of the pivot row): no LDS, no barrier.  What remains in the x rows is the Schur complement
// P' = Q_xx - Q_ux^T Q_uu^{-1} Q_ux (exactly symmetric: mirrored entries subtract the same product), in the u rows the gains K = Q_uu^{-1} Q_ux and
// k = Q_uu^{-1} q_u; the pivots are those of the Cholesky factorisation (positive iff Q_uu is positive definite).  The roll-outs and the costate
// recursion use the same layout: entry-wise work on the lanes that hold it, row sums by three shuffles.  numpy: oracle/torque_ipm.py.
//   1. merit of the trial  f + mu_b B  and the ratio test against the decrease the damped model predicted for the scaled step; Levenberg-Marquardt update
//   2. reduced gradient of the accepted point by the costate recursion, for the barrier parameter in force and for the next one
//   3. convergence / barrier update (the stage gradient is  g_f + mu_b g_b,  the merit  f + mu_b B:  a new mu_b needs no re-evaluation)
//   4. Riccati sweep; a stage block of the exact Hessian that leaves Q_uu indefinite raises the damping and the sweep is repeated
//   5. closed-loop rollout of the unit step: the control steps go to LDS, the linearised rows give the fraction to the boundary alpha
//   6. open-loop rollout of  u + alpha du  on the trial values themselves (the Euler rows hold to the rounding of one operation)
// A trial that left the domain of the arithmetic, or a boundary-shortened step that was rejected, is retried shorter: steps 5-6 only.
#ifndef OH_TQ_STEP_WAVES
#define OH_TQ_STEP_WAVES 2
#endif
template <int N, bool VEL = false>
__global__ __launch_bounds__(64, OH_TQ_STEP_WAVES) void k_tq_step(TqParams P, TqBuffers D) {
  static_assert(N >= 1 && N <= 7, "the lane layout is 8 rows x 8 columns: up to 7 joints and the vector column (column N)");
  constexpr int NX = 2 * N, NZ = 3 * N;
  extern __shared__ double du_dyn[];  // [T][8]: the control steps of the unit step
  const int T = P.T;
  const int lane = threadIdx.x;
  const int r = lane >> 3, c = lane & 7;
  const bool mat = r < N && c < N;  // an entry of the N x N blocks
  const bool vec = r < N && c == N;  // an entry of the vectors
  if ((int)blockIdx.x >= D.n_run) return;
  const int b = D.list[blockIdx.x];
  if (D.status[b] >= 0) return;
  const double dt = P.dt;
  int cur = D.cur[b];
  const int ts = 1 - cur;
  // 1. merit of the trial point
  double fsum = 0.0, bsum_t = 0.0, nrel_t = 0.0, viol_t = 0.0;
  for (int t = lane; t < T; t += 64) {
    const double* sr = D.st + st_off(D, T, ts, b, t);
    fsum += sr[252];
    bsum_t += sr[253];
    nrel_t += sr[254];
    viol_t = fmax(viol_t, sr[255]);
  }
  fsum = wave_sum(fsum);
  bsum_t = wave_sum(bsum_t);
  nrel_t = wave_sum(nrel_t);
  viol_t = wave_max(viol_t);

  const bool first = D.first[b] != 0;
  double f_cur = D.f_cur[b], f_true = D.f_true[b], bsum = D.bsum[b], mu = D.mu[b], nun = D.nun[b], mub = D.mub[b], alpha = D.alpha[b], viol = D.viol[b];
  double qk = D.qk[b], ndx = D.ndx[b];
  int iters = D.iters[b], rejected = D.rejected[b], n_barrier = D.n_barrier[b], nrel = D.nrel[b], n_back = D.n_back[b], stall = D.stall[b];
  const double f_t = fsum + mub * bsum_t;
  bool accept, new_gains = true;
  if (first) {
    accept = true;
  } else if (!isfinite(f_t)) {
    accept = false;
    new_gains = false;
    alpha *= 0.1;
    rejected += 1;
  } else {
    const double pred = (alpha - 0.5 * alpha * alpha) * qk + 0.5 * alpha * alpha * mu * ndx;
    const double ratio = (f_cur - f_t) / fmax(pred, 1e-300);
    accept = ratio > 1e-4 || (pred <= 1e-15 * fabs(f_cur) && f_t <= f_cur + 1e-14 * fabs(f_cur));
    if (accept) {
      const double w = 2.0 * ratio - 1.0;
      mu *= ratio > 0.9 ? P.mu_dec : fmax(1.0 / 3.0, 1.0 - w * w * w);
      if (mu < 1e-7) mu = 0.0;
      nun = 4.0;
    } else if ((alpha < 1.0 || (P.ls_curv && D.curv[b] != 0)) && n_back < P.max_back) {
      // a step the boundary rule had shortened already: the rows near their bounds are to blame (the logarithm is far from its quadratic model
      // there), not the model of the states -- a quarter of the feed-forward, same gains, same damping.  Round 5: likewise a full Newton step
      // (exact curvature) that was rejected -- measured on stragglers (tools/gpu_tq_param_sweep.py, HISTORY): the full step gives up more barrier
      // than it gains (a slack drops to a third), 0.3 of it follows the model to 8 %, and the damping, which acts on the states, does not shorten a
      // step that lives in the accelerations: five rejections and six careful steps afterwards, or one quartering
      new_gains = false;
      alpha *= 0.25;
      n_back += 1;
      rejected += 1;
    } else {
      mu = fmax(mu * nun, 0.1);
      nun *= 2.0;
      rejected += 1;
    }
  }
  if (accept || new_gains) n_back = 0;
  if (accept) {
    cur = ts;
    f_cur = f_t;
    f_true = fsum;
    bsum = bsum_t;
    nrel = (int)nrel_t;
    viol = viol_t;
  }
  const int nts = 1 - cur;  // slot of the next trial

  // 2. reduced gradient of the accepted point: gradient of the rolled-out merit w.r.t. u_t by the costate recursion, for mu_b and for its successor.
  // Entry-wise per joint: lanes (r, 7) carry the costates of q_r and dq_r for the cost and the barrier part; what a knot needs from memory is
  // requested one knot ahead.
  const double mu_min = 0.1 * P.tol_compl;
  const double mub_next = fmax(mu_min, fmin(P.kappa_mu * mub, pow(mub, P.theta_mu)));
  const double mub_up = fmin(P.mu_b0, 100.0 * mub);  // the watchdog's way back up (step 3)
  double stat = 0.0, stat_next = 0.0, stat_up = 0.0;
  {
    double lfq = 0.0, lfd = 0.0, lbq = 0.0, lbd = 0.0;
    double g_n[6] = {0, 0, 0, 0, 0, 0};
    auto fetch = [&](const int t) {
      if (vec) {
        const double* sr = D.st + st_off(D, T, cur, b, t);
        g_n[0] = sr[231 + r]; g_n[1] = sr[231 + N + r]; g_n[2] = sr[231 + NX + r];
        g_n[3] = sr[TQ_SD_GB + r]; g_n[4] = sr[TQ_SD_GB + N + r]; g_n[5] = sr[TQ_SD_GB + NX + r];
      }
    };
    fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      const double gfq = g_n[0], gfd = g_n[1], gfu = g_n[2], gbq = g_n[3], gbd = g_n[4], gbu = g_n[5];
      if (t > 0) fetch(t - 1);
      const double rf = fma(dt, lfd, gfu), rb = fma(dt, lbd, gbu);
      stat = fmax(stat, fabs(fma(mub, rb, rf)));
      stat_next = fmax(stat_next, fabs(fma(mub_next, rb, rf)));
      stat_up = fmax(stat_up, fabs(fma(mub_up, rb, rf)));
      const double nfd = gfd + fma(dt, lfq, lfd), nbd = gbd + fma(dt, lbq, lbd);
      lfq = gfq + lfq;
      lbq = gbq + lbq;
      lfd = nfd;
      lbd = nbd;
    }
    if (!vec) { stat = 0.0; stat_next = 0.0; stat_up = 0.0; }
    stat = wave_max(stat);
    stat_next = wave_max(stat_next);
    stat_up = wave_max(stat_up);
    if (!(stat == stat)) stat = 1e300;
    if (!(stat_next == stat_next)) stat_next = 1e300;
    if (!(stat_up == stat_up)) stat_up = 1e300;
  }

  // 3. convergence, barrier update
  int status = -1;
  bool do_gains = false, do_roll = false;
  bool reeval = false;  // the next "trial" is this point itself under a lower barrier parameter (round 6)
  int curv = D.curv[b] != 0;  // the pending trial was evaluated with exact curvature (1 computed, 2 the stored term: k_tq_curv)
  if (!isfinite(f_cur)) {
    status = OH_STATUS_NUMERICAL;
  } else if (stat <= P.tol && mub <= P.tol_compl && nrel == 0) {
    status = OH_STATUS_CONVERGED;
  } else if (iters >= P.max_iter) {
    status = OH_STATUS_MAX_ITER;
  } else {
    iters += 1;
    do_roll = true;
    if (new_gains) {
      do_gains = true;
      if (accept && stat <= P.kappa_eps * mub && nrel == 0 && mub > mu_min) {
        mub = mub_next;
        f_cur = f_true + mub * bsum;
        stat = stat_next;
        n_barrier += 1;
        stall = 0;
      } else if (accept && stat <= P.kappa_eps * mub && nrel > 0 && mub > mu_min) {
        // Round 6: stationary for this mu_b with rows inside the relaxed zone (slack below theta mu_b: a row whose multiplier exceeds 1 / theta -- a velocity limit
        // the tracking cost pushes hard against).  Merit and gradient are not affine in mu_b there, so the update above does not apply; without one the instance sat at
        // this point until the cap (null steps "rejected" at rounding level, the damping doubling).  The barrier parameter is lowered all the same and the point itself
        // evaluated again under it: a null step (alpha = 0 on the gains of the last sweep), accepted as it is (D.first).  oracle/torque_ipm.py alike.
        mub = mub_next;
        n_barrier += 1;
        stall = 0;
        reeval = true;
        do_gains = false;
        alpha = 0.0;
      } else if (accept && stat <= 10.0 * P.tol && nrel > 0 && mub <= mu_min && viol > P.tol_compl) {
        // ... and at the floor of the barrier parameter a stationary point that still violates a row has no feasible neighbour (the relaxed barrier is a penalty of
        // weight 1 / (theta^2 mu_b) by now): IPOPT's Infeasible_Problem_Detected, did_solve() False (solver.py:133-134, 407-412)
        status = OH_STATUS_INFEASIBLE;
        iters -= 1;
        do_roll = false;
        do_gains = false;
      } else {
        stall += 1;
        if (stall >= P.stall_max && nrel == 0 && mub <= mu_min && stat <= 10.0 * P.tol) {
          // "acceptable level" (IPOPT's acceptable_tol / acceptable_iter in spirit): stall_max steps at the floor of the barrier parameter without
          // meeting tol while within ten times of it -- the arithmetic floor of the reduced gradient: an active row with slack s carries mu_b / s,
          // and s = up - tau inherits the ~1e-14 of the recursion, so at s ~ 1e-8 the multiplier is good to ~1e-6 relative and the reduced gradient to
          // ~1e-6 absolute.  Without this such an instance (1 of 8192 in one of nine batches) fed the watchdog below and ended NUMERICAL at the optimum.
          status = OH_STATUS_ACCEPTABLE;
          iters -= 1;
          do_roll = false;
          do_gains = false;
        } else if (stall >= P.stall_max && nrel == 0 && accept) {
          // watchdog: stall_max steps without reaching the barrier test -- the iterate sits far from the central path of this mu_b (slacks of the active
          // rows collapse and recover in turn).  Back to a larger barrier parameter: the path is regained there and followed down again.
          mub = mub_up;
          f_cur = f_true + mub * bsum;
          stat = stat_up;
          stall = 0;
        }
      }
      if (!reeval) curv = (stat <= P.curv_from || (n_barrier >= P.curv_after && stat <= P.curv_late)) ? 1 : 0;
    }
  }

  // 4. Riccati sweep (everything below is uniform over the wavefront: one instance)
  // offsets of this lane's entries in the packed lower stage block: row(row + 1) / 2 + col, row >= col; q at 0, dq at N, u at 2 N
  auto pk = [](const int row, const int col) { return row >= col ? row * (row + 1) / 2 + col : col * (col + 1) / 2 + row; };
  const int rr = mat ? r : 0, cc = mat ? c : 0;
  const int o_qq = pk(rr, cc), o_qd = pk(rr, N + cc), o_dq = pk(N + rr, cc), o_dd = pk(N + rr, N + cc), o_qu = pk(rr, NX + cc), o_du = pk(N + rr, NX + cc),
            o_uq = pk(NX + rr, cc), o_ud = pk(NX + rr, N + cc), o_uu = pk(NX + rr, NX + cc);
  const int rv = vec ? r : 0;
  if (do_gains) {
    bool failed = true;
    for (int attempt = 0; attempt < 12 && failed; ++attempt) {
      failed = false;
      qk = 0.0;
      double Pqq = 0.0, Pqd = 0.0, Pdq = 0.0, Pdd = 0.0;  // mat lanes: blocks of P; vec lanes: Pqq = p_q[r], Pdd = p_d[r]
      double h_n[9];
      auto fetch_rec = [&](const int t) {
        const double* sr = D.st + st_off(D, T, cur, b, t);
        if (mat) {
          h_n[0] = sr[o_qq]; h_n[1] = sr[o_qd]; h_n[2] = sr[o_dq]; h_n[3] = sr[o_dd]; h_n[4] = sr[o_qu]; h_n[5] = sr[o_du]; h_n[6] = sr[o_uq]; h_n[7] = sr[o_ud];
          h_n[8] = sr[o_uu];
        } else {  // the stage gradient g_f + mu_b g_b
          h_n[0] = fma(mub, sr[TQ_SD_GB + rv], sr[231 + rv]);
          h_n[1] = fma(mub, sr[TQ_SD_GB + N + rv], sr[231 + N + rv]);
          h_n[2] = fma(mub, sr[TQ_SD_GB + NX + rv], sr[231 + NX + rv]);
        }
      };
      fetch_rec(T - 1);
      for (int t = T - 1; t >= 0; --t) {
        double Mqq, Mqd, Mdq, Mdd, Mqu, Mdu, Muq, Mud, Muu;
        if (c < N) {
          const double dg = (r == c) ? mu : 0.0;
          Mqq = h_n[0] + Pqq + dg;
          Mqd = h_n[1] + fma(dt, Pqq, Pqd);
          Mdq = h_n[2] + fma(dt, Pqq, Pdq);
          Mdd = h_n[3] + fma(dt, fma(dt, Pqq, Pqd), fma(dt, Pdq, Pdd)) + dg;
          Mqu = h_n[4] + dt * Pqd;
          Mdu = h_n[5] + dt * fma(dt, Pqd, Pdd);
          Muq = h_n[6] + dt * Pdq;
          Mud = h_n[7] + dt * fma(dt, Pdq, Pdd);
          Muu = h_n[8] + dt * dt * Pdd;
        } else {  // vectors: m_q, m_d, m_u in the slots of column "u" of their block row (Mqu, Mdu, Muu)
          Mqq = Mqd = Mdq = Mdd = Muq = Mud = 0.0;
          Mqu = h_n[0] + Pqq;
          Mdu = h_n[1] + fma(dt, Pqq, Pdd);
          Muu = h_n[2] + dt * Pdd;
        }
        if (t > 0) fetch_rec(t - 1);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          // column j of this lane's row (blocks X u) and row j of this lane's column (blocks u Y; for the vector column: m_u[j])
          const double cq = __shfl(Mqu, 8 * r + j), cd = __shfl(Mdu, 8 * r + j), cu = __shfl(Muu, 8 * r + j);
          const double rq = __shfl(Muq, 8 * j + c), rd = __shfl(Mud, 8 * j + c), ru = __shfl(Muu, 8 * j + c);
          const double piv = readlane_f64(Muu, 9 * j);
          if (!(piv > 0.0) || !isfinite(piv)) failed = true;
          double d = __builtin_amdgcn_rcp(piv);  // reciprocal to the last bit or two (two Newton steps): an IEEE division is 30 instructions on the critical path of every pivot
          d = d * fma(-piv, d, 2.0);
          d = d * fma(-piv, d, 2.0);
          if (c < N) {
            Mqq = fma(-(cq * rq), d, Mqq);
            Mqd = fma(-(cq * rd), d, Mqd);
            Mdq = fma(-(cd * rq), d, Mdq);
            Mdd = fma(-(cd * rd), d, Mdd);
            Mqu = fma(-(cq * ru), d, Mqu);
            Mdu = fma(-(cd * ru), d, Mdu);
            if (r == j) {
              Muq *= d;
              Mud *= d;
              Muu *= d;
            } else {
              Muq = fma(-(cu * rq), d, Muq);
              Mud = fma(-(cu * rd), d, Mud);
              Muu = fma(-(cu * ru), d, Muu);
            }
          } else {  // ru = m_u[j] (forward-eliminated: row j has only been updated by the pivots before it)
            if (r == 0) qk = fma(ru * ru, d, qk);
            Mqu = fma(-(cq * ru), d, Mqu);
            Mdu = fma(-(cd * ru), d, Mdu);
            if (r == j) Muu *= d;
            else Muu = fma(-(cu * ru), d, Muu);
          }
        }
        // gains of knot t: K_q[r][c], K_d[r][c] on the matrix lanes, k[r] on the vector lanes; two doubles per lane
        double* gn = D.gains + ((size_t)b * T + t) * TQ_GN;
        gn[2 * lane] = c < N ? Muq : Muu;
        gn[2 * lane + 1] = c < N ? Mud : 0.0;
        if (c < N) {
          Pqq = Mqq; Pqd = Mqd; Pdq = Mdq; Pdd = Mdd;
        } else {
          Pqq = Mqu; Pdd = Mdu; Pqd = Pdq = 0.0;
        }
      }
      qk = __shfl(qk, N);  // lane (0, 7)
      failed = __any(failed && r < N) || !isfinite(qk);
      if (failed) {  // an indefinite stage block of the exact Hessian: more damping, same point
        mu = fmax(mu * nun, 0.1);
        nun *= 2.0;
      }
    }
    if (failed) {
      status = OH_STATUS_NUMERICAL;
      do_roll = false;
    }
  }

  // 5. closed-loop rollout of the unit step (du to LDS), fraction to the boundary on the linearised rows.  Lane (r, c) carries the state step of joint
  // c (replicated over the rows) and multiplies it with its entries of the gains and of d tau / d z; row sums give du_r and d tau_r.
  const double delta = P.theta * mub;
  double (*du_all)[8] = reinterpret_cast<double (*)[8]>(du_dyn);
  if (do_roll) {
    double dq = 0.0, dd = 0.0;
    double a_ftb = 1.0, ndx_acc = 0.0;
    double k_n[2], j_n[3], s_n[4] = {0, 0, 0, 0};
    auto fetch_knot = [&](const int t) {
      const double* gn = D.gains + ((size_t)b * T + t) * TQ_GN;
      k_n[0] = gn[2 * lane];
      k_n[1] = gn[2 * lane + 1];
      const double* sr = D.st + st_off(D, T, cur, b, t);
      if (mat) {
        j_n[0] = sr[TQ_SD_J + r * NZ + c];
        j_n[1] = sr[TQ_SD_J + r * NZ + N + c];
        j_n[2] = sr[TQ_SD_J + r * NZ + NX + c];
      } else {
        j_n[0] = j_n[1] = j_n[2] = 0.0;
      }
      if (r < N && c == 0) {
        const double tv = sr[256 + r];
        s_n[0] = tv - P.tau_lo[r];
        s_n[1] = P.tau_up[r] - tv;
        if constexpr (VEL) {
          const double dv = D.xs[xs_off(D, T, cur, b, t) + 8 + r];
          s_n[2] = dv - P.dq_lo[r];
          s_n[3] = P.dq_up[r] - dv;
        }
      }
    };
    fetch_knot(0);
    for (int t = 0; t < T; ++t) {
      const double kq = k_n[0], kd = k_n[1], jq = j_n[0], jd = j_n[1], ju = j_n[2];
      const double s_lo = s_n[0], s_up = s_n[1], v_lo = s_n[2], v_up = s_n[3];
      if (t + 1 < T) fetch_knot(t + 1);
      // du_r = -k_r - sum_c (K_q[r][c] dq_c + K_d[r][c] dd_c): the vector lane contributes k_r (its kq slot)
      const double part = c < N ? fma(kq, dq, kd * dd) : (c == N ? kq : 0.0);  // (columns beyond N exist on chains shorter than seven joints: idle lanes)
      const double du_r = -row_sum8(r < N ? part : 0.0);
      const double du_c = __shfl(du_r, 8 * c);  // du of joint c, from row c
      if (r == 0 && c < N) du_all[t][c] = du_c;
      if (r == 0 && c < N) ndx_acc = fma(dq, dq, fma(dd, dd, ndx_acc));
      const double ds = row_sum8(mat ? fma(jq, dq, fma(jd, dd, ju * du_c)) : 0.0);  // d tau_r of the unit step
      double dv = 0.0;
      if constexpr (VEL) dv = __shfl(dd, r);  // the velocity step of joint r sits on lane (0, r) (every lane takes part in the shuffle)
      if (r < N && c == 0) {
        if (ds < 0.0 && s_lo >= delta) a_ftb = fmin(a_ftb, -P.tau_ftb * s_lo / ds);
        if (ds > 0.0 && s_up >= delta) a_ftb = fmin(a_ftb, P.tau_ftb * s_up / ds);
        if constexpr (VEL) {
          if (dv < 0.0 && v_lo >= delta) a_ftb = fmin(a_ftb, -P.tau_ftb * v_lo / dv);
          if (dv > 0.0 && v_up >= delta) a_ftb = fmin(a_ftb, P.tau_ftb * v_up / dv);
        }
      }
      // dx_{t+1} = A dx_t + B du_t
      dq = fma(dt, dd, dq);
      dd = fma(dt, du_c, dd);
    }
    if (do_gains) {
      ndx = wave_sum(ndx_acc);
      alpha = wave_min(a_ftb);
    }
  }
  __syncthreads();  // du_all

  // 6. open-loop rollout of u + alpha du from the fixed initial state: lane c < N carries joint c
  if (do_roll && lane < N) {
    const double* x0r = D.xs + xs_off(D, T, cur, b, 0);
    double q = x0r[lane], dqv = x0r[8 + lane];
    double u_n = x0r[16 + lane];
    for (int t = 0; t < T; ++t) {
      const double ucur = u_n;
      if (t + 1 < T) u_n = D.xs[xs_off(D, T, cur, b, t + 1) + 16 + lane];
      double* xn = D.xs + xs_off(D, T, nts, b, t);
      const double un = fma(alpha, du_all[t][lane], ucur);
      xn[16 + lane] = un;
      xn[lane] = q;
      xn[8 + lane] = dqv;
      // x_{t+1} = A x_t + B u_t on the trial values themselves
      q = fma(dt, dqv, q);
      dqv = fma(dt, un, dqv);
    }
  }
  if (lane == 0) {
    D.cur[b] = cur;
    D.first[b] = reeval ? 1 : 0;
    {  // next evaluation: the curvature term afresh, or the stored one while it is younger than curv_lag evaluations
      int age = D.curv_age[b], mode = 0;
      if (!accept && D.curv[b] == 1) age = -1;  // a term computed at a point that was refused (possibly outside the domain of the arithmetic) is not kept
      if (curv) {
        if (age >= 0 && age < P.curv_lag) { mode = 2; age += 1; }
        else { mode = 1; age = 0; }
      } else age = -1;
      D.curv[b] = mode;
      D.curv_age[b] = age;
    }
    D.f_cur[b] = f_cur;
    D.f_true[b] = f_true;
    D.bsum[b] = bsum;
    D.mu[b] = mu;
    D.nun[b] = nun;
    D.mub[b] = mub;
    D.stat[b] = stat;
    D.alpha[b] = alpha;
    D.qk[b] = qk;
    D.ndx[b] = ndx;
    D.viol[b] = viol;
    D.nrel[b] = nrel;
    D.n_back[b] = n_back;
    D.stall[b] = stall;
    D.iters[b] = iters;
    D.rejected[b] = rejected;
    D.n_barrier[b] = n_barrier;
    D.status[b] = status;
    if (status < 0) atomicAdd(D.n_running, 1);
  }
}

// ---- results in the reference layout ------------------------------------------------------------------------------------------------
// Multipliers: lam_i = mu_b / s_i, the point of the central path the iteration stopped at (lam_i s_i = mu_b <= tol_compl on every row, and with them
// the reduced gradient of the Lagrangian is the `stat` the iteration tested).
template <int N>
__global__ __launch_bounds__(64) void k_tq_finalize(TqParams P, TqBuffers D, double* __restrict__ x, double* __restrict__ f, double* __restrict__ kkt,
                                                    int* __restrict__ iters, int* __restrict__ status, double* __restrict__ mult) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.B) return;
  const int T = P.T, cur = D.cur[b];
  const double mub = D.mub[b];
  double cmpl = 0.0;
  for (int t = 0; t < T; ++t) {
    const double* xr = D.xs + xs_off(D, T, cur, b, t);
    const double* sr = D.st + st_off(D, T, cur, b, t);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (x) {
        double* xb = x + (size_t)b * P.nx;
        xb[N * t + j] = xr[j];
        xb[N * T + N * t + j] = xr[8 + j];
        xb[2 * N * T + N * t + j] = xr[16 + j];
        xb[3 * N * T + N * t + j] = sr[256 + j];
      }
      const double tv = sr[256 + j];
      const double s_lo = tv - P.tau_lo[j], s_up = P.tau_up[j] - tv;
      const double l_lo = mub / fmax(s_lo, 1e-300), l_up = mub / fmax(s_up, 1e-300);
      cmpl = fmax(cmpl, fmax(fabs(l_lo * s_lo), fabs(l_up * s_up)));
      const int NR = P.vel ? 4 * N : 2 * N;  // rows per knot: effort rows, then (with velocity limits) [dq - dq_lo; dq_up - dq]
      if (mult) {
        mult[((size_t)b * T + t) * NR + j] = l_lo;
        mult[((size_t)b * T + t) * NR + N + j] = l_up;
      }
      if (P.vel) {
        const double v_lo = xr[8 + j] - P.dq_lo[j], v_up = P.dq_up[j] - xr[8 + j];
        const double m_lo = mub / fmax(v_lo, 1e-300), m_up = mub / fmax(v_up, 1e-300);
        cmpl = fmax(cmpl, fmax(fabs(m_lo * v_lo), fabs(m_up * v_up)));
        if (mult) {
          mult[((size_t)b * T + t) * NR + 2 * N + j] = m_lo;
          mult[((size_t)b * T + t) * NR + 3 * N + j] = m_up;
        }
      }
    }
  }
  if (f) f[b] = D.f_true[b];
  if (kkt) {
    kkt[3 * b] = D.stat[b];
    kkt[3 * b + 1] = fmax(0.0, D.viol[b]);
    kkt[3 * b + 2] = cmpl;
  }
  if (iters) iters[b] = D.iters[b];
  int st = D.status[b] < 0 ? OH_STATUS_MAX_ITER : D.status[b];
  if (P.vel) {
    // dq_0 = dqc is pinned (fix_configuration on the velocity state), so its velocity-limit rows are constants of the instance: outside them there is no
    // feasible point -- IPOPT's "infeasible problem" (solver.py:407-412); the interior point above can only sit in the relaxed barrier until its cap
    const double* x0r = D.xs + xs_off(D, T, cur, b, 0);
    double worst = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) worst = fmin(worst, fmin(x0r[8 + j] - P.dq_lo[j], P.dq_up[j] - x0r[8 + j]));
    if (worst < -1e-9) {
      st = OH_STATUS_INFEASIBLE;
      if (kkt) kkt[3 * b + 1] = fmax(kkt[3 * b + 1], -worst);
    }
  }
  if (status) status[b] = st;
}

}  // namespace

// The solver's kernels by chain length: 2 .. 7 joints (k_tq_step's lane layout is 8 x 8: N joints and the vector column)
#define OH_TQ_DISPATCH(n, C) \
  switch (n) {               \
    case 2: C(2); break;     \
    case 3: C(3); break;     \
    case 4: C(4); break;     \
    case 5: C(5); break;     \
    case 6: C(6); break;     \
    case 7: C(7); break;     \
    default: return false;   \
  }
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
bool oh_launch_rnea_hess(hipStream_t s, const oh_dynamics* d_dyn, int nbodies, int n, const double* q, const double* qd, const double* qdd, const double* c, double* H) {
#define OH_RH(NN)                                                                                                                       \
  case NN + 1:                                                                                                                          \
    hipLaunchKernelGGL(k_rnea_hess<NN>, dim3((unsigned)((n + (64 / NN) - 1) / (64 / NN))), dim3(64), 0, s, d_dyn, n, q, qd, qdd, c, H);  \
    return true;
  switch (nbodies) {
    OH_RH(1) OH_RH(2) OH_RH(3) OH_RH(4) OH_RH(5) OH_RH(6) OH_RH(7) OH_RH(8)
    default: return false;
  }
#undef OH_RH
}
bool oh_launch_tq_setup(hipStream_t s, const TqParams& P, const TqBuffers& D, const double* x0, const double* p) {
#define C(NN) hipLaunchKernelGGL(k_tq_setup<NN>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, x0, p)
  OH_TQ_DISPATCH(P.N, C)
#undef C
  return true;
}
void oh_launch_tq_list(hipStream_t s, const TqBuffers& D) { hipLaunchKernelGGL(k_tq_list, dim3((D.B + 255) / 256), dim3(256), 0, s, D); }
bool oh_launch_tq_eval(hipStream_t s, const TqParams& P, const TqBuffers& D) {
  const long long units = (long long)D.n_run * P.T;
  // (the variants with joint-velocity rows are instantiations of their own: as a run-time branch the rows cost k_tq_eval3 21 % at 8192 instances)
  // a wavefront holds 64 / N units (nine of the 7-joint arm)
#define C(NN)                                                                                             \
  {                                                                                                       \
    const dim3 grid((unsigned)((units + (64 / NN) - 1) / (64 / NN)));                                     \
    if (P.jac_closed_form) {                                                                              \
      if (P.vel) hipLaunchKernelGGL((k_tq_eval3<NN, true, true>), grid, dim3(64), 0, s, P, D);            \
      else hipLaunchKernelGGL((k_tq_eval3<NN, false, true>), grid, dim3(64), 0, s, P, D);                 \
    } else {                                                                                              \
      if (P.vel) hipLaunchKernelGGL((k_tq_eval3<NN, true, false>), grid, dim3(64), 0, s, P, D);           \
      else hipLaunchKernelGGL((k_tq_eval3<NN, false, false>), grid, dim3(64), 0, s, P, D);                \
    }                                                                                                     \
    hipLaunchKernelGGL(k_tq_curv<NN>, grid, dim3(64), 0, s, P, D); /* exits at once where no instance takes Newton steps */ \
  }
  OH_TQ_DISPATCH(P.N, C)
#undef C
  return true;
}
bool oh_launch_tq_step(hipStream_t s, const TqParams& P, const TqBuffers& D) {
#define C(NN)                                                                                                              \
  {                                                                                                                        \
    if (P.vel) hipLaunchKernelGGL((k_tq_step<NN, true>), dim3(D.n_run), dim3(64), sizeof(double) * 8 * P.T, s, P, D);      \
    else hipLaunchKernelGGL((k_tq_step<NN>), dim3(D.n_run), dim3(64), sizeof(double) * 8 * P.T, s, P, D);                  \
  }
  OH_TQ_DISPATCH(P.N, C)
#undef C
  return true;
}
// ---- receding horizon, resident on the device (oh_tq_rollout, round 5) ----------------------------------------------------------------------
// One thread per plant.  The reference's pattern (example/point_mass_mpc.py:156-175): parameters of the tick from the plant's state, seed = the
// previous solution, solve, the plant takes the plan's next state.  Reference layouts: x [B][4 N T] = [vec(Q); vec(dQ); vec(ddQ); vec(TAU)],
// p [B][2 N + 3 T] = [qc; dqc; vec(goal 3 x T)].
__global__ __launch_bounds__(64) void k_tq_tick_params(const int B, const int T, const int N, const int first_row, const int n_rows, const double* __restrict__ state,
                                                       const double* __restrict__ goal_table, double* __restrict__ p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double* pb = p + (size_t)b * (2 * N + 3 * T);
  for (int i = 0; i < 2 * N; ++i) pb[i] = state[(size_t)b * 2 * N + i];
  const double* gt = goal_table + ((size_t)b * n_rows + first_row) * 3;
  for (int i = 0; i < 3 * T; ++i) pb[2 * N + i] = gt[i];
}
// seed of the next tick: the accelerations of the plan shifted by `advance` knots, the last one repeated (only the ddQ block of a seed is read:
// the states are rolled out from the plant's state, the torques follow from the dynamics rows)
__global__ __launch_bounds__(64) void k_tq_shift_seed(const int B, const int T, const int N, const int advance, const double* __restrict__ x_prev,
                                                      double* __restrict__ x_seed) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* u = x_prev + (size_t)b * 4 * N * T + 2 * N * T;
  double* us = x_seed + (size_t)b * 4 * N * T + 2 * N * T;
  for (int t = 0; t < T; ++t) {
    const int ts = t + advance < T ? t + advance : T - 1;
    for (int j = 0; j < N; ++j) us[N * t + j] = u[N * ts + j];
  }
}
// the plant follows the plan for `advance` knots (the plan's states ARE the Euler roll-out of its accelerations, and its torques those of the
// inverse dynamics at its states: tau_t = rnea(q_t, dq_t, ddq_t) to 1e-14); the torque applied at the tick is the one of knot 0
__global__ __launch_bounds__(64) void k_tq_advance(const int B, const int T, const int N, const int advance, const double* __restrict__ x, double* __restrict__ state_next,
                                                   double* __restrict__ tau0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* xb = x + (size_t)b * 4 * N * T;
  for (int j = 0; j < N; ++j) {
    state_next[(size_t)b * 2 * N + j] = xb[N * advance + j];
    state_next[(size_t)b * 2 * N + N + j] = xb[N * T + N * advance + j];
    if (tau0) tau0[(size_t)b * N + j] = xb[3 * N * T + j];
  }
}
void oh_launch_tq_tick_params(hipStream_t s, int B, int T, int N, int first_row, int n_rows, const double* state, const double* goal_table, double* p) {
  hipLaunchKernelGGL(k_tq_tick_params, dim3((B + 63) / 64), dim3(64), 0, s, B, T, N, first_row, n_rows, state, goal_table, p);
}
void oh_launch_tq_shift_seed(hipStream_t s, int B, int T, int N, int advance, const double* x_prev, double* x_seed) {
  hipLaunchKernelGGL(k_tq_shift_seed, dim3((B + 63) / 64), dim3(64), 0, s, B, T, N, advance, x_prev, x_seed);
}
void oh_launch_tq_advance(hipStream_t s, int B, int T, int N, int advance, const double* x, double* state_next, double* tau0) {
  hipLaunchKernelGGL(k_tq_advance, dim3((B + 63) / 64), dim3(64), 0, s, B, T, N, advance, x, state_next, tau0);
}
bool oh_launch_tq_finalize(hipStream_t s, const TqParams& P, const TqBuffers& D, double* x, double* f, double* kkt, int* iters, int* status, double* mult) {
#define C(NN) hipLaunchKernelGGL(k_tq_finalize<NN>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, x, f, kkt, iters, status, mult)
  OH_TQ_DISPATCH(P.N, C)
#undef C
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
  if (n == "k_tq_curv") return kernel_info(k_tq_curv<7>, 64, out);
  if (n == "k_tq_step") return kernel_info(k_tq_step<7>, 64, out);
  return false;
}
