// Small dense QP family (the QuadraticCost* classes the reference hands to OSQP / CVXOPT / qpOASES,
// solver.py:421-584; its own known-answer solver test, tests/test_solver.py:22-54, is one of them):
//
//     min_x  x^T P x + q^T x      s.t.  M x + c >= 0,   A x + b = 0           (optimization.py:219-260: no factor 1/2)
//
// One thread owns one instance: infeasible-start primal-dual interior point with slacks s = Mx + c, Newton system reduced to
// H = 2P + M^T (lam/s) M (dense Cholesky, n <= OH_QP_MAX_N) and the Schur complement A H^{-1} A^T for the equality rows.
// The matrices differ per instance (P, M, A may depend on the parameters: the Booth test has a * y in its cost), so every
// instance brings its own [P | q | M | c | A | b] row; work arrays are thread-private slices of one global buffer.
// numpy restatement of the same iteration: oracle/qp_ipm.py.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "oh_kernels.h"

namespace {

// A work array of one instance: element k lives at p[k * stride].  The instance's slices sit either in LDS ([k][lane], when the whole work set of
// a block fits: the iteration is a chain of dependent accesses, and from global memory every one of them is a round trip -- 2 ms per solve at
// B = 1) or in the global work buffer ([k][instance]: the lanes of a wavefront touch one line per access).
struct QArr {
  double* p;
  int stride;
  __device__ double& operator[](const int i) const { return p[(size_t)i * stride]; }
  __device__ QArr at(const int off) const { return QArr{p + (size_t)off * stride, stride}; }
};

// in-place Cholesky of the n x n row-major SPD matrix H (lower triangle); returns false on a non-positive pivot
__device__ bool qp_chol(const QArr H, const int n) {
  for (int j = 0; j < n; ++j) {
    double d = H[j * n + j];
    for (int k = 0; k < j; ++k) d -= H[j * n + k] * H[j * n + k];
    if (!(d > 0.0)) return false;
    const double l = sqrt(d);
    H[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = H[i * n + j];
      for (int k = 0; k < j; ++k) v -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = v / l;
    }
  }
  return true;
}
__device__ void qp_solve_chol(const QArr L, const int n, const QArr x) {  // x <- (L L^T)^{-1} x
  for (int i = 0; i < n; ++i) {
    double v = x[i];
    for (int k = 0; k < i; ++k) v -= L[i * n + k] * x[k];
    x[i] = v / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = x[i];
    for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * x[k];
    x[i] = v / L[i * n + i];
  }
}

// MODE 0: work set in the global buffer; 1: in LDS; 2: in LDS together with the instance's [P | q | M | c | A | b] row
template <int MODE>
__global__ __launch_bounds__(64) void k_qp_solve(QpParams Q, int B, int Bp, const double* __restrict__ x0, const double* __restrict__ par, double* __restrict__ work,
                                                 double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters,
                                                 int* __restrict__ status, double* __restrict__ mult) {
  extern __shared__ double qp_sm[];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = Q.n, m = Q.m, me = Q.me;
  QArr w = MODE ? QArr{qp_sm + threadIdx.x, (int)blockDim.x} : QArr{work + b, Bp};
  // (the row is read-only; const_cast only to share the accessor type)
  QArr pr{const_cast<double*>(par) + (size_t)b * Q.np, 1};
  if (MODE == 2) {
    for (int k = 0; k < Q.np; ++k) w[k] = pr[k];
    pr = w;
    w = w.at(Q.np);
  }
  const QArr P = pr;                 // [n][n]
  const QArr q = P.at(n * n);        // [n]
  const QArr M = q.at(n);            // [m][n]
  const QArr c = M.at(m * n);        // [m]
  const QArr A = c.at(m);            // [me][n]
  const QArr bv = A.at(me * n);      // [me]
  const QArr x = w; w = w.at(n);
  const QArr s = w; w = w.at(m);
  const QArr lam = w; w = w.at(m);
  const QArr nu = w; w = w.at(me);
  const QArr H = w; w = w.at(n * n);
  const QArr rhs = w; w = w.at(n);
  const QArr dx = w; w = w.at(n);
  const QArr ds = w; w = w.at(m);
  const QArr dl = w; w = w.at(m);
  const QArr Y = w; w = w.at(me * n);   // H^{-1} A^T, row i = H^{-1} A_i
  const QArr S = w; w = w.at(me * me);
  const QArr dnu = w; w = w.at(me);
  const QArr rd = w; w = w.at(n);
  for (int i = 0; i < n; ++i) x[i] = x0[(size_t)b * n + i];
  double mu = 1.0;
  for (int i = 0; i < m; ++i) {
    double v = c[i];
    for (int j = 0; j < n; ++j) v += M[i * n + j] * x[j];
    s[i] = fmax(v, 1.0);
    lam[i] = mu / s[i];
  }
  for (int i = 0; i < me; ++i) nu[i] = 0.0;
  int st = OH_STATUS_MAX_ITER, it = 0;
  double stat = 0.0, feas = 0.0, gap = 0.0;
  for (; it <= Q.max_iter; ++it) {
    // residuals
    stat = 0.0; feas = 0.0; gap = 0.0;
    bool finite = true;
    for (int i = 0; i < n; ++i) {
      double v = q[i];
      for (int j = 0; j < n; ++j) v += 2.0 * P[i * n + j] * x[j];
      for (int k = 0; k < m; ++k) v -= M[k * n + i] * lam[k];
      for (int k = 0; k < me; ++k) v -= A[k * n + i] * nu[k];
      rd[i] = v;
      stat = fmax(stat, fabs(v));
      finite = finite && (v == v) && (fabs(v) < 1e300);
    }
    for (int i = 0; i < m; ++i) {
      double v = c[i] - s[i];
      for (int j = 0; j < n; ++j) v += M[i * n + j] * x[j];
      ds[i] = v;  // r_p
      feas = fmax(feas, fabs(v));
      gap = fmax(gap, s[i] * lam[i]);
    }
    for (int i = 0; i < me; ++i) {
      double v = bv[i];
      for (int j = 0; j < n; ++j) v += A[i * n + j] * x[j];
      dnu[i] = v;  // r_e
      feas = fmax(feas, fabs(v));
    }
    if (!finite || !(feas == feas)) { st = OH_STATUS_NUMERICAL; break; }
    if (stat <= Q.tol && feas <= Q.tol && gap <= Q.tol) { st = OH_STATUS_CONVERGED; break; }
    if (it == Q.max_iter) break;
    // H = 2P + M^T diag(lam/s) M (+ tiny shift), rhs = -rd + M^T [(mu/s - lam) - (lam/s) r_p]
    double dmax = 0.0;
    for (int k = 0; k < m; ++k) dl[k] = lam[k] / s[k];  // (dl is free until the step: one division per row instead of one per use)
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j <= i; ++j) {
        double v = P[i * n + j] + P[j * n + i];
        for (int k = 0; k < m; ++k) v += M[k * n + i] * dl[k] * M[k * n + j];
        H[i * n + j] = v;
      }
      dmax = fmax(dmax, fabs(H[i * n + i]));
      double r = -rd[i];
      for (int k = 0; k < m; ++k) r += M[k * n + i] * ((mu / s[k] - lam[k]) - dl[k] * ds[k]);
      rhs[i] = r;
    }
    double shift = 1e-13 * fmax(dmax, 1.0);
    bool ok = false;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
      if (attempt > 0) {  // rebuild with a larger shift (singular P without enough active rows)
        for (int i = 0; i < n; ++i)
          for (int j = 0; j <= i; ++j) {
            double v = P[i * n + j] + P[j * n + i];
            for (int k = 0; k < m; ++k) v += M[k * n + i] * dl[k] * M[k * n + j];
            H[i * n + j] = v;
          }
        shift *= 1e3;
      }
      for (int i = 0; i < n; ++i) H[i * n + i] += shift;
      ok = qp_chol(H, n);
    }
    if (!ok) { st = OH_STATUS_NUMERICAL; break; }
    for (int i = 0; i < n; ++i) dx[i] = rhs[i];
    qp_solve_chol(H, n, dx);  // H^{-1} rhs
    if (me > 0) {
      // A dx = -r_e with dx = H^{-1}(rhs + A^T dnu):  (A H^{-1} A^T) dnu = -r_e - A H^{-1} rhs
      for (int i = 0; i < me; ++i) {
        for (int j = 0; j < n; ++j) Y[i * n + j] = A[i * n + j];
        qp_solve_chol(H, n, Y.at(i * n));
      }
      for (int i = 0; i < me; ++i) {
        double r = -dnu[i];
        for (int j = 0; j < n; ++j) r -= A[i * n + j] * dx[j];
        for (int k = 0; k <= i; ++k) {
          double v = 0.0;
          for (int j = 0; j < n; ++j) v += A[i * n + j] * Y[k * n + j];
          S[i * me + k] = v;
        }
        S[i * me + i] += 1e-14 * fmax(1.0, S[i * me + i]);
        dnu[i] = r;
      }
      if (!qp_chol(S, me)) { st = OH_STATUS_NUMERICAL; break; }
      qp_solve_chol(S, me, dnu);
      for (int i = 0; i < me; ++i)
        for (int j = 0; j < n; ++j) dx[j] += Y[i * n + j] * dnu[i];
    }
    // ds = M dx + r_p ; dlam = (mu/s - lam) - (lam/s) ds ; fraction to the boundary
    double ap = 1.0, ad = 1.0;
    for (int i = 0; i < m; ++i) {
      double v = ds[i];
      for (int j = 0; j < n; ++j) v += M[i * n + j] * dx[j];
      const double d2 = (mu / s[i] - lam[i]) - dl[i] * v;
      ds[i] = v;
      dl[i] = d2;
      if (v < 0.0) ap = fmin(ap, -0.995 * s[i] / v);
      if (d2 < 0.0) ad = fmin(ad, -0.995 * lam[i] / d2);
    }
    for (int i = 0; i < n; ++i) x[i] += ap * dx[i];
    double comp = 0.0;
    for (int i = 0; i < m; ++i) {
      s[i] += ap * ds[i];
      lam[i] += ad * dl[i];
      comp += s[i] * lam[i];
    }
    for (int i = 0; i < me; ++i) nu[i] += ad * dnu[i];
    if (m > 0) {
      const double am = fmin(ap, ad);
      const double sigma = (am > 0.9) ? 0.1 : ((am > 0.5) ? 0.3 : 0.8);
      mu = fmax(sigma * comp / m, 1e-2 * Q.tol);
    }
  }
  double fval = 0.0;
  for (int i = 0; i < n; ++i) {
    double v = q[i];
    for (int j = 0; j < n; ++j) v += P[i * n + j] * x[j];
    fval += v * x[i];
    if (xo) xo[(size_t)b * n + i] = x[i];
  }
  if (fo) fo[b] = fval;
  if (kkt) { kkt[3 * (size_t)b] = stat; kkt[3 * (size_t)b + 1] = feas; kkt[3 * (size_t)b + 2] = gap; }
  if (iters) iters[b] = it;
  if (status) status[b] = st;
  if (mult) {
    for (int i = 0; i < m; ++i) mult[(size_t)b * (m + me) + i] = lam[i];
    for (int i = 0; i < me; ++i) mult[(size_t)b * (m + me) + m + i] = nu[i];
  }
}

// The same iteration for a FEW instances (a controller's tick is one): one wavefront per instance, everything of the instance in LDS, the loops
// over rows, columns and matrix entries spread over the lanes, the short serial parts (Cholesky factor, substitutions, the Schur complement of
// the equality rows) executed by all lanes alike.  One thread per instance is a chain of ~3000 dependent LDS accesses per iteration (80 us);
// here an iteration is a few hundred.  Sums over lanes associate differently than the thread's loop: results agree to rounding, not bit for bit.
__device__ double qp_wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ double qp_wave_max(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}
__device__ double qp_wave_min(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
  return v;
}
__global__ __launch_bounds__(64) void k_qp_solve_wave(QpParams Q, int B, const double* __restrict__ x0, const double* __restrict__ par, double* __restrict__ xo,
                                                      double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters, int* __restrict__ status,
                                                      double* __restrict__ mult) {
  extern __shared__ double qp_sm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = Q.n, m = Q.m, me = Q.me;
  QArr w{qp_sm, 1};
  for (int k = lane; k < Q.np; k += 64) w[k] = par[(size_t)b * Q.np + k];
  const QArr P = w;                  // [n][n]
  const QArr q = P.at(n * n);        // [n]
  const QArr M = q.at(n);            // [m][n]
  const QArr c = M.at(m * n);        // [m]
  const QArr A = c.at(m);            // [me][n]
  const QArr bv = A.at(me * n);      // [me]
  w = w.at(Q.np);
  const QArr x = w; w = w.at(n);
  const QArr s = w; w = w.at(m);
  const QArr lam = w; w = w.at(m);
  const QArr nu = w; w = w.at(me);
  const QArr H = w; w = w.at(n * n);
  const QArr rhs = w; w = w.at(n);
  const QArr dx = w; w = w.at(n);
  const QArr ds = w; w = w.at(m);
  const QArr dl = w; w = w.at(m);
  const QArr Y = w; w = w.at(me * n);
  const QArr S = w; w = w.at(me * me);
  const QArr dnu = w; w = w.at(me);
  const QArr rd = w; w = w.at(n);
  for (int i = lane; i < n; i += 64) x[i] = x0[(size_t)b * n + i];
  for (int i = lane; i < me; i += 64) nu[i] = 0.0;
  __syncthreads();
  double mu = 1.0;
  for (int i = lane; i < m; i += 64) {
    double v = c[i];
    for (int j = 0; j < n; ++j) v += M[i * n + j] * x[j];
    s[i] = fmax(v, 1.0);
    lam[i] = mu / s[i];
  }
  __syncthreads();
  int st = OH_STATUS_MAX_ITER, it = 0;
  double stat = 0.0, feas = 0.0, gap = 0.0;
  for (; it <= Q.max_iter; ++it) {
    stat = 0.0; feas = 0.0; gap = 0.0;
    bool finite = true;
    for (int i = lane; i < n; i += 64) {
      double v = q[i];
      for (int j = 0; j < n; ++j) v += 2.0 * P[i * n + j] * x[j];
      for (int k = 0; k < m; ++k) v -= M[k * n + i] * lam[k];
      for (int k = 0; k < me; ++k) v -= A[k * n + i] * nu[k];
      rd[i] = v;
      stat = fmax(stat, fabs(v));
      finite = finite && (v == v) && (fabs(v) < 1e300);
    }
    for (int i = lane; i < m; i += 64) {
      double v = c[i] - s[i];
      for (int j = 0; j < n; ++j) v += M[i * n + j] * x[j];
      ds[i] = v;  // r_p
      feas = fmax(feas, fabs(v));
      gap = fmax(gap, s[i] * lam[i]);
      dl[i] = lam[i] / s[i];
    }
    for (int i = lane; i < me; i += 64) {
      double v = bv[i];
      for (int j = 0; j < n; ++j) v += A[i * n + j] * x[j];
      dnu[i] = v;  // r_e
      feas = fmax(feas, fabs(v));
    }
    finite = __all(finite);
    const bool feas_nan = __any(!(feas == feas));
    stat = qp_wave_max(stat);
    feas = qp_wave_max(feas);
    gap = qp_wave_max(gap);
    __syncthreads();
    if (!finite || feas_nan) { st = OH_STATUS_NUMERICAL; break; }
    if (stat <= Q.tol && feas <= Q.tol && gap <= Q.tol) { st = OH_STATUS_CONVERGED; break; }
    if (it == Q.max_iter) break;
    // H = 2P + M^T diag(lam/s) M, entry (i, j <= i) per lane; rhs = -rd + M^T [(mu/s - lam) - (lam/s) r_p]
    const int nh = n * (n + 1) / 2;
    auto build = [&]() {
      for (int e = lane; e < nh; e += 64) {
        int i = 0, r = e;
        while (r > i) { r -= i + 1; ++i; }  // e = i (i + 1) / 2 + j
        const int j = r;
        double v = P[i * n + j] + P[j * n + i];
        for (int k = 0; k < m; ++k) v += M[k * n + i] * dl[k] * M[k * n + j];
        H[i * n + j] = v;
      }
    };
    build();
    for (int i = lane; i < n; i += 64) {
      double r = -rd[i];
      for (int k = 0; k < m; ++k) r += M[k * n + i] * ((mu / s[k] - lam[k]) - dl[k] * ds[k]);
      rhs[i] = r;
    }
    __syncthreads();
    double dmax = 0.0;
    for (int i = 0; i < n; ++i) dmax = fmax(dmax, fabs(H[i * n + i]));
    double shift = 1e-13 * fmax(dmax, 1.0);
    bool ok = false;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
      if (attempt > 0) {  // rebuild with a larger shift (singular P without enough active rows)
        __syncthreads();
        build();
        shift *= 1e3;
        __syncthreads();
      }
      for (int i = lane; i < n; i += 64) H[i * n + i] += shift;
      __syncthreads();
      // left-looking Cholesky: the pivot of column j by every lane alike, the rows below it one per lane
      ok = true;
      for (int j = 0; j < n && ok; ++j) {
        double d = H[j * n + j];
        for (int k = 0; k < j; ++k) d -= H[j * n + k] * H[j * n + k];
        if (!(d > 0.0)) { ok = false; break; }
        const double l = sqrt(d);
        double vmine = 0.0;
        const int i = j + 1 + lane;
        if (i < n) {
          vmine = H[i * n + j];
          for (int k = 0; k < j; ++k) vmine -= H[i * n + k] * H[j * n + k];
        }
        __syncthreads();
        if (lane == 0) H[j * n + j] = l;
        if (i < n) H[i * n + j] = vmine / l;
        __syncthreads();
      }
    }
    if (!ok) { st = OH_STATUS_NUMERICAL; break; }
    // the substitutions and the equality rows: short, serial, by all lanes alike (the same values land in the same places)
    for (int i = lane; i < n; i += 64) dx[i] = rhs[i];
    __syncthreads();
    if (lane == 0) qp_solve_chol(H, n, dx);  // H^{-1} rhs
    __syncthreads();
    if (me > 0) {
      for (int i = lane; i < me; i += 64) {  // row i of Y = H^{-1} A_i: one lane per equality row
        for (int j = 0; j < n; ++j) Y[i * n + j] = A[i * n + j];
        qp_solve_chol(H, n, Y.at(i * n));
      }
      __syncthreads();
      for (int i = lane; i < me; i += 64) {
        double r = -dnu[i];
        for (int j = 0; j < n; ++j) r -= A[i * n + j] * dx[j];
        for (int k = 0; k <= i; ++k) {
          double v = 0.0;
          for (int j = 0; j < n; ++j) v += A[i * n + j] * Y[k * n + j];
          S[i * me + k] = v;
        }
        S[i * me + i] += 1e-14 * fmax(1.0, S[i * me + i]);
        dnu[i] = r;
      }
      __syncthreads();
      bool oks = true;
      if (lane == 0) {
        oks = qp_chol(S, me);
        if (oks) {
          qp_solve_chol(S, me, dnu);
          for (int i = 0; i < me; ++i)
            for (int j = 0; j < n; ++j) dx[j] += Y[i * n + j] * dnu[i];
        }
      }
      oks = __all(oks);
      __syncthreads();
      if (!oks) { st = OH_STATUS_NUMERICAL; break; }
    }
    // ds = M dx + r_p ; dlam = (mu/s - lam) - (lam/s) ds ; fraction to the boundary
    double ap = 1.0, ad = 1.0;
    for (int i = lane; i < m; i += 64) {
      double v = ds[i];
      for (int j = 0; j < n; ++j) v += M[i * n + j] * dx[j];
      const double d2 = (mu / s[i] - lam[i]) - dl[i] * v;
      ds[i] = v;
      dl[i] = d2;
      if (v < 0.0) ap = fmin(ap, -0.995 * s[i] / v);
      if (d2 < 0.0) ad = fmin(ad, -0.995 * lam[i] / d2);
    }
    ap = qp_wave_min(ap);
    ad = qp_wave_min(ad);
    for (int i = lane; i < n; i += 64) x[i] += ap * dx[i];
    double comp = 0.0;
    for (int i = lane; i < m; i += 64) {
      s[i] += ap * ds[i];
      lam[i] += ad * dl[i];
      comp += s[i] * lam[i];
    }
    comp = qp_wave_sum(comp);
    for (int i = lane; i < me; i += 64) nu[i] += ad * dnu[i];
    if (m > 0) {
      const double am = fmin(ap, ad);
      const double sigma = (am > 0.9) ? 0.1 : ((am > 0.5) ? 0.3 : 0.8);
      mu = fmax(sigma * comp / m, 1e-2 * Q.tol);
    }
    __syncthreads();
  }
  double fval = 0.0;
  for (int i = lane; i < n; i += 64) {
    double v = q[i];
    for (int j = 0; j < n; ++j) v += P[i * n + j] * x[j];
    fval += v * x[i];
    if (xo) xo[(size_t)b * n + i] = x[i];
  }
  fval = qp_wave_sum(fval);
  if (lane == 0) {
    if (fo) fo[b] = fval;
    if (kkt) { kkt[3 * (size_t)b] = stat; kkt[3 * (size_t)b + 1] = feas; kkt[3 * (size_t)b + 2] = gap; }
    if (iters) iters[b] = it;
    if (status) status[b] = st;
  }
  if (mult) {
    for (int i = lane; i < m; i += 64) mult[(size_t)b * (m + me) + i] = lam[i];
    for (int i = lane; i < me; i += 64) mult[(size_t)b * (m + me) + m + i] = nu[i];
  }
}

// ---- QP data read off the problem's instruction tape on the device (oh_qp_set_tape) -----------------------------------------------------
// The reference's QuadraticCost* classes hold P, q, M, c, A, b as cs.Functions of the parameters (optimization.py:219-260); the mirror reads
// them off f, k, a by probing on the host (optas_amd/optimization.py: values at 0, +-e_i, e_i + e_j -- exact for a quadratic cost and affine
// rows), 36 tree evaluations per instance for n = 7: 1.8 ms.  Here one thread per instance runs the same probes through the problem's
// instruction tape (optas_amd/tape.py; the interpreter of oh_tape.hip, forward sweep only, registers val[i][b] in a global work array) and
// leaves the [P | q | M | c | A | b] row k_qp_solve reads, plus the cost's constant term f(0, p).  Same formulas in the same order as the host.
struct QpTape {
  int len, out_cost;
  const int* __restrict__ op;
  const int* __restrict__ a;
  const int* __restrict__ bb;
  const double* __restrict__ c;
  const int* __restrict__ rows;
  int n_xdep;                     // instructions whose value depends on x (the kinematics of a velocity-IK problem depend on p only:
  const int* __restrict__ xdep;   // 139 of its 474 instructions are left), in tape order
};
#define QIDX(i) ((size_t)(i) * Bp + b)
// FULL: every instruction; else only the x-dependent ones (the others keep the values of an earlier full sweep at the same p)
template <bool FULL>
__device__ double qp_tape_forward(const QpTape& tp, const double* xs, const double* __restrict__ pb, double* __restrict__ val, const int Bp, const int b) {
#pragma clang fp contract(off)
  const int cnt = FULL ? tp.len : tp.n_xdep;
  for (int k = 0; k < cnt; ++k) {
    const int i = FULL ? k : tp.xdep[k];
    const int o = tp.op[i], ia = tp.a[i], ib = tp.bb[i];
    double v;
    switch (o) {
      case 0: v = tp.c[i]; break;
      case 1: v = xs[ia]; break;
      case 2: v = pb[ia]; break;
      case 3: v = val[QIDX(ia)] + val[QIDX(ib)]; break;
      case 4: v = val[QIDX(ia)] - val[QIDX(ib)]; break;
      case 5: v = val[QIDX(ia)] * val[QIDX(ib)]; break;
      default: v = tape_op_value(o, val[QIDX(ia)], tape_op_arity(o) == 2 ? val[QIDX(ib)] : 0.0); break;  // 6 .. 26 (coefficients may be any function of p)
    }
    val[QIDX(i)] = v;
  }
  return val[QIDX(tp.out_cost)];
}
__global__ __launch_bounds__(64) void k_qp_assemble(QpParams Q, QpTape tp, int np_raw, int B, int Bp, const double* __restrict__ par, double* __restrict__ val,
                                                    double* __restrict__ rows_out, double* __restrict__ f0_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = Q.n, m = Q.m, me = Q.me;
  const double* pb = par + (size_t)b * np_raw;
  double* P = rows_out + (size_t)b * Q.np;
  double* q = P + n * n;
  double* M = q + n;
  double* c = M + m * n;
  double* A = c + m;
  double* bv = A + me * n;
  double x[OH_QP_MAX_N], f1[OH_QP_MAX_N];
  for (int i = 0; i < n; ++i) x[i] = 0.0;
  const double f0 = qp_tape_forward<true>(tp, x, pb, val, Bp, b);
  for (int r = 0; r < m; ++r) c[r] = val[QIDX(tp.rows[r])];
  for (int r = 0; r < me; ++r) bv[r] = val[QIDX(tp.rows[m + r])];
  for (int i = 0; i < n; ++i) {
    x[i] = 1.0;
    f1[i] = qp_tape_forward<false>(tp, x, pb, val, Bp, b);
    for (int r = 0; r < m; ++r) M[r * n + i] = val[QIDX(tp.rows[r])] - c[r];
    for (int r = 0; r < me; ++r) A[r * n + i] = val[QIDX(tp.rows[m + r])] - bv[r];
    x[i] = -1.0;
    const double fm = qp_tape_forward<false>(tp, x, pb, val, Bp, b);
    x[i] = 0.0;
    q[i] = 0.5 * (f1[i] - fm);
  }
  for (int i = 0; i < n; ++i) {
    P[i * n + i] = f1[i] - f0 - q[i];
    for (int j = 0; j < i; ++j) {
      x[i] = x[j] = 1.0;
      const double fij = qp_tape_forward<false>(tp, x, pb, val, Bp, b);
      x[i] = x[j] = 0.0;
      P[i * n + j] = P[j * n + i] = 0.5 * (fij - f1[i] - f1[j] + f0);
    }
  }
  f0_out[b] = f0;
}
// The same for a few instances: one BLOCK per instance, one lane per probe point (1 + 2 n + n (n - 1) / 2 of them, 64 at a time), registers of
// lane l of instance b at val[(i * B + b) * 64 + l].  A tick of a velocity-IK controller is one instance: 36 probes side by side instead of one
// after the other (the sweep is a chain of dependent memory round trips either way).
__device__ void qp_probe_point(const int pid, const int n, double* x, int* pi, int* pj) {  // x of probe pid; (i, j) of a pair probe, else (-1, -1)
  for (int k = 0; k < n; ++k) x[k] = 0.0;
  *pi = *pj = -1;
  if (pid == 0) return;
  if (pid <= n) { x[pid - 1] = 1.0; return; }
  if (pid <= 2 * n) { x[pid - 1 - n] = -1.0; return; }
  int k = pid - 1 - 2 * n, i = 1;
  while (k >= i) { k -= i; ++i; }  // pair k of row i: (i, j = k), j < i
  x[i] = x[k] = 1.0;
  *pi = i;
  *pj = k;
}
__global__ __launch_bounds__(64) void k_qp_assemble_par(QpParams Q, QpTape tp, int np_raw, int B, const double* __restrict__ par, double* __restrict__ val,
                                                        double* __restrict__ rows_out, double* __restrict__ f0_out) {
  __shared__ double fs[1 + 2 * OH_QP_MAX_N + OH_QP_MAX_N * (OH_QP_MAX_N - 1) / 2];
  const int inst = blockIdx.x, lane = threadIdx.x;
  const int n = Q.n, m = Q.m, me = Q.me;
  const int n_probe = 1 + 2 * n + n * (n - 1) / 2;
  const double* pb = par + (size_t)inst * np_raw;
  double* P = rows_out + (size_t)inst * Q.np;
  double* q = P + n * n;
  double* M = q + n;
  double* c = M + m * n;
  double* A = c + m;
  double* bv = A + me * n;
  const int Bp = B * 64, b = inst * 64 + lane;  // QIDX(i) = (i * B + inst) * 64 + lane
  double x[OH_QP_MAX_N];
  for (int base = 0; base < n_probe; base += 64) {
    const int pid = base + lane;
    const bool live = pid < n_probe;
    int pi, pj;
    qp_probe_point(live ? pid : 0, n, x, &pi, &pj);
    const double f = qp_tape_forward<true>(tp, x, pb, val, Bp, b);
    if (live) fs[pid] = f;
    if (base == 0) {  // the rows at 0 first, then the columns of M and A relative to them
      if (lane == 0) {
        for (int r = 0; r < m; ++r) c[r] = val[QIDX(tp.rows[r])];
        for (int r = 0; r < me; ++r) bv[r] = val[QIDX(tp.rows[m + r])];
      }
      __syncthreads();
    }
    if (live && pid >= 1 && pid <= n) {
      const int i = pid - 1;
      for (int r = 0; r < m; ++r) M[r * n + i] = val[QIDX(tp.rows[r])] - c[r];
      for (int r = 0; r < me; ++r) A[r * n + i] = val[QIDX(tp.rows[m + r])] - bv[r];
    }
  }
  __syncthreads();
  const double f0 = fs[0];
  for (int i = lane; i < n; i += 64) {
    const double qi = 0.5 * (fs[1 + i] - fs[1 + n + i]);
    q[i] = qi;
    P[i * n + i] = fs[1 + i] - f0 - qi;
  }
  for (int k = lane; k < n * (n - 1) / 2; k += 64) {
    int kk = k, i = 1;
    while (kk >= i) { kk -= i; ++i; }
    P[i * n + kk] = P[kk * n + i] = 0.5 * (fs[1 + 2 * n + k] - fs[1 + i] - fs[1 + kk] + f0);
  }
  if (lane == 0) f0_out[inst] = f0;
}
__global__ __launch_bounds__(256) void k_qp_add_constant(int B, double* __restrict__ f, const double* __restrict__ f0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) f[b] += f0[b];
}

}  // namespace

void oh_launch_qp_assemble(hipStream_t s, const QpParams& Q, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows,
                           const int* xdep, int n_xdep, int B, int Bp, const double* p_raw, double* val, double* rows_out, double* f0) {
  const QpTape tp{T.len, T.out_cost, op, a, b, c, rows, n_xdep, xdep};
  if ((size_t)B * 64 <= (size_t)Bp) hipLaunchKernelGGL(k_qp_assemble_par, dim3(B), dim3(64), 0, s, Q, tp, T.np, B, p_raw, val, rows_out, f0);  // room for a lane per probe
  else hipLaunchKernelGGL(k_qp_assemble, dim3((B + 63) / 64), dim3(64), 0, s, Q, tp, T.np, B, Bp, p_raw, val, rows_out, f0);
}
void oh_launch_qp_add_constant(hipStream_t s, int B, double* f, const double* f0) {
  hipLaunchKernelGGL(k_qp_add_constant, dim3((B + 255) / 256), dim3(256), 0, s, B, f, f0);
}

void oh_launch_qp_solve(hipStream_t s, const QpParams& Q, int B, int Bp, const double* x0, const double* p, double* work, double* x, double* f, double* kkt,
                        int* iters, int* status, double* mult) {
  // the work set of a block in LDS when it fits 48 KB at 64, 32 or 16 instances per block; for a few instances the problem row as well
  auto fit = [](const size_t doubles) {
    for (int c : {64, 32, 16})
      if (sizeof(double) * doubles * c <= 48 * 1024) return c;
    return 0;
  };
  const int forced = oh_launch_opts().qp_mode;  // option "qp_mode" (experiments)
  const int bs2 = fit((size_t)Q.nwork + Q.np), bs1 = fit((size_t)Q.nwork);
  const size_t wave_bytes = sizeof(double) * ((size_t)Q.nwork + Q.np);
  if (forced != 0 && forced != 1 && forced != 2 && B <= 64 && wave_bytes <= 48 * 1024) {  // a few instances: one wavefront each
    hipLaunchKernelGGL(k_qp_solve_wave, dim3(B), dim3(64), wave_bytes, s, Q, B, x0, p, x, f, kkt, iters, status, mult);
    return;
  }
  int mode = bs2 ? 2 : (bs1 ? 1 : 0);  // (velocity-IK QP, n = 7, m = 16: B = 1 in 1.77 / 1.49 / 1.15 ms wall, 65 536 in 13.0 / 12.7 / 11.5 ms for modes 0 / 1 / 2)
  if (forced == 0 || (forced == 1 && bs1) || (forced == 2 && bs2)) mode = forced;
  if (mode == 2)
    hipLaunchKernelGGL(k_qp_solve<2>, dim3((B + bs2 - 1) / bs2), dim3(bs2), sizeof(double) * ((size_t)Q.nwork + Q.np) * bs2, s, Q, B, Bp, x0, p, work, x, f, kkt, iters,
                       status, mult);
  else if (mode == 1)
    hipLaunchKernelGGL(k_qp_solve<1>, dim3((B + bs1 - 1) / bs1), dim3(bs1), sizeof(double) * (size_t)Q.nwork * bs1, s, Q, B, Bp, x0, p, work, x, f, kkt, iters, status, mult);
  else hipLaunchKernelGGL(k_qp_solve<0>, dim3((B + 63) / 64), dim3(64), 0, s, Q, B, Bp, x0, p, work, x, f, kkt, iters, status, mult);
}
