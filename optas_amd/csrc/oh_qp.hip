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

#include "oh_kernels.h"

namespace {

// in-place Cholesky of the n x n row-major SPD matrix H (lower triangle); returns false on a non-positive pivot
__device__ bool qp_chol(double* H, const int n) {
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
__device__ void qp_solve_chol(const double* L, const int n, double* x) {  // x <- (L L^T)^{-1} x
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

__global__ __launch_bounds__(64) void k_qp_solve(QpParams Q, int B, const double* __restrict__ x0, const double* __restrict__ par, double* __restrict__ work,
                                                 double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters,
                                                 int* __restrict__ status, double* __restrict__ mult) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = Q.n, m = Q.m, me = Q.me;
  const double* pb = par + (size_t)b * Q.np;
  const double* P = pb;              // [n][n]
  const double* q = P + n * n;       // [n]
  const double* M = q + n;           // [m][n]
  const double* c = M + m * n;       // [m]
  const double* A = c + m;           // [me][n]
  const double* bv = A + me * n;     // [me]
  double* w = work + (size_t)b * Q.nwork;
  double* x = w; w += n;
  double* s = w; w += m;
  double* lam = w; w += m;
  double* nu = w; w += me;
  double* H = w; w += n * n;
  double* rhs = w; w += n;
  double* dx = w; w += n;
  double* ds = w; w += m;
  double* dl = w; w += m;
  double* Y = w; w += me * n;   // H^{-1} A^T, row i = H^{-1} A_i
  double* S = w; w += me * me;
  double* dnu = w; w += me;
  double* rd = w; w += n;
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
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j <= i; ++j) {
        double v = P[i * n + j] + P[j * n + i];
        for (int k = 0; k < m; ++k) v += M[k * n + i] * (lam[k] / s[k]) * M[k * n + j];
        H[i * n + j] = v;
      }
      dmax = fmax(dmax, fabs(H[i * n + i]));
      double r = -rd[i];
      for (int k = 0; k < m; ++k) r += M[k * n + i] * ((mu / s[k] - lam[k]) - (lam[k] / s[k]) * ds[k]);
      rhs[i] = r;
    }
    double shift = 1e-13 * fmax(dmax, 1.0);
    bool ok = false;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
      if (attempt > 0) {  // rebuild with a larger shift (singular P without enough active rows)
        for (int i = 0; i < n; ++i)
          for (int j = 0; j <= i; ++j) {
            double v = P[i * n + j] + P[j * n + i];
            for (int k = 0; k < m; ++k) v += M[k * n + i] * (lam[k] / s[k]) * M[k * n + j];
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
        qp_solve_chol(H, n, Y + i * n);
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
      const double d2 = (mu / s[i] - lam[i]) - (lam[i] / s[i]) * v;
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

}  // namespace

void oh_launch_qp_solve(hipStream_t s, const QpParams& Q, int B, const double* x0, const double* p, double* work, double* x, double* f, double* kkt,
                        int* iters, int* status, double* mult) {
  hipLaunchKernelGGL(k_qp_solve, dim3((B + 63) / 64), dim3(64), 0, s, Q, B, x0, p, work, x, f, kkt, iters, status, mult);
}
