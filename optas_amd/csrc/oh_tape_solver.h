// Solver of the generic tape family, shared by the two ways a tape is evaluated on the GPU:
//   * oh_tape.hip compiles it ahead of time around an interpreter of the instruction arrays (registers in HBM/L2);
//   * oh_tape_jit.hip hands this very text to hiprtc in front of straight-line code generated from the tape (registers in VGPRs),
//     the counterpart of CasADi's "jit" code generation for its SX virtual machine.
// Self-contained on purpose (no #include, only builtins and the device math library): build.py embeds the file as a string.
// The includer defines OH_TAPE_ST_CONVERGED / OH_TAPE_ST_MAX_ITER / OH_TAPE_ST_NUMERICAL (the oh_status values of include/optas_hip.h).
//
// Outer loop: Powell-Hestenes-Rockafellar augmented Lagrangian; inner solver: BFGS on the inverse Hessian with Armijo backtracking -- the
// dense n x n matrix for small problems (T.lbfgs == 0), the limited-memory form (Nocedal's two-loop recursion over the last T.lbfgs
// (s, y) pairs, 2 m n doubles instead of n^2) for the trajectory-sized ones (round 3: nx up to OH_TAPE_MAX_N = 4096).
// numpy restatement: oracle/tape_ref.py (solve_tape_al).
#ifndef OH_TAPE_SOLVER_H
#define OH_TAPE_SOLVER_H

struct TapeParams {
  int len, nx, np, n_ineq, n_eq, out_cost, max_iter;
  double tol, tol_feas, rho0;
  int lbfgs;  // 0: dense inverse-Hessian BFGS; m > 0: limited-memory BFGS with m pairs
  // Initial metric of the limited-memory form (oh_tape_set_metric): a symmetric positive definite [nx][nx] matrix H0 in device memory, shared by
  // every instance of the handle -- the two-loop recursion takes r = H0 q where it otherwise scales q by s.y / y.y of the newest pair.  The host
  // hands over the inverse of the constant part of the cost's Hessian (tape.py:quadratic_cost_metric): what the pairs then have to learn is the
  // curvature of the rows alone.  nullptr: the scaled identity.
  const double* h0;
};

#define TIDX(i) ((size_t)(i) * Bp + b)

struct TapeWork {  // SoA slices [k][Bp]: instance index fastest, every access of a wavefront is one coalesced line
  double *x, *xt, *g, *gt, *d, *H, *lam, *mu, *s, *hy, *rowv;
};

__host__ __device__ inline size_t tape_solver_rows(const TapeParams& T) {
  const size_t n = T.nx;
  const size_t hrows = T.lbfgs > 0 ? 2 * (size_t)T.lbfgs * n + 2 * (size_t)T.lbfgs : n * n;  // (s, y) pairs + their 1 / s.y and the loop's alphas
  return 7 * n + hrows + 2 * (size_t)(T.n_ineq > 0 ? T.n_ineq : 1) + 2 * (size_t)(T.n_eq > 0 ? T.n_eq : 1);
}

__device__ inline TapeWork tape_carve(const TapeParams& T, double* w, const int Bp) {
  const size_t n = T.nx, ni = T.n_ineq > 0 ? T.n_ineq : 1, ne = T.n_eq > 0 ? T.n_eq : 1;
  TapeWork W;
  auto take = [&](size_t rows) { double* o = w; w += rows * (size_t)Bp; return o; };
  W.x = take(n); W.xt = take(n); W.g = take(n); W.gt = take(n); W.d = take(n); W.s = take(n); W.hy = take(n);
  W.H = take(T.lbfgs > 0 ? 2 * (size_t)T.lbfgs * n + 2 * (size_t)T.lbfgs : n * n); W.lam = take(ni); W.mu = take(ne); W.rowv = take(ni + ne);
  return W;
}

// PHR terms of one row: add the row's share of the merit and of the two constraint measures, return the seed of the reverse sweep
// Value of one tape instruction with operand values va, vb (ops 6 .. 26 of include/optas_hip.h; CONST / X / P / ADD / SUB / MUL are the callers' business).
// One IEEE operation or one libm call per instruction, as the numpy restatement and casadi's SX machine compute them.
__device__ inline double tape_op_value(const int o, const double va, const double vb) {
  switch (o) {
    case 6: return va / vb;
    case 7: return -va;
    case 8: return sin(va);
    case 9: return cos(va);
    case 10: return atan2(va, vb);
    case 11: return sqrt(va);
    case 12: return va * va;
    case 13: return asin(va);
    case 14: return fabs(va);
    case 15: return fmin(va, vb);
    case 16: return fmax(va, vb);
    case 17: return va < vb ? 1.0 : 0.0;
    case 18: return va <= vb ? 1.0 : 0.0;
    case 19: return va == vb ? 1.0 : 0.0;
    case 20: return va != vb ? 1.0 : 0.0;
    case 21: return va == 0.0 ? 1.0 : 0.0;
    case 22: return (va != 0.0 && vb != 0.0) ? 1.0 : 0.0;
    case 23: return (va != 0.0 || vb != 0.0) ? 1.0 : 0.0;
    case 24: return va != 0.0 ? vb : 0.0;
    case 25: return exp(va);
    default: return log(va);
  }
}
// operands an instruction reads: 0 (CONST, X, P), 1 or 2
__host__ __device__ inline int tape_op_arity(const int o) {
  if (o <= 2) return 0;
  return ((o >= 3 && o <= 6) || o == 10 || (o >= 15 && o <= 20) || (o >= 22 && o <= 24)) ? 2 : 1;
}
__device__ inline double tape_al_ineq(const double g, const double lam, const double rho, double& val, double& cm, double& ms) {
  const double s = fmax(0.0, lam - rho * g);
  val += (s * s - lam * lam) / (2.0 * rho);
  cm = fmax(cm, fmax(0.0, -g));
  ms = fmax(ms, fabs(fmin(g, lam / rho)));
  return -s;
}
__device__ inline double tape_al_eq(const double c, const double mu, const double rho, double& val, double& cm, double& ms) {
  val += -mu * c + 0.5 * rho * c * c;
  cm = fmax(cm, fabs(c));
  ms = fmax(ms, fabs(c));
  return -mu + rho * c;
}

// E::phi(xs, gout, rho, &f, &cmax, &meas): merit value at the point in xs (SoA), its gradient into gout (SoA), the row values into W.rowv
template <class E>
// (Bp, b) address the work arrays (TIDX: [row][lane], in the global buffer or in LDS); gb is the instance's index in the batch
__device__ inline void tape_solve_instance(const TapeParams& T, E& ev, const TapeWork& W, const int Bp, const int b, const int gb, const double* __restrict__ x0,
                                           double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters,
                                           int* __restrict__ status, double* __restrict__ mult) {
  const int n = T.nx;
  for (int k = 0; k < n; ++k) W.x[TIDX(k)] = x0[(size_t)gb * n + k];
  for (int i = 0; i < T.n_ineq; ++i) W.lam[TIDX(i)] = 0.0;
  for (int i = 0; i < T.n_eq; ++i) W.mu[TIDX(i)] = 0.0;
  // limited-memory form: W.H holds S [m][n], then Y [m][n], then 1 / (s_i . y_i) [m], then the two-loop alphas [m]; pair j of the `hist` stored
  // ones (oldest first) sits in ring slot (head - hist + j) mod m
  const int m = T.lbfgs;
  const bool metric = T.h0 != nullptr && m > 0;
  int hist = 0, head = 0;
  auto Sr = [&](int slot, int k) -> double& { return W.H[TIDX((size_t)slot * n + k)]; };
  auto Yr = [&](int slot, int k) -> double& { return W.H[TIDX((size_t)(m + slot) * n + k)]; };
  auto Rr = [&](int slot) -> double& { return W.H[TIDX((size_t)2 * m * n + slot)]; };
  auto Ar = [&](int slot) -> double& { return W.H[TIDX((size_t)2 * m * n + m + slot)]; };
  auto eye = [&]() {
    if (m > 0) { hist = 0; head = 0; return; }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) W.H[TIDX(i * n + j)] = (i == j) ? 1.0 : 0.0;
  };
  double rho = T.rho0, omega = fmax(T.tol, 1e-2), meas_prev = 1e300;
  double alpha_prev = 1.0;  // step length of the last accepted step (handles with a metric)
  double msum = 0.0;  // sum of the multipliers' magnitudes: scales what the merit resolves (see the line search)
  double fval, cmax, meas;
  double val = ev.phi(W.x, W.g, rho, &fval, &cmax, &meas);
  int evals = 1, st = OH_TAPE_ST_MAX_ITER;
  bool H_is_eye = true;
  eye();
  double stat = 0.0;
  for (;;) {
    stat = 0.0;
    for (int k = 0; k < n; ++k) stat = fmax(stat, fabs(W.g[TIDX(k)]));
    bool finite = (val == val) && (fabs(val) < 1e300);
    for (int k = 0; k < n; ++k) finite = finite && (W.g[TIDX(k)] == W.g[TIDX(k)]);
    if (!finite) { st = OH_TAPE_ST_NUMERICAL; break; }
    if (stat <= omega) {
      if (stat <= T.tol && meas <= T.tol_feas) { st = OH_TAPE_ST_CONVERGED; break; }
      if (evals >= T.max_iter) break;
      // outer iteration: multiplier update at x with the current penalty (rowv holds the rows of the last evaluation, which was at x)
      msum = 0.0;
      for (int i = 0; i < T.n_eq; ++i) {
        const double v = W.mu[TIDX(i)] - rho * W.rowv[TIDX(T.n_ineq + i)];
        W.mu[TIDX(i)] = v;
        msum += fabs(v);
      }
      for (int i = 0; i < T.n_ineq; ++i) {
        const double v = fmax(0.0, W.lam[TIDX(i)] - rho * W.rowv[TIDX(i)]);
        W.lam[TIDX(i)] = v;
        msum += v;
      }
      if (meas > 0.25 * meas_prev) {
        rho = fmin(rho * 10.0, 1e8);
        // with a metric of the handle the pairs hold nothing but the curvature of the rows under the OLD penalty: misleading by a factor of ten, and
        // cheap to measure again on top of H0 (planner, numpy port, 32 instances: stalls of hundreds of evaluations at the rounding floor without this)
        if (metric) { hist = 0; head = 0; H_is_eye = true; }
      }
      meas_prev = meas;
      omega = fmax(T.tol, fmin(omega, 0.1 * meas));
      val = ev.phi(W.x, W.g, rho, &fval, &cmax, &meas);
      ++evals;
      // the metric is kept (round 3): the multiplier update shifts the merit, its curvature -- cost + penalty of the rows in reach -- stays what
      // the pairs have measured.  Rebuilding it from the identity at every outer update cost 4 of every 5 evaluations on the 7-variable IK.
      continue;
    }
    if (evals >= T.max_iter) break;
    double slope = 0.0;  // d = -H g
    if (m > 0) {
      // two-loop recursion (Nocedal 1980): q = g; newest to oldest a_i = rho_i s_i.q, q -= a_i y_i; r = gamma q with gamma = s.y / y.y of the
      // newest pair; oldest to newest r += s_i (a_i - rho_i y_i.r); d = -r
      for (int k = 0; k < n; ++k) W.d[TIDX(k)] = W.g[TIDX(k)];
      for (int j = hist - 1; j >= 0; --j) {
        const int sl = ((head - hist + j) % m + m) % m;
        double sq = 0.0;
        for (int k = 0; k < n; ++k) sq += Sr(sl, k) * W.d[TIDX(k)];
        const double al = Rr(sl) * sq;
        Ar(sl) = al;
        for (int k = 0; k < n; ++k) W.d[TIDX(k)] -= al * Yr(sl, k);
      }
      if (metric) {  // r = H0 q (H0 symmetric: row k read as column k, the same matrix for every lane -> scalar loads); W.hy is free in this form
        for (int k = 0; k < n; ++k) {
          double acc = 0.0;
          for (int j = 0; j < n; ++j) acc += T.h0[(size_t)j * n + k] * W.d[TIDX(j)];
          W.hy[TIDX(k)] = acc;
        }
        for (int k = 0; k < n; ++k) W.d[TIDX(k)] = W.hy[TIDX(k)];
      } else if (hist > 0) {
        const int sl = ((head - 1) % m + m) % m;
        double sy = 0.0, yy = 0.0;
        for (int k = 0; k < n; ++k) { sy += Sr(sl, k) * Yr(sl, k); yy += Yr(sl, k) * Yr(sl, k); }
        const double gam = sy / yy;
        for (int k = 0; k < n; ++k) W.d[TIDX(k)] *= gam;
      }
      for (int j = 0; j < hist; ++j) {
        const int sl = ((head - hist + j) % m + m) % m;
        double yr = 0.0;
        for (int k = 0; k < n; ++k) yr += Yr(sl, k) * W.d[TIDX(k)];
        const double be = Ar(sl) - Rr(sl) * yr;
        for (int k = 0; k < n; ++k) W.d[TIDX(k)] += be * Sr(sl, k);
      }
      for (int k = 0; k < n; ++k) {
        const double v = -W.d[TIDX(k)];
        W.d[TIDX(k)] = v;
        slope += W.g[TIDX(k)] * v;
      }
    } else {
      for (int i = 0; i < n; ++i) {
        double v = 0.0;
        for (int j = 0; j < n; ++j) v -= W.H[TIDX(i * n + j)] * W.g[TIDX(j)];
        W.d[TIDX(i)] = v;
        slope += W.g[TIDX(i)] * v;
      }
    }
    if (!(slope < 0.0)) {
      eye();
      H_is_eye = true;
      slope = 0.0;
      for (int i = 0; i < n; ++i) { W.d[TIDX(i)] = -W.g[TIDX(i)]; slope -= W.g[TIDX(i)] * W.g[TIDX(i)]; }
    }
    // a fresh (identity) metric knows nothing about the scale of the problem: keep the first step within unit length
    double alpha = 1.0, vt = 0.0, ft = 0.0, ct = 0.0, mt = 0.0;
    if (H_is_eye) {
      double dmax = 0.0;
      for (int k = 0; k < n; ++k) dmax = fmax(dmax, fabs(W.d[TIDX(k)]));
      alpha = fmin(1.0, 1.0 / dmax);
    }
    // with a metric: early on its unit step is far too long across the rows' curvature (penalty 1e4) and every line search walked down from 1 again, 3-5 trials
    // a step; the first trial is now four times the last accepted fraction (planner, numpy port, 2 x 32 instances: 54 / 56 -> 40 / 41 evaluations, slowest 85 -> 63).
    // Fractions below 1e-4 are end-game steps at the rounding floor, not a scale to carry over.
    if (metric && alpha_prev >= 1e-4 && alpha_prev < 1.0) alpha = fmin(alpha, 4.0 * alpha_prev);
    bool ok = false;
    // what the merit resolves: its own rounding plus the rows' rounding (1e-16 of quantities of order one) times their multipliers -- under
    // multipliers of 30 the term -mu c moves by 3e-15 between two evaluations of the same point
    const double slack = 4e-16 * (fmax(1.0, fabs(val)) + msum);
    double gg = 0.0;
    for (int k = 0; k < n; ++k) gg += W.g[TIDX(k)] * W.g[TIDX(k)];
    for (int ls = 0; ls < 40; ++ls) {
      for (int k = 0; k < n; ++k) W.xt[TIDX(k)] = W.x[TIDX(k)] + alpha * W.d[TIDX(k)];
      vt = ev.phi(W.xt, W.gt, rho, &ft, &ct, &mt);
      ++evals;
      const double need = -1e-4 * alpha * slope;
      if (need > slack) {
        if ((vt == vt) && vt <= val - need + slack) { ok = true; break; }
      } else if ((vt == vt) && vt <= val - slack) {  // a decrease the merit does resolve, larger than the one asked for
        ok = true;
        break;
      } else if ((vt == vt) && vt <= val + slack) {
        // the decrease asked for is below that resolution (end game under a large penalty: a gradient of 4e-6 across a curvature of 1e4 is worth
        // 7e-16 of merit): the value cannot judge the step, the gradient can -- a step that keeps the merit within its rounding is taken if it
        // shrinks the gradient; one that leaves the gradient where it was is not a step (alpha -> 0 used to pass as one: 20 of 65 536 IK
        // instances walked in place until the evaluation cap)
        double ggt = 0.0;
        for (int k = 0; k < n; ++k) ggt += W.gt[TIDX(k)] * W.gt[TIDX(k)];
        // (strictly smaller: below alpha ~ 1e-12 the factor rounds to one and a trial that is x itself would pass as a step, for ever -- seen on the planner
        // under penalty 1e5 with the stationarity measure resting at 1.01 tol)
        if (ggt <= (1.0 - 1e-4 * alpha) * gg && ggt < gg) { ok = true; break; }
      }
      // backtracking: halving; with a metric (whose unit step is a Newton step of the cost, too long only across the rows' curvature) the minimiser of
      // the parabola through phi(0), phi'(0), phi(alpha), kept inside [0.1, 0.5] alpha
      const double bend = vt - val - alpha * slope;
      if (metric && (vt == vt) && fabs(vt) < 1e300 && bend > 0.0) alpha = fmin(0.5 * alpha, fmax(0.1 * alpha, -slope * alpha * alpha / (2.0 * bend)));
      else alpha *= 0.5;
      if (evals >= T.max_iter) break;
    }
    if (!ok) {
      // rowv belongs to the rejected trial: re-evaluate at x before anything reads the rows again
      val = ev.phi(W.x, W.g, rho, &fval, &cmax, &meas);
      ++evals;
      if (H_is_eye || evals >= T.max_iter) break;  // steepest descent cannot improve: rounding floor
      eye();
      H_is_eye = true;
      continue;
    }
    // BFGS update of the inverse Hessian with s = xt - x, y = gt - g (y kept in d, which is free now)
    double sy = 0.0, ss = 0.0, yy = 0.0;
    for (int k = 0; k < n; ++k) {
      const double sv = W.xt[TIDX(k)] - W.x[TIDX(k)], yv = W.gt[TIDX(k)] - W.g[TIDX(k)];
      W.s[TIDX(k)] = sv;
      W.d[TIDX(k)] = yv;
      sy += sv * yv; ss += sv * sv; yy += yv * yv;
    }
    if (m > 0) {
      if (sy > 1e-12 * sqrt(ss) * sqrt(yy)) {
        for (int k = 0; k < n; ++k) { Sr(head, k) = W.s[TIDX(k)]; Yr(head, k) = W.d[TIDX(k)]; }
        Rr(head) = 1.0 / sy;
        head = (head + 1) % m;
        if (hist < m) ++hist;
        H_is_eye = false;
      }
    } else if (sy > 1e-12 * sqrt(ss) * sqrt(yy)) {
      if (H_is_eye) {  // Nocedal & Wright (6.20): the first pair sets the scale of the identity before it updates it
        const double gam = sy / yy;
        for (int i = 0; i < n; ++i) W.H[TIDX(i * n + i)] = gam;
      }
      double yHy = 0.0;
      for (int i = 0; i < n; ++i) {
        double v = 0.0;
        for (int j = 0; j < n; ++j) v += W.H[TIDX(i * n + j)] * W.d[TIDX(j)];
        W.hy[TIDX(i)] = v;
        yHy += W.d[TIDX(i)] * v;
      }
      const double c1 = (sy + yHy) / (sy * sy);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
          W.H[TIDX(i * n + j)] += c1 * W.s[TIDX(i)] * W.s[TIDX(j)] - (W.hy[TIDX(i)] * W.s[TIDX(j)] + W.s[TIDX(i)] * W.hy[TIDX(j)]) / sy;
      H_is_eye = false;
    }
    for (int k = 0; k < n; ++k) { W.x[TIDX(k)] = W.xt[TIDX(k)]; W.g[TIDX(k)] = W.gt[TIDX(k)]; }
    val = vt; fval = ft; cmax = ct; meas = mt;
    alpha_prev = alpha;
  }
  for (int k = 0; k < n; ++k)
    if (xo) xo[(size_t)gb * n + k] = W.x[TIDX(k)];
  if (fo) fo[gb] = fval;
  if (kkt) { kkt[3 * (size_t)gb] = stat; kkt[3 * (size_t)gb + 1] = cmax; kkt[3 * (size_t)gb + 2] = meas; }
  if (iters) iters[gb] = evals;
  if (status) status[gb] = st;
  if (mult) {
    for (int i = 0; i < T.n_ineq; ++i) mult[(size_t)gb * (T.n_ineq + T.n_eq) + i] = W.lam[TIDX(i)];
    for (int i = 0; i < T.n_eq; ++i) mult[(size_t)gb * (T.n_ineq + T.n_eq) + T.n_ineq + i] = W.mu[TIDX(i)];
  }
}

#endif  // OH_TAPE_SOLVER_H
