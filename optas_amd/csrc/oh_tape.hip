// Generic small dense NLP family (SURVEY 8(f) rank 1 in spirit: the reference hands CasADi SX tapes to IPOPT; here the host compiles the
// problem's expression trees into ONE scalar instruction tape, optas_amd/tape.py, and the GPU evaluates it):
//
//     min f(x, p)   s.t.  rows[0 .. n_ineq) >= 0,   rows[n_ineq .. n_ineq + n_eq) = 0            (optimization.py:27-51 without the mirrored rows)
//
// One thread owns one instance.  An evaluation is one forward sweep and ONE reverse sweep seeded with the augmented-Lagrangian weights of
// all rows at once (cost 1, equality rows -mu + rho c, active inequality rows -(lam - rho g)), so the gradient of the merit costs two
// passes over the tape whatever the number of rows.  Two evaluators share the solver of oh_tape_solver.h:
//   * the interpreter below: all lanes of a wavefront execute the same tape instruction on registers that live in HBM/L2 as val[i][b]
//     (every access one coalesced line).  No set-up cost, but each instruction is a dependent memory round trip: latency bound;
//   * straight-line HIP code generated from the tape and compiled with hiprtc for gfx950 when the handle is created (desc.jit): SSA values
//     become VGPRs, the device compiler schedules, folds and shares the transcendentals.  Both compute the same IEEE operations in the
//     same order (contraction is switched off inside the generated evaluator), so they agree bit for bit.
// numpy restatement of evaluator and solver: oracle/tape_ref.py.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "oh_jit.h"  // the disk cache of run-time compiled code objects
#include "oh_kernels.h"
#include "oh_tape_solver_src.h"  // OH_TAPE_SOLVER_SRC: the text of oh_tape_solver.h, written by optas_amd/build.py

namespace {

struct TapeView {
  const int* __restrict__ op;
  const int* __restrict__ a;
  const int* __restrict__ bb;
  const double* __restrict__ c;
  const int* __restrict__ rows;
};

struct InterpEval {
  const TapeParams& T;
  const TapeView& tv;
  const TapeWork& W;
  double* __restrict__ val;
  double* __restrict__ adj;
  const double* __restrict__ pb;
  const int Bp, b;

  // contraction off: one IEEE operation per tape operation, like the generated code and the numpy restatement
  __device__ void forward(const double* __restrict__ xs) {
#pragma clang fp contract(off)
    for (int i = 0; i < T.len; ++i) {
      const int o = tv.op[i], ia = tv.a[i], ib = tv.bb[i];
      double v;
      switch (o) {
        case 0: v = tv.c[i]; break;
        case 1: v = xs[TIDX(ia)]; break;
        case 2: v = pb[ia]; break;
        case 3: v = val[TIDX(ia)] + val[TIDX(ib)]; break;
        case 4: v = val[TIDX(ia)] - val[TIDX(ib)]; break;
        case 5: v = val[TIDX(ia)] * val[TIDX(ib)]; break;
        case 6: v = val[TIDX(ia)] / val[TIDX(ib)]; break;
        case 7: v = -val[TIDX(ia)]; break;
        case 8: v = sin(val[TIDX(ia)]); break;
        case 9: v = cos(val[TIDX(ia)]); break;
        case 10: v = atan2(val[TIDX(ia)], val[TIDX(ib)]); break;
        case 11: v = sqrt(val[TIDX(ia)]); break;
        case 12: { const double t = val[TIDX(ia)]; v = t * t; } break;
        // round 4: the rest of what the reference's graphs emit (Quaternion.getrpy, optas.clip): values as casadi's SX machine computes them
        case 13: v = asin(val[TIDX(ia)]); break;
        case 14: v = fabs(val[TIDX(ia)]); break;
        case 15: v = fmin(val[TIDX(ia)], val[TIDX(ib)]); break;
        case 16: v = fmax(val[TIDX(ia)], val[TIDX(ib)]); break;
        case 17: v = val[TIDX(ia)] < val[TIDX(ib)] ? 1.0 : 0.0; break;
        case 18: v = val[TIDX(ia)] <= val[TIDX(ib)] ? 1.0 : 0.0; break;
        case 19: v = val[TIDX(ia)] == val[TIDX(ib)] ? 1.0 : 0.0; break;
        case 20: v = val[TIDX(ia)] != val[TIDX(ib)] ? 1.0 : 0.0; break;
        case 21: v = val[TIDX(ia)] == 0.0 ? 1.0 : 0.0; break;
        case 22: v = (val[TIDX(ia)] != 0.0 && val[TIDX(ib)] != 0.0) ? 1.0 : 0.0; break;
        case 23: v = (val[TIDX(ia)] != 0.0 || val[TIDX(ib)] != 0.0) ? 1.0 : 0.0; break;
        case 24: v = val[TIDX(ia)] != 0.0 ? val[TIDX(ib)] : 0.0; break;  // if_else_zero
        // round 5: exp and log -- with them the walker expresses pow, tanh, sinh, cosh, acos, atan, the inverse hyperbolics (casadi_tape.py)
        case 25: v = exp(val[TIDX(ia)]); break;
        default: v = log(val[TIDX(ia)]); break;  // 26
      }
      val[TIDX(i)] = v;
    }
  }

  // adj holds the seeds on entry (zero elsewhere); grad[k][b] accumulates d/dx_k
  __device__ void reverse(double* __restrict__ grad) {
#pragma clang fp contract(off)
    for (int k = 0; k < T.nx; ++k) grad[TIDX(k)] = 0.0;
    for (int i = T.len - 1; i >= 0; --i) {
      const double w = adj[TIDX(i)];
      const int o = tv.op[i], ia = tv.a[i], ib = tv.bb[i];
      switch (o) {
        case 0: case 2: break;
        case 1: grad[TIDX(ia)] += w; break;
        case 3: adj[TIDX(ia)] += w; adj[TIDX(ib)] += w; break;
        case 4: adj[TIDX(ia)] += w; adj[TIDX(ib)] -= w; break;
        case 5: { const double va = val[TIDX(ia)], vb = val[TIDX(ib)]; adj[TIDX(ia)] += w * vb; adj[TIDX(ib)] += w * va; } break;
        case 6: { const double va = val[TIDX(ia)], vb = val[TIDX(ib)]; adj[TIDX(ia)] += w / vb; adj[TIDX(ib)] -= w * va / (vb * vb); } break;
        case 7: adj[TIDX(ia)] -= w; break;
        case 8: adj[TIDX(ia)] += w * cos(val[TIDX(ia)]); break;
        case 9: adj[TIDX(ia)] -= w * sin(val[TIDX(ia)]); break;
        case 10: { const double va = val[TIDX(ia)], vb = val[TIDX(ib)], d = va * va + vb * vb; adj[TIDX(ia)] += w * vb / d; adj[TIDX(ib)] -= w * va / d; } break;
        case 11: adj[TIDX(ia)] += w * 0.5 / val[TIDX(i)]; break;
        case 12: adj[TIDX(ia)] += w * 2.0 * val[TIDX(ia)]; break;
        case 13: { const double va = val[TIDX(ia)]; adj[TIDX(ia)] += w / sqrt(1.0 - va * va); } break;
        case 14: { const double va = val[TIDX(ia)]; adj[TIDX(ia)] += w * (va > 0.0 ? 1.0 : (va < 0.0 ? -1.0 : 0.0)); } break;
        case 15: if (val[TIDX(ia)] <= val[TIDX(ib)]) adj[TIDX(ia)] += w; else adj[TIDX(ib)] += w; break;  // casadi: d fmin = (x <= y, !(x <= y))
        case 16: if (val[TIDX(ia)] >= val[TIDX(ib)]) adj[TIDX(ia)] += w; else adj[TIDX(ib)] += w; break;
        case 24: if (val[TIDX(ia)] != 0.0) adj[TIDX(ib)] += w; break;
        case 25: adj[TIDX(ia)] += w * val[TIDX(i)]; break;
        case 26: adj[TIDX(ia)] += w / val[TIDX(ia)]; break;
        default: break;  // 17..23: comparisons and logic are piecewise constant
      }
    }
  }

  __device__ double phi(const double* __restrict__ xs, double* __restrict__ gout, const double rho, double* fout, double* cmax, double* meas) {
    forward(xs);
    for (int i = 0; i < T.len; ++i) adj[TIDX(i)] = 0.0;
    const double f = val[TIDX(T.out_cost)];
    adj[TIDX(T.out_cost)] = 1.0;
    double v = f, cm = 0.0, ms = 0.0;
    for (int i = 0; i < T.n_ineq; ++i) {
      const int r = tv.rows[i];
      const double g = val[TIDX(r)];
      adj[TIDX(r)] += tape_al_ineq(g, W.lam[TIDX(i)], rho, v, cm, ms);
      W.rowv[TIDX(i)] = g;
    }
    for (int i = 0; i < T.n_eq; ++i) {
      const int r = tv.rows[T.n_ineq + i];
      const double c = val[TIDX(r)];
      adj[TIDX(r)] += tape_al_eq(c, W.mu[TIDX(i)], rho, v, cm, ms);
      W.rowv[TIDX(T.n_ineq + i)] = c;
    }
    reverse(gout);
    *fout = f;
    *cmax = cm;
    *meas = ms;
    return v;
  }
};

__global__ __launch_bounds__(64) void k_tape_solve(TapeParams T, TapeView tv, int B, int Bp, const double* __restrict__ x0, const double* __restrict__ par,
                                                   double* __restrict__ work, double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt,
                                                   int* __restrict__ iters, int* __restrict__ status, double* __restrict__ mult) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const TapeWork W = tape_carve(T, work + 2 * (size_t)T.len * Bp, Bp);
  InterpEval ev{T, tv, W, work, work + (size_t)T.len * Bp, par + (size_t)b * T.np, Bp, b};
  tape_solve_instance(T, ev, W, Bp, b, b, x0, xo, fo, kkt, iters, status, mult);
}

// One forward sweep (and, seeded, one reverse sweep) at given points: values / adjoints of chosen registers and the gradient wrt x (oh_tape_probe).
// work: [2 len + 2 nx][Bp] -- registers, adjoints, the point and the gradient in the interpreter's SoA layout.
__global__ __launch_bounds__(64) void k_tape_probe(TapeParams T, TapeView tv, int B, int Bp, const double* __restrict__ x, const double* __restrict__ par,
                                                   double* __restrict__ work, int n_regs, const int* __restrict__ regs, double* __restrict__ val_out,
                                                   const double* __restrict__ seeds, double* __restrict__ adj_out, double* __restrict__ grad_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double* val = work;
  double* adj = work + (size_t)T.len * Bp;
  double* xs = work + 2 * (size_t)T.len * Bp;
  double* gr = xs + (size_t)T.nx * Bp;
  for (int k = 0; k < T.nx; ++k) xs[TIDX(k)] = x[(size_t)b * T.nx + k];
  const TapeWork none{};
  InterpEval ev{T, tv, none, val, adj, par + (size_t)b * T.np, Bp, b};
  ev.forward(xs);
  if (val_out)
    for (int i = 0; i < n_regs; ++i) val_out[(size_t)b * n_regs + i] = val[TIDX(regs[i])];
  if (!seeds) return;
  for (int i = 0; i < T.len; ++i) adj[TIDX(i)] = 0.0;
  const int nrow = T.n_ineq + T.n_eq;
  const double* sd = seeds + (size_t)b * (1 + nrow);
  adj[TIDX(T.out_cost)] += sd[0];
  for (int i = 0; i < nrow; ++i) adj[TIDX(tv.rows[i])] += sd[1 + i];
  ev.reverse(gr);
  // (the reverse sweep leaves in adj[i] the derivative of the seeded combination with respect to register i)
  if (adj_out)
    for (int i = 0; i < n_regs; ++i) adj_out[(size_t)b * n_regs + i] = adj[TIDX(regs[i])];
  if (grad_out)
    for (int k = 0; k < T.nx; ++k) grad_out[(size_t)b * T.nx + k] = gr[TIDX(k)];
}

// ---- code generation ------------------------------------------------------------------------------------------------------------------
void emit(std::string& s, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  s += buf;
}

std::string generate(const TapeParams& T, const int* op, const int* a, const int* bb, const double* c, const int* rows) {
  std::string s;
  s.reserve(64 * (size_t)T.len + sizeof(OH_TAPE_SOLVER_SRC) + 4096);
  emit(s, "#define OH_TAPE_ST_CONVERGED %d\n#define OH_TAPE_ST_MAX_ITER %d\n#define OH_TAPE_ST_NUMERICAL %d\n", (int)OH_STATUS_CONVERGED, (int)OH_STATUS_MAX_ITER,
       (int)OH_STATUS_NUMERICAL);
  s += OH_TAPE_SOLVER_SRC;
  s += "\nstruct JitEval {\n  const TapeWork& W;\n  const double* __restrict__ pb;\n  const int Bp, b;\n"
       "  __device__ double phi(const double* __restrict__ xs, double* __restrict__ gout, const double rho, double* fout, double* cmax, double* meas) {\n"
       "#pragma clang fp contract(off)\n";
  // which registers the reverse sweep reaches: everything the cost and the rows depend on
  std::vector<char> live(T.len, 0);
  live[T.out_cost] = 1;
  for (int i = 0; i < T.n_ineq + T.n_eq; ++i) live[rows[i]] = 1;
  auto is_binary = [](int o) { return (o >= 3 && o <= 6) || o == 10 || (o >= 15 && o <= 20) || (o >= 22 && o <= 24); };
  for (int i = T.len - 1; i >= 0; --i)
    if (live[i] && op[i] >= 3) {
      live[a[i]] = 1;
      if (is_binary(op[i])) live[bb[i]] = 1;
    }
  for (int i = 0; i < T.len; ++i) {
    if (!live[i]) continue;
    switch (op[i]) {
      case 0: emit(s, "    const double v%d = %a;\n", i, c[i]); break;
      case 1: emit(s, "    const double v%d = xs[TIDX(%d)];\n", i, a[i]); break;
      case 2: emit(s, "    const double v%d = pb[%d];\n", i, a[i]); break;
      case 3: emit(s, "    const double v%d = v%d + v%d;\n", i, a[i], bb[i]); break;
      case 4: emit(s, "    const double v%d = v%d - v%d;\n", i, a[i], bb[i]); break;
      case 5: emit(s, "    const double v%d = v%d * v%d;\n", i, a[i], bb[i]); break;
      case 6: emit(s, "    const double v%d = v%d / v%d;\n", i, a[i], bb[i]); break;
      case 7: emit(s, "    const double v%d = -v%d;\n", i, a[i]); break;
      case 8: emit(s, "    const double v%d = sin(v%d);\n", i, a[i]); break;
      case 9: emit(s, "    const double v%d = cos(v%d);\n", i, a[i]); break;
      case 10: emit(s, "    const double v%d = atan2(v%d, v%d);\n", i, a[i], bb[i]); break;
      case 11: emit(s, "    const double v%d = sqrt(v%d);\n", i, a[i]); break;
      case 13: emit(s, "    const double v%d = asin(v%d);\n", i, a[i]); break;
      case 14: emit(s, "    const double v%d = fabs(v%d);\n", i, a[i]); break;
      case 15: emit(s, "    const double v%d = fmin(v%d, v%d);\n", i, a[i], bb[i]); break;
      case 16: emit(s, "    const double v%d = fmax(v%d, v%d);\n", i, a[i], bb[i]); break;
      case 17: emit(s, "    const double v%d = v%d < v%d ? 1.0 : 0.0;\n", i, a[i], bb[i]); break;
      case 18: emit(s, "    const double v%d = v%d <= v%d ? 1.0 : 0.0;\n", i, a[i], bb[i]); break;
      case 19: emit(s, "    const double v%d = v%d == v%d ? 1.0 : 0.0;\n", i, a[i], bb[i]); break;
      case 20: emit(s, "    const double v%d = v%d != v%d ? 1.0 : 0.0;\n", i, a[i], bb[i]); break;
      case 21: emit(s, "    const double v%d = v%d == 0.0 ? 1.0 : 0.0;\n", i, a[i]); break;
      case 22: emit(s, "    const double v%d = (v%d != 0.0 && v%d != 0.0) ? 1.0 : 0.0;\n", i, a[i], bb[i]); break;
      case 23: emit(s, "    const double v%d = (v%d != 0.0 || v%d != 0.0) ? 1.0 : 0.0;\n", i, a[i], bb[i]); break;
      case 24: emit(s, "    const double v%d = v%d != 0.0 ? v%d : 0.0;\n", i, a[i], bb[i]); break;
      case 25: emit(s, "    const double v%d = exp(v%d);\n", i, a[i]); break;
      case 26: emit(s, "    const double v%d = log(v%d);\n", i, a[i]); break;
      default: emit(s, "    const double v%d = v%d * v%d;\n", i, a[i], a[i]); break;
    }
  }
  for (int i = 0; i < T.len; ++i)
    if (live[i] && op[i] != 0 && op[i] != 2) emit(s, "    double a%d = 0.0;\n", i);
  for (int k = 0; k < T.nx; ++k) emit(s, "    double g%d = 0.0;\n", k);
  emit(s, "    double val = v%d, cm = 0.0, ms = 0.0;\n", T.out_cost);
  auto has_adj = [&](int i) { return op[i] != 0 && op[i] != 2; };
  if (has_adj(T.out_cost)) emit(s, "    a%d = 1.0;\n", T.out_cost);
  for (int i = 0; i < T.n_ineq; ++i) {
    const int r = rows[i];
    emit(s, "    { const double w = tape_al_ineq(v%d, W.lam[TIDX(%d)], rho, val, cm, ms); W.rowv[TIDX(%d)] = v%d;", r, i, i, r);
    if (has_adj(r)) emit(s, " a%d += w;", r);
    s += " }\n";
  }
  for (int i = 0; i < T.n_eq; ++i) {
    const int r = rows[T.n_ineq + i];
    emit(s, "    { const double w = tape_al_eq(v%d, W.mu[TIDX(%d)], rho, val, cm, ms); W.rowv[TIDX(%d)] = v%d;", r, i, T.n_ineq + i, r);
    if (has_adj(r)) emit(s, " a%d += w;", r);
    s += " }\n";
  }
  for (int i = T.len - 1; i >= 0; --i) {
    if (!live[i]) continue;
    const int ia = a[i], ib = bb[i];
    const bool da = op[i] >= 3 && has_adj(ia), db = is_binary(op[i]) && has_adj(ib);
    switch (op[i]) {
      case 0: case 2: break;
      case 1: emit(s, "    g%d += a%d;\n", ia, i); break;
      case 3:
        if (da) emit(s, "    a%d += a%d;\n", ia, i);
        if (db) emit(s, "    a%d += a%d;\n", ib, i);
        break;
      case 4:
        if (da) emit(s, "    a%d += a%d;\n", ia, i);
        if (db) emit(s, "    a%d -= a%d;\n", ib, i);
        break;
      case 5:
        if (da) emit(s, "    a%d += a%d * v%d;\n", ia, i, ib);
        if (db) emit(s, "    a%d += a%d * v%d;\n", ib, i, ia);
        break;
      case 6:
        if (da) emit(s, "    a%d += a%d / v%d;\n", ia, i, ib);
        if (db) emit(s, "    a%d -= a%d * v%d / (v%d * v%d);\n", ib, i, ia, ib, ib);
        break;
      case 7:
        if (da) emit(s, "    a%d -= a%d;\n", ia, i);
        break;
      case 8:
        if (da) emit(s, "    a%d += a%d * cos(v%d);\n", ia, i, ia);
        break;
      case 9:
        if (da) emit(s, "    a%d -= a%d * sin(v%d);\n", ia, i, ia);
        break;
      case 10:
        emit(s, "    { const double d = v%d * v%d + v%d * v%d;", ia, ia, ib, ib);
        if (da) emit(s, " a%d += a%d * v%d / d;", ia, i, ib);
        if (db) emit(s, " a%d -= a%d * v%d / d;", ib, i, ia);
        s += " }\n";
        break;
      case 11:
        if (da) emit(s, "    a%d += a%d * 0.5 / v%d;\n", ia, i, i);
        break;
      case 12:
        if (da) emit(s, "    a%d += a%d * 2.0 * v%d;\n", ia, i, ia);
        break;
      case 13:
        if (da) emit(s, "    a%d += a%d / sqrt(1.0 - v%d * v%d);\n", ia, i, ia, ia);
        break;
      case 14:
        if (da) emit(s, "    a%d += a%d * (v%d > 0.0 ? 1.0 : (v%d < 0.0 ? -1.0 : 0.0));\n", ia, i, ia, ia);
        break;
      case 15:
      case 16: {
        const char* cmp = op[i] == 15 ? "<=" : ">=";
        if (da && db) emit(s, "    if (v%d %s v%d) a%d += a%d; else a%d += a%d;\n", ia, cmp, ib, ia, i, ib, i);
        else if (da) emit(s, "    if (v%d %s v%d) a%d += a%d;\n", ia, cmp, ib, ia, i);
        else if (db) emit(s, "    if (!(v%d %s v%d)) a%d += a%d;\n", ia, cmp, ib, ib, i);
      } break;
      case 24:
        if (db) emit(s, "    if (v%d != 0.0) a%d += a%d;\n", ia, ib, i);
        break;
      case 25:
        if (da) emit(s, "    a%d += a%d * v%d;\n", ia, i, i);
        break;
      case 26:
        if (da) emit(s, "    a%d += a%d / v%d;\n", ia, i, ia);
        break;
      default:  // 17..23: comparisons and logic are piecewise constant
        break;
    }
  }
  for (int k = 0; k < T.nx; ++k) emit(s, "    gout[TIDX(%d)] = g%d;\n", k, k);
  emit(s, "    *fout = v%d; *cmax = cm; *meas = ms;\n    return val;\n  }\n};\n", T.out_cost);
  s += "extern \"C\" __global__ __launch_bounds__(64) void k_tape_jit(TapeParams T, int B, int Bp, const double* __restrict__ x0, const double* __restrict__ par,\n"
       "    double* __restrict__ work, double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters,\n"
       "    int* __restrict__ status, double* __restrict__ mult) {\n"
       "  const int b = blockIdx.x * blockDim.x + threadIdx.x;\n  if (b >= B) return;\n"
       "  const TapeWork W = tape_carve(T, work, Bp);\n  JitEval ev{W, par + (size_t)b * T.np, Bp, b};\n"
       "  tape_solve_instance(T, ev, W, Bp, b, b, x0, xo, fo, kkt, iters, status, mult);\n}\n";
  // the same with the solver's work arrays (x, gradients, the BFGS matrix, multipliers, rows) in LDS, [row][lane of the block]: for small
  // batches every access of the quasi-Newton loop is otherwise a dependent round trip to the global buffer
  s += "extern \"C\" __global__ __launch_bounds__(64) void k_tape_jit_lds(TapeParams T, int B, int Bp_unused, const double* __restrict__ x0, const double* __restrict__ par,\n"
       "    double* __restrict__ work_unused, double* __restrict__ xo, double* __restrict__ fo, double* __restrict__ kkt, int* __restrict__ iters,\n"
       "    int* __restrict__ status, double* __restrict__ mult) {\n"
       "  extern __shared__ double tape_lds[];\n"
       "  const int gb = blockIdx.x * blockDim.x + threadIdx.x;\n  if (gb >= B) return;\n"
       "  const int Bp = blockDim.x, b = threadIdx.x;\n"
       "  const TapeWork W = tape_carve(T, tape_lds, Bp);\n  JitEval ev{W, par + (size_t)gb * T.np, Bp, b};\n"
       "  tape_solve_instance(T, ev, W, Bp, b, gb, x0, xo, fo, kkt, iters, status, mult);\n}\n";
  return s;
}

std::mutex g_cache_mutex;
std::unordered_map<std::string, std::vector<char>> g_code_cache;  // generated source -> code object (same problem built again: no recompile)

}  // namespace

size_t oh_tape_work_rows(const TapeParams& T, bool jit) { return tape_solver_rows(T) + (jit ? 0 : 2 * (size_t)T.len); }

void oh_launch_tape_solve(hipStream_t s, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows, int B, int Bp,
                          const double* x0, const double* p, double* work, double* x, double* f, double* kkt, int* iters, int* status, double* mult) {
  TapeView tv{op, a, b, c, rows};
  hipLaunchKernelGGL(k_tape_solve, dim3((B + 63) / 64), dim3(64), 0, s, T, tv, B, Bp, x0, p, work, x, f, kkt, iters, status, mult);
}

void oh_launch_tape_probe(hipStream_t s, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows, int B, int Bp,
                          const double* x, const double* p, double* work, int n_regs, const int* regs, double* val, const double* seeds, double* adj, double* grad) {
  TapeView tv{op, a, b, c, rows};
  hipLaunchKernelGGL(k_tape_probe, dim3((B + 63) / 64), dim3(64), 0, s, T, tv, B, Bp, x, p, work, n_regs, regs, val, seeds, adj, grad);
}

std::string oh_tape_jit_source(const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows) {
  return generate(T, op, a, b, c, rows);
}

int oh_tape_jit_compile(const std::string& src, std::vector<char>* code, std::string* err) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    auto it = g_code_cache.find(src);
    if (it != g_code_cache.end()) { *code = it->second; return 0; }
  }
  // a trajectory-sized tape is minutes of compilation: its object is kept on disk like the chain-specialised kernels' (the key is the
  // generated text, which contains the solver and every constant of the problem)
  // (the disk key is the generated text plus the compile options below: an object built with other options is another object)
  static const std::string kOptKey = "// --offload-arch=gfx950 -O3 -std=c++17\n";
  if (src.size() > 100000 && oh_jit_disk_lookup(kOptKey + src, "tape", code)) {
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    g_code_cache.emplace(src, *code);
    return 0;
  }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "oh_tape_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *err = "hiprtcCreateProgram failed"; return 1; }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls > 1) hiprtcGetProgramLog(prog, log.data());
    *err = std::string("hiprtc: ") + hiprtcGetErrorString(r) + ": " + log.substr(0, 1500);
    hiprtcDestroyProgram(&prog);
    return 1;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->resize(cs);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  if (src.size() > 100000) oh_jit_disk_store(kOptKey + src, "tape", *code);
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  g_code_cache.emplace(src, *code);
  return 0;
}

// Forget a code object that did not load (a truncated or stale file of the disk cache): the next oh_tape_jit_compile of the same source recompiles.
void oh_tape_jit_forget(const std::string& src) {
  oh_jit_disk_drop(std::string("// --offload-arch=gfx950 -O3 -std=c++17\n") + src, "tape");
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  g_code_cache.erase(src);
}

int oh_tape_jit_load(const std::vector<char>& code, TapeJit* out, std::string* err) {
  if (hipModuleLoadData(&out->mod, code.data()) != hipSuccess) {
    (void)hipGetLastError();
    *err = "hipModuleLoadData failed for the generated tape kernel";
    return 1;
  }
  if (hipModuleGetFunction(&out->fn, out->mod, "k_tape_jit") != hipSuccess) {
    hipModuleUnload(out->mod);
    out->mod = nullptr;
    *err = "hipModuleGetFunction(k_tape_jit) failed";
    return 1;
  }
  if (hipModuleGetFunction(&out->fn_lds, out->mod, "k_tape_jit_lds") != hipSuccess) out->fn_lds = nullptr;
  return 0;
}

void oh_tape_jit_release(TapeJit* j) {
  if (j->mod) hipModuleUnload(j->mod);
  j->mod = nullptr;
  j->fn = nullptr;
  j->fn_lds = nullptr;
}

hipError_t oh_launch_tape_jit(hipStream_t s, const TapeJit& j, TapeParams T, int B, int Bp, const double* x0, const double* p, double* work, double* x, double* f,
                              double* kkt, int* iters, int* status, double* mult) {
  void* args[] = {&T, &B, &Bp, &x0, &p, &work, &x, &f, &kkt, &iters, &status, &mult};
  // the solver's work set in LDS when it fits 48 KB at 64, 32 or 16 instances per block (tools/gpu_tape_sweep.py, the 7-joint IK problem: one
  // instance 5.0 -> 3.4 ms, 2048 14.3 -> 10.9 ms, 32 768 25.8 -> 21.0 ms; option tape_lds_max = 0 switches it off)
  // (round 2: level above 32 768 instances -- 65 536: 30.2 / 33.5 ms, 131 072: 47.5 / 44.9 ms global / LDS; with the solver of round 3's end, which
  // spends 65 evaluations instead of 266 on the median instance, the LDS set wins at every size: 65 536: 9.8 / 7.7 ms, 131 072: 17.5 / 13.2 ms)
  const int lds_max = oh_launch_opts().tape_lds_max;  // option "tape_lds_max"
  if (j.fn_lds && B <= lds_max) {
    const size_t per = sizeof(double) * tape_solver_rows(T);
    for (int bs : {64, 32, 16})
      if (per * bs <= 48 * 1024) return hipModuleLaunchKernel(j.fn_lds, (B + bs - 1) / bs, 1, 1, bs, 1, 1, (unsigned)(per * bs), s, args, nullptr);
  }
  return hipModuleLaunchKernel(j.fn, (B + 63) / 64, 1, 1, 64, 1, 1, 0, s, args, nullptr);
}
