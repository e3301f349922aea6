// ORACLE-SIDE CPU BASELINE (test/measurement infrastructure, never product): the figure-eight state machine of
// optas_amd/csrc (eval_knot / couple_knot / step_instance, the very functions the HIP kernels call) compiled for the host
// cores and driven by plain loops over knots and instances.  It exists so that bench.py's `cpu_baseline` times the same
// algorithm at compiled-code speed on the box's CPU cores next to the GPU (SURVEY 8(d) "C++ host path on 1 core and on
// all cores"); the independent parity oracle remains the numpy restatement in oracle/*.py.
// Only bench.py's cpu_baseline leg and tests/ may load the library built from this file (oracle/_build/liboracle_port.so).
// Host definitions of the platform hooks the device headers ask for (oh_device.h, oh_figure8_units.h); the product's are in
// optas_amd/csrc/oh_platform_gfx950.h.
#include <cmath>
#define OH_DEV __host__ __device__ __forceinline__
#define OH_RSQRT(x) (1.0 / sqrt(x))
#include "../../optas_amd/csrc/oh_device.h"
struct RowBuf {
  char* p;
};
OH_DEV RowBuf rowbuf(const double* knot_base) { return RowBuf{(char*)knot_base}; }
OH_DEV double rb_ld(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes) { return *(const double*)(rb.p + row_bytes + lane_bytes); }
OH_DEV void rb_st(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes, const double x) { *(double*)(rb.p + row_bytes + lane_bytes) = x; }
OH_DEV void oh_count(unsigned long long* c) { *c += 1ULL; }
OH_DEV int oh_take_ticket(int* c) { return (*c)++; }
OH_DEV void oh_fence(const double) {}
#include "../../optas_amd/csrc/oh_figure8_units.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Workspace {
  std::vector<double> pool;
  std::vector<int> ipool;
  unsigned long long work[3] = {0, 0, 0};
  FigBuffers D{};
};

void carve(Workspace& w, const oh_problem_desc& d, const oh_chain* chain) {
  const int N = d.ndof, NZ = N - 3, T = d.T, Bp = 1;
  const size_t per_q = (size_t)T * N * Bp, per_Z = (size_t)T * N * NZ * Bp, per_Dr = (size_t)T * (NZ * (NZ + 1) / 2) * Bp, per_t = (size_t)T * Bp;
  size_t nd = 2 * per_q + 2 * per_Z + 2 * per_Dr + 2 * per_q + 4 * per_t + 2 * per_q + 2 * (size_t)T * NZ * NZ + 2 * (size_t)T * NZ + 2 * per_t +
              (size_t)T * NZ + (size_t)T * NZ * NZ + (size_t)T * NZ + 12 + 7 + (size_t)4 * T + 2 * (size_t)T * (3 + 3 * NZ) + 64;
  w.pool.assign(nd, 0.0);
  w.ipool.assign(16, 0);
  double* p = w.pool.data();
  auto take = [&](size_t n) { double* r = p; p += n; return r; };
  FigBuffers& D = w.D;
  D.B = 1; D.Bp = 1; D.chain = chain;
  for (int s = 0; s < 2; ++s) D.q[s] = take(per_q);
  for (int s = 0; s < 2; ++s) D.Z[s] = take(per_Z);
  for (int s = 0; s < 2; ++s) D.Dr[s] = take(per_Dr);
  for (int s = 0; s < 2; ++s) D.g[s] = take(per_q);
  for (int s = 0; s < 2; ++s) D.phi[s] = take(per_t);
  for (int s = 0; s < 2; ++s) D.cv[s] = take(per_t);
  for (int s = 0; s < 2; ++s) D.Gfull[s] = take(per_q);
  for (int s = 0; s < 2; ++s) D.mdl[s] = take((size_t)T * (3 + 3 * NZ));
  for (int s = 0; s < 2; ++s) D.E[s] = take((size_t)T * NZ * NZ);
  for (int s = 0; s < 2; ++s) D.gt[s] = take((size_t)T * NZ);
  for (int s = 0; s < 2; ++s) D.merit[s] = take(per_t);
  D.zstep = take((size_t)T * NZ);
  D.Kmat = take((size_t)T * NZ * NZ);
  D.kvec = take((size_t)T * NZ);
  D.ref = take(12);
  D.fconst = take(1); D.f_cur = take(1); D.pred = take(1); D.mu = take(1); D.nun = take(1); D.stat = take(1); D.feas = take(1);
  D.fpsi = nullptr;
  D.lam_h = take((size_t)4 * T);
  int* ip = w.ipool.data();
  D.cur = ip++; D.first = ip++; D.skip = ip++; D.polish = ip++; D.stale = ip++; D.status = ip++; D.iters = ip++; D.orig = ip++; D.newidx = ip++; D.n_running = ip++; D.n_new = ip++;
  D.work = w.work;
}

template <int N>
void solve_one(const FigParams& P, Workspace& w, const double* x0, const double* p, double* x, double* f, double* kkt, int* iters, int* status) {
  const FigBuffers& D = w.D;
  for (int tt = P.T - 1; tt >= 0; --tt) setup_unit<N>(P, D, x0, p, 0, tt);
  const int hard_cap = 2 * P.max_iter + 42;
  for (int it = 0; it < hard_cap && D.status[0] < 0; ++it) {
    const int slot = it & 1;
    for (int t = P.t0; t < P.T; ++t) eval_unit<N>(P, D, slot, 0, t);      // both return at once for a skipping instance
    for (int t = P.t0; t < P.T; ++t) couple_unit<N>(P, D, slot, 0, t);
    if (D.skip[0]) D.skip[0] = 0;
    else step_instance<N, false>(P, D, 0, slot);
  }
  for (int t = 0; t < P.T; ++t) finalize_unit<N>(P, D, 0, x, f, kkt, iters, status, 0, t);
}

}  // namespace

// desc->local_path is read on the host; x0 [B][nx], p [B][ndof] in; x [B][nx], f [B], kkt [B][3], iters [B], status [B] out.
extern "C" int oh_port_solve(const oh_problem_desc* desc, const oh_chain* chain, int B, const double* x0, const double* p, double* x, double* f,
                             double* kkt, int* iters, int* status, int threads) {
  if (!desc || !chain || B < 1 || !x0 || !p || !x || !f || !kkt || !iters || !status) return 1;
  if (chain->has_lead) return 1;  // parameterised lead joints are not part of the baseline workload
  if (desc->kind != OH_PROBLEM_FIGURE_EIGHT || !desc->lock_orientation || (desc->ndof != 6 && desc->ndof != 7) || !desc->local_path) return 1;
  FigParams P{};
  P.T = desc->T;
  P.t0 = desc->fix_dq0 ? 2 : 1;
  P.lock = 1;
  P.path_in_frame = desc->path_in_frame;
  P.nx = desc->ndof * desc->T + desc->ndof * (desc->T - 1);
  P.dt = desc->dt;
  P.w_path = desc->w_path;
  P.kappa = desc->w_vel / (desc->dt * desc->dt);
  P.tol = desc->tol > 0.0 ? desc->tol : 1e-6;
  P.tol_feas = desc->tol_feas > 0.0 ? desc->tol_feas : 1e-9;
  P.tol_retract = fmin(1e-10, P.tol_feas);
  P.tol_retract_min = fmin(1e-13, P.tol_retract);  // (fill_params in csrc/oh_api.hip)
  P.feas_accept = fmax(1e-8, 10.0 * P.tol_feas);
  P.max_retract = 4;
  P.max_iter = desc->max_iter > 0 ? desc->max_iter : 200;
  P.hessian = desc->hessian;
  P.hyb_switch = 1e-5 * desc->w_path;
  P.mu0 = desc->mu0 > 0.0 ? desc->mu0 : 0.0;
  P.relax = 1.5;  // the library's defaults (oh_api.hip:fill_params)
  P.relax_from = 4;
  P.settle_k = 1.0;
  P.al_fuse = 1;
  P.local_path = desc->local_path;
  P.np = desc->ndof;
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  std::atomic<int> next{0};
  auto worker = [&]() {
    Workspace w;
    carve(w, *desc, chain);
    for (;;) {
      const int b = next.fetch_add(1);
      if (b >= B) break;
      const size_t nx = (size_t)P.nx;
      if (desc->ndof == 7) solve_one<7>(P, w, x0 + b * nx, p + (size_t)b * 7, x + b * nx, f + b, kkt + 3 * (size_t)b, iters + b, status + b);
      else solve_one<6>(P, w, x0 + b * nx, p + (size_t)b * 6, x + b * nx, f + b, kkt + 3 * (size_t)b, iters + b, status + b);
    }
  };
  std::vector<std::thread> pool;
  for (int i = 1; i < threads; ++i) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  return 0;
}
