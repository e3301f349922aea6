// HIP kernels of liboptas_hip (gfx950): the figure-eight family.  See DESIGN.md for the data layout and per-kernel rooflines.
//
// All per-(instance, knot) arrays are structure-of-arrays with the instance index fastest:
//   a[(t*K + k)*Bp + b]        (Bp = B rounded up to 64)
// so that the 64 lanes of a wavefront, which always hold 64 consecutive instances b at one knot t,
// read and write full 512-byte lines.  What a lane computes lives in oh_figure8_units.h / oh_figure8.h; this file holds the __global__
// entry points (lane -> (instance, knot) mapping, launch bounds), the compaction kernels, the persistent tail kernel and the launchers.
#include "oh_platform_gfx950.h"
#include "oh_figure8_kernels.h"
#include "oh_kernels.h"

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

// Chain lengths (round 5).  Orientation-locked family: the null space of the three orientation rows has N - 3 dimensions, so 4 ... 8 actuated
// joints; kernels shared with the position-tracking family (set-up, finalisation, compaction): 2 ... 8.
#define OH_DISPATCH_N(n, call)         \
  switch (n) {                         \
    case 4: call(4); break;            \
    case 5: call(5); break;            \
    case 6: call(6); break;            \
    case 7: call(7); break;            \
    case 8: call(8); break;            \
    default: return false;             \
  }
#define OH_DISPATCH_N_ANY(n, call)     \
  switch (n) {                         \
    case 2: call(2); break;            \
    case 3: call(3); break;            \
    case 4: call(4); break;            \
    case 5: call(5); break;            \
    case 6: call(6); break;            \
    case 7: call(7); break;            \
    case 8: call(8); break;            \
    default: return false;             \
  }

template <int N>
__global__ __launch_bounds__(64) void k_setup(FigParams P, FigBuffers D, const double* __restrict__ x0, const double* __restrict__ pin) {
  setup_unit<N>(P, D, x0, pin, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
// Blocks per CU of the fused evaluation kernels (lead-joint and guarded variants): with the six-row retraction the fused kernel needs ~360
// live registers in its loop; at 2 waves/SIMD (256) it spills 99 of them and runs 10 % slower than at 1 wave/SIMD with none.
#ifndef OH_EVAL_WAVES
#define OH_EVAL_WAVES 1
#endif
// The same knot in two launches, each at two waves per SIMD (see EVAL_RETRACT_ONLY / EVAL_ONLY in oh_figure8.h): k_retract leaves the
// retracted trial knot in the slot, k_evalb evaluates it.  One kinematics pass more than fused, both kernels without the register
// overflow of the fused loop.
// Grid: (instance block, knot), instance block fastest.  The XCD-aware 1-D order of k_couple (one XCD walks the knots of one instance
// block, so that the 12 reference doubles and the per-instance scalars are L2 hits after the first knot) was tried here in round 2 and is
// 2.4 % SLOWER (k_eval pair 1236 vs 1207 us per launch at 3.99 M units, A/B on one box): these two kernels are bound by f64 issue, not by
// the re-fetched lines, and the knots of one instance block retire less evenly than a knot row of the batch.
template <int N>
__global__ __launch_bounds__(256, 2) void k_retract(FigParams P, FigBuffers D, const int slot) { retract_block<N>(P, D, slot); }
template <int N>
__global__ __launch_bounds__(256, 2) void k_evalb(FigParams P, FigBuffers D, const int slot) { evalb_block<N>(P, D, slot); }
// ... with the neighbour coupling of the Lagrangian gradient and of the merit folded in (eval_unit<.., ZC>): no k_couple launch follows
template <int N>
__global__ __launch_bounds__(256, 2) void k_evalb_zc(FigParams P, FigBuffers D, const int slot) { evalb_block<N, true>(P, D, slot); }
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
// the same with joint-velocity rows (oh_guards.vel_limits), and the multiplier refresh of those rows that precedes it at an outer update
template <int N>
__global__ __launch_bounds__(256) void k_couple_vel(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *D.n_running = 0;
  const int Tn = P.T - P.t0;
  const int w = blockIdx.x;
  const int chunk = w / (8 * Tn), r = w - chunk * 8 * Tn;
  const int bx = chunk * 8 + (r & 7);
  couple_unit<N, true>(P, D, slot, bx * blockDim.x + threadIdx.x, (r >> 3) + P.t0, &GP, &GB);
}
template <int N>
__global__ __launch_bounds__(256) void k_vel_update(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot) {
  vel_update_unit<N>(P, D, GP, GB, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0);
}
template <int N>
__global__ __launch_bounds__(256) void k_eval_lg(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot) {
  eval_unit<N, true>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0, &GP, &GB);
}
// ... and, for handles without sphere rows, split like the plain family's: k_retract (the kernel compiled for the chain when it is loaded), then
// the evaluation of the retracted knot with the limit rows at two wavefronts per SIMD (256 registers, 116 B of scratch; the fused kernel needs
// 256 + 96).  65 536 velocity-limited instances: evaluation 28.2 -> 23.0 ms of 88.  With sphere rows the evaluation stays at one wavefront per
// SIMD either way and the second launch costs what the split saves (16 384 instances: 134 against 130 ms): those handles keep the fused kernel.
template <int N>
__global__ __launch_bounds__(256, 2) void k_evalb_lg(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot) {
  eval_unit<N, true, false, EVAL_ONLY, false, false>(P, D, slot, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y + P.t0, &GP, &GB);
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
bool oh_launch_eval_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot, int part) {
  // part 0: the fused kernel; 1: k_retract only; 2: the evaluation of the retracted knots only (1 then 2 = the split form; no sphere rows)
  const dim3 g((D.B + 255) / 256, P.T - P.t0), b(256);
#define C(NN)                                                                              \
  if (part == 0) hipLaunchKernelGGL(k_eval_lg<NN>, g, b, 0, s, P, D, GP, GB, slot);       \
  else if (part == 1) hipLaunchKernelGGL(k_retract<NN>, g, b, 0, s, P, D, slot);          \
  else hipLaunchKernelGGL(k_evalb_lg<NN>, g, b, 0, s, P, D, GP, GB, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
bool oh_launch_step_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
  const dim3 g((D.B + 63) / 64), b(64);
#define C(NN) hipLaunchKernelGGL(k_step_lg<NN>, g, b, 0, s, P, D, GB, slot)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
}
#ifndef OH_STEP_WAVES
#define OH_STEP_WAVES 2
#endif
template <int N>
__global__ __launch_bounds__(64, OH_STEP_WAVES) void k_step(FigParams P, FigBuffers D, const int slot) {
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

// K3 with the coupling folded in (step_instance_zc): Z_{t+1} of every lane in a private column of LDS
#ifndef OH_STEP_ZC_WAVES
#define OH_STEP_ZC_WAVES 1
#endif
template <int N>
__global__ __launch_bounds__(64, OH_STEP_ZC_WAVES) void k_step_zc(FigParams P, FigBuffers D, const int slot) {
  __shared__ double zl[N * (N - 3)][64];
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
  if (running) still = step_instance_zc<N>(P, D, b, slot, &zl[0][threadIdx.x]);
  const unsigned long long m2 = __ballot(still);
  if ((threadIdx.x & 63) == 0 && m2) atomicAdd(D.n_running, __popcll(m2));
}

// The instances k_step_zc deferred in this launch (list of parity `slot`): their older Lagrangian gradient goes from the trial slot into the slot the evaluation
// reads its multiplier estimate from, a thread per (instance, knot).  A fixed small grid that strides over the list: no host round trip, no launch shaped by a count.
template <int N>
__global__ __launch_bounds__(256) void k_defer_copy(FigParams P, FigBuffers D, const int slot) {
  const int n = D.n_defer[slot];
  if (blockIdx.x == 0 && threadIdx.x == 0) D.n_defer[1 - slot] = 0;  // (the other parity's list was consumed one launch ago: ready for the next k_step_zc)
  const int Bp = D.Bp, nk = P.T - P.t0;
  const long long total = (long long)n * nk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i / nk), t = P.t0 + (int)(i % nk);
    const int b = D.defer_list[(size_t)slot * Bp + e];
    const int cur = D.cur[b];
    const double* __restrict__ Gold = D.Gfull[1 - cur];
    double* __restrict__ Gnew = D.Gfull[cur];
#pragma unroll
    for (int k = 0; k < N; ++k) Gnew[IDX(t, N, k)] = Gold[IDX(t, N, k)];
  }
}

template <int N>
__global__ __launch_bounds__(64) void k_tail(FigParams P, FigBuffers D, const int slot) { tail_block<N>(P, D, slot); }
// ... for handles whose inequality rows are joint-velocity limits only (tail_block<N, true>)
template <int N>
__global__ __launch_bounds__(64) void k_tail_vel(FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, const int slot) {
  tail_block<N, true>(P, D, slot, &GP, &GB);
}

template <int N>
__global__ __launch_bounds__(256) void k_finalize(FigParams P, FigBuffers D, int only_done, double* __restrict__ x, double* __restrict__ f,
                                                  double* __restrict__ kkt, int* __restrict__ iters, int* __restrict__ status) {
  finalize_unit<N>(P, D, only_done, x, f, kkt, iters, status, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
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
  if (P.hessian != OH_HESSIAN_GAUSS_NEWTON || P.zc) {
#pragma unroll
    for (int j = 0; j < N; ++j) D.g[1][((size_t)t * N + j) * Bp + nb] = D.Gfull[D.cur[b]][IDX(t, N, j)];
  }
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) ts[(size_t)i * Bp + nb] = D.ref[(size_t)i * Bp + b];
    ts[(size_t)12 * Bp + nb] = D.fconst[b];
    ts[(size_t)13 * Bp + nb] = D.mu[b];
    ts[(size_t)14 * Bp + nb] = (double)(D.iters[b] + (D.polish[b] == 2 ? 1 : 0));  // (a deferred sweep counted no step: nothing to take back at the restart)
    ts[(size_t)15 * Bp + nb] = (double)D.orig[b];
    ts[(size_t)16 * Bp + nb] = D.nun[b];
    ts[(size_t)17 * Bp + nb] = D.stat[b];
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
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON || P.zc) D.Gfull[1 - slot][IDX(t, N, j)] = D.g[1][IDX(t, N, j)];
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
    // 2: a restart, not a seed -- the reduced gradient and the Lagrangian gradient of the point came along, so the evaluation builds the same
    // (exact, where the hybrid rule has switched) curvature the interrupted iteration had.  With first = 1 it fell back to Gauss-Newton blocks for
    // that one step: harmless on most instances, but one whose Gauss-Newton iteration crawls (1 of 20 000 velocity-limited T = 100 instances,
    // tools/gpu_T100_vel_probe2.py) was thrown back above the switch at every one of 23 compactions and sat at the iteration cap.
    D.stat[b] = ts[(size_t)17 * Bp + b];
    D.first[b] = 2;
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
// (Writing the leaving instances out from here instead of from a k_finalize launch before the scan was tried in round 3: the gather then
// carries the kinematics walk of the multiplier map, 128 registers instead of 20, and a bench step takes 96.5 instead of 89.5 ms.)
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
  double* __restrict__ t_trial = D.q_spare[0];  // the host swaps these with q[slot], q[1 - slot], Gfull[1 - slot] after the launch
  double* __restrict__ t_cur = D.q_spare[1];
  double* __restrict__ t_G = D.G_spare;
  double* __restrict__ ts = D.Dr[1];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    t_trial[((size_t)t * N + j) * Bp + nb] = D.q[slot][IDX(t, N, j)];
    t_cur[((size_t)t * N + j) * Bp + nb] = D.q[cur][IDX(t, N, j)];
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON || P.zc) t_G[((size_t)t * N + j) * Bp + nb] = D.Gfull[cur][IDX(t, N, j)];
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
    ts[(size_t)21 * Bp + nb] = (double)((restart ? 1 : 0) + 2 * (D.polish[b] != 0 ? 1 : 0) + 4 * (D.polish[b] == 2 ? 1 : 0));
  }
}
template <int N>
__global__ __launch_bounds__(256) void k_carry_scatter(FigParams P, FigBuffers D, int Bnew, const int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int Bp = D.Bp;
  if (b >= Bnew) return;
  const double* __restrict__ ts = D.Dr[1];
  {  // per-instance scalars only: the knots stay where k_carry_gather put them
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
    D.polish[b] = (!restart && (flags & 2)) ? ((flags & 4) ? 2 : 1) : 0;
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
template <int N>
static void launch_setup_t(hipStream_t s, const FigParams& P, const FigBuffers& D, const double* x0, const double* p) {
  // (instances, knots); the last wavefront's padding lanes up to a multiple of 64 are marked finished
  hipLaunchKernelGGL(k_setup<N>, dim3((D.B + 63) / 64, P.T), dim3(64), 0, s, P, D, x0, p);
}
template <int N>
static void launch_eval_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot, int part) {
  // part 0: the whole evaluation; 1: k_retract only; 2: k_evalb only (the compaction that carries the trial along sits between them)
  const dim3 ge((D.B + 255) / 256, P.T - P.t0);
  if (part != 2) hipLaunchKernelGGL(k_retract<N>, ge, dim3(256), 0, s, P, D, slot);
  if (part != 1) {
    if (P.zc) hipLaunchKernelGGL(k_evalb_zc<N>, dim3((((D.B + 255) / 256) + 7) / 8 * 8 * (P.T - P.t0)), dim3(256), 0, s, P, D, slot);
    else hipLaunchKernelGGL(k_evalb<N>, ge, dim3(256), 0, s, P, D, slot);
  }
}
template <int N>
static void launch_carry_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
  if (phase == 0) hipLaunchKernelGGL(k_carry_gather<N>, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D, slot);
  else hipLaunchKernelGGL(k_carry_scatter<N>, dim3((Bnew + 255) / 256), dim3(256), 0, s, P, D, Bnew, slot);
}
template <int N>
static void launch_couple_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot) {
  const int nbx8 = (((D.B + 255) / 256) + 7) / 8 * 8;  // instance blocks, padded to whole groups of 8 XCDs (couple_unit drops b >= B)
  hipLaunchKernelGGL(k_couple<N>, dim3(nbx8 * (P.T - P.t0)), dim3(256), 0, s, P, D, slot);
}
template <int N>
static void launch_step_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot) {
  if (P.zc) {
    hipLaunchKernelGGL(k_step_zc<N>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, slot);
    if (P.hessian != OH_HESSIAN_GAUSS_NEWTON) hipLaunchKernelGGL(k_defer_copy<N>, dim3(512), dim3(256), 0, s, P, D, slot);
  } else hipLaunchKernelGGL(k_step<N>, dim3((D.B + 63) / 64), dim3(64), 0, s, P, D, slot);
}
template <int N>
static void launch_tail_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int slot) {
  hipLaunchKernelGGL(k_tail<N>, dim3(D.B), dim3(64), 0, s, P, D, slot);
}
// The solution x = [vec(Q); vec(dQ)] out through an LDS transpose (round 4): the knots live in [row][instance] order, the reference layout is
// [instance][row].  k_finalize writes 7 doubles per lane at a stride of nx doubles between lanes (half-filled cache lines, one per lane and store:
// 3.3 ms per solve at B = 262 144, 3.7 % of the headline).  Here a block reads 64 instances x 9 knots with the instances along the lanes (512-byte
// runs), and writes every instance's 56 q and 56 dq values of the tile as one contiguous run each.  Same arithmetic per entry as finalize_unit.
template <int N>
__global__ __launch_bounds__(256) void k_finalize_x(FigParams P, FigBuffers D, const int only_done, double* __restrict__ x) {
  constexpr int TK = 8, ROWS = (TK + 1) * N;
  __shared__ double tile[ROWS][65];
  __shared__ long long orig_s[64];
  const int Bp = D.Bp;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane, t0 = blockIdx.y * TK;
  bool emit = b < D.B;
  if (emit && only_done && D.status[b] < 0) emit = false;
  if (w == 0) orig_s[lane] = emit ? (long long)D.orig[b] : -1;
  const int nk = min(TK + 1, P.T - t0);  // knots of this tile, the one the last velocity needs included
  if (emit) {
    const double* __restrict__ qs = D.q[D.cur[b]];
    for (int row = w; row < nk * N; row += 4) tile[row][lane] = qs[IDX(t0 + row / N, N, row % N)];
  }
  __syncthreads();
  const int nq = min(TK, P.T - t0) * N, ndq = min(TK, P.T - 1 - t0) * N;
  const double inv_dt = 1.0 / P.dt;
  for (int i = w; i < 64; i += 4) {
    const long long ob = orig_s[i];
    if (ob < 0) continue;
    double* xb = x + (size_t)ob * P.nx;
    if (lane < nq) xb[(size_t)t0 * N + lane] = tile[lane][i];
    if (lane < ndq) xb[(size_t)P.T * N + (size_t)t0 * N + lane] = (tile[lane + N][i] - tile[lane][i]) * inv_dt;
  }
}
template <int N>
static void launch_finalize_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int only_done, double* x, double* f, double* kkt,
                              int* iters, int* status, const int parts) {
  if (x && (parts & 1)) hipLaunchKernelGGL(k_finalize_x<N>, dim3((D.B + 63) / 64, (P.T + 7) / 8), dim3(256), 0, s, P, D, only_done, x);
  if (!(parts & 2)) return;
  // the scalars of knot 0, and the multipliers of the quaternion rows where they are asked for (every knot)
  hipLaunchKernelGGL(k_finalize<N>, dim3((D.B + 255) / 256, D.lam_h ? P.T : 1), dim3(256), 0, s, P, D, only_done, (double*)nullptr, f, kkt, iters, status);
}
template <int N>
static void launch_compact_t(hipStream_t s, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
  if (phase == 0) hipLaunchKernelGGL(k_compact_gather<N>, dim3((D.B + 255) / 256, P.T), dim3(256), 0, s, P, D);
  else hipLaunchKernelGGL(k_compact_scatter<N>, dim3((Bnew + 255) / 256, P.T), dim3(256), 0, s, P, D, Bnew, slot);
}

bool oh_launch_setup(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const double* x0, const double* p) {
#define C(NN) launch_setup_t<NN>(s, P, D, x0, p)
  OH_DISPATCH_N_ANY(n, C)
#undef C
  return true;
}
bool oh_launch_eval(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot, int part) {
#define C(NN) launch_eval_t<NN>(s, P, D, slot, part)
  OH_DISPATCH_N(n, C)
#undef C
  return true;
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
bool oh_launch_couple_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
  const int Tn = P.T - P.t0;
  const dim3 gu((D.B + 255) / 256, Tn), gc((unsigned)(((D.B + 255) / 256 + 7) / 8 * 8 * Tn)), b(256);
#define C(NN)                                                              \
  hipLaunchKernelGGL(k_vel_update<NN>, gu, b, 0, s, P, D, GP, GB, slot);  \
  hipLaunchKernelGGL(k_couple_vel<NN>, gc, b, 0, s, P, D, GP, GB, slot)
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
bool oh_launch_tail_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot) {
#define C(NN) hipLaunchKernelGGL(k_tail_vel<NN>, dim3(D.B), dim3(64), 0, s, P, D, GP, GB, slot)
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
                        int* iters, int* status, int parts) {
#define C(NN) launch_finalize_t<NN>(s, P, D, only_done, x, f, kkt, iters, status, parts)
  OH_DISPATCH_N_ANY(n, C)
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
// ---- compaction that moves EVERYTHING (round 5; orientation-locked handles with inequality rows, horizons beyond the persistent kernel) ---------------
// The restart compaction above lays the accepted knots down densely and lets the survivors evaluate them again: cheap, but the stage data of the accepted
// point is then rebuilt with that point's own multiplier estimates instead of its predecessor's and a pending line search is dropped -- an instance's
// iterates depended on when its batch was compacted.  Here every array of both slots moves with the instance, one array at a time through a scratch
// array: the state machine does not notice, and an instance takes the same steps, bit for bit, with and without compaction, alone and in any batch.
template <class V>
__global__ __launch_bounds__(256) void k_move_gather(const V* __restrict__ src, V* __restrict__ scr, const int Bp, const int B, const int* __restrict__ newidx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nb = newidx[b];
  if (nb < 0) return;
  const size_t row = (size_t)blockIdx.y * Bp;
  scr[row + nb] = src[row + b];
}
template <class V>
__global__ __launch_bounds__(256) void k_move_scatter(V* __restrict__ dst, const V* __restrict__ scr, const int Bp, const int Bnew) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bnew) return;
  const size_t row = (size_t)blockIdx.y * Bp;
  dst[row + b] = scr[row + b];
}
// The same for an array that exists once per slot when only the slot of the accepted point (cur) is live: row of slot cur[b] -> scratch -> the same slot at the
// new index (curn = cur gathered to the new order beforehand; D.cur itself moves last).
__global__ __launch_bounds__(256) void k_move_gather_live(const double* __restrict__ src0, const double* __restrict__ src1, double* __restrict__ scr, const int Bp,
                                                          const int B, const int* __restrict__ newidx, const int* __restrict__ cur) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nb = newidx[b];
  if (nb < 0) return;
  const size_t row = (size_t)blockIdx.y * Bp;
  scr[row + nb] = (cur[b] ? src1 : src0)[row + b];
}
__global__ __launch_bounds__(256) void k_move_scatter_live(double* __restrict__ dst0, double* __restrict__ dst1, const double* __restrict__ scr, const int Bp,
                                                           const int Bnew, const int* __restrict__ curn) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bnew) return;
  const size_t row = (size_t)blockIdx.y * Bp;
  (curn[b] ? dst1 : dst0)[row + b] = scr[row + b];
}
void oh_launch_move_rows_live(hipStream_t s, double* a0, double* a1, double* scr, int rows, int Bp, int B, int Bnew, const int* newidx, const int* cur, const int* curn) {
  if (!a0 || !a1 || rows <= 0) return;
  const dim3 gg((B + 255) / 256, rows), gs((Bnew + 255) / 256, rows), blk(256);
  hipLaunchKernelGGL(k_move_gather_live, gg, blk, 0, s, (const double*)a0, (const double*)a1, scr, Bp, B, newidx, cur);
  hipLaunchKernelGGL(k_move_scatter_live, gs, blk, 0, s, a0, a1, (const double*)scr, Bp, Bnew, curn);
}
void oh_launch_move_rows(hipStream_t s, void* arr, void* scr, int rows, int Bp, int B, int Bnew, const int* newidx, bool is_int) {
  if (!arr || rows <= 0) return;
  const dim3 gg((B + 255) / 256, rows), gs((Bnew + 255) / 256, rows), blk(256);
  if (is_int) {
    hipLaunchKernelGGL(k_move_gather<int>, gg, blk, 0, s, (const int*)arr, (int*)scr, Bp, B, newidx);
    hipLaunchKernelGGL(k_move_scatter<int>, gs, blk, 0, s, (int*)arr, (const int*)scr, Bp, Bnew);
  } else {
    hipLaunchKernelGGL(k_move_gather<double>, gg, blk, 0, s, (const double*)arr, (double*)scr, Bp, B, newidx);
    hipLaunchKernelGGL(k_move_scatter<double>, gs, blk, 0, s, (double*)arr, (const double*)scr, Bp, Bnew);
  }
}

bool oh_launch_compact(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot) {
#define C(NN) launch_compact_t<NN>(s, P, D, phase, Bnew, slot)
  OH_DISPATCH_N_ANY(n, C)
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
bool oh_kernel_info_figure8(const char* name, OhKernelInfo* out) {
  const std::string n(name);
  if (n == "k_retract") return kernel_info(k_retract<7>, 256, out);
  if (n == "k_evalb") return kernel_info(k_evalb<7>, 256, out);
  if (n == "k_couple") return kernel_info(k_couple<7>, 256, out);
  if (n == "k_step") return kernel_info(k_step<7>, 64, out);
  if (n == "k_step_zc") return kernel_info(k_step_zc<7>, 64, out);
  if (n == "k_evalb_zc") return kernel_info(k_evalb_zc<7>, 256, out);
  if (n == "k_tail") return kernel_info(k_tail<7>, 64, out);
  return false;
}
