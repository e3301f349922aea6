// C ABI of liboptas_hip.so (see include/optas_hip.h).  Host-side orchestration only: buffer
// ownership, the SQP launch loop (eval kernel + Riccati/step kernel per iteration), HIP-event timing.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is opened with dlopen when a communicator is first asked for

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "oh_kernels.h"
#include "oh_jit.h"

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      return fail(OH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                     \
    }                                                                                                  \
  } while (0)

static thread_local OhLaunchOpts g_launch_opts;
OhLaunchOpts& oh_launch_opts() { return g_launch_opts; }
#define OH_PINNED_STAGE_BYTES (256 * 1024)

struct oh_handle {
  oh_problem_desc desc;
  std::vector<double> local_path;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evt0 = nullptr, evt1 = nullptr;
  bool have_chain = false;
  oh_chain chain_host;
  oh_chain* d_chain = nullptr;
  bool have_dyn = false;
  oh_dynamics dyn_host;
  oh_dynamics* d_dyn = nullptr;
  double* d_local_path = nullptr;
  // point-mass family
  oh_pointmass_desc pm{};
  PmParams PmP{};
  PmBuffers PmD{};
  // inequality rows of the position-tracking family
  bool have_guards = false;
  oh_guards guards{};
  GuardParams GP{};
  GuardBuffers GB{};
  void* gpool = nullptr;
  int gcap = 0;
  void* move_scr = nullptr;  // scratch of the compaction that moves every array (move_everything)
  size_t move_scr_bytes = 0;
  // tape family
  TapeParams TP{};
  int *d_tape_op = nullptr, *d_tape_a = nullptr, *d_tape_b = nullptr, *d_tape_rows = nullptr;
  double* d_tape_c = nullptr;
  double* d_tape_work = nullptr;
  double* d_tape_mult = nullptr;
  double* d_tape_h0 = nullptr;  // oh_tape_set_metric: initial metric of the limited-memory form [nx][nx]
  int tape_cap = 0;
  TapeJit tape_jit;
  TapeWave tape_wave;  // trajectory-sized tapes: one wavefront per instance (oh_tape_wave.hip)
  // dense QP family
  oh_qp_desc qp{};
  void* h_stage = nullptr;  // pinned mirror of the staging area for small oh_solve calls (OH_PINNED_STAGE_BYTES)
  double* d_qp_work = nullptr;
  double* d_qp_mult = nullptr;
  int qp_cap = 0;
  bool qp_tape = false;       // oh_qp_set_tape: p of a solve is the problem's parameter vector, the QP data is read off the tape (h->TP, d_tape_*) on the device
  double* d_qp_rows = nullptr;  // [B][qp_np] assembled [P | q | M | c | A | b]
  double* d_qp_val = nullptr;   // [TP.len][Bp] registers of the tape interpreter
  double* d_qp_f0 = nullptr;    // [B] f(0, p)
  int* d_qp_xdep = nullptr;     // indices of the tape's x-dependent instructions
  int qp_n_xdep = 0;
  int qp_tape_cap = 0;
  // inverse-kinematics family
  oh_ik_desc ik{};
  double* d_ik_mult = nullptr;
  int ik_cap = 0;
  // torque-MPC family
  oh_torque_desc tq{};
  TqParams TqP{};
  TqBuffers TqD{};
  void* tq_pool = nullptr;
  double* d_tq_mult = nullptr;
  double* d_tq_hc = nullptr;  // [tq_cap][T][TQ_HC] stored curvature terms (k_tq_curv)
  int tq_cap = 0;
  int tq_check = 4;  // the host looks at the running count every tq_check iterations
  std::array<double, 4> inv_saved{1.0, 16384.0, 1.0, -2.0};  // compaction, tail_threshold, tail_vel, free_persist (-2: unset) as they were before batch_invariant
  // solver buffers
  int cap_B = 0;
  FigBuffers D{};
  FigParams P{};
  void* pool = nullptr;
  size_t pool_bytes = 0;
  int last_B = 0;
  // staging for the host-buffer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;
  double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double timing_couple = 0;
  double rejects = 0;
  double tail_iters = 0;
  int* h_flag = nullptr;  // pinned
  // a large batch of the plain orientation-locked family is solved in parts on handles (streams, host threads) of their own: solve_split
  std::vector<oh_handle*> peers;
  std::vector<int> split_parts;  // instances per part of the last solve (empty: it was not split)
  bool pipe_last = false;        // the last solve was a pipelined oh_solve: its multipliers wait in d_pipe_mult (every chunk's, in instance order)
  double* d_pipe_mult = nullptr;
  size_t pipe_mult_cap = 0;      // doubles
  bool is_peer = false;
  bool compaction = true;
  int compact_carry = 1;     // compaction carries the pending trial along instead of restarting the survivors (k_carry_*)
  int tail_vel_threshold = 1 << 30;  // ... from this many instances down: always (see oh_solve_device)
  int lg_split = 1;          // orientation-locked handles with limit rows and no sphere rows: k_retract + k_evalb_lg instead of the fused k_eval_lg (OH_LG_SPLIT=0)
  int tail_vel = 1;          // velocity-limited handles drain in the persistent kernel too (k_tail_vel; OH_TAIL_VEL=0: batched launches to the end)
  int compact_sort = 1;      // order the survivors of a compaction by progress (k_scan_*)
  double compact_frac = 0.97;  // compact the batch once this fraction of it (or less) is still running (0.9 until the carried compaction
                               // stopped copying back: 0.95 ... 0.99 are +1 ... 2 % over 0.9 on two boxes, interleaved runs)
  double compact_frac_restart = 0.9;  // the same for the compaction that restarts the survivors (guarded handles, free family): it costs an evaluation
  int tail_threshold = 16384;  // hand the last instances to the persistent one-wave-per-instance kernel (round 2: with the kernel compiled for the
                               // chain 8192 against 2048 was +1.3 ... 3 % at B = 262 144 and -21 % on a batch of 4096; with four of its blocks per CU
                               // (two-pass exchange, 40 KB of LDS) 16 384 is level at B = 262 144 and -6 ... 13 % on batches of 16 ... 24 k)
  int sparse_check_below = 2048;  // OH_SPARSE_CHECK_BELOW (0: look every iteration whatever the batch)
  int fuse_couple = 1;        // OH_FUSE_COUPLE=0 restores the three-kernel iteration (k_couple between evaluation and sweep) for A/B runs
  int free_pcr_max = 1536;    // position-tracking family: K3 by cyclic reduction, one block per instance, while at most this many are in the launch
  // run-time specialised evaluation kernels of the orientation-locked figure-eight family (oh_jit.hip)
  int specialize = OH_SPECIALIZE_AUTO;  // option "specialize": 0 never, 1 at the first solve, 2 auto: at the first solve of >= specialize_min_B instances
  int specialize_min_B = 4096;
  const FigSpec* spec = nullptr;
  bool spec_failed = false;
  bool spec_cache_checked = false;  // automatic mode has looked for a cached code object once (reset with the constants)
  const FkSpec* fk_spec = nullptr;  // K1 for this chain (any handle with constants)
  bool fk_spec_failed = false;
  int specialize_min_units = 1 << 16;
  double spec_seconds = 0.0;  // of the last oh_specialize
  std::vector<int> prof_tags;  // per recorded event: 0 base marker, 1 after eval, 2 after step
  // options set by name (oh_set_option; OH_DEBUG_OPTIONS at creation) that are not backed by a field above: read with optv() where they are used
  std::map<std::string, double> opt;
  // host copy of the tape of an OH_PROBLEM_TAPE handle: the evaluator is rebuilt when an option that shapes it changes (tape_wave, tape_lbfgs, ...)
  std::vector<int> t_op, t_a, t_b, t_rows;
  std::vector<double> t_c;
  oh_tape_desc t_desc{};
};

extern "C" void oh_destroy(oh_handle* h);
extern "C" int oh_specialize(oh_handle* h);
static bool spec_applies(const oh_handle* h);
static bool spec_tail_vel_applies(const oh_handle* h);
static int specialize_fk(oh_handle* h);
extern "C" const char* oh_last_error(void) { return g_err.c_str(); }
extern "C" const char* oh_version(void) { return "optas_hip 0.2 (gfx950)"; }
extern "C" int oh_abi_version(void) { return OH_ABI_VERSION; }

// ---- per-handle options (round 5) ----------------------------------------------------------------------------------------------------
// Until round 4 forty OH_* environment variables were read inside the library: two handles of one process could not differ and a result
// depended on the caller's environment.  Now every knob is an option of ONE handle, set by name; the single environment hook left is
// OH_DEBUG_OPTIONS ("name=value,name=value"), applied to every handle when it is created (tools/, A/B runs).
static double optv(const oh_handle* h, const char* name, double dflt) {
  auto it = h->opt.find(name);
  return it == h->opt.end() ? dflt : it->second;
}
struct OptDoc { const char* name; double dflt; };
// map-backed options and their defaults (field-backed ones are handled in set_option_impl / oh_get_option)
static const OptDoc OPT_TABLE[] = {
    {"check_every", 1},        {"row_pad", 13},          {"retract_min", 1e-13}, {"hyb_switch", 1e-5},   {"relax", 1.5},          {"relax_from", 4},         {"settle_k", 1.0},   {"al_fuse", 1},   {"streams", 2},         {"split_min", 65536},  {"tq_split_min", 1024}, {"free_split_min", 256},
    {"free_bb", 1},            {"free_persist", -1},     {"free_cp_max", 512},   {"pm_wave_max", 20480}, {"qp_mode", -1},         {"tape_lds_max", 1 << 30},
    {"tape_wave", 1},          {"tape_lbfgs", -1},       {"tape_wave_nt", 256},  {"tape_wave_regs", -1}, {"tape_wave_hist", -1},  {"tq_stall", 25},
    {"tq_curv_after", 3},      {"tq_ftb", 0.995},        {"tq_theta_mu", 1.35},  {"tq_kappa_mu", 0.4},   {"tq_curv_from", 0.1},   {"tq_jac_dual", 0},
    {"tq_rebuild", 0.9},         {"compact_move_all", 1},  {"tq_curv_late", 1.0},  {"tq_kappa_eps", 10.0}, {"tq_max_back", 3},     {"tq_mu_dec", 1.0 / 3.0},     {"tq_ls_curv", 1},   {"tq_mu_dec_warm", 0.1}, {"tq_curv_lag", 3},
    {"tol", 0},                {"pipe", 1},               {"pipe_chunk", 32768},                {"invariant_compact_frac", 0.65}, {"invariant_split", 1}, {"invariant_move_slim", 1}, {"invariant_move_live", 1},
};
static int tape_configure(oh_handle* h);
static int set_option_impl(oh_handle* h, const std::string& name, double v) {
  if (name == "tail_threshold") h->tail_threshold = (int)v;
  else if (name == "free_pcr_max") h->free_pcr_max = (int)v;
  else if (name == "compaction") h->compaction = v != 0.0;
  else if (name == "compact_frac") h->compact_frac = h->compact_frac_restart = v;
  else if (name == "compact_sort") h->compact_sort = (int)v;
  else if (name == "compact_carry") h->compact_carry = (int)v;
  else if (name == "tail_vel") h->tail_vel = (int)v;
  else if (name == "lg_split") h->lg_split = (int)v;
  else if (name == "tail_vel_threshold") h->tail_vel_threshold = (int)v;
  else if (name == "fuse_couple") h->fuse_couple = v != 0.0;
  else if (name == "sparse_check_below") h->sparse_check_below = (int)v;
  else if (name == "specialize") h->specialize = v == 0.0 ? OH_SPECIALIZE_NEVER : (v == 1.0 ? OH_SPECIALIZE_ALWAYS : OH_SPECIALIZE_AUTO);
  else if (name == "tq_check") h->tq_check = v >= 1.0 ? (int)v : 1;
  else if (name == "batch_invariant") {
    // every instance takes the batched launches to the end, nothing restarts, nothing is handed to the persistent kernel; the batch is compacted only
    // by moving a survivor with everything it owns (move_everything): the path of an instance is then a function of the instance alone, bit for bit
    // (the price: 1.3 x the device time of a 262 144 batch -- 3.0 x before the moving compaction --, and small batches pay launches)
    // (the scheduling fields the option overrides are parked and come back when it is switched off: a threshold the user set earlier is not replaced
    //  by the default -- ADVICE r5)
    const bool was = optv(h, "batch_invariant", 0.0) != 0.0;
    h->opt[name] = v;
    if (v != 0.0 && !was) {
      h->inv_saved = {h->compaction ? 1.0 : 0.0, (double)h->tail_threshold, (double)h->tail_vel, h->opt.count("free_persist") ? h->opt["free_persist"] : -2.0};
      h->compaction = false; h->tail_threshold = 0; h->tail_vel = 0; h->opt["free_persist"] = 0;
    } else if (v == 0.0 && was) {
      h->compaction = h->inv_saved[0] != 0.0; h->tail_threshold = (int)h->inv_saved[1]; h->tail_vel = (int)h->inv_saved[2];
      if (h->inv_saved[3] == -2.0) h->opt.erase("free_persist"); else h->opt["free_persist"] = h->inv_saved[3];
    }
  } else {
    bool known = false;
    for (const OptDoc& d : OPT_TABLE) known = known || name == d.name;
    if (!known) return fail(OH_ERR_INVALID, "oh_set_option: unknown option '" + name + "'");
    h->opt[name] = v;
    if (h->desc.kind == OH_PROBLEM_TAPE && !h->t_op.empty() && name.rfind("tape_", 0) == 0 && name != "tape_lds_max") return tape_configure(h);
  }
  return OH_OK;
}
extern "C" int oh_set_option(oh_handle* h, const char* name, double value) {
  if (!h || !name) return fail(OH_ERR_INVALID, "oh_set_option: null argument");
  return set_option_impl(h, name, value);
}
extern "C" int oh_get_option(oh_handle* h, const char* name, double* value) {
  if (!h || !name || !value) return fail(OH_ERR_INVALID, "oh_get_option: null argument");
  const std::string n(name);
  if (n == "tail_threshold") *value = h->tail_threshold;
  else if (n == "free_pcr_max") *value = h->free_pcr_max;
  else if (n == "compaction") *value = h->compaction ? 1 : 0;
  else if (n == "compact_frac") *value = h->compact_frac;
  else if (n == "compact_sort") *value = h->compact_sort;
  else if (n == "compact_carry") *value = h->compact_carry;
  else if (n == "tail_vel") *value = h->tail_vel;
  else if (n == "lg_split") *value = h->lg_split;
  else if (n == "tail_vel_threshold") *value = h->tail_vel_threshold;
  else if (n == "fuse_couple") *value = h->fuse_couple ? 1 : 0;
  else if (n == "sparse_check_below") *value = h->sparse_check_below;
  else if (n == "specialize") *value = h->specialize;
  else if (n == "tq_check") *value = h->tq_check;
  else if (n == "batch_invariant") *value = optv(h, "batch_invariant", 0.0);
  else {
    for (const OptDoc& d : OPT_TABLE)
      if (n == d.name) { *value = optv(h, d.name, d.dflt); return OH_OK; }
    return fail(OH_ERR_INVALID, "oh_get_option: unknown option '" + n + "'");
  }
  return OH_OK;
}
// OH_DEBUG_OPTIONS="name=value,name=value": the one environment hook of the library, applied to a handle when it is created
static int apply_debug_options(oh_handle* h) {
  const char* e = getenv("OH_DEBUG_OPTIONS");
  if (!e || !*e) return OH_OK;
  std::string str(e);
  size_t pos = 0;
  while (pos < str.size()) {
    size_t end = str.find_first_of(",;", pos);
    if (end == std::string::npos) end = str.size();
    const std::string item = str.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) return fail(OH_ERR_INVALID, "OH_DEBUG_OPTIONS: expected name=value, got '" + item + "'");
    char* endp = nullptr;
    const std::string val = item.substr(eq + 1);
    const double v = strtod(val.c_str(), &endp);
    if (endp == val.c_str()) return fail(OH_ERR_INVALID, "OH_DEBUG_OPTIONS: '" + item + "' has no numeric value");
    if (const int rc = set_option_impl(h, item.substr(0, eq), v)) return rc;
  }
  return OH_OK;
}
// the launchers' share of the options, for the call that is starting
static void load_launch_opts(const oh_handle* h) {
  OhLaunchOpts& o = g_launch_opts;
  o = OhLaunchOpts{};
  o.free_bb = (int)optv(h, "free_bb", o.free_bb);
  o.free_cp_max = (int)optv(h, "free_cp_max", o.free_cp_max);
  o.pm_wave_max = (int)optv(h, "pm_wave_max", o.pm_wave_max);
  o.qp_mode = (int)optv(h, "qp_mode", o.qp_mode);
  o.tape_lds_max = (int)optv(h, "tape_lds_max", o.tape_lds_max);
  o.tape_wave_nt = (int)optv(h, "tape_wave_nt", o.tape_wave_nt);
  o.tape_wave_regs = (int)optv(h, "tape_wave_regs", o.tape_wave_regs);
  o.tape_wave_hist = (int)optv(h, "tape_wave_hist", o.tape_wave_hist);
}
// every oh_create*: a fresh handle with the debug options applied (nullptr: a bad OH_DEBUG_OPTIONS, oh_last_error says which)
static oh_handle* new_handle() {
  oh_handle* h = new oh_handle();
  if (apply_debug_options(h) != OH_OK) {
    delete h;
    return nullptr;
  }
  return h;
}

extern "C" int oh_device_count(int* n) {
  if (!n) return fail(OH_ERR_INVALID, "oh_device_count: null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return fail(OH_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *n = c;
  return OH_OK;
}
extern "C" int oh_set_device(int index) {
  HIPCHK(hipSetDevice(index));
  return OH_OK;
}
extern "C" int oh_device_malloc(void** ptr, size_t nbytes) {
  if (!ptr) return fail(OH_ERR_INVALID, "oh_device_malloc: null");
  HIPCHK(hipMalloc(ptr, nbytes ? nbytes : 8));
  return OH_OK;
}
extern "C" int oh_device_free(void* ptr) {
  HIPCHK(hipFree(ptr));
  return OH_OK;
}
extern "C" int oh_memcpy_h2d(void* dst, const void* src, size_t nbytes) {
  HIPCHK(hipMemcpy(dst, src, nbytes, hipMemcpyHostToDevice));
  return OH_OK;
}
extern "C" int oh_memcpy_d2h(void* dst, const void* src, size_t nbytes) {
  HIPCHK(hipMemcpy(dst, src, nbytes, hipMemcpyDeviceToHost));
  return OH_OK;
}
extern "C" int oh_device_synchronize(void) {
  HIPCHK(hipDeviceSynchronize());
  return OH_OK;
}

extern "C" int oh_create(const oh_problem_desc* desc, oh_handle** out) {
  if (!desc || !out) return fail(OH_ERR_INVALID, "oh_create: null argument");
  *out = nullptr;
  if (desc->kind != OH_PROBLEM_FIGURE_EIGHT && desc->kind != OH_PROBLEM_KINEMATICS)
    return fail(OH_ERR_INVALID, "oh_create: unknown problem kind");
  if (desc->kind == OH_PROBLEM_KINEMATICS) {
    if (desc->ndof < 1 || desc->ndof > OH_MAX_CHAIN) return fail(OH_ERR_INVALID, "oh_create: ndof must be in [1, OH_MAX_CHAIN]");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1)
      return fail(OH_ERR_HIP, "oh_create: no HIP device available (this library has no CPU path)");
    oh_handle* hk = new_handle();
    if (!hk) return OH_ERR_INVALID;
    hk->desc = *desc;
    hk->desc.local_path = nullptr;
    hipGetDevice(&hk->device);
    if (hipStreamCreate(&hk->stream) != hipSuccess || hipEventCreate(&hk->ev0) != hipSuccess ||
        hipEventCreate(&hk->ev1) != hipSuccess || hipEventCreate(&hk->evt0) != hipSuccess ||
        hipEventCreate(&hk->evt1) != hipSuccess || hipMalloc((void**)&hk->d_chain, sizeof(oh_chain)) != hipSuccess) {
      delete hk;
      return fail(OH_ERR_HIP, "oh_create: stream/event/allocation failed");
    }
    *out = hk;
    return OH_OK;
  }
  if (desc->T < 3 || desc->T > OH_MAX_T) return fail(OH_ERR_INVALID, "oh_create: T must be in [3, OH_MAX_T]");
  if (desc->ndof < (desc->lock_orientation ? 4 : 2) || desc->ndof > 8)
    return fail(OH_ERR_INVALID, "oh_create: the trajectory kernels are instantiated for 2 ... 8 actuated joints, the orientation-locked ones (three rows per "
                                "knot, null space of ndof - 3 dimensions) for 4 ... 8");
  if (!(desc->dt > 0.0)) return fail(OH_ERR_INVALID, "oh_create: dt must be positive");
  if (!desc->local_path) return fail(OH_ERR_INVALID, "oh_create: local_path is null");
  if (desc->hessian != OH_HESSIAN_GAUSS_NEWTON && desc->hessian != OH_HESSIAN_EXACT && desc->hessian != OH_HESSIAN_HYBRID)
    return fail(OH_ERR_INVALID, "oh_create: bad hessian mode");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(OH_ERR_HIP, "oh_create: no HIP device available (this library has no CPU path)");
  oh_handle* h = new_handle();
  if (!h) return OH_ERR_INVALID;
  h->desc = *desc;
  h->local_path.assign(desc->local_path, desc->local_path + 3 * (size_t)desc->T);
  h->desc.local_path = nullptr;
  if (h->desc.max_iter <= 0) h->desc.max_iter = 200;
  if (!(h->desc.tol > 0.0)) h->desc.tol = 1e-6;
  if (!(h->desc.tol_feas > 0.0)) h->desc.tol_feas = 1e-9;
  if (h->desc.mu0 < 0.0) h->desc.mu0 = 0.0;
  hipGetDevice(&h->device);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
      hipEventCreate(&h->ev1) != hipSuccess || hipEventCreate(&h->evt0) != hipSuccess ||
      hipEventCreate(&h->evt1) != hipSuccess) {
    delete h;
    return fail(OH_ERR_HIP, "oh_create: stream/event creation failed");
  }
  if (hipMalloc((void**)&h->d_chain, sizeof(oh_chain)) != hipSuccess ||
      hipMalloc((void**)&h->d_local_path, sizeof(double) * 3 * (size_t)desc->T) != hipSuccess ||
      hipHostMalloc((void**)&h->h_flag, sizeof(int)) != hipSuccess) {
    delete h;
    return fail(OH_ERR_HIP, "oh_create: device allocation failed");
  }
  hipMemcpy(h->d_local_path, h->local_path.data(), sizeof(double) * 3 * (size_t)desc->T, hipMemcpyHostToDevice);
  *out = h;
  return OH_OK;
}

extern "C" int oh_create_pointmass(const oh_pointmass_desc* desc, oh_handle** out) {
  if (!desc || !out) return fail(OH_ERR_INVALID, "oh_create_pointmass: null argument");
  *out = nullptr;
  if (desc->T < 2 || desc->T > OH_MAX_T) return fail(OH_ERR_INVALID, "oh_create_pointmass: T must be in [2, OH_MAX_T]");
  if (!(desc->dt > 0.0) || !(desc->w_acc > 0.0) || !(desc->ylim > 0.0) || !(desc->vlim > 0.0) || !(desc->safe >= 0.0) || !(desc->w_vel >= 0.0) ||
      (desc->fix_final_velocity && desc->T < 3))
    return fail(OH_ERR_INVALID, "oh_create_pointmass: dt, w_acc, ylim, vlim must be positive and safe non-negative");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1)
    return fail(OH_ERR_HIP, "oh_create_pointmass: no HIP device available (this library has no CPU path)");
  oh_handle* h = new_handle();
  if (!h) return OH_ERR_INVALID;
  h->desc = oh_problem_desc{};
  h->desc.kind = OH_PROBLEM_POINT_MASS_MPC;
  h->desc.T = desc->T;
  h->desc.ndof = 2;
  h->pm = *desc;
  if (h->pm.max_iter <= 0) h->pm.max_iter = 100;
  if (!(h->pm.tol > 0.0)) h->pm.tol = 1e-8;
  hipGetDevice(&h->device);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipEventCreate(&h->evt0) != hipSuccess || hipEventCreate(&h->evt1) != hipSuccess) {
    delete h;
    return fail(OH_ERR_HIP, "oh_create_pointmass: stream/event creation failed");
  }
  *out = h;
  return OH_OK;
}

static int tape_validate(const oh_tape_desc* d, const char* who) {
  const std::string w(who);
  if (d->nx < 1 || d->nx > OH_TAPE_MAX_N || d->np < 0 || d->len < 1 || d->len > OH_TAPE_MAX_LEN || d->n_ineq < 0 || d->n_eq < 0 || !d->op || !d->a ||
      !d->b || !d->c || (d->n_ineq + d->n_eq > 0 && !d->rows) || d->out_cost < 0 || d->out_cost >= d->len)
    return fail(OH_ERR_INVALID, (w + ": bad sizes or null arrays").c_str());
  for (int i = 0; i < d->len; ++i) {
    const int o = d->op[i];
    const bool two = (o >= 3 && o <= 6) || o == 10 || (o >= 15 && o <= 20) || (o >= 22 && o <= 24);
    const bool one = o == 7 || o == 8 || o == 9 || o == 11 || o == 12 || o == 13 || o == 14 || o == 21 || o == 25 || o == 26;
    if (o < 0 || o > 26 || (o == 1 && (d->a[i] < 0 || d->a[i] >= d->nx)) || (o == 2 && (d->a[i] < 0 || d->a[i] >= d->np)) ||
        ((one || two) && (d->a[i] < 0 || d->a[i] >= i)) || (two && (d->b[i] < 0 || d->b[i] >= i)))
      return fail(OH_ERR_INVALID, (w + ": malformed instruction (operands must be earlier registers / valid indices)").c_str());
  }
  for (int i = 0; i < d->n_ineq + d->n_eq; ++i)
    if (d->rows[i] < 0 || d->rows[i] >= d->len) return fail(OH_ERR_INVALID, (w + ": row register out of range").c_str());
  return OH_OK;
}

static TapeParams tape_params(const oh_tape_desc* d, const int lbfgs_opt = -1) {
  // dense inverse-Hessian BFGS up to 48 variables (n^2 doubles per instance), the limited-memory form with 12 pairs beyond (option tape_lbfgs overrides:
  // 0 forces the dense matrix, m > 0 the m-pair form at any size)
  int lb = d->nx > 48 ? 12 : 0;
  if (lbfgs_opt >= 0) lb = lbfgs_opt > 64 ? 64 : lbfgs_opt;
  return TapeParams{d->len, d->nx, d->np, d->n_ineq, d->n_eq, d->out_cost, d->max_iter > 0 ? d->max_iter : 2000, d->tol > 0.0 ? d->tol : 1e-6,
                    d->tol_feas > 0.0 ? d->tol_feas : 1e-9, d->rho0 > 0.0 ? d->rho0 : 10.0, lb, nullptr};
}

extern "C" int oh_tape_compile(const oh_tape_desc* d, size_t* code_bytes, char* source, size_t source_cap, size_t* source_len) {
  if (!d) return fail(OH_ERR_INVALID, "oh_tape_compile: null argument");
  if (const int rc = tape_validate(d, "oh_tape_compile")) return rc;
  const std::string src = oh_tape_jit_source(tape_params(d), d->op, d->a, d->b, d->c, d->rows);
  if (source_len) *source_len = src.size();
  if (source && source_cap > 0) {
    const size_t k = src.size() < source_cap - 1 ? src.size() : source_cap - 1;
    memcpy(source, src.data(), k);
    source[k] = 0;
  }
  std::vector<char> code;
  std::string err;
  if (oh_tape_jit_compile(src, &code, &err)) return fail(OH_ERR_HIP, ("oh_tape_compile: " + err).c_str());
  if (code_bytes) *code_bytes = code.size();
  return OH_OK;
}

// (Re)build the evaluator of an OH_PROBLEM_TAPE handle from its host copy of the tape and its options: the wavefront-per-instance schedule where it
// applies (tape_wave != 0, limited-memory regime, LDS fit), otherwise generated code (desc.jit) or the interpreter.
static int tape_configure(oh_handle* h) {
  oh_tape_desc d = h->t_desc;
  d.op = h->t_op.data(); d.a = h->t_a.data(); d.b = h->t_b.data(); d.c = h->t_c.data(); d.rows = h->t_rows.empty() ? nullptr : h->t_rows.data();
  h->TP = tape_params(&d, (int)optv(h, "tape_lbfgs", -1.0));
  h->TP.h0 = h->d_tape_h0;  // (a metric handed over before an option rebuilt the evaluator stays)
  load_launch_opts(h);  // oh_tape_wave_build reads tape_wave_nt / _regs / _hist
  oh_tape_wave_release(&h->tape_wave);
  h->tape_wave = TapeWave{};
  if (h->tape_cap) {  // the work arrays were sized for the other evaluator
    if (h->d_tape_work) hipFree(h->d_tape_work);
    if (h->d_tape_mult) hipFree(h->d_tape_mult);
    h->d_tape_work = h->d_tape_mult = nullptr;
    h->tape_cap = 0;
  }
  int lds_limit = 0;
  if (optv(h, "tape_wave", 1.0) != 0.0 && hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) == hipSuccess) {
    std::string err;
    if (oh_tape_wave_build(h->TP, d.op, d.a, d.b, d.c, d.rows, (size_t)lds_limit, &h->tape_wave, &err)) return fail(OH_ERR_HIP, ("oh_create_tape: " + err).c_str());
  }
  if (d.jit && !h->tape_wave.ready && !h->tape_jit.fn) {
    std::vector<char> code;
    std::string err;
    const std::string src = oh_tape_jit_source(h->TP, d.op, d.a, d.b, d.c, d.rows);
    bool ok = !oh_tape_jit_compile(src, &code, &err) && !oh_tape_jit_load(code, &h->tape_jit, &err);
    if (!ok && !code.empty()) {  // an object that compiled (or came from the disk cache) and does not load: drop it, recompile once (as oh_jit_figure8 does)
      oh_tape_jit_forget(src);
      code.clear();
      ok = !oh_tape_jit_compile(src, &code, &err) && !oh_tape_jit_load(code, &h->tape_jit, &err);
    }
    if (!ok) return fail(OH_ERR_HIP, ("oh_create_tape: " + err).c_str());
  }
  return OH_OK;
}

extern "C" int oh_create_tape(const oh_tape_desc* d, oh_handle** out) {
  if (!d || !out) return fail(OH_ERR_INVALID, "oh_create_tape: null argument");
  *out = nullptr;
  if (const int rc = tape_validate(d, "oh_create_tape")) return rc;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return fail(OH_ERR_HIP, "oh_create_tape: no HIP device available (this library has no CPU path)");
  oh_handle* h = new_handle();
  if (!h) return OH_ERR_INVALID;
  h->desc = oh_problem_desc{};
  h->desc.kind = OH_PROBLEM_TAPE;
  h->desc.T = 1;
  h->desc.ndof = d->nx;
  hipGetDevice(&h->device);
  // the tape stays with the handle: the evaluator is rebuilt when an option that shapes it changes
  h->t_op.assign(d->op, d->op + d->len);
  h->t_a.assign(d->a, d->a + d->len);
  h->t_b.assign(d->b, d->b + d->len);
  h->t_c.assign(d->c, d->c + d->len);
  h->t_rows.assign(d->rows ? d->rows : d->op, (d->rows ? d->rows : d->op) + (d->rows ? d->n_ineq + d->n_eq : 0));
  h->t_desc = *d;
  if (d->no_wave && !h->opt.count("tape_wave")) h->opt["tape_wave"] = 0.0;
  if (d->lbfgs != 0 && !h->opt.count("tape_lbfgs")) h->opt["tape_lbfgs"] = d->lbfgs > 0 ? (double)d->lbfgs : 0.0;
  if (const int rc = tape_configure(h)) {
    oh_destroy(h);
    return rc;
  }
  const size_t li = sizeof(int) * (size_t)d->len, ld = sizeof(double) * (size_t)d->len, lr = sizeof(int) * (size_t)(d->n_ineq + d->n_eq + 1);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipEventCreate(&h->evt0) != hipSuccess || hipEventCreate(&h->evt1) != hipSuccess || hipMalloc((void**)&h->d_tape_op, li) != hipSuccess ||
      hipMalloc((void**)&h->d_tape_a, li) != hipSuccess || hipMalloc((void**)&h->d_tape_b, li) != hipSuccess ||
      hipMalloc((void**)&h->d_tape_c, ld) != hipSuccess || hipMalloc((void**)&h->d_tape_rows, lr) != hipSuccess) {
    oh_destroy(h);  // (releases the wavefront schedule / the generated module built above as well)
    return fail(OH_ERR_HIP, "oh_create_tape: stream/event/allocation failed");
  }
  hipMemcpy(h->d_tape_op, d->op, li, hipMemcpyHostToDevice);
  hipMemcpy(h->d_tape_a, d->a, li, hipMemcpyHostToDevice);
  hipMemcpy(h->d_tape_b, d->b, li, hipMemcpyHostToDevice);
  hipMemcpy(h->d_tape_c, d->c, ld, hipMemcpyHostToDevice);
  if (d->n_ineq + d->n_eq > 0) hipMemcpy(h->d_tape_rows, d->rows, sizeof(int) * (size_t)(d->n_ineq + d->n_eq), hipMemcpyHostToDevice);
  *out = h;
  return OH_OK;
}

static int ensure_stage(oh_handle* h, size_t bytes);
extern "C" int oh_tape_probe(oh_handle* h, int B, const double* x, const double* p, int n_regs, const int* regs, double* val, const double* seeds, double* adj,
                             double* grad) {
  if (!h || !x) return fail(OH_ERR_INVALID, "oh_tape_probe: null argument");
  if (h->desc.kind != OH_PROBLEM_TAPE) return fail(OH_ERR_STATE, "oh_tape_probe: handle is not an OH_PROBLEM_TAPE problem");
  const TapeParams& T = h->TP;
  if (B < 1 || n_regs < 0 || (n_regs > 0 && !regs) || (T.np > 0 && !p)) return fail(OH_ERR_INVALID, "oh_tape_probe: bad sizes");
  for (int i = 0; i < n_regs; ++i)
    if (regs[i] < 0 || regs[i] >= T.len) return fail(OH_ERR_INVALID, "oh_tape_probe: register out of range");
  HIPCHK(hipSetDevice(h->device));
  const int Bp = (B + 63) / 64 * 64;
  const int nrow = T.n_ineq + T.n_eq;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t b_x = sizeof(double) * (size_t)T.nx * B, b_p = sizeof(double) * (size_t)(T.np > 0 ? T.np : 1) * B, b_r = sizeof(int) * (size_t)(n_regs + 1),
               b_v = sizeof(double) * (size_t)(n_regs + 1) * B, b_s = sizeof(double) * (size_t)(1 + nrow) * B, b_w = sizeof(double) * (2 * (size_t)T.len + 2 * (size_t)T.nx) * Bp;
  int rc = ensure_stage(h, al(b_x) * 2 + al(b_p) + al(b_r) + 2 * al(b_v) + al(b_s) + al(b_w));
  if (rc) return rc;
  char* base = (char*)h->stage;
  double* d_x = (double*)base; base += al(b_x);
  double* d_g = (double*)base; base += al(b_x);
  double* d_p = (double*)base; base += al(b_p);
  int* d_r = (int*)base; base += al(b_r);
  double* d_v = (double*)base; base += al(b_v);
  double* d_a = (double*)base; base += al(b_v);
  double* d_s = (double*)base; base += al(b_s);
  double* d_w = (double*)base;
  hipStream_t s = h->stream;
  HIPCHK(hipMemcpyAsync(d_x, x, b_x, hipMemcpyHostToDevice, s));
  if (T.np > 0) HIPCHK(hipMemcpyAsync(d_p, p, sizeof(double) * (size_t)T.np * B, hipMemcpyHostToDevice, s));
  if (n_regs > 0) HIPCHK(hipMemcpyAsync(d_r, regs, sizeof(int) * (size_t)n_regs, hipMemcpyHostToDevice, s));
  if (seeds) HIPCHK(hipMemcpyAsync(d_s, seeds, b_s, hipMemcpyHostToDevice, s));
  oh_launch_tape_probe(s, T, h->d_tape_op, h->d_tape_a, h->d_tape_b, h->d_tape_c, h->d_tape_rows, B, Bp, d_x, d_p, d_w, n_regs, d_r, val ? d_v : nullptr,
                       seeds ? d_s : nullptr, (seeds && adj) ? d_a : nullptr, (seeds && grad) ? d_g : nullptr);
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  if (val && n_regs > 0) HIPCHK(hipMemcpy(val, d_v, sizeof(double) * (size_t)n_regs * B, hipMemcpyDeviceToHost));
  if (seeds && adj && n_regs > 0) HIPCHK(hipMemcpy(adj, d_a, sizeof(double) * (size_t)n_regs * B, hipMemcpyDeviceToHost));
  if (seeds && grad) HIPCHK(hipMemcpy(grad, d_g, b_x, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_tape_set_metric(oh_handle* h, const double* H0) {
  if (!h) return fail(OH_ERR_INVALID, "oh_tape_set_metric: null argument");
  if (h->desc.kind != OH_PROBLEM_TAPE) return fail(OH_ERR_INVALID, "oh_tape_set_metric: not an OH_PROBLEM_TAPE handle");
  HIPCHK(hipSetDevice(h->device));
  const size_t n = (size_t)h->TP.nx;
  if (!H0) {
    if (h->d_tape_h0) hipFree(h->d_tape_h0);
    h->d_tape_h0 = nullptr;
    h->TP.h0 = nullptr;
    return OH_OK;
  }
  // symmetric with a positive diagonal is what can be checked here without factorising; a matrix that is not positive definite costs the solver a
  // reset to steepest descent whenever the direction it gives does not descend (oh_tape_solver.h), never a wrong answer
  for (size_t i = 0; i < n; ++i) {
    if (!(H0[i * n + i] > 0.0)) return fail(OH_ERR_INVALID, "oh_tape_set_metric: diagonal entry not positive");
    for (size_t j = 0; j < i; ++j) {
      const double a = H0[i * n + j], b = H0[j * n + i];
      if (!(fabs(a - b) <= 1e-10 * (fabs(a) + fabs(b)) + 1e-300)) return fail(OH_ERR_INVALID, "oh_tape_set_metric: matrix not symmetric");
    }
  }
  if (!h->d_tape_h0) HIPCHK(hipMalloc((void**)&h->d_tape_h0, sizeof(double) * n * n));
  HIPCHK(hipMemcpy(h->d_tape_h0, H0, sizeof(double) * n * n, hipMemcpyHostToDevice));
  h->TP.h0 = h->d_tape_h0;
  return OH_OK;
}

static int tape_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt, void* d_iters, void* d_status) {
  HIPCHK(hipSetDevice(h->device));
  const int Bp = (B + 63) / 64 * 64;
  if (Bp > h->tape_cap) {
    if (h->d_tape_work) hipFree(h->d_tape_work);
    if (h->d_tape_mult) hipFree(h->d_tape_mult);
    h->d_tape_work = h->d_tape_mult = nullptr;
    h->tape_cap = 0;
    if (!h->tape_wave.ready) HIPCHK(hipMalloc((void**)&h->d_tape_work, sizeof(double) * oh_tape_work_rows(h->TP, h->tape_jit.fn != nullptr) * Bp));
    HIPCHK(hipMalloc((void**)&h->d_tape_mult, sizeof(double) * (size_t)(h->TP.n_ineq + h->TP.n_eq + 1) * Bp));
    h->tape_cap = Bp;
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  if (h->tape_wave.ready)
    HIPCHK(oh_launch_tape_wave(h->stream, h->tape_wave, h->TP, B, (const double*)d_x0, (const double*)d_p, (double*)d_x, (double*)d_f, (double*)d_kkt, (int*)d_iters,
                               (int*)d_status, h->d_tape_mult));
  else if (h->tape_jit.fn)
    HIPCHK(oh_launch_tape_jit(h->stream, h->tape_jit, h->TP, B, h->tape_cap, (const double*)d_x0, (const double*)d_p, h->d_tape_work, (double*)d_x, (double*)d_f,
                              (double*)d_kkt, (int*)d_iters, (int*)d_status, h->d_tape_mult));
  else
    oh_launch_tape_solve(h->stream, h->TP, h->d_tape_op, h->d_tape_a, h->d_tape_b, h->d_tape_c, h->d_tape_rows, B, h->tape_cap, (const double*)d_x0,
                         (const double*)d_p, h->d_tape_work, (double*)d_x, (double*)d_f, (double*)d_kkt, (int*)d_iters, (int*)d_status, h->d_tape_mult);
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms;
  h->timing[5] = 1;
  h->last_B = B;
  return OH_OK;
}

extern "C" int oh_create_qp(const oh_qp_desc* desc, oh_handle** out) {
  if (!desc || !out) return fail(OH_ERR_INVALID, "oh_create_qp: null argument");
  *out = nullptr;
  if (desc->n < 1 || desc->n > OH_QP_MAX_N || desc->m < 0 || desc->m > OH_QP_MAX_M || desc->me < 0 || desc->me > OH_QP_MAX_ME || desc->me > desc->n)
    return fail(OH_ERR_INVALID, "oh_create_qp: sizes out of range (n <= 32, m <= 256, me <= min(32, n))");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return fail(OH_ERR_HIP, "oh_create_qp: no HIP device available (this library has no CPU path)");
  oh_handle* h = new_handle();
  if (!h) return OH_ERR_INVALID;
  h->desc = oh_problem_desc{};
  h->desc.kind = OH_PROBLEM_QP;
  h->desc.T = 1;
  h->desc.ndof = desc->n;
  h->qp = *desc;
  if (h->qp.max_iter <= 0) h->qp.max_iter = 100;
  if (!(h->qp.tol > 0.0)) h->qp.tol = 1e-9;
  hipGetDevice(&h->device);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipEventCreate(&h->evt0) != hipSuccess || hipEventCreate(&h->evt1) != hipSuccess) {
    delete h;
    return fail(OH_ERR_HIP, "oh_create_qp: stream/event creation failed");
  }
  *out = h;
  return OH_OK;
}

static size_t qp_np(const oh_qp_desc& q) { return (size_t)q.n * q.n + q.n + (size_t)q.m * q.n + q.m + (size_t)q.me * q.n + q.me; }

extern "C" int oh_qp_set_tape(oh_handle* h, const oh_tape_desc* d) {
  if (!h || !d) return fail(OH_ERR_INVALID, "oh_qp_set_tape: null argument");
  if (h->desc.kind != OH_PROBLEM_QP) return fail(OH_ERR_STATE, "oh_qp_set_tape: not an OH_PROBLEM_QP handle");
  if (const int rc = tape_validate(d, "oh_qp_set_tape")) return rc;
  if (d->nx != h->qp.n || d->n_ineq != h->qp.m || d->n_eq != h->qp.me)
    return fail(OH_ERR_INVALID, "oh_qp_set_tape: the tape's nx / n_ineq / n_eq differ from the handle's n / m / me");
  HIPCHK(hipSetDevice(h->device));
  for (void** q2 : {(void**)&h->d_tape_op, (void**)&h->d_tape_a, (void**)&h->d_tape_b, (void**)&h->d_tape_c, (void**)&h->d_tape_rows}) {
    if (*q2) hipFree(*q2);
    *q2 = nullptr;
  }
  h->qp_tape = false;
  const size_t li = sizeof(int) * (size_t)d->len, ld = sizeof(double) * (size_t)d->len, lr = sizeof(int) * (size_t)(d->n_ineq + d->n_eq + 1);
  HIPCHK(hipMalloc((void**)&h->d_tape_op, li));
  HIPCHK(hipMalloc((void**)&h->d_tape_a, li));
  HIPCHK(hipMalloc((void**)&h->d_tape_b, li));
  HIPCHK(hipMalloc((void**)&h->d_tape_c, ld));
  HIPCHK(hipMalloc((void**)&h->d_tape_rows, lr));
  HIPCHK(hipMemcpy(h->d_tape_op, d->op, li, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_tape_a, d->a, li, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_tape_b, d->b, li, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_tape_c, d->c, ld, hipMemcpyHostToDevice));
  if (d->n_ineq + d->n_eq > 0) HIPCHK(hipMemcpy(h->d_tape_rows, d->rows, sizeof(int) * (size_t)(d->n_ineq + d->n_eq), hipMemcpyHostToDevice));
  h->TP = tape_params(d);
  {
    // instructions whose value depends on x: the probes after the first re-run only these
    std::vector<char> dep((size_t)d->len, 0);
    std::vector<int> list;
    for (int i = 0; i < d->len; ++i) {
      const int o = d->op[i];
      const bool two = tape_op_arity(o) == 2, one = tape_op_arity(o) == 1;
      dep[i] = o == 1 || ((one || two) && dep[d->a[i]]) || (two && dep[d->b[i]]);
      if (dep[i]) list.push_back(i);
    }
    if (h->d_qp_xdep) hipFree(h->d_qp_xdep);
    h->d_qp_xdep = nullptr;
    h->qp_n_xdep = (int)list.size();
    HIPCHK(hipMalloc((void**)&h->d_qp_xdep, sizeof(int) * (list.size() + 1)));
    if (!list.empty()) HIPCHK(hipMemcpy(h->d_qp_xdep, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice));
  }
  if (h->qp_tape_cap) {  // the register file of another tape: size it again at the next solve
    for (double** q2 : {&h->d_qp_rows, &h->d_qp_val, &h->d_qp_f0}) {
      if (*q2) hipFree(*q2);
      *q2 = nullptr;
    }
    h->qp_tape_cap = 0;
  }
  h->qp_tape = true;
  return OH_OK;
}

static int qp_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt, void* d_iters, void* d_status) {
  HIPCHK(hipSetDevice(h->device));
  const oh_qp_desc& q = h->qp;
  QpParams Q{};
  Q.n = q.n; Q.m = q.m; Q.me = q.me; Q.np = (int)qp_np(q); Q.max_iter = q.max_iter; Q.tol = q.tol;
  Q.nwork = q.n + 2 * q.m + q.me + q.n * q.n + 2 * q.n + 2 * q.m + q.me * q.n + q.me * q.me + q.me + q.n;
  const int Bp = (B + 63) / 64 * 64;
  if (Bp > h->qp_cap) {
    if (h->d_qp_work) hipFree(h->d_qp_work);
    if (h->d_qp_mult) hipFree(h->d_qp_mult);
    h->d_qp_work = h->d_qp_mult = nullptr;
    h->qp_cap = 0;
    HIPCHK(hipMalloc((void**)&h->d_qp_work, sizeof(double) * (size_t)Q.nwork * Bp));
    HIPCHK(hipMalloc((void**)&h->d_qp_mult, sizeof(double) * (size_t)(q.m + q.me + 1) * Bp));
    h->qp_cap = Bp;
  }
  // register file of the tape interpreter: a lane per instance, or (a few instances: B <= 64) a lane per probe point of every instance
  const int Bv = B <= 64 ? (B * 64 > Bp ? B * 64 : Bp) : Bp;
  if (h->qp_tape && Bv > h->qp_tape_cap) {
    for (double** q2 : {&h->d_qp_rows, &h->d_qp_val, &h->d_qp_f0}) {
      if (*q2) hipFree(*q2);
      *q2 = nullptr;
    }
    h->qp_tape_cap = 0;
    HIPCHK(hipMalloc((void**)&h->d_qp_rows, sizeof(double) * (size_t)Q.np * Bv));
    HIPCHK(hipMalloc((void**)&h->d_qp_val, sizeof(double) * (size_t)h->TP.len * Bv));
    HIPCHK(hipMalloc((void**)&h->d_qp_f0, sizeof(double) * (size_t)Bv));
    h->qp_tape_cap = Bv;
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  if (h->qp_tape) {
    oh_launch_qp_assemble(h->stream, Q, h->TP, h->d_tape_op, h->d_tape_a, h->d_tape_b, h->d_tape_c, h->d_tape_rows, h->d_qp_xdep, h->qp_n_xdep, B, B <= 64 ? B * 64 : h->qp_tape_cap, (const double*)d_p,
                          h->d_qp_val, h->d_qp_rows, h->d_qp_f0);
    d_p = h->d_qp_rows;
  }
  oh_launch_qp_solve(h->stream, Q, B, h->qp_cap, (const double*)d_x0, (const double*)d_p, h->d_qp_work, (double*)d_x, (double*)d_f, (double*)d_kkt,
                     (int*)d_iters, (int*)d_status, h->d_qp_mult);
  if (h->qp_tape && d_f) oh_launch_qp_add_constant(h->stream, B, (double*)d_f, h->d_qp_f0);
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms;
  h->timing[5] = 1;
  h->last_B = B;
  return OH_OK;
}

extern "C" int oh_create_ik(const oh_ik_desc* desc, oh_handle** out) {
  if (!desc || !out) return fail(OH_ERR_INVALID, "oh_create_ik: null argument");
  *out = nullptr;
  if (desc->ndof < 2 || desc->ndof > 8) return fail(OH_ERR_INVALID, "oh_create_ik: kernels are instantiated for 2 ... 8 actuated joints");
  if (!(desc->w_nominal > 0.0)) return fail(OH_ERR_INVALID, "oh_create_ik: w_nominal must be positive");
  for (int i = 0; i < desc->ndof; ++i)
    if (!(desc->q_lo[i] <= desc->q_up[i])) return fail(OH_ERR_INVALID, "oh_create_ik: q_lo must not exceed q_up");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1)
    return fail(OH_ERR_HIP, "oh_create_ik: no HIP device available (this library has no CPU path)");
  oh_handle* h = new_handle();
  if (!h) return OH_ERR_INVALID;
  h->desc = oh_problem_desc{};
  h->desc.kind = OH_PROBLEM_IK;
  h->desc.T = 1;
  h->desc.ndof = desc->ndof;
  h->ik = *desc;
  if (h->ik.max_iter <= 0) h->ik.max_iter = 200;
  if (!(h->ik.tol > 0.0)) h->ik.tol = 1e-6;
  if (!(h->ik.tol_feas > 0.0)) h->ik.tol_feas = 1e-9;
  if (!(h->ik.rho0 > 0.0)) h->ik.rho0 = 100.0 * h->ik.w_nominal;
  hipGetDevice(&h->device);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipEventCreate(&h->evt0) != hipSuccess || hipEventCreate(&h->evt1) != hipSuccess ||
      hipMalloc((void**)&h->d_chain, sizeof(oh_chain)) != hipSuccess) {
    delete h;
    return fail(OH_ERR_HIP, "oh_create_ik: stream/event/allocation failed");
  }
  *out = h;
  return OH_OK;
}

static bool solver_chain_ok(const oh_chain& c);

extern "C" int oh_create_torque(const oh_torque_desc* desc, oh_handle** out) {
  if (!desc || !out) return fail(OH_ERR_INVALID, "oh_create_torque: null argument");
  *out = nullptr;
  if (desc->ndof < 2 || desc->ndof > 7) return fail(OH_ERR_INVALID, "oh_create_torque: kernels are instantiated for ndof 2 .. 7");
  if (desc->T < 2 || desc->T > OH_MAX_T) return fail(OH_ERR_INVALID, "oh_create_torque: T must be in [2, OH_MAX_T]");
  if (!(desc->dt > 0.0) || !(desc->w_tau > 0.0) || !(desc->w_path >= 0.0) || !(desc->w_vel >= 0.0))
    return fail(OH_ERR_INVALID, "oh_create_torque: dt and w_tau must be positive, w_path and w_vel non-negative");
  for (int i = 0; i < desc->ndof; ++i)
    if (!(desc->tau_lo[i] < desc->tau_up[i])) return fail(OH_ERR_INVALID, "oh_create_torque: tau_lo must be below tau_up");
  if (desc->vel_limits)
    for (int i = 0; i < desc->ndof; ++i)
      if (!(desc->dq_lo[i] < desc->dq_up[i])) return fail(OH_ERR_INVALID, "oh_create_torque: dq_lo must be below dq_up");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1)
    return fail(OH_ERR_HIP, "oh_create_torque: no HIP device available (this library has no CPU path)");
  oh_handle* h = new_handle();
  if (!h) return OH_ERR_INVALID;
  h->desc = oh_problem_desc{};
  h->desc.kind = OH_PROBLEM_TORQUE_MPC;
  h->desc.T = desc->T;
  h->desc.ndof = desc->ndof;
  h->tq = *desc;
  if (h->tq.max_iter <= 0) h->tq.max_iter = 300;
  if (!(h->tq.tol > 0.0)) h->tq.tol = 1e-6;
  if (!(h->tq.tol_compl > 0.0)) h->tq.tol_compl = 1e-8;
  if (!(h->tq.mu_barrier0 > 0.0)) h->tq.mu_barrier0 = 0.1;
  if (!(h->tq.mu0 >= 0.0)) h->tq.mu0 = 0.0;
  hipGetDevice(&h->device);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipEventCreate(&h->evt0) != hipSuccess || hipEventCreate(&h->evt1) != hipSuccess ||
      hipMalloc((void**)&h->d_chain, sizeof(oh_chain)) != hipSuccess || hipHostMalloc((void**)&h->h_flag, sizeof(int)) != hipSuccess) {
    oh_destroy(h);
    return fail(OH_ERR_HIP, "oh_create_torque: stream/event/allocation failed");
  }
  *out = h;
  return OH_OK;
}

static int tq_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt, void* d_iters, void* d_status,
                           const double mu_b0_warm = 0.0) {
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_solve_device: call oh_set_constants first");
  if (!h->have_dyn) return fail(OH_ERR_STATE, "oh_solve_device: call oh_set_dynamics first");
  if (!solver_chain_ok(h->chain_host) || h->chain_host.has_lead)
    return fail(OH_ERR_INVALID, "oh_solve_device: the solver needs a chain that covers every model joint in order");
  const int N = h->tq.ndof, T = h->tq.T;
  if (h->dyn_host.ndof != N) return fail(OH_ERR_INVALID, "oh_solve_device: the inverse-dynamics tables must have ndof + 1 bodies");
  HIPCHK(hipSetDevice(h->device));
  TqParams& P = h->TqP;
  P = TqParams{};
  P.T = T; P.N = N; P.max_iter = h->tq.max_iter;
  P.dt = h->tq.dt; P.w_path = h->tq.w_path; P.w_vel = h->tq.w_vel; P.w_tau = h->tq.w_tau;
  P.tol = h->tq.tol; P.tol_compl = h->tq.tol_compl; P.mu_b0 = mu_b0_warm > 0.0 ? mu_b0_warm : h->tq.mu_barrier0; P.mu0 = h->tq.mu0;
  // interior point: relaxed barrier below theta mu_b; monotone barrier update of Waechter & Biegler (2006, eq. 7) -- IPOPT's constants except theta_mu (1.35 for 1.5: the hardest of 8192 instances needs 127 steps instead of 198) and, round 5, kappa_mu (0.4 for 0.2: tools/gpu_tq_param_sweep.py); exact
  // curvature of the Lagrangian once the reduced gradient is below curv_from (oracle/torque_ipm.py:solve_torque_ipm has the same defaults)
  P.theta = 0.01; P.kappa_eps = 10.0; P.kappa_mu = 0.4; P.theta_mu = 1.35; P.curv_from = 0.1; P.curv_late = 1.0; P.curv_after = 3; P.tau_ftb = 0.995; P.max_back = 3; P.stall_max = 25;
  P.stall_max = (int)optv(h, "tq_stall", P.stall_max);  // options (oh_set_option)
  P.curv_after = (int)optv(h, "tq_curv_after", P.curv_after);
  P.tau_ftb = optv(h, "tq_ftb", P.tau_ftb);
  P.theta_mu = optv(h, "tq_theta_mu", P.theta_mu);
  P.kappa_mu = optv(h, "tq_kappa_mu", P.kappa_mu);
  P.curv_from = optv(h, "tq_curv_from", P.curv_from);  // 0: Gauss-Newton blocks throughout (A/B)
  P.curv_late = optv(h, "tq_curv_late", P.curv_late);
  P.kappa_eps = optv(h, "tq_kappa_eps", P.kappa_eps);
  // (a warm-started tick of oh_tq_rollout starts next to its optimum: there the damping comes down faster)
  P.mu_dec = mu_b0_warm > 0.0 ? optv(h, "tq_mu_dec_warm", 0.1) : optv(h, "tq_mu_dec", 1.0 / 3.0);
  P.ls_curv = (int)optv(h, "tq_ls_curv", 1);
  P.curv_lag = (int)optv(h, "tq_curv_lag", 3);
  P.max_back = (int)optv(h, "tq_max_back", P.max_back);
  P.vel = h->tq.vel_limits ? 1 : 0;
  // d tau / dz in closed form needs the tables to describe a rigid-body chain: unit joint axes that the joint-origin rotation leaves in place (then
  // the angular velocity the reference adds, iRp @ axis, is the axis its rotation turns about; models.py:1821-1823).  Otherwise: dual numbers.
  P.jac_closed_form = 1;
  for (int i = 0; i < N; ++i) {
    const double* a = h->dyn_host.axis[i];
    const double* R = h->dyn_host.R0[i];
    double dev = fabs(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] - 1.0);
    for (int k = 0; k < 3; ++k) dev = fmax(dev, fabs(R[k] * a[0] + R[3 + k] * a[1] + R[6 + k] * a[2] - a[k]));
    if (!(dev <= 1e-12)) P.jac_closed_form = 0;
  }
  if (optv(h, "tq_jac_dual", 0.0) != 0.0) P.jac_closed_form = 0;  // the dual-number path whatever the tables (A/B, tests)
  for (int i = 0; i < N; ++i) {
    P.tau_lo[i] = h->tq.tau_lo[i];
    P.tau_up[i] = h->tq.tau_up[i];
    P.dq_lo[i] = h->tq.dq_lo[i];
    P.dq_up[i] = h->tq.dq_up[i];
  }
  P.nx = 4 * N * T;
  P.np = 2 * N + 3 * T;
  TqBuffers& D = h->TqD;
  if (B > h->tq_cap) {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->tq_pool) hipFree(h->tq_pool);
    if (h->d_tq_mult) hipFree(h->d_tq_mult);
    if (h->d_tq_hc) hipFree(h->d_tq_hc);
    h->d_tq_hc = nullptr;
    h->tq_pool = nullptr;
    h->d_tq_mult = nullptr;
    h->tq_cap = 0;
    const size_t BT = (size_t)B * T;
    const size_t nd = 2 * BT * TQ_XS + 2 * BT * TQ_SD + 2 * BT * TQ_LAM + BT * TQ_GN + BT * 4 + 11 * (size_t)B;
    const size_t bytes = nd * sizeof(double) + (12 * (size_t)B + 16) * sizeof(int);
    HIPCHK(hipMalloc(&h->tq_pool, bytes));
    HIPCHK(hipMalloc((void**)&h->d_tq_mult, sizeof(double) * BT * 4 * N));  // effort rows, and room for the velocity rows
    HIPCHK(hipMalloc((void**)&h->d_tq_hc, sizeof(double) * BT * TQ_HC));
    HIPCHK(hipMemsetAsync(h->d_tq_hc, 0, sizeof(double) * BT * TQ_HC, h->stream));  // entries the adjoint never writes stay zero
    h->tq_cap = B;
  }
  {
    // carve for the capacity the pool was allocated with; D.B is the live batch (strides of the [slot][B][T] arrays follow it)
    const size_t BT = (size_t)B * T;
    double* d = (double*)h->tq_pool;
    auto take = [&](size_t n) { double* r = d; d += n; return r; };
    D.B = B;
    D.chain = h->d_chain;
    D.dyn = h->d_dyn;
    D.xs = take(2 * BT * TQ_XS);
    D.st = take(2 * BT * TQ_SD);
    D.lam = take(2 * BT * TQ_LAM);
    D.gains = take(BT * TQ_GN);
    D.goal = take(BT * 4);
    D.hc = h->d_tq_hc;
    D.f_cur = take(B); D.f_true = take(B); D.bsum = take(B); D.mu = take(B); D.nun = take(B); D.mub = take(B); D.stat = take(B);
    D.alpha = take(B); D.qk = take(B); D.ndx = take(B); D.viol = take(B);
    int* ip = (int*)d;
    D.cur = ip; ip += B; D.first = ip; ip += B; D.curv = ip; ip += B; D.status = ip; ip += B; D.iters = ip; ip += B; D.rejected = ip; ip += B;
    D.n_barrier = ip; ip += B;
    D.nrel = ip; ip += B;
    D.n_back = ip; ip += B;
    D.stall = ip; ip += B;
    D.curv_age = ip; ip += B;
    D.list = ip; ip += B;
    D.n_running = ip;
    D.n_list = ip + 1;
    D.n_run = B;
  }
  hipStream_t s = h->stream;
  HIPCHK(hipEventRecord(h->ev0, s));
  if (!oh_launch_tq_setup(s, P, D, (const double*)d_x0, (const double*)d_p)) return fail(OH_ERR_INVALID, "oh_solve_device: unsupported ndof");
  // iteration k: evaluate the pending trial of every running instance, then ratio test / Riccati sweep / next trial.  The host only looks at
  // the running count every tq_check iterations (instances that finished in between cost nothing: their lanes exit at once).
  int launched = 0;
  double work = 0.0;
  int running = B;
  const int cap = P.max_iter + 2;
  // oh_set_profiling(1): one event before the evaluation pair (k_tq_eval3 + k_tq_curv), one after it, one after k_tq_step, every iteration -- the
  // per-kernel device times behind the family's roofline object (tools/bench_configs.py); such a solve runs on one stream
  const bool prof = h->profiling;
  size_t ne = 0;
  if (prof) {
    const size_t need = 3 * (size_t)cap + 4;
    while (h->prof_events.size() < need) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      h->prof_events.push_back(e);
    }
  }
  while (launched < cap) {
    if (prof) HIPCHK(hipEventRecord(h->prof_events[ne++], s));
    oh_launch_tq_eval(s, P, D);  // also resets the running count
    if (prof) HIPCHK(hipEventRecord(h->prof_events[ne++], s));
    oh_launch_tq_step(s, P, D);
    if (prof) HIPCHK(hipEventRecord(h->prof_events[ne++], s));
    ++launched;
    work += running;
    if (launched % h->tq_check == 0 || launched == cap) {
      HIPCHK(hipMemcpyAsync(h->h_flag, D.n_running, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      running = *h->h_flag;
      if (running == 0) break;
      const double rebuild = optv(h, "tq_rebuild", 0.9);
      if (running <= rebuild * D.n_run) {  // rebuild the list of running instances: grids shrink with the batch
        HIPCHK(hipMemsetAsync(D.n_list, 0, sizeof(int), s));
        oh_launch_tq_list(s, D);
        D.n_run = running;
      }
    }
  }
  oh_launch_tq_finalize(s, P, D, (double*)d_x, (double*)d_f, (double*)d_kkt, (int*)d_iters, (int*)d_status, h->d_tq_mult);
  HIPCHK(hipEventRecord(h->ev1, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  for (double& t : h->timing) t = 0.0;
  if (prof) {
    for (size_t i = 0; i + 3 <= ne; i += 3) {
      float a = 0.f, b2 = 0.f;
      hipEventElapsedTime(&a, h->prof_events[i], h->prof_events[i + 1]);
      hipEventElapsedTime(&b2, h->prof_events[i + 1], h->prof_events[i + 2]);
      h->timing[0] += a;
      h->timing[2] += b2;
    }
    h->timing[1] = h->timing[3] = launched;
  }
  h->timing[4] = ms;
  h->timing[5] = launched;
  h->timing[6] = work;
  h->timing_couple = 0;
  h->rejects = 0;
  h->tail_iters = 0;
  h->last_B = B;
  return OH_OK;
}

static int ik_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt, void* d_iters,
                           void* d_status) {
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_solve_device: call oh_set_constants first");
  if (!solver_chain_ok(h->chain_host))
    return fail(OH_ERR_INVALID, "oh_solve_device: the solver needs a chain that covers every model joint in order");
  HIPCHK(hipSetDevice(h->device));
  const int N = h->ik.ndof;
  if (B > h->ik_cap) {
    if (h->d_ik_mult) hipFree(h->d_ik_mult);
    h->d_ik_mult = nullptr;
    h->ik_cap = 0;
    HIPCHK(hipMalloc((void**)&h->d_ik_mult, sizeof(double) * (3 + 2 * (size_t)N) * B));
    h->ik_cap = B;
  }
  IkParams P{};
  P.ndof = N;
  P.max_iter = h->ik.max_iter;
  P.w = h->ik.w_nominal;
  P.tol = h->ik.tol;
  P.tol_feas = h->ik.tol_feas;
  P.rho0 = h->ik.rho0;
  for (int i = 0; i < N; ++i) {
    P.lo[i] = h->ik.q_lo[i];
    P.up[i] = h->ik.q_up[i];
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  if (!oh_launch_ik_solve(h->stream, h->d_chain, P, B, (const double*)d_x0, (const double*)d_p, (double*)d_x, (double*)d_f, (double*)d_kkt,
                          (int*)d_iters, (int*)d_status, h->d_ik_mult))
    return fail(OH_ERR_INVALID, "oh_solve_device: unsupported ndof");
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms;
  h->timing[5] = 1;
  h->last_B = B;
  return OH_OK;
}

static int pm_prepare(oh_handle* h, int B) {
  HIPCHK(hipSetDevice(h->device));
  const int T = h->pm.T;
  const int Bp = (B + 63) / 64 * 64;  // (padding the row stride like the trajectory families do was measured: no effect, the solve is latency bound)
  const size_t rows = 2 * (size_t)(T - 1) + 4 * (size_t)T + 9 * (size_t)T + 9 * (size_t)T + 8 * (size_t)(T - 1) + 2 * (size_t)(T - 1) +
                      4 * (size_t)T + 2 * (size_t)(T - 1);
  if (Bp > h->cap_B || !h->pool) {
    if (h->pool) hipFree(h->pool);
    h->pool = nullptr;
    hipError_t e = hipMalloc(&h->pool, rows * Bp * sizeof(double));
    if (e != hipSuccess) return fail(OH_ERR_HIP, std::string("device pool allocation failed: ") + hipGetErrorString(e));
    h->cap_B = Bp;
    h->PmD.Bp = Bp;
    double* d = (double*)h->pool;
    auto take = [&](size_t r) { double* o = d; d += r * Bp; return o; };
    h->PmD.a = take(2 * (size_t)(T - 1));
    h->PmD.X = take(4 * (size_t)T);
    h->PmD.s = take(9 * (size_t)T);
    h->PmD.lam = take(9 * (size_t)T);
    h->PmD.K = take(8 * (size_t)(T - 1));
    h->PmD.kk = take(2 * (size_t)(T - 1));
    h->PmD.dX = take(4 * (size_t)T);
    h->PmD.da = take(2 * (size_t)(T - 1));
  }
  h->PmD.B = B;
  h->PmP = PmParams{T, h->pm.dt, h->pm.w_acc, h->pm.ylim, h->pm.vlim, h->pm.safe * h->pm.safe, h->pm.tol, h->pm.max_iter,
                    h->pm.track_final_only ? 1 : 0, h->pm.w_vel, h->pm.fix_final_velocity ? 1 : 0};
  return OH_OK;
}

static int pm_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt, void* d_iters,
                           void* d_status) {
  int rc = pm_prepare(h, B);
  if (rc) return rc;
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  oh_launch_pm_solve(h->stream, h->PmP, h->PmD, (const double*)d_x0, (const double*)d_p, (double*)d_x, (double*)d_f, (double*)d_kkt,
                     (int*)d_iters, (int*)d_status);
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms;
  h->timing[5] = 1;
  h->last_B = B;
  return OH_OK;
}

static int validate_chain(const oh_handle* h, const oh_chain& c) {
  if (c.n_chain < 1 || c.n_chain > OH_MAX_CHAIN || c.ndof < c.n_chain || c.ndof > OH_MAX_CHAIN)
    return fail(OH_ERR_INVALID, "oh_set_constants: bad n_chain/ndof");
  for (int k = 0; k < c.n_chain; ++k) {
    if (c.jtype[k] != 0 && c.jtype[k] != 1) return fail(OH_ERR_INVALID, "oh_set_constants: joint type not supported");
    if (c.qidx[k] < 0 || c.qidx[k] >= c.ndof) return fail(OH_ERR_INVALID, "oh_set_constants: qidx out of range");
  }
  if (c.ndof != h->desc.ndof) return fail(OH_ERR_INVALID, "oh_set_constants: chain.ndof != desc.ndof");
  return OH_OK;
}

// The parts of a split solve run on peer handles that were given the chain, dynamics and inequality rows of this handle when they were created
// (solve_split).  Any setter that changes one of those on the main handle destroys the peers: the next split solve builds them again from the new
// data (parts 1.. used to go on solving with the old chain / limits / obstacles, and guards set after a first split solve left the peers
// striding p by ndof only).
static void drop_peers(oh_handle* h) {
  if (h->is_peer) return;
  for (oh_handle* p : h->peers) oh_destroy(p);
  h->peers.clear();
  h->split_parts.clear();
}

// the handle has new constants: everything compiled for, or remembered about, the previous chain goes (one place for the three entry
// points that set constants)
static void adopt_chain(oh_handle* h, const oh_chain& c) {
  drop_peers(h);
  h->chain_host = c;
  h->have_chain = true;
  h->spec = nullptr;
  h->spec_failed = false;
  h->spec_cache_checked = false;
  h->fk_spec = nullptr;
  h->fk_spec_failed = false;
}

extern "C" int oh_set_constants(oh_handle* h, const oh_chain* chain) {
  if (!h || !chain) return fail(OH_ERR_INVALID, "oh_set_constants: null argument");
  int rc = validate_chain(h, *chain);
  if (rc) return rc;
  HIPCHK(hipMemcpy(h->d_chain, chain, sizeof(oh_chain), hipMemcpyHostToDevice));
  adopt_chain(h, *chain);
  return OH_OK;
}

extern "C" int oh_set_constants_device(oh_handle* h, const void* d_chain, size_t nbytes) {
  if (!h || !d_chain) return fail(OH_ERR_INVALID, "oh_set_constants_device: null argument");
  if (nbytes != sizeof(oh_chain)) return fail(OH_ERR_INVALID, "oh_set_constants_device: nbytes != sizeof(oh_chain)");
  oh_chain tmp;
  HIPCHK(hipMemcpy(&tmp, d_chain, sizeof(oh_chain), hipMemcpyDeviceToHost));
  int rc = validate_chain(h, tmp);
  if (rc) return rc;
  HIPCHK(hipMemcpy(h->d_chain, d_chain, sizeof(oh_chain), hipMemcpyDeviceToDevice));
  adopt_chain(h, tmp);
  return OH_OK;
}

static bool solver_chain_ok(const oh_chain& c) {
  if (c.n_chain != c.ndof) return false;
  for (int k = 0; k < c.n_chain; ++k)
    if (c.qidx[k] != k) return false;
  return true;
}

// row stride of the SoA stage arrays for a batch of B (see ensure_capacity)
static int row_stride(const oh_handle* h, int B) {
  int Bp = (B + 63) / 64 * 64;
  if (Bp >= 4096) Bp += 64 * (int)optv(h, "row_pad", 13);
  return Bp;
}
static bool stage_fits(const oh_handle* h, int B) {
  const int N = h->desc.ndof, NZ = h->desc.lock_orientation ? N - 3 : N, T = h->desc.T;
  if (!h->desc.lock_orientation) return true;
  if (!(((double)T * NZ * NZ + 1.0) * (double)row_stride(h, B) * 8.0 < 4294967296.0)) return false;
  if (!(((double)T * (3 * N - 3) + 1.0) * (double)row_stride(h, B) * 8.0 < 4294967296.0)) return false;  // slot offset of the Householder vectors
  // handles with the coupling folded in: the sweep addresses G of either slot and the spare as one 32-bit offset from the lowest of the three
  // adjacent arrays (oh_figure8_units.h:step_instance_zc): 3 T N doubles per instance have to stay below 4 GiB (round 3: found by the batch
  // sweep -- 524 288 instances ran through with wrapped offsets and converged nowhere; they are now refused here and split by the host)
  const bool zc = h->fuse_couple && !h->have_guards && !h->chain_host.has_lead;
  return !zc || 3.0 * (double)T * N * (double)row_stride(h, B) * 8.0 < 4294967296.0;
}

// Largest batch one oh_solve / oh_solve_device call of this handle takes (0: no bound of the library's own, memory permitting): hosts chunk
// bigger batches.  The bound is the 32-bit byte offset of the sweep kernels' stage arrays, row pad included.
extern "C" int oh_max_batch(oh_handle* h, int* out) {
  if (!h || !out) return fail(OH_ERR_INVALID, "oh_max_batch: null argument");
  *out = 0;
  if (h->desc.kind != OH_PROBLEM_FIGURE_EIGHT || !h->desc.lock_orientation) return OH_OK;
  int lo = 64, hi = 1 << 30;  // largest multiple of 64 that fits, by bisection over the predicate ensure_capacity applies
  if (!stage_fits(h, lo)) return OH_OK;
  while (hi - lo > 64) {
    const int mid = ((lo + (hi - lo) / 2) / 64) * 64;
    if (stage_fits(h, mid)) lo = mid; else hi = mid;
  }
  *out = lo;
  return OH_OK;
}

// carve the handle's device pool for B instances
static int ensure_capacity(oh_handle* h, int B) {
  const int N = h->desc.ndof, NZ = h->desc.lock_orientation ? N - 3 : N, T = h->desc.T;
  // Row stride of the SoA stage arrays: B rounded up to whole wavefronts, plus -- for large batches -- 13 x 512 B.  With a power-of-two batch
  // every row of every knot starts at a multiple of 2 MiB: the ~30 rows a sweep wave streams side by side then sit at the same offset of
  // their pages, and how the channel hash happens to spread them differed from process to process (k_couple 494 or 525 us per launch,
  // k_step 650...740, interleaved repeats on one box).  Off the power of two the spread is even: k_couple 455, k_step ~650, +3 % solves/s
  // (any of 1...13 x 512 B does it; OH_ROW_PAD overrides, 0 restores the old layout).
  const int Bp = row_stride(h, B);
  // the sweep kernels address one slot of a stage array with a 32-bit byte offset (buffer resources, oh_kernels.hip): the largest such
  // array, T x NZ^2 doubles per instance, has to stay below 4 GiB (about 670 000 instances at T = 50, N = 7; oh_max_batch says exactly)
  if (!stage_fits(h, B)) return fail(OH_ERR_INVALID, "batch too large for one call (stage array beyond 4 GiB): split the batch (oh_max_batch)");
  if (B <= h->cap_B && h->pool) {
    h->D.B = B;
    // keep the Bp the pool was carved with (stride), only the active count changes
    return OH_OK;
  }
  if (h->pool) {
    hipFree(h->pool);
    h->pool = nullptr;
  }
  const size_t per_q = (size_t)T * N * Bp;
  // Householder vectors of the null-space basis: 3N - 3 rows per knot (HV_ROWS).  (Until round 3 this was carved as N x NZ rows, the size of Z
  // itself: at 393 216 instances the second slot then started 4.4 GB after the first and the sweep's 32-bit slot offset wrapped.)
  const size_t per_Z = (size_t)T * (3 * N - 3) * Bp;
  const size_t per_Dr = (size_t)T * (NZ * (NZ + 1) / 2) * Bp;
  const size_t per_t = (size_t)T * Bp;
  size_t nd = 0;  // doubles
  nd += 2 * per_q + 2 * per_Z + 2 * per_Dr + 2 * per_q /*g*/ + 4 * per_t /*phi,cv*/;
  nd += 3 * per_q;  // q_spare, G_spare
  nd += 2 * per_q /*Gfull*/ + (size_t)T * NZ * NZ * Bp + (size_t)T * NZ * Bp;
  nd += 2 * (size_t)T * (3 + 3 * NZ) * Bp;  // mdl
  nd += 2 * (size_t)T * NZ * NZ * Bp + 2 * (size_t)T * NZ * Bp + 2 * per_t + (size_t)T * NZ * Bp;  // E, gt, merit, zstep
  nd += (size_t)12 * Bp + 7 * (size_t)Bp;
  nd += (size_t)4 * T * Bp;  // lam_h
  nd += per_t;               // lead-joint angles
  size_t ni = 11 * (size_t)Bp + 32 + 8 * 1024;  // + n_running, n_new, work (8-byte aligned), n_defer; defer_list [2][Bp]
  size_t bytes = nd * sizeof(double) + ni * sizeof(int);
  void* pool = nullptr;
  hipError_t e = hipMalloc(&pool, bytes);
  if (e != hipSuccess) return fail(OH_ERR_HIP, std::string("device pool allocation failed: ") + hipGetErrorString(e));
  hipMemsetAsync(pool, 0, bytes, h->stream);
  h->pool = pool;
  h->pool_bytes = bytes;
  h->cap_B = Bp;
  double* d = (double*)pool;
  auto take = [&](size_t n) {
    double* r = d;
    d += n;
    return r;
  };
  FigBuffers& D = h->D;
  D.B = B;
  D.Bp = Bp;
  D.chain = h->d_chain;
  for (int s = 0; s < 2; ++s) D.q[s] = take(per_q);
  for (int s = 0; s < 2; ++s) D.q_spare[s] = take(per_q);
  for (int s = 0; s < 2; ++s) D.Z[s] = take(per_Z);
  for (int s = 0; s < 2; ++s) D.Dr[s] = take(per_Dr);
  for (int s = 0; s < 2; ++s) D.g[s] = take(per_q);
  for (int s = 0; s < 2; ++s) D.phi[s] = take(per_t);
  for (int s = 0; s < 2; ++s) D.cv[s] = take(per_t);
  // (the sweep with the coupling folded in addresses G of either slot as a 32-bit offset from the lowest of these three: the carried
  //  compaction swaps Gfull[] with the spare, so the three stay next to each other in the pool)
  for (int s = 0; s < 2; ++s) D.Gfull[s] = take(per_q);
  D.G_spare = take(per_q);
  for (int s = 0; s < 2; ++s) D.mdl[s] = take((size_t)T * (3 + 3 * NZ) * Bp);
  for (int s = 0; s < 2; ++s) D.E[s] = take((size_t)T * NZ * NZ * Bp);
  for (int s = 0; s < 2; ++s) D.gt[s] = take((size_t)T * NZ * Bp);
  for (int s = 0; s < 2; ++s) D.merit[s] = take(per_t);
  D.zstep = take((size_t)T * NZ * Bp);
  D.Kmat = take((size_t)T * NZ * NZ * Bp);
  D.kvec = take((size_t)T * NZ * Bp);
  D.ref = take((size_t)12 * Bp);
  D.fconst = take(Bp);
  D.f_cur = take(Bp);
  D.pred = take(Bp);
  D.mu = take(Bp);
  D.nun = take(Bp);
  D.stat = take(Bp);
  D.feas = take(Bp);
  D.lam_h = take((size_t)4 * T * Bp);
  D.lead = take(per_t);
  if (!h->desc.lock_orientation) D.lam_h = nullptr;  // no quaternion rows, no multipliers to report
  int* ip = (int*)d;
  D.cur = ip; ip += Bp;
  D.first = ip; ip += Bp;
  D.skip = ip; ip += Bp;
  D.polish = ip; ip += Bp;
  D.stale = ip; ip += Bp;
  D.status = ip; ip += Bp;
  D.iters = ip; ip += Bp;
  D.orig = ip; ip += Bp;
  D.newidx = ip; ip += Bp;
  D.n_running = ip; ip += 1;
  D.n_new = ip; ip += 1;
  D.work = (unsigned long long*)ip; ip += 28;
  D.n_defer = (int*)(D.work + 3);  // (two ints inside the spare part of the counter block: zeroed with it at the start of a solve)
  D.scan_blk = ip; ip += 8 * 1024;
  D.defer_list = ip;
  return OH_OK;
}

extern "C" int oh_set_guards(oh_handle* h, const oh_guards* g) {
  if (!h || !g) return fail(OH_ERR_INVALID, "oh_set_guards: null argument");
  if (h->desc.kind != OH_PROBLEM_FIGURE_EIGHT) return fail(OH_ERR_INVALID, "oh_set_guards: not a trajectory family");
  if (g->n_links < 0 || g->n_links > OH_MAX_SPHERE_LINKS || g->n_obstacles < 0 || g->n_obstacles > OH_MAX_OBSTACLES)
    return fail(OH_ERR_INVALID, "oh_set_guards: too many sphere links / obstacles");
  if ((g->n_links == 0) != (g->n_obstacles == 0)) return fail(OH_ERR_INVALID, "oh_set_guards: sphere rows need both links and obstacles");
  if (!g->limits && g->n_links == 0 && !g->vel_limits) return fail(OH_ERR_INVALID, "oh_set_guards: no rows");
  // (the orientation-locked kernels that carry inequality rows -- k_eval_lg, k_tail_vel -- are instantiated for 6 and 7 joints only; any
  //  other chain used to launch nothing, finalise the seed and still return OH_OK)
  if (h->desc.lock_orientation && (h->desc.ndof < 4 || h->desc.ndof > 8))
    return fail(OH_ERR_INVALID, "oh_set_guards: inequality rows on an orientation-locked handle need 4 ... 8 joints");
  if (g->vel_limits) {
    for (int j = 0; j < h->desc.ndof; ++j)
      if (!(g->dq_lo[j] < g->dq_up[j])) return fail(OH_ERR_INVALID, "oh_set_guards: dq_lo must be below dq_up");
  }
  for (int l = 0; l < g->n_links; ++l)
    if (g->link_joint[l] < 0 || g->link_joint[l] >= h->desc.ndof)
      return fail(OH_ERR_INVALID, "oh_set_guards: sphere links must hang on an actuated joint of the chain");
  if (g->limits)
    for (int j = 0; j < h->desc.ndof; ++j)
      if (!(g->q_lo[j] < g->q_up[j])) return fail(OH_ERR_INVALID, "oh_set_guards: q_lo must be below q_up");
  if (h->gpool) {  // the pool was carved for the previous row count: rebuild it at the next solve
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->gpool);
    h->gpool = nullptr;
    h->gcap = 0;
  }
  h->guards = *g;
  h->have_guards = true;
  drop_peers(h);
  return OH_OK;
}

static int ensure_guards(oh_handle* h) {
  const int N = h->desc.ndof, T = h->desc.T;
  const oh_guards& g = h->guards;
  GuardParams& GP = h->GP;
  GP = GuardParams{};
  GP.limits = g.limits ? 1 : 0;
  GP.n_links = g.n_links;
  GP.n_obs = g.n_obstacles;
  GP.NC = (g.limits ? 2 * N : 0) + g.n_links * g.n_obstacles;
  for (int l = 0; l < g.n_links; ++l) {
    GP.link_joint[l] = g.link_joint[l];
    for (int i = 0; i < 3; ++i) GP.link_off[l][i] = g.link_offset[l][i];
  }
  for (int j = 0; j < N; ++j) {
    GP.lo[j] = g.q_lo[j];
    GP.up[j] = g.q_up[j];
  }
  GP.rho0 = g.rho0 > 0.0 ? g.rho0 : 10.0 * h->desc.w_path;
  GP.vel = g.vel_limits ? 1 : 0;
  for (int j = 0; j < N; ++j) {
    GP.vlo[j] = g.dq_lo[j];
    GP.vup[j] = g.dq_up[j];
  }
  GP.vscale = h->desc.dt * h->desc.dt / 40.0;
  const int Bp = h->D.Bp;
  if (!h->gpool || h->gcap != Bp) {
    if (h->gpool) hipFree(h->gpool);
    h->gpool = nullptr;
    const size_t npar = (size_t)g.n_links + 4 * (size_t)g.n_obstacles;
    const size_t n_lam = (size_t)T * GP.NC * Bp, n_lamv = GP.vel ? (size_t)T * 2 * N * Bp : 0;
    const size_t nd = 3 * (n_lam + n_lamv) + 2 * npar * Bp + 4 * (size_t)T * Bp + (6 + 8 + 2) * (size_t)Bp;  // lam, lam_out, the compaction scratch, ls_*
    const size_t bytes = nd * sizeof(double) + 3 * (size_t)Bp * sizeof(int);
    hipError_t e = hipMalloc(&h->gpool, bytes);
    if (e != hipSuccess) return fail(OH_ERR_HIP, std::string("guard pool allocation failed: ") + hipGetErrorString(e));
    hipMemsetAsync(h->gpool, 0, bytes, h->stream);
    h->gcap = Bp;
    double* d = (double*)h->gpool;
    auto take = [&](size_t n) { double* r = d; d += n; return r; };
    GuardBuffers& GB = h->GB;
    GB.lam = take((size_t)T * GP.NC * Bp);
    GB.par = take(npar * Bp);
    GB.psi[0] = take((size_t)T * Bp);
    GB.psi[1] = take((size_t)T * Bp);
    GB.rho = take(Bp);
    GB.rho_next = take(Bp);
    GB.omega = take(Bp);
    GB.meas_prev = take(Bp);
    h->D.fpsi = take(Bp);
    GB.mcv[0] = take((size_t)T * Bp);
    GB.mcv[1] = take((size_t)T * Bp);
    GB.meas = take(Bp);
    GB.lamv = GP.vel ? take((size_t)T * 2 * N * Bp) : nullptr;
    GB.lam_out = take(n_lam);
    GB.lamv_out = GP.vel ? take(n_lamv) : nullptr;
    GB.scr = take(n_lam + n_lamv + (npar + 8) * Bp);
    GB.ls_gd = take(Bp);
    GB.ls_q = take(Bp);
    int* ip = (int*)d;
    GB.outer = ip; ip += Bp;
    GB.n_outer = ip; ip += Bp;
    GB.ls_count = ip;
  } else {
    // D.fpsi lives in the guard pool; ensure_capacity may have rebuilt FigBuffers
    h->D.fpsi = h->GB.meas_prev + Bp;
  }
  return OH_OK;
}

static void fill_params(oh_handle* h) {
  FigParams& P = h->P;
  const oh_problem_desc& d = h->desc;
  P.T = d.T;
  P.t0 = d.fix_dq0 ? 2 : 1;
  P.lock = d.lock_orientation;
  P.path_in_frame = d.path_in_frame;
  P.nx = d.ndof * d.T + d.ndof * (d.T - 1);
  P.dt = d.dt;
  P.w_path = d.w_path;
  P.kappa = d.w_vel / (d.dt * d.dt);
  P.tol = optv(h, "tol", 0.0) > 0.0 ? optv(h, "tol", 0.0) : d.tol;  // (option "tol": the stopping tolerance of an existing trajectory handle, bench.py's second pass)
  P.tol_feas = d.tol_feas;
  P.tol_retract = fmin(1e-10, d.tol_feas);
  // (every handle since the end of round 3.  First built for handles with inequality rows; a tolerance sweep then showed the plain family in the
  //  same retraction-noise end game below tol = 1e-7 -- a tenth of a batch sitting at 1.5 x tol until the cap at 1e-9 -- and at the default 1e-6
  //  the rules cut the rejected steps from 1.7 % to 0.4 % and the median instance from 13 to 10 steps: 2.92 -> 3.0-3.1 M solves/s)
  P.tol_retract_min = fmin(1e-13, P.tol_retract);
  P.tol_retract_min = fmin(optv(h, "retract_min", 1e-13), P.tol_retract);  // (option: experiments)
  P.feas_accept = fmax(1e-8, 10.0 * d.tol_feas);
  P.max_retract = 4;
  P.max_iter = d.max_iter;
  P.hessian = d.hessian;
  P.hyb_switch = 1e-5 * d.w_path;
  // (The same switch for handles with limit rows.  Switching those earlier was measured and dropped: at 3e-5 x w_path 16 384 velocity-limited
  // instances take 20.4 instead of 22.0 steps on average with 99.95 % of them at the same optimum, at 1e-4 19.6 steps with 0.8 % forking -- but the
  // tail of the distribution moves about: of 24 576 instances one sits at the 600-step cap at 3e-5, takes 565 steps at 2e-5 and 166 at 5e-5, where
  // 1e-5 finishes every instance of every batch measured, 16 384 to 262 144, in at most 508; tools/gpu_vel_switch_quality.py, gpu_vel_24576_probe.py.)
  P.hyb_switch = optv(h, "hyb_switch", 1e-5) * d.w_path;  // (option: experiments)
  P.mu0 = d.mu0;
  P.relax = 1.5;
  P.relax_from = 4;
  P.relax = optv(h, "relax", P.relax);
  P.relax_from = (int)optv(h, "relax_from", P.relax_from);
  P.settle_k = optv(h, "settle_k", 1.0);
  P.al_fuse = (int)optv(h, "al_fuse", 1.0);
  P.local_path = h->d_local_path;
  P.np = d.ndof + (h->have_guards ? h->guards.n_links + 4 * h->guards.n_obstacles : 0);
  if (h->chain_host.has_lead) P.np = d.ndof + 1 + d.T;
  // coupling folded into evaluation and sweep (no k_couple launch): the plain orientation-locked handles, i.e. the batched path of config 2
  P.zc = (h->fuse_couple && d.lock_orientation && !h->have_guards && !h->chain_host.has_lead) ? 1 : 0;
  P.zc_free = (h->fuse_couple && !d.lock_orientation && !(h->have_guards && h->guards.vel_limits)) ? 1 : 0;
}

__global__ __launch_bounds__(256) static void k_gather_int(const int* __restrict__ src, int* __restrict__ dst, const int B, const int* __restrict__ newidx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && newidx[b] >= 0) dst[newidx[b]] = src[b];
}
// Every array of a running orientation-locked handle with inequality rows to its instance's new index (D.newidx from oh_launch_scan_running): both slots
// of the stage data, the pending step, multipliers, obstacle parameters, every per-instance scalar and flag.  Nothing restarts.
static int move_everything(oh_handle* h, hipStream_t s, int Bnew) {
  const int N = h->desc.ndof, NZ = N - 3, T = h->desc.T, Bp = h->D.Bp, B = h->D.B;
  FigBuffers& D = h->D;
  GuardBuffers& GB = h->GB;
  const GuardParams& GP = h->GP;
  struct Item { void* p; int rows; bool is_int; };
  std::vector<Item> items;
  auto dbl = [&](double* p, int rows) { if (p) items.push_back({p, rows, false}); };
  auto itg = [&](int* p) { if (p) items.push_back({p, 1, true}); };
  // What a handle's kernels never carry from one launch to the next stays behind: the folded-coupling family (P.zc) keeps no coupling blocks, reduced
  // gradients or per-knot tracking cost of its own (E, gt, phi: oh_figure8_units.h writes merit[] instead and the sweep rebuilds the rest), the Riccati gains
  // are written by the backward pass of a sweep and read by the forward pass of the same launch, and a chain without a lead joint has no lead[].
  const bool zc = h->P.zc != 0, slim = optv(h, "invariant_move_slim", 1.0) != 0.0;
  // invariant_move_live (plain folded-coupling family only): of the per-slot stage data only the slot of the accepted point travels -- the other slot is the
  // trial the sweep has just judged: accepted, it IS the accepted slot by now; rejected, it is rewritten by the next evaluation before anything reads it.
  // The knots themselves travel in both slots (the pinned end knots of a slot are written once, by k_setup).
  struct Pair { double* a0; double* a1; int rows; };
  std::vector<Pair> live;
  const bool live_only = zc && !h->have_guards && optv(h, "invariant_move_live", 1.0) != 0.0;
  if (live_only) {
    live.push_back({D.Z[0], D.Z[1], T * (3 * N - 3)}); live.push_back({D.Dr[0], D.Dr[1], T * (NZ * (NZ + 1) / 2)}); live.push_back({D.g[0], D.g[1], T * N});
    live.push_back({D.cv[0], D.cv[1], T}); live.push_back({D.Gfull[0], D.Gfull[1], T * N}); live.push_back({D.mdl[0], D.mdl[1], T * (3 + 3 * NZ)});
    live.push_back({D.merit[0], D.merit[1], T});
  }
  for (int sl = 0; sl < 2; ++sl) {
    dbl(D.q[sl], T * N);
    if (!live_only) {
      dbl(D.Z[sl], T * (3 * N - 3)); dbl(D.Dr[sl], T * (NZ * (NZ + 1) / 2)); dbl(D.g[sl], T * N); dbl(D.cv[sl], T);
      dbl(D.Gfull[sl], T * N); dbl(D.mdl[sl], T * (3 + 3 * NZ)); dbl(D.merit[sl], T);
    }
    if (!(zc && slim)) { dbl(D.phi[sl], T); dbl(D.E[sl], T * NZ * NZ); dbl(D.gt[sl], T * NZ); }
    dbl(GB.psi[sl], T); dbl(GB.mcv[sl], T);
  }
  dbl(D.zstep, T * NZ); dbl(D.ref, 12);
  if (!(zc && slim)) { dbl(D.Kmat, T * NZ * NZ); dbl(D.kvec, T * NZ); }
  if (h->chain_host.has_lead || !slim) dbl(D.lead, T);
  for (double* p : {D.fconst, D.f_cur, D.pred, D.mu, D.nun, D.stat, D.feas, D.fpsi, GB.rho, GB.rho_next, GB.omega, GB.meas_prev, GB.meas, GB.ls_gd, GB.ls_q}) dbl(p, 1);
  dbl(GB.lam, T * GP.NC); dbl(GB.lamv, GP.vel ? T * 2 * N : 0); dbl(GB.par, GP.n_links + 4 * GP.n_obs);
  for (int* p : {D.cur, D.first, D.skip, D.polish, D.stale, D.status, D.iters, D.orig, GB.outer, GB.n_outer, GB.ls_count}) itg(p);
  int max_rows = 1;
  for (const Item& it : items) max_rows = it.rows > max_rows ? it.rows : max_rows;
  for (const Pair& pr : live) max_rows = pr.rows > max_rows ? pr.rows : max_rows;
  const size_t need = sizeof(double) * ((size_t)max_rows + 1) * Bp;  // (+ one row: cur in the new order, for the live-slot moves)
  if (need > h->move_scr_bytes) {
    if (h->move_scr) hipFree(h->move_scr);
    h->move_scr = nullptr;
    h->move_scr_bytes = 0;
    HIPCHK(hipMalloc(&h->move_scr, need));
    h->move_scr_bytes = need;
  }
  if (!live.empty()) {
    int* curn = (int*)((double*)h->move_scr + (size_t)max_rows * Bp);
    hipLaunchKernelGGL(k_gather_int, dim3((B + 255) / 256), dim3(256), 0, s, (const int*)D.cur, curn, B, (const int*)D.newidx);
    for (const Pair& pr : live) oh_launch_move_rows_live(s, pr.a0, pr.a1, (double*)h->move_scr, pr.rows, Bp, B, Bnew, D.newidx, D.cur, curn);
  }
  for (const Item& it : items) oh_launch_move_rows(s, it.p, h->move_scr, it.rows, Bp, B, Bnew, D.newidx, it.is_int);
  return OH_OK;
}

// ---- one batch, several streams (round 5) ----------------------------------------------------------------------------------------------------
// A solve of the plain orientation-locked family alternates bandwidth-bound launches over the whole batch with phases that leave the machine
// mostly idle: the persistent tail kernel (one wavefront per SIMD for milliseconds), the last launches before the hand-over, the compactions, the
// host's look at the running count after every iteration.  Two halves of the batch on two streams, each driven by a host thread of its own, fill
// each other's gaps: 262 144 instances 91.1 -> 84.6 ms on one box (2.88 -> 3.10 M solves/s; three parts 86.3, four 90.9: tools/gpu_two_streams.py).
// Instances are independent, so this is the multi-GPU sharding of DESIGN section 7 applied once more inside a GPU.  Each part is a handle of its own (peer:
// same description, constants, options, compiled kernels); profiling runs (events after every kernel) stay on one stream.
static void copy_options(oh_handle* dst, const oh_handle* src) {
  dst->tail_threshold = src->tail_threshold; dst->free_pcr_max = src->free_pcr_max; dst->compaction = src->compaction;
  dst->compact_frac = src->compact_frac; dst->compact_frac_restart = src->compact_frac_restart; dst->compact_sort = src->compact_sort;
  dst->compact_carry = src->compact_carry; dst->tail_vel = src->tail_vel; dst->lg_split = src->lg_split; dst->tail_vel_threshold = src->tail_vel_threshold;
  dst->fuse_couple = src->fuse_couple; dst->sparse_check_below = src->sparse_check_below; dst->specialize = src->specialize; dst->tq_check = src->tq_check; dst->opt = src->opt;
}
// peer handles of a trajectory / torque handle: same description, constants, dynamics and inequality rows, a stream of their own (parts of a split solve,
// lanes of the pipelined host-buffer solve)
static int ensure_peers(oh_handle* h, const int n) {
  const bool tqk = h->desc.kind == OH_PROBLEM_TORQUE_MPC;
  HIPCHK(hipSetDevice(h->device));  // peers are created on, and every part's host thread is bound to, the device of the handle (not the calling thread's)
  while ((int)h->peers.size() < n) {
    oh_handle* p = nullptr;
    int rc;
    if (tqk) rc = oh_create_torque(&h->tq, &p);
    else {
      oh_problem_desc d = h->desc;
      d.local_path = h->local_path.data();
      rc = oh_create(&d, &p);
    }
    if (rc) return rc;
    p->is_peer = true;
    p->device = h->device;
    rc = oh_set_constants(p, &h->chain_host);
    if (!rc && tqk) rc = oh_set_dynamics(p, &h->dyn_host);
    if (!rc && !tqk && h->have_guards) rc = oh_set_guards(p, &h->guards);
    if (rc) { oh_destroy(p); return rc; }
    h->peers.push_back(p);
  }
  return OH_OK;
}
static int solve_split(oh_handle* h, const int S, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt, void* d_iters, void* d_status) {
  const bool tqk = h->desc.kind == OH_PROBLEM_TORQUE_MPC;
  int prc = ensure_peers(h, S - 1);
  if (prc) return prc;
  const int N = h->desc.ndof, T = h->desc.T;
  const size_t nx = tqk ? 4 * (size_t)N * T : (size_t)N * T + (size_t)N * (T - 1);
  const size_t npar = tqk ? 2 * (size_t)N + 3 * (size_t)T : (size_t)N + (h->have_guards ? h->guards.n_links + 4 * (size_t)h->guards.n_obstacles : 0);
  std::vector<int> lo(S + 1, 0);
  for (int i = 1; i <= S; ++i) lo[i] = (int)((long long)B * i / S / 64 * 64);  // (parts start on multiples of 64: whole wavefronts of the thread-per-instance kernels)
  lo[S] = B;
  std::vector<int> rcs(S, OH_OK);
  std::vector<std::string> errs(S);
  auto part = [&](const int i) {
    oh_handle* q = i == 0 ? h : h->peers[i - 1];
    if (hipSetDevice(h->device) != hipSuccess) { rcs[i] = OH_ERR_HIP; errs[i] = "solve_split: hipSetDevice failed"; return; }  // (a new host thread starts on device 0)
    const size_t o = (size_t)lo[i];
    const int n = lo[i + 1] - lo[i];
    auto off = [&](const void* ptr, const size_t bytes_per) -> void* { return ptr ? (void*)((char*)ptr + o * bytes_per) : nullptr; };
    rcs[i] = oh_solve_device(q, n, off(d_x0, nx * 8), off(d_p, npar * 8), off(d_x, nx * 8), off(d_f, 8), off(d_kkt, 24), off(d_iters, 4), off(d_status, 4));
    if (rcs[i]) errs[i] = oh_last_error();
  };
  for (int i = 1; i < S; ++i) {
    copy_options(h->peers[i - 1], h);
    h->peers[i - 1]->spec = h->spec;
    h->peers[i - 1]->spec_failed = h->spec_failed;
    h->peers[i - 1]->spec_cache_checked = true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  h->is_peer = true;  // (this handle takes part 0 itself and must not split again)
  for (int i = 1; i < S; ++i) th.emplace_back(part, i);
  part(0);
  for (std::thread& t : th) t.join();
  h->is_peer = false;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (int i = 0; i < S; ++i)
    if (rcs[i]) return fail(rcs[i], errs[i]);
  h->split_parts.assign(S, 0);
  for (int i = 0; i < S; ++i) h->split_parts[i] = lo[i + 1] - lo[i];
  h->timing[4] = ms;
  for (int i = 1; i < S; ++i) {
    const oh_handle* q = h->peers[i - 1];
    h->timing[5] = std::max(h->timing[5], q->timing[5]);
    h->timing[6] += q->timing[6];
    h->timing[7] += q->timing[7];
    h->rejects += q->rejects;
    h->tail_iters += q->tail_iters;
  }
  h->last_B = B;
  return OH_OK;
}

extern "C" int oh_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt,
                               void* d_iters, void* d_status) {
  if (!h) return fail(OH_ERR_INVALID, "oh_solve_device: null handle");
  if (B < 1) return fail(OH_ERR_INVALID, "oh_solve_device: B must be >= 1");
  if (!d_x0 || !d_p) return fail(OH_ERR_INVALID, "oh_solve_device: x0 and p are required");
  load_launch_opts(h);
  if (h->desc.kind == OH_PROBLEM_POINT_MASS_MPC) return pm_solve_device(h, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
  if (h->desc.kind == OH_PROBLEM_IK) return ik_solve_device(h, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
  if (h->desc.kind == OH_PROBLEM_QP) return qp_solve_device(h, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
  if (h->desc.kind == OH_PROBLEM_TAPE) return tape_solve_device(h, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
  if (h->desc.kind == OH_PROBLEM_TORQUE_MPC) {
    if (!h->is_peer) { h->split_parts.clear(); h->pipe_last = false; }
    const int S = std::min(8, (int)optv(h, "streams", 2.0));
    if (!h->is_peer && !h->profiling && S >= 2 && B >= (int)optv(h, "tq_split_min", 1024.0) && B / S >= 64 && h->have_chain && h->have_dyn)
      return solve_split(h, S, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
    return tq_solve_device(h, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
  }
  if (h->desc.kind != OH_PROBLEM_FIGURE_EIGHT) return fail(OH_ERR_STATE, "oh_solve_device: handle was created without a problem (OH_PROBLEM_KINEMATICS)");
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_solve_device: call oh_set_constants first");
  if (!solver_chain_ok(h->chain_host))
    return fail(OH_ERR_INVALID, "oh_solve_device: the solver needs a chain that covers every model joint in order");
  HIPCHK(hipSetDevice(h->device));
  if (!h->is_peer) { h->split_parts.clear(); h->pipe_last = false; }
  // (batch_invariant handles take part in the split too since their compaction moves everything: an instance's answer does not depend on its part)
  if (!h->is_peer && !h->profiling && spec_applies(h) && (optv(h, "batch_invariant", 0.0) == 0.0 || optv(h, "invariant_split", 1.0) != 0.0)) {
    const int S = std::min(8, (int)optv(h, "streams", 2.0));
    if (S >= 2 && B >= (int)optv(h, "split_min", 65536.0) && B / S >= 4096 && stage_fits(h, B)) {  // (a batch beyond oh_max_batch is refused below, as ever)
      // (the kernels compiled for the chain are shared: make sure they exist before the parts look for them)
      if (!h->spec && !h->spec_failed && h->specialize != OH_SPECIALIZE_NEVER && oh_specialize(h) != OH_OK) h->spec_failed = true;
      return solve_split(h, S, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
    }
  }
  if (!h->is_peer && !h->profiling && !h->desc.lock_orientation && !h->chain_host.has_lead && optv(h, "batch_invariant", 0.0) == 0.0) {
    // position-tracking family (every launch latency-bound, the machine far from full): two halves side by side -- 1024 arms 9.5 -> 7.3 ms
    const int S = std::min(8, (int)optv(h, "streams", 2.0));
    if (S >= 2 && B >= (int)optv(h, "free_split_min", 256.0) && B / S >= 64) return solve_split(h, S, B, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status);
  }
  int rc = ensure_capacity(h, B);
  if (rc) return rc;
  fill_params(h);
  if ((spec_applies(h) || spec_tail_vel_applies(h)) && !h->spec && !h->spec_failed) {
    bool want = h->specialize == OH_SPECIALIZE_ALWAYS || (h->specialize == OH_SPECIALIZE_AUTO && B >= h->specialize_min_B);
    if (!want && h->specialize == OH_SPECIALIZE_AUTO && !h->spec_cache_checked) {  // a compiled object is at hand: milliseconds, whatever the batch
      h->spec_cache_checked = true;
      want = oh_jit_figure8_cached(h->chain_host, h->desc.ndof);
    }
    if (want && oh_specialize(h) != OH_OK) h->spec_failed = true;
  }  // the generic kernels run (and no further attempt is made); oh_last_error keeps the reason
  const bool guarded = h->have_guards;
  const bool lead = h->chain_host.has_lead != 0;
  {  // position-tracking family, 7 joints, every launch of this solve a block per instance (k_step_free_bb): stage blocks instance-major
    // (batch_invariant: the size-independent path -- knot-major blocks, the serial sweep on one lane per instance -- whatever the batch: the block-per-instance
    //  factorisations chosen by size round differently)
    h->P.inst_major = (!h->desc.lock_orientation && h->desc.ndof == 7 && B <= h->free_pcr_max && h->desc.T - (h->desc.fix_dq0 ? 2 : 1) <= 128 && optv(h, "free_bb", 1.0) != 0.0 &&
                       optv(h, "batch_invariant", 0.0) == 0.0) ? 1 : 0;
  }
  if (lead && (guarded || !h->desc.lock_orientation || h->desc.ndof != 6))
    return fail(OH_ERR_INVALID, "oh_solve_device: a parameterised lead joint is lowered for the orientation-locked family with 6 optimised joints, "
                                "without inequality rows");
  if (guarded) {
    rc = ensure_guards(h);
    if (rc) return rc;
  } else {
    h->D.fpsi = nullptr;
  }
  const int N = h->desc.ndof;
  hipStream_t s = h->stream;
  const bool prof = h->profiling;
  if (prof) {
    // events: [0] start, then per iteration (after eval, after step), last = end
    const size_t need = 4 * (size_t)(2 * h->desc.max_iter + 44) + 64;
    while (h->prof_events.size() < need) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      h->prof_events.push_back(e);
    }
  }
  HIPCHK(hipEventRecord(h->ev0, s));
  HIPCHK(hipMemsetAsync(h->D.work, 0, 4 * sizeof(unsigned long long), s));  // (work[0..2] and the two deferral counters)
  if (!oh_launch_setup(s, N, h->P, h->D, (const double*)d_x0, (const double*)d_p))
    return fail(OH_ERR_INVALID, "oh_solve_device: unsupported ndof");
  if (guarded) oh_launch_setup_guards(s, N, h->P, h->D, h->GP, h->GB, (const double*)d_p);
  size_t ne = 0;
  h->prof_tags.clear();
  if (prof) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(0); }
  int launched = 0;
  int compactions = 0;
  int check_every = 1;  // one small D2H read per iteration: with the cheap compaction timely decisions beat the saved round trips (every 2nd: -1.5 %)
  if (optv(h, "check_every", 1.0) >= 1.0) check_every = (int)optv(h, "check_every", 1.0);  // (option: experiments)
  double* ox = (double*)d_x; double* of = (double*)d_f; double* ok = (double*)d_kkt;
  int* oi = (int*)d_iters; int* os = (int*)d_status;
  // Every instance needs at most max_iter steps (accepted + rejected) plus its first evaluation; each
  // plain compaction re-evaluates the survivors once.  The batch is compacted whenever 3 % of it have finished (compact_frac);
  // the regular compactions carry the pending trial along and cost no evaluation (k_carry_*), the hand-over to the tail kernel
  // restarts the survivors.
  const int hard_cap = 2 * h->desc.max_iter + 2 + 40 + 64;  // a rejected step costs two launches, a compaction one
  const int NV = (guarded && h->GP.vel) ? 2 * N : 0;  // velocity rows per knot
  // persistent tail kernel: plain orientation-locked handles, and (round 3, k_tail_vel) those whose inequality rows are joint and / or joint-velocity limits (no sphere rows)
  const bool tail_vel = guarded && (h->GP.vel || h->GP.limits) && h->GP.n_links == 0 && h->tail_vel;
  const bool tail_ok = (h->desc.T - h->P.t0 <= 64) && h->tail_threshold > 0 && h->desc.lock_orientation && (!guarded || tail_vel) && !lead;
  // limit / velocity-limit handles: the batched launches (evaluation, velocity coupling in a launch of its own, sweep: 1.45 ms per iteration of
  // 40 000 instances) lose against the persistent kernel at every size measured -- 65 536 instances: 87.5 ms with the hand-over at 16 384,
  // 80.5 at 32 768, 76.3 when the whole batch starts there -- so these handles go to it whatever the batch (OH_TAIL_VEL_THRESHOLD overrides)
  const int tail_threshold = tail_vel ? h->tail_vel_threshold : h->tail_threshold;
  const FigSpec* const spec = spec_applies(h) ? h->spec : nullptr;
  // evaluation / tail launches: the kernels compiled for this handle's chain when they are loaded, the generic ones otherwise
  auto launch_eval = [&](int slot, int part) {
    if (spec) return oh_spec_launch_eval(*spec, s, h->P, h->D, slot, part) == hipSuccess;
    return oh_launch_eval(s, N, h->P, h->D, slot, part);
  };
  auto launch_tail = [&](int slot) {
    if (tail_vel && h->spec && spec_tail_vel_applies(h)) return oh_spec_launch_tail_vel(*h->spec, s, h->P, h->D, h->GP, h->GB, slot) == hipSuccess;
    if (tail_vel) return oh_launch_tail_vel(s, N, h->P, h->D, h->GP, h->GB, slot);
    if (spec) return oh_spec_launch_tail(*spec, s, h->P, h->D, slot) == hipSuccess;
    return oh_launch_tail(s, N, h->P, h->D, slot);
  };
  // results of finished instances: x through the library's LDS transpose, scalars and multipliers through the kernel compiled for the chain where there is one
  auto finalize = [&](const int only_done) {
    if (spec && spec->finalize) {
      oh_launch_finalize(s, N, h->P, h->D, only_done, ox, of, ok, oi, os, 1);
      if (oh_spec_launch_finalize(*spec, s, h->P, h->D, only_done, of, ok, oi, os) == hipSuccess) return;
      (void)hipGetLastError();
      oh_launch_finalize(s, N, h->P, h->D, only_done, ox, of, ok, oi, os, 2);
      return;
    }
    oh_launch_finalize(s, N, h->P, h->D, only_done, ox, of, ok, oi, os);
  };
  bool tail_done = false;
  {  // position-tracking family with limit / sphere rows, every instance a block of its own: the whole solve in one launch (k_free_persist)
    const int fp = (int)optv(h, "free_persist", -1.0);  // -1: where it pays, 1: always, 0: never
    // measured (HISTORY): same time per iteration as the launch pair (the iteration is memory round trips inside the phases, not launch gaps), so it
    // only wins where the host's sparse looks at the running count cost idle launches and every block is resident: horizons up to 64 free knots,
    // at most 512 instances (option free_persist = 1: always, 0: never)
    const bool fits = (h->desc.T - h->P.t0 <= 64 && B <= 512) || fp == 1;
    if (!h->P.lock && guarded && !h->GP.vel && h->P.inst_major && h->P.zc_free && !prof && !lead && fits && fp != 0) {
      if (oh_launch_free_persist(s, N, h->P, h->D, h->GP, h->GB)) {
        tail_done = true;
        launched = 1;
      }
    }
  }
  if (!tail_done && tail_ok && B <= tail_threshold) {  // small batch: the whole solve is one persistent launch
    if (!launch_tail(0)) return fail(OH_ERR_INVALID, "oh_solve_device: no persistent kernel for this handle (ndof / rows)");
    tail_done = true;
  }
  bool rebase = false;
  const bool invariant = optv(h, "batch_invariant", 0.0) != 0.0;
  const double inv_frac = optv(h, "invariant_compact_frac", 0.65);
  int carry_pending = 0;  // > 0: survivors to compact the batch to, between k_retract and k_evalb of the next iteration
  for (int it = 0; it < hard_cap && !tail_done; ++it) {
    if (prof && rebase && ne + 3 < h->prof_events.size()) {  // host synced: do not bill the idle gap to the eval kernel
      HIPCHK(hipEventRecord(h->prof_events[ne++], s));
      h->prof_tags.push_back(0);
    }
    rebase = false;
    const int slot = it & 1;
    if (h->P.lock && guarded) {
      if (h->lg_split && h->GP.n_links == 0) {  // retraction and evaluation as two launches, like the plain family (k_evalb_lg in oh_kernels.hip)
        if (!(h->spec && spec_tail_vel_applies(h) && oh_spec_launch_eval(*h->spec, s, h->P, h->D, slot, 1) == hipSuccess))
          oh_launch_eval_locked_guarded(s, N, h->P, h->D, h->GP, h->GB, slot, 1);
        oh_launch_eval_locked_guarded(s, N, h->P, h->D, h->GP, h->GB, slot, 2);
      } else oh_launch_eval_locked_guarded(s, N, h->P, h->D, h->GP, h->GB, slot, 0);
    } else if (lead) oh_launch_eval_lead(s, N, h->P, h->D, slot);
    else if (h->P.lock && carry_pending > 0) {
      // compaction with the trial carried along: retract on the old layout, move, evaluate on the dense one (k_carry_* in oh_kernels.hip)
      launch_eval(slot, 1);
      if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(4); }
      finalize(1);
      oh_launch_scan_running(s, h->D, h->compact_sort);
      oh_launch_carry(s, N, h->P, h->D, 0, 0, slot);
      oh_launch_carry(s, N, h->P, h->D, 1, carry_pending, slot);
      // the knots were laid down densely in the spare arrays: they become q[] / Gfull[] for the launches that follow (no copy back)
      std::swap(h->D.q[slot], h->D.q_spare[0]);
      std::swap(h->D.q[1 - slot], h->D.q_spare[1]);
      if (h->P.hessian != OH_HESSIAN_GAUSS_NEWTON || h->P.zc) std::swap(h->D.Gfull[1 - slot], h->D.G_spare);
      h->D.B = carry_pending;
      carry_pending = 0;
      ++compactions;
      if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(0); }
      launch_eval(slot, 2);
    } else if (h->P.lock) launch_eval(slot, 0);
    else if (guarded) oh_launch_eval_guarded(s, N, h->P, h->D, h->GP, h->GB, slot);
    else oh_launch_eval_free(s, N, h->P, h->D, slot);
    if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(1); }
    if (h->P.zc || h->P.zc_free) {}  // folded into k_evalb_zc / k_step_zc (position tracking: into the evaluation)
    else if (h->P.lock && guarded && h->GP.vel) oh_launch_couple_vel(s, N, h->P, h->D, h->GP, h->GB, slot);
    else if (h->P.lock) oh_launch_couple(s, N, h->P, h->D, slot);
    else if (guarded && h->GP.vel) oh_launch_couple_free_vel(s, N, h->P, h->D, h->GP, h->GB, slot);
    else oh_launch_couple_free(s, N, h->P, h->D, slot);
    // handles without a persistent tail kernel (inequality rows, position-only tracking, lead joint) end in launches of a few hundred instances that
    // are pure latency: there the host looks at the running count every 8th iteration only (a look is a copy + stream synchronisation, 20-30 us
    // of a ~150 us iteration; the price is up to 7 idle iterations of finished instances at the very end)
    const int ce = (!tail_ok && check_every == 1 && h->D.B <= h->sparse_check_below) ? 8 : check_every;
    const bool check = ((it + 1) % ce == 0);
    if (check && !h->P.lock) HIPCHK(hipMemsetAsync(h->D.n_running, 0, sizeof(int), s));  // the locked family's k_couple resets it
    if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(3); }
    if (h->P.lock && guarded) oh_launch_step_locked_guarded(s, N, h->P, h->D, h->GP, h->GB, slot);
    else if (h->P.lock) oh_launch_step(s, N, h->P, h->D, slot);
    else if (guarded) oh_launch_step_guarded(s, N, h->P, h->D, h->GP, h->GB, slot, !invariant && h->D.B <= h->free_pcr_max);
    else oh_launch_step_free(s, N, h->P, h->D, slot, !invariant && h->D.B <= h->free_pcr_max);
    if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(2); }
    ++launched;
    if (check) {
      HIPCHK(hipMemcpyAsync(h->h_flag, h->D.n_running, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      const int nrun = *h->h_flag;
      rebase = true;
      if (nrun == 0) break;
      if (tail_ok && nrun <= tail_threshold) {
        // drain: compact the survivors and let one wavefront per instance finish them without further launches
        finalize(1);
        if (guarded) oh_launch_guard_emit(s, h->P, h->D, h->GP, h->GB, NV, 1);
        oh_launch_scan_running(s, h->D, h->compact_sort);
        oh_launch_compact(s, N, h->P, h->D, 0, 0, 0);
        if (guarded) oh_launch_guard_compact(s, h->P, h->D, h->GP, h->GB, NV, 0, 0);
        oh_launch_compact(s, N, h->P, h->D, 1, nrun, (it + 1) & 1);
        if (guarded) oh_launch_guard_compact(s, h->P, h->D, h->GP, h->GB, NV, 1, nrun);
        h->D.B = nrun;
        ++compactions;
        if (!launch_tail((it + 1) & 1)) return fail(OH_ERR_INVALID, "oh_solve_device: no persistent kernel for this handle (ndof / rows)");
        tail_done = true;
        break;
      }
      // (k_carry_* park 22 per-instance scalars in the first rows of the spare Dr slot: T x NZ(NZ+1)/2 rows must hold them)
      const bool carry_fits = (size_t)h->desc.T * ((N - 3) * (N - 2) / 2) >= 22;
      if (h->compaction && h->compact_carry && carry_fits && h->P.lock && !guarded && !lead && h->D.B >= 512 && (double)nrun <= h->compact_frac * (double)h->D.B) {
        carry_pending = nrun;  // done after the next k_retract
      } else if (invariant && h->P.lock && !lead && inv_frac > 0.0 && h->D.B >= 512 && (double)nrun <= inv_frac * (double)h->D.B) {
        // batch_invariant handles: the survivors move with everything they own (both slots' stage data, the pending step, every scalar and flag), so
        // the compaction is invisible to the state machine -- same iterates, bit for bit, as without it -- and is done on a coarse schedule (when
        // a third of the batch has finished, by default: a move costs an instance a few iterations' worth of bandwidth; tools/gpu_invariant_compaction.py)
        finalize(1);
        if (guarded) oh_launch_guard_emit(s, h->P, h->D, h->GP, h->GB, NV, 1);
        oh_launch_scan_running(s, h->D, h->compact_sort);
        const int mrc = move_everything(h, s, nrun);
        if (mrc) return mrc;
        h->D.B = nrun;
        ++compactions;
        if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(0); }
      } else if (h->compaction && !lead && h->D.B >= 512 && (double)nrun <= h->compact_frac_restart * (double)h->D.B) {
        // restart compaction: the survivors' accepted knots (and, with inequality rows, their multipliers and outer-loop state) are laid
        // down densely and re-evaluated
        finalize(1);
        if (guarded) oh_launch_guard_emit(s, h->P, h->D, h->GP, h->GB, NV, 1);
        oh_launch_scan_running(s, h->D, h->compact_sort);
        if (h->P.lock && guarded && optv(h, "compact_move_all", 1.0) != 0.0) {
          // orientation-locked handles with inequality rows (horizons beyond the persistent kernel): the instance moves with everything it owns,
          // nothing restarts -- the same iterates, bit for bit, as without compaction (round 5)
          const int mrc = move_everything(h, s, nrun);
          if (mrc) return mrc;
        } else {
          oh_launch_compact(s, N, h->P, h->D, 0, 0, 0);
          if (guarded) oh_launch_guard_compact(s, h->P, h->D, h->GP, h->GB, NV, 0, 0);
          oh_launch_compact(s, N, h->P, h->D, 1, nrun, (it + 1) & 1);
          if (guarded) oh_launch_guard_compact(s, h->P, h->D, h->GP, h->GB, NV, 1, nrun);
        }
        h->D.B = nrun;
        ++compactions;
        // the compaction kernels are accounted to neither eval nor step: restart the event pair
        if (prof && ne + 2 < h->prof_events.size()) { HIPCHK(hipEventRecord(h->prof_events[ne++], s)); h->prof_tags.push_back(0); }
      }
    }
  }
  finalize(0);
  if (guarded) oh_launch_guard_emit(s, h->P, h->D, h->GP, h->GB, NV, 0);
  h->D.B = B;
  if (guarded) oh_launch_guard_infeasible(s, N, h->P, h->D, h->GP, (const double*)d_p, B, ok, os);  // constant rows of the pinned knots
  HIPCHK(hipEventRecord(h->ev1, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  h->last_B = B;
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->timing[4] = ms;
  h->timing[5] = launched;
  unsigned long long work[3] = {0, 0, 0};
  // (on the handle's own stream: a hipMemcpy is an operation of the legacy null stream and waits for the kernels of every other handle's stream -- the other
  //  part of a split solve, the other lane of a pipelined one)
  HIPCHK(hipMemcpyAsync(work, h->D.work, sizeof(work), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  h->timing[6] = (double)work[0];
  h->rejects = (double)work[1];
  h->tail_iters = (double)work[2];
  if (prof) {
    double te = 0, tsx = 0, tc = 0;
    int n_e = 0, n_s = 0;
    for (size_t i = 1; i < ne; ++i) {
      const int tag = h->prof_tags[i];
      if (tag == 0) continue;
      float ms2 = 0.f;
      hipEventElapsedTime(&ms2, h->prof_events[i - 1], h->prof_events[i]);
      if (tag == 1) { te += ms2; ++n_e; }
      else if (tag == 4) { te += ms2; }
      else if (tag == 3) { tc += ms2; }
      else { tsx += ms2; ++n_s; }
    }
    h->timing[0] = te;
    h->timing[1] = n_e;
    h->timing[2] = tsx;
    h->timing[3] = n_s;
    h->timing_couple = tc;
  }
  h->timing[7] = compactions;
  return OH_OK;
}

static int ensure_stage(oh_handle* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return OH_OK;
  if (h->stage) hipFree(h->stage);
  h->stage = nullptr;
  h->stage_bytes = 0;
  hipError_t e = hipMalloc(&h->stage, bytes);
  if (e != hipSuccess) return fail(OH_ERR_HIP, std::string("staging allocation failed: ") + hipGetErrorString(e));
  h->stage_bytes = bytes;
  return OH_OK;
}

// oh_solve of a large batch: chunks of `chunk` instances alternate between two lanes; a lane uploads its chunk on its own stream, solves it on its own
// handle (lane 0: the handle itself, lane 1: a peer) and downloads the results, while the other lane is one phase ahead or behind.  Every chunk is solved
// as a batch of its own (like the parts of solve_split); timing[4] is the wall clock of the whole call.
static int solve_pipelined(oh_handle* h, const int B, const int chunk, const size_t nx, const size_t npar, const double* x0, const double* p, double* x,
                           double* f, double* kkt, int* iters, int* status) {
  int rc = ensure_peers(h, 1);
  if (rc) return rc;
  oh_handle* lanes[2] = {h, h->peers[0]};
  copy_options(lanes[1], h);
  if (h->desc.kind == OH_PROBLEM_FIGURE_EIGHT) {
    if (spec_applies(h) && !h->spec && !h->spec_failed && h->specialize != OH_SPECIALIZE_NEVER && chunk >= h->specialize_min_B && oh_specialize(h) != OH_OK) h->spec_failed = true;
    lanes[1]->spec = h->spec; lanes[1]->spec_failed = h->spec_failed; lanes[1]->spec_cache_checked = true;
  }
  const int C = (B + chunk - 1) / chunk;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  // multipliers of every chunk, kept on the device in instance order for oh_get_multipliers (a handle only remembers its last solve)
  const bool tqk = h->desc.kind == OH_PROBLEM_TORQUE_MPC;
  const size_t per = tqk ? (size_t)h->tq.T * (h->tq.vel_limits ? 4 : 2) * h->tq.ndof : (h->desc.lock_orientation ? 4 * (size_t)h->desc.T : 0);
  if (per * B > h->pipe_mult_cap) {
    if (h->d_pipe_mult) hipFree(h->d_pipe_mult);
    h->d_pipe_mult = nullptr;
    h->pipe_mult_cap = 0;
    HIPCHK(hipMalloc((void**)&h->d_pipe_mult, sizeof(double) * per * B));
    h->pipe_mult_cap = per * B;
  }
  int rcs[2] = {OH_OK, OH_OK};
  std::string errs[2];
  double launched = 0, work = 0, compactions = 0, rejects = 0, tails = 0;
  std::mutex mtx;
  auto lane = [&](const int l) {
    oh_handle* q = lanes[l];
    if (hipSetDevice(h->device) != hipSuccess) { rcs[l] = OH_ERR_HIP; errs[l] = "oh_solve: hipSetDevice failed"; return; }
    for (int c = l; c < C; c += 2) {
      const size_t lo = (size_t)c * chunk;
      const int n = (int)std::min((size_t)chunk, (size_t)B - lo);
      const size_t b_x = sizeof(double) * nx * n, b_p = sizeof(double) * npar * n, b_f = sizeof(double) * n, b_k = sizeof(double) * 3 * n, b_i = sizeof(int) * (size_t)n;
      if (ensure_stage(q, al(b_x) * 2 + al(b_p) + al(b_f) + al(b_k) + 2 * al(b_i))) { rcs[l] = OH_ERR_HIP; errs[l] = oh_last_error(); return; }
      char* base = (char*)q->stage;
      void* d_x0 = base; void* d_p = base + al(b_x); void* d_x = (char*)d_p + al(b_p); void* d_f = (char*)d_x + al(b_x); void* d_k = (char*)d_f + al(b_f);
      void* d_it = (char*)d_k + al(b_k); void* d_st = (char*)d_it + al(b_i);
      hipStream_t s = q->stream;
      bool ok = hipMemcpyAsync(d_x0, x0 + lo * nx, b_x, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(d_p, p + lo * npar, b_p, hipMemcpyHostToDevice, s) == hipSuccess &&
                hipStreamSynchronize(s) == hipSuccess;
      if (!ok) { rcs[l] = OH_ERR_HIP; errs[l] = "oh_solve: upload failed"; return; }
      const bool was_peer = q->is_peer;
      q->is_peer = true;  // (a chunk is not split again: the two lanes ARE the two streams)
      const int r = oh_solve_device(q, n, d_x0, d_p, d_x, d_f, d_k, d_it, d_st);
      q->is_peer = was_peer;
      if (r) { rcs[l] = r; errs[l] = oh_last_error(); return; }
      ok = true;
      const double* mult = tqk ? q->d_tq_mult : q->D.lam_h;
      if (per && mult) ok = ok && hipMemcpyAsync(h->d_pipe_mult + lo * per, mult, sizeof(double) * per * n, hipMemcpyDeviceToDevice, s) == hipSuccess;
      if (x) ok = ok && hipMemcpyAsync(x + lo * nx, d_x, b_x, hipMemcpyDeviceToHost, s) == hipSuccess;
      if (f) ok = ok && hipMemcpyAsync(f + lo, d_f, b_f, hipMemcpyDeviceToHost, s) == hipSuccess;
      if (kkt) ok = ok && hipMemcpyAsync(kkt + 3 * lo, d_k, b_k, hipMemcpyDeviceToHost, s) == hipSuccess;
      if (iters) ok = ok && hipMemcpyAsync(iters + lo, d_it, b_i, hipMemcpyDeviceToHost, s) == hipSuccess;
      if (status) ok = ok && hipMemcpyAsync(status + lo, d_st, b_i, hipMemcpyDeviceToHost, s) == hipSuccess;
      ok = ok && hipStreamSynchronize(s) == hipSuccess;
      if (!ok) { rcs[l] = OH_ERR_HIP; errs[l] = "oh_solve: download failed"; return; }
      std::lock_guard<std::mutex> g(mtx);
      launched = std::max(launched, q->timing[5]); work += q->timing[6]; compactions += q->timing[7]; rejects += q->rejects; tails += q->tail_iters;
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::thread other(lane, 1);
  lane(0);
  other.join();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (int l = 0; l < 2; ++l)
    if (rcs[l]) return fail(rcs[l], errs[l]);
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms; h->timing[5] = launched; h->timing[6] = work; h->timing[7] = compactions;
  h->rejects = rejects; h->tail_iters = tails;
  h->split_parts.clear();
  h->pipe_last = true;
  h->last_B = B;
  return OH_OK;
}

extern "C" int oh_solve(oh_handle* h, int B, const double* x0, const double* p, double* x, double* f, double* kkt, int* iters,
                        int* status) {
  if (!h) return fail(OH_ERR_INVALID, "oh_solve: null handle");
  if (B < 1) return fail(OH_ERR_INVALID, "oh_solve: B must be >= 1");
  if (!x0 || !p) return fail(OH_ERR_INVALID, "oh_solve: x0 and p are required");
  if (h->desc.kind != OH_PROBLEM_FIGURE_EIGHT && h->desc.kind != OH_PROBLEM_POINT_MASS_MPC && h->desc.kind != OH_PROBLEM_IK &&
      h->desc.kind != OH_PROBLEM_QP && h->desc.kind != OH_PROBLEM_TAPE && h->desc.kind != OH_PROBLEM_TORQUE_MPC)
    return fail(OH_ERR_STATE, "oh_solve: handle was created without a problem (OH_PROBLEM_KINEMATICS)");
  HIPCHK(hipSetDevice(h->device));
  const int N = h->desc.ndof, T = h->desc.T;
  const bool pmk = h->desc.kind == OH_PROBLEM_POINT_MASS_MPC;
  const bool ikk = h->desc.kind == OH_PROBLEM_IK;
  const bool qpk = h->desc.kind == OH_PROBLEM_QP;
  const bool tpk = h->desc.kind == OH_PROBLEM_TAPE;
  const bool tqk = h->desc.kind == OH_PROBLEM_TORQUE_MPC;
  const size_t nx = tqk ? 4 * (size_t)N * T : tpk ? (size_t)h->TP.nx : qpk ? (size_t)h->qp.n : pmk ? 4 * (size_t)T : (ikk ? (size_t)N : (size_t)N * T + (size_t)N * (T - 1));
  const size_t npar = tqk ? 2 * (size_t)N + 3 * (size_t)T : tpk ? (size_t)(h->TP.np > 0 ? h->TP.np : 1) : qpk ? (h->qp_tape ? (size_t)(h->TP.np > 0 ? h->TP.np : 1) : qp_np(h->qp)) : pmk ? 4 + 4 * (size_t)T : (ikk ? (size_t)N + 3 : (h->chain_host.has_lead ? (size_t)N + 1 + T : (size_t)N + (h->have_guards ? h->guards.n_links + 4 * (size_t)h->guards.n_obstacles : 0)));
  const size_t b_x = sizeof(double) * nx * B, b_p = sizeof(double) * npar * (size_t)B;
  const size_t b_f = sizeof(double) * B, b_k = sizeof(double) * 3 * (size_t)B, b_i = sizeof(int) * (size_t)B;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t total = al(b_x) * 2 + al(b_p) + al(b_f) + al(b_k) + 2 * al(b_i);
  int rc = ensure_stage(h, total);
  if (rc) return rc;
  // device layout: inputs [x0 | p], then outputs [x | f | kkt | iters | status]
  char* base = (char*)h->stage;
  const size_t o_x0 = 0, o_p = o_x0 + al(b_x), o_x = o_p + al(b_p), o_f = o_x + al(b_x), o_k = o_f + al(b_f), o_it = o_k + al(b_k), o_st = o_it + al(b_i);
  void* d_x0 = base + o_x0; void* d_p = base + o_p; void* d_x = base + o_x; void* d_f = base + o_f; void* d_k = base + o_k;
  void* d_it = base + o_it; void* d_st = base + o_st;
  // a controller's tick moves a few hundred bytes in seven pieces: through one pinned mirror of the staging area that is two transfers (each
  // hipMemcpy of pageable memory is ~10-15 us of driver work whatever its size)
  const bool small = total <= OH_PINNED_STAGE_BYTES;
  if (small && !h->h_stage) {
    if (hipHostMalloc((void**)&h->h_stage, OH_PINNED_STAGE_BYTES) != hipSuccess) h->h_stage = nullptr;
  }
  if (small && h->h_stage) {
    char* hb = (char*)h->h_stage;
    memcpy(hb + o_x0, x0, b_x);
    memcpy(hb + o_p, p, b_p);
    HIPCHK(hipMemcpy(base, hb, o_p + b_p, hipMemcpyHostToDevice));
    rc = oh_solve_device(h, B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st);
    if (rc) return rc;
    HIPCHK(hipMemcpy(hb + o_x, base + o_x, o_st + b_i - o_x, hipMemcpyDeviceToHost));
    if (x) memcpy(x, hb + o_x, b_x);
    if (f) memcpy(f, hb + o_f, b_f);
    if (kkt) memcpy(kkt, hb + o_k, b_k);
    if (iters) memcpy(iters, hb + o_it, b_i);
    if (status) memcpy(status, hb + o_st, b_i);
    return OH_OK;
  }
  // large batches of the trajectory families from host buffers: chunks on two lanes (handles, streams, host threads), so that one lane's transfers over
  // PCIe run under the other lane's kernels (round 6; until then: upload everything, solve, download everything -- 38 % of the resident rate)
  if ((h->desc.kind == OH_PROBLEM_FIGURE_EIGHT || tqk) && !h->is_peer && !h->profiling && optv(h, "pipe", 1.0) != 0.0 && h->have_chain && !h->have_guards &&
      !h->chain_host.has_lead) {
    const int chunk = std::max(1024, (int)optv(h, "pipe_chunk", 32768.0));
    if (B >= 2 * chunk && (!tqk || h->have_dyn)) return solve_pipelined(h, B, chunk, nx, npar, x0, p, x, f, kkt, iters, status);
  }
  HIPCHK(hipMemcpy(d_x0, x0, b_x, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_p, p, b_p, hipMemcpyHostToDevice));
  rc = oh_solve_device(h, B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st);
  if (rc) return rc;
  if (x) HIPCHK(hipMemcpy(x, d_x, b_x, hipMemcpyDeviceToHost));
  if (f) HIPCHK(hipMemcpy(f, d_f, b_f, hipMemcpyDeviceToHost));
  if (kkt) HIPCHK(hipMemcpy(kkt, d_k, b_k, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(hipMemcpy(iters, d_it, b_i, hipMemcpyDeviceToHost));
  if (status) HIPCHK(hipMemcpy(status, d_st, b_i, hipMemcpyDeviceToHost));
  return OH_OK;
}

static int ensure_stage(oh_handle* h, size_t bytes);

extern "C" int oh_pm_rollout(oh_handle* h, int B, int n_ticks, int advance, double ramp, const double* state0, const double* obs_table,
                             double* states, double* f, int* iters, int* status) {
  if (!h || !state0 || !obs_table) return fail(OH_ERR_INVALID, "oh_pm_rollout: null argument");
  if (h->desc.kind != OH_PROBLEM_POINT_MASS_MPC) return fail(OH_ERR_STATE, "oh_pm_rollout: handle is not a point-mass MPC problem");
  const int T = h->pm.T;
  if (B < 1 || n_ticks < 1 || advance < 1 || advance >= T) return fail(OH_ERR_INVALID, "oh_pm_rollout: need B >= 1, n_ticks >= 1, 1 <= advance < T");
  int rc = pm_prepare(h, B);
  if (rc) return rc;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t n_obs = (size_t)n_ticks * advance + T;
  const size_t b_states = sizeof(double) * 4 * (size_t)B * (n_ticks + 1), b_obs = sizeof(double) * 2 * n_obs;
  const size_t b_p = sizeof(double) * (4 + 4 * (size_t)T) * B, b_x = sizeof(double) * 4 * (size_t)T * B;
  const size_t b_f = sizeof(double) * (size_t)B * n_ticks, b_i = sizeof(int) * (size_t)B * n_ticks;
  rc = ensure_stage(h, al(b_states) + al(b_obs) + al(b_p) + 2 * al(b_x) + al(b_f) + 2 * al(b_i));
  if (rc) return rc;
  char* base = (char*)h->stage;
  double* d_states = (double*)base; base += al(b_states);
  double* d_obs = (double*)base; base += al(b_obs);
  double* d_p = (double*)base; base += al(b_p);
  double* d_xa = (double*)base; base += al(b_x);
  double* d_xb = (double*)base; base += al(b_x);
  double* d_f = (double*)base; base += al(b_f);
  int* d_it = (int*)base; base += al(b_i);
  int* d_st = (int*)base;
  hipStream_t s = h->stream;
  HIPCHK(hipMemcpyAsync(d_states, state0, sizeof(double) * 4 * (size_t)B, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d_obs, obs_table, b_obs, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(d_xa, 0, b_x, s));  // first tick: zero seed (solver.py:76)
  HIPCHK(hipEventRecord(h->ev0, s));
  for (int k = 0; k < n_ticks; ++k) {
    double* st_k = d_states + 4 * (size_t)B * k;
    oh_launch_pm_tick_params(s, B, T, k, advance, ramp, st_k, d_obs, d_p);
    double* x_seed = (k & 1) ? d_xb : d_xa;  // previous solution = warm start of this tick
    double* x_sol = (k & 1) ? d_xa : d_xb;
    oh_launch_pm_solve(s, h->PmP, h->PmD, x_seed, d_p, x_sol, d_f + (size_t)B * k, nullptr, d_it + (size_t)B * k, d_st + (size_t)B * k);
    oh_launch_pm_advance(s, B, T, advance, x_sol, st_k + 4 * (size_t)B);
  }
  HIPCHK(hipEventRecord(h->ev1, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms;
  h->timing[5] = n_ticks;
  h->last_B = B;
  if (states) HIPCHK(hipMemcpy(states, d_states, b_states, hipMemcpyDeviceToHost));
  if (f) HIPCHK(hipMemcpy(f, d_f, b_f, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(hipMemcpy(iters, d_it, b_i, hipMemcpyDeviceToHost));
  if (status) HIPCHK(hipMemcpy(status, d_st, b_i, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_tq_rollout(oh_handle* h, int B, int n_ticks, int advance, double mu_warm, const double* state0, const double* goal_table, double* states,
                             double* tau0, double* f, int* iters, int* status) {
  if (!h || !state0 || !goal_table) return fail(OH_ERR_INVALID, "oh_tq_rollout: null argument");
  if (h->desc.kind != OH_PROBLEM_TORQUE_MPC) return fail(OH_ERR_STATE, "oh_tq_rollout: handle is not a torque-MPC problem");
  const int T = h->tq.T, N = h->tq.ndof;
  if (B < 1 || n_ticks < 1 || advance < 1 || advance >= T) return fail(OH_ERR_INVALID, "oh_tq_rollout: need B >= 1, n_ticks >= 1, 1 <= advance < T");
  if (!(mu_warm > 0.0)) mu_warm = 1e-6;
  HIPCHK(hipSetDevice(h->device));
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t n_rows = (size_t)n_ticks * advance + T;
  const size_t nx = 4 * (size_t)N * T, np_ = 2 * (size_t)N + 3 * (size_t)T;
  const size_t b_states = sizeof(double) * 2 * N * (size_t)B * (n_ticks + 1), b_goal = sizeof(double) * 3 * n_rows * B, b_p = sizeof(double) * np_ * B,
               b_x = sizeof(double) * nx * B, b_tau = sizeof(double) * N * (size_t)B * n_ticks, b_f = sizeof(double) * (size_t)B * n_ticks,
               b_i = sizeof(int) * (size_t)B * n_ticks;
  int rc = ensure_stage(h, al(b_states) + al(b_goal) + al(b_p) + 2 * al(b_x) + al(b_tau) + al(b_f) + 2 * al(b_i));
  if (rc) return rc;
  char* base = (char*)h->stage;
  double* d_states = (double*)base; base += al(b_states);
  double* d_goal = (double*)base; base += al(b_goal);
  double* d_p = (double*)base; base += al(b_p);
  double* d_xa = (double*)base; base += al(b_x);
  double* d_xb = (double*)base; base += al(b_x);
  double* d_tau = (double*)base; base += al(b_tau);
  double* d_f = (double*)base; base += al(b_f);
  int* d_it = (int*)base; base += al(b_i);
  int* d_st = (int*)base;
  hipStream_t s = h->stream;
  HIPCHK(hipMemcpyAsync(d_states, state0, sizeof(double) * 2 * N * (size_t)B, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d_goal, goal_table, b_goal, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(d_xa, 0, b_x, s));  // first tick: zero accelerations (the cold solve of oh_solve from a constant-configuration seed)
  double ms_total = 0.0, launched = 0.0, work = 0.0;
  for (int k = 0; k < n_ticks; ++k) {
    double* st_k = d_states + 2 * (size_t)N * B * k;
    oh_launch_tq_tick_params(s, B, T, N, k * advance, (int)n_rows, st_k, d_goal, d_p);
    double* x_seed = d_xa;  // the seed of this tick; the solution lands in d_xb and is shifted back into d_xa for the next one
    double* x_sol = d_xb;
    rc = tq_solve_device(h, B, x_seed, d_p, x_sol, d_f + (size_t)B * k, nullptr, d_it + (size_t)B * k, d_st + (size_t)B * k, k > 0 ? mu_warm : 0.0);
    if (rc) return rc;
    ms_total += h->timing[4];
    launched += h->timing[5];
    work += h->timing[6];
    oh_launch_tq_advance(s, B, T, N, advance, x_sol, st_k + 2 * (size_t)N * B, tau0 ? d_tau + (size_t)N * B * k : nullptr);
    oh_launch_tq_shift_seed(s, B, T, N, advance, x_sol, x_seed);
  }
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  for (double& t : h->timing) t = 0.0;
  h->timing[4] = ms_total;  // device time of the solves (HIP events around each)
  h->timing[5] = launched;
  h->timing[6] = work;
  h->last_B = B;
  if (states) HIPCHK(hipMemcpy(states, d_states, b_states, hipMemcpyDeviceToHost));
  if (tau0) HIPCHK(hipMemcpy(tau0, d_tau, b_tau, hipMemcpyDeviceToHost));
  if (f) HIPCHK(hipMemcpy(f, d_f, b_f, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(hipMemcpy(iters, d_it, b_i, hipMemcpyDeviceToHost));
  if (status) HIPCHK(hipMemcpy(status, d_st, b_i, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_get_multipliers(oh_handle* h, int B, double* lam_h) {
  if (!h || !lam_h) return fail(OH_ERR_INVALID, "oh_get_multipliers: null argument");
  if ((h->desc.kind != OH_PROBLEM_FIGURE_EIGHT && h->desc.kind != OH_PROBLEM_IK && h->desc.kind != OH_PROBLEM_QP && h->desc.kind != OH_PROBLEM_TAPE &&
       h->desc.kind != OH_PROBLEM_TORQUE_MPC) ||
      B != h->last_B || B < 1)
    return fail(OH_ERR_STATE, "oh_get_multipliers: B does not match the last solve");
  HIPCHK(hipSetDevice(h->device));
  if (h->pipe_last) {  // the last solve was a pipelined oh_solve: every chunk left its multipliers in the handle's cache
    const size_t per = h->desc.kind == OH_PROBLEM_TORQUE_MPC ? (size_t)h->tq.T * (h->tq.vel_limits ? 4 : 2) * h->tq.ndof : (h->desc.lock_orientation ? 4 * (size_t)h->desc.T : 0);
    if (!per) return fail(OH_ERR_STATE, "oh_get_multipliers: this problem has no nonlinear equality rows");
    HIPCHK(hipMemcpy(lam_h, h->d_pipe_mult, sizeof(double) * per * B, hipMemcpyDeviceToHost));
    return OH_OK;
  }
  if (h->desc.kind == OH_PROBLEM_TORQUE_MPC) {
    const size_t per = (size_t)h->tq.T * (h->tq.vel_limits ? 4 : 2) * h->tq.ndof;
    if (!h->split_parts.empty()) {  // the last solve ran in parts (solve_split)
      size_t o = 0;
      for (size_t i = 0; i < h->split_parts.size(); ++i) {
        const oh_handle* q = i == 0 ? h : h->peers[i - 1];
        HIPCHK(hipMemcpy(lam_h + o * per, q->d_tq_mult, sizeof(double) * per * h->split_parts[i], hipMemcpyDeviceToHost));
        o += (size_t)h->split_parts[i];
      }
      return OH_OK;
    }
    HIPCHK(hipMemcpy(lam_h, h->d_tq_mult, sizeof(double) * (size_t)B * per, hipMemcpyDeviceToHost));
    return OH_OK;
  }
  if (h->desc.kind == OH_PROBLEM_TAPE) {
    if (h->TP.n_ineq + h->TP.n_eq > 0)
      HIPCHK(hipMemcpy(lam_h, h->d_tape_mult, sizeof(double) * (size_t)(h->TP.n_ineq + h->TP.n_eq) * B, hipMemcpyDeviceToHost));
    return OH_OK;
  }
  if (h->desc.kind == OH_PROBLEM_QP) {
    if (h->qp.m + h->qp.me > 0) HIPCHK(hipMemcpy(lam_h, h->d_qp_mult, sizeof(double) * (size_t)(h->qp.m + h->qp.me) * B, hipMemcpyDeviceToHost));
    return OH_OK;
  }
  if (h->desc.kind == OH_PROBLEM_IK) {
    HIPCHK(hipMemcpy(lam_h, h->d_ik_mult, sizeof(double) * (3 + 2 * (size_t)h->ik.ndof) * B, hipMemcpyDeviceToHost));
    return OH_OK;
  }
  if (h->have_guards) {
    // SoA [T][NC][Bp] on the device -> [B][T][NC (+ 2 ndof velocity rows)] for the caller
    auto fetch = [&](const oh_handle* q, const int n, double* out) -> int {
      const int T = q->desc.T, NC = q->GP.NC, Bp = q->D.Bp, NV = q->GP.vel ? 2 * q->desc.ndof : 0, NT = NC + NV;
      std::vector<double> tmp((size_t)T * NC * Bp);
      if (NC) HIPCHK(hipMemcpy(tmp.data(), q->GB.lam_out, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));  // original order (k_guard_emit)
      for (int b = 0; b < n; ++b)
        for (int t = 0; t < T; ++t)
          for (int i = 0; i < NC; ++i) out[((size_t)b * T + t) * NT + i] = tmp[((size_t)t * NC + i) * Bp + b];
      if (NV) {  // the device keeps the rows of dq_t = interval (t, t+1) in row block t + 1
        std::vector<double> tv((size_t)T * NV * Bp);
        HIPCHK(hipMemcpy(tv.data(), q->GB.lamv_out, tv.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < n; ++b)
          for (int t = 0; t < T; ++t)
            for (int i = 0; i < NV; ++i) out[((size_t)b * T + t) * NT + NC + i] = (t + 1 < T) ? tv[((size_t)(t + 1) * NV + i) * Bp + b] : 0.0;
      }
      return OH_OK;
    };
    if (h->split_parts.empty()) return fetch(h, B, lam_h);
    const size_t per = (size_t)h->desc.T * ((h->guards.limits ? 2 * h->desc.ndof : 0) + h->guards.n_links * h->guards.n_obstacles + (h->guards.vel_limits ? 2 * h->desc.ndof : 0));
    size_t o = 0;
    for (size_t i = 0; i < h->split_parts.size(); ++i) {  // the last solve ran in parts (solve_split)
      const int rc = fetch(i == 0 ? h : h->peers[i - 1], h->split_parts[i], lam_h + o * per);
      if (rc) return rc;
      o += (size_t)h->split_parts[i];
    }
    return OH_OK;
  }
  if (!h->D.lam_h) return fail(OH_ERR_STATE, "oh_get_multipliers: this problem has no nonlinear equality rows");
  if (!h->split_parts.empty()) {  // the last solve ran in parts (solve_split): every part keeps the multipliers of its instances
    size_t o = 0;
    for (size_t i = 0; i < h->split_parts.size(); ++i) {
      const oh_handle* q = i == 0 ? h : h->peers[i - 1];
      HIPCHK(hipMemcpy(lam_h + o * 4 * (size_t)h->desc.T, q->D.lam_h, sizeof(double) * 4 * (size_t)h->desc.T * h->split_parts[i], hipMemcpyDeviceToHost));
      o += (size_t)h->split_parts[i];
    }
    return OH_OK;
  }
  HIPCHK(hipMemcpy(lam_h, h->D.lam_h, sizeof(double) * 4 * (size_t)h->desc.T * B, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_set_dynamics(oh_handle* h, const oh_dynamics* dyn) {
  if (!h || !dyn) return fail(OH_ERR_INVALID, "oh_set_dynamics: null argument");
  if (dyn->n < 2 || dyn->n > OH_MAX_BODIES - 1 || dyn->ndof != dyn->n - 1)
    return fail(OH_ERR_INVALID, "oh_set_dynamics: need 2 <= n <= 9 bodies and ndof == n - 1");
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_dyn) HIPCHK(hipMalloc((void**)&h->d_dyn, sizeof(oh_dynamics)));
  HIPCHK(hipMemcpy(h->d_dyn, dyn, sizeof(oh_dynamics), hipMemcpyHostToDevice));
  h->dyn_host = *dyn;
  h->have_dyn = true;
  drop_peers(h);
  return OH_OK;
}
extern "C" int oh_rnea_device(oh_handle* h, int n, const void* d_q, const void* d_qd, const void* d_qdd, void* d_tau) {
  if (!h) return fail(OH_ERR_INVALID, "oh_rnea: null handle");
  if (n < 1 || !d_q || !d_qd || !d_qdd || !d_tau) return fail(OH_ERR_INVALID, "oh_rnea: bad arguments");
  if (!h->have_dyn) return fail(OH_ERR_STATE, "oh_rnea: call oh_set_dynamics first");
  HIPCHK(hipSetDevice(h->device));
  if (!oh_launch_rnea(h->stream, h->d_dyn, h->dyn_host.n, n, (const double*)d_q, (const double*)d_qd, (const double*)d_qdd, (double*)d_tau))
    return fail(OH_ERR_INVALID, "oh_rnea: unsupported number of bodies");
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return OH_OK;
}
extern "C" int oh_rnea(oh_handle* h, int n, const double* q, const double* qd, const double* qdd, double* tau) {
  if (!h) return fail(OH_ERR_INVALID, "oh_rnea: null handle");
  if (n < 1 || !q || !qd || !qdd || !tau) return fail(OH_ERR_INVALID, "oh_rnea: bad arguments");
  if (!h->have_dyn) return fail(OH_ERR_STATE, "oh_rnea: call oh_set_dynamics first");
  HIPCHK(hipSetDevice(h->device));
  const size_t bq = (sizeof(double) * h->dyn_host.ndof * (size_t)n + 255) / 256 * 256;
  int rc = ensure_stage(h, 4 * bq);
  if (rc) return rc;
  char* base = (char*)h->stage;
  HIPCHK(hipMemcpy(base, q, sizeof(double) * h->dyn_host.ndof * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + bq, qd, sizeof(double) * h->dyn_host.ndof * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + 2 * bq, qdd, sizeof(double) * h->dyn_host.ndof * (size_t)n, hipMemcpyHostToDevice));
  rc = oh_rnea_device(h, n, base, base + bq, base + 2 * bq, base + 3 * bq);
  if (rc) return rc;
  HIPCHK(hipMemcpy(tau, base + 3 * bq, sizeof(double) * h->dyn_host.ndof * (size_t)n, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_rnea_jac(oh_handle* h, int n, const double* q, const double* qd, const double* qdd, double* J) {
  if (!h) return fail(OH_ERR_INVALID, "oh_rnea_jac: null handle");
  if (n < 1 || !q || !qd || !qdd || !J) return fail(OH_ERR_INVALID, "oh_rnea_jac: bad arguments");
  if (!h->have_dyn) return fail(OH_ERR_STATE, "oh_rnea_jac: call oh_set_dynamics first");
  HIPCHK(hipSetDevice(h->device));
  const size_t nd = h->dyn_host.ndof;
  const size_t bq = (sizeof(double) * nd * (size_t)n + 255) / 256 * 256, bj = sizeof(double) * nd * 3 * nd * (size_t)n;
  int rc = ensure_stage(h, 3 * bq + bj);
  if (rc) return rc;
  char* base = (char*)h->stage;
  HIPCHK(hipMemcpy(base, q, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + bq, qd, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + 2 * bq, qdd, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  if (!oh_launch_rnea_jac(h->stream, h->d_dyn, h->dyn_host.n, n, (const double*)base, (const double*)(base + bq), (const double*)(base + 2 * bq),
                          (double*)(base + 3 * bq)))
    return fail(OH_ERR_INVALID, "oh_rnea_jac: unsupported number of bodies");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(J, base + 3 * bq, bj, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_rnea_hess(oh_handle* h, int n, const double* q, const double* qd, const double* qdd, const double* c, double* H) {
  if (!h) return fail(OH_ERR_INVALID, "oh_rnea_hess: null handle");
  if (n < 1 || !q || !qd || !qdd || !c || !H) return fail(OH_ERR_INVALID, "oh_rnea_hess: bad arguments");
  if (!h->have_dyn) return fail(OH_ERR_STATE, "oh_rnea_hess: call oh_set_dynamics first");
  HIPCHK(hipSetDevice(h->device));
  const size_t nd = h->dyn_host.ndof;
  const size_t bq = (sizeof(double) * nd * (size_t)n + 255) / 256 * 256, bh = sizeof(double) * 9 * nd * nd * (size_t)n;
  int rc = ensure_stage(h, 4 * bq + bh);
  if (rc) return rc;
  char* base = (char*)h->stage;
  HIPCHK(hipMemcpy(base, q, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + bq, qd, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + 2 * bq, qdd, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + 3 * bq, c, sizeof(double) * nd * (size_t)n, hipMemcpyHostToDevice));
  if (!oh_launch_rnea_hess(h->stream, h->d_dyn, h->dyn_host.n, n, (const double*)base, (const double*)(base + bq), (const double*)(base + 2 * bq),
                           (const double*)(base + 3 * bq), (double*)(base + 4 * bq)))
    return fail(OH_ERR_INVALID, "oh_rnea_hess: unsupported number of bodies");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(H, base + 4 * bq, bh, hipMemcpyDeviceToHost));
  return OH_OK;
}

static int fk_common(oh_handle* h, int n, bool soa, const void* d_q, void* d_pose, void* d_J) {
  if (!h) return fail(OH_ERR_INVALID, "oh_fk_jac: null handle");
  if (n < 1 || !d_q) return fail(OH_ERR_INVALID, "oh_fk_jac: bad arguments");
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_fk_jac: call oh_set_constants first");
  HIPCHK(hipSetDevice(h->device));
  if (!h->fk_spec && !h->fk_spec_failed && (h->specialize == OH_SPECIALIZE_ALWAYS || (h->specialize == OH_SPECIALIZE_AUTO && n >= h->specialize_min_units)))
    specialize_fk(h);  // on failure the generic kernel runs; oh_last_error keeps the reason
  if (h->fk_spec) HIPCHK(oh_spec_launch_fk(*h->fk_spec, h->stream, soa, n, (const double*)d_q, (double*)d_pose, (double*)d_J));
  else oh_launch_fk_jac(h->stream, soa, h->d_chain, h->chain_host.n_chain, h->chain_host.ndof, n, (const double*)d_q, (double*)d_pose, (double*)d_J);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return OH_OK;
}
extern "C" int oh_fk_jac_device(oh_handle* h, int n, const void* d_q, void* d_pose, void* d_J) {
  return fk_common(h, n, false, d_q, d_pose, d_J);
}
extern "C" int oh_fk_jac_soa_device(oh_handle* h, int n, const void* d_q, void* d_pose, void* d_J) {
  return fk_common(h, n, true, d_q, d_pose, d_J);
}
extern "C" int oh_fk_jac(oh_handle* h, int n, const double* q, double* pose, double* J) {
  if (!h) return fail(OH_ERR_INVALID, "oh_fk_jac: null handle");
  if (n < 1 || !q) return fail(OH_ERR_INVALID, "oh_fk_jac: bad arguments");
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_fk_jac: call oh_set_constants first");
  HIPCHK(hipSetDevice(h->device));
  const int ndof = h->chain_host.ndof;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t b_q = sizeof(double) * ndof * (size_t)n, b_p = sizeof(double) * 7 * (size_t)n,
               b_J = sizeof(double) * 6 * ndof * (size_t)n;
  int rc = ensure_stage(h, al(b_q) + al(b_p) + al(b_J));
  if (rc) return rc;
  char* base = (char*)h->stage;
  void* d_q = base; base += al(b_q);
  void* d_pose = base; base += al(b_p);
  void* d_J = base;
  HIPCHK(hipMemcpy(d_q, q, b_q, hipMemcpyHostToDevice));
  rc = fk_common(h, n, false, d_q, pose ? d_pose : nullptr, J ? d_J : nullptr);
  if (rc) return rc;
  if (pose) HIPCHK(hipMemcpy(pose, d_pose, b_p, hipMemcpyDeviceToHost));
  if (J) HIPCHK(hipMemcpy(J, d_J, b_J, hipMemcpyDeviceToHost));
  return OH_OK;
}

extern "C" int oh_set_profiling(oh_handle* h, int enable) {
  if (!h) return fail(OH_ERR_INVALID, "oh_set_profiling: null handle");
  h->profiling = enable != 0;
  return OH_OK;
}
extern "C" int oh_get_timing(oh_handle* h, double* out8) {
  if (!h || !out8) return fail(OH_ERR_INVALID, "oh_get_timing: null argument");
  for (int i = 0; i < 8; ++i) out8[i] = h->timing[i];
  out8[8] = h->timing_couple;
  out8[9] = h->rejects;
  out8[10] = h->tail_iters;
  return OH_OK;
}
extern "C" int oh_event_timer_start(oh_handle* h) {
  if (!h) return fail(OH_ERR_INVALID, "oh_event_timer_start: null handle");
  HIPCHK(hipEventRecord(h->evt0, h->stream));
  return OH_OK;
}
extern "C" int oh_event_timer_stop(oh_handle* h, double* ms) {
  if (!h || !ms) return fail(OH_ERR_INVALID, "oh_event_timer_stop: null argument");
  HIPCHK(hipEventRecord(h->evt1, h->stream));
  HIPCHK(hipEventSynchronize(h->evt1));
  float f = 0.f;
  HIPCHK(hipEventElapsedTime(&f, h->evt0, h->evt1));
  *ms = f;
  return OH_OK;
}

extern "C" void oh_destroy(oh_handle* h) {
  if (!h) return;
  for (oh_handle* p : h->peers) oh_destroy(p);
  h->peers.clear();
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (hipEvent_t e : h->prof_events) hipEventDestroy(e);
  if (h->pool) hipFree(h->pool);
  if (h->tq_pool) hipFree(h->tq_pool);
  if (h->d_tq_mult) hipFree(h->d_tq_mult);
  if (h->d_tq_hc) hipFree(h->d_tq_hc);
  if (h->d_ik_mult) hipFree(h->d_ik_mult);
  if (h->gpool) hipFree(h->gpool);
  if (h->move_scr) hipFree(h->move_scr);
  if (h->d_pipe_mult) hipFree(h->d_pipe_mult);
  if (h->d_qp_work) hipFree(h->d_qp_work);
  for (void* q : {(void*)h->d_tape_op, (void*)h->d_tape_a, (void*)h->d_tape_b, (void*)h->d_tape_rows, (void*)h->d_tape_c, (void*)h->d_tape_work, (void*)h->d_tape_mult,
                  (void*)h->d_tape_h0})
    if (q) hipFree(q);
  if (h->d_qp_mult) hipFree(h->d_qp_mult);
  for (void* q : {(void*)h->d_qp_rows, (void*)h->d_qp_val, (void*)h->d_qp_f0, (void*)h->d_qp_xdep})
    if (q) hipFree(q);
  if (h->stage) hipFree(h->stage);
  if (h->d_chain) hipFree(h->d_chain);
  if (h->d_dyn) hipFree(h->d_dyn);
  if (h->d_local_path) hipFree(h->d_local_path);
  if (h->h_flag) hipHostFree(h->h_flag);
  if (h->h_stage) hipHostFree(h->h_stage);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->evt0) hipEventDestroy(h->evt0);
  if (h->evt1) hipEventDestroy(h->evt1);
  oh_tape_jit_release(&h->tape_jit);
  oh_tape_wave_release(&h->tape_wave);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Multi-GPU: one process per GPU, instances sharded by the host, no data-path collective.  The single exchange of a job is the
// broadcast of the URDF-derived constants from one rank; the library owns the RCCL communicator for it (and for the
// barrier / MAX / SUM reductions a benchmark harness needs), so a ctypes host needs no other GPU runtime.  librccl is opened on first
// use: single-GPU processes never load it.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;  // optional (oh_comm_allgather)
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // optional
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;  // optional
};
RcclApi g_rccl;
ncclComm_t g_comm = nullptr;
int g_comm_rank = -1, g_comm_world = 0, g_comm_device = 0;
hipStream_t g_comm_stream = nullptr;
double* g_comm_scratch = nullptr;  // device, 2 doubles

int rccl_load() {
  if (g_rccl.lib) return OH_OK;
  void* lib = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) return fail(OH_ERR_HIP, std::string("oh_comm: cannot open librccl: ") + dlerror());
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
  a.Broadcast = (decltype(a.Broadcast))dlsym(lib, "ncclBroadcast");
  a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
  a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
  a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
  a.CommUserRank = (decltype(a.CommUserRank))dlsym(lib, "ncclCommUserRank");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.Broadcast || !a.AllReduce || !a.GetErrorString) {
    dlclose(lib);
    return fail(OH_ERR_HIP, "oh_comm: librccl lacks an expected symbol");
  }
  g_rccl = a;
  return OH_OK;
}
int rccl_fail(const char* what, ncclResult_t r) { return fail(OH_ERR_HIP, std::string(what) + ": " + g_rccl.GetErrorString(r)); }
#define RCCLCHK(expr)                                  \
  do {                                                 \
    ncclResult_t _r = (expr);                          \
    if (_r != ncclSuccess) return rccl_fail(#expr, _r); \
  } while (0)
}  // namespace

extern "C" int oh_comm_unique_id(char* id) {
  if (!id) return fail(OH_ERR_INVALID, "oh_comm_unique_id: null");
  static_assert(OH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (const int rc = rccl_load()) return rc;
  ncclUniqueId u;
  RCCLCHK(g_rccl.GetUniqueId(&u));
  memcpy(id, u.internal, OH_COMM_ID_BYTES);
  return OH_OK;
}

extern "C" int oh_comm_init(int rank, int world, const char* id) {
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(OH_ERR_INVALID, "oh_comm_init: bad rank / world / id");
  if (g_comm) return fail(OH_ERR_STATE, "oh_comm_init: this process already holds a communicator");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return fail(OH_ERR_HIP, "oh_comm_init: no HIP device available (this library has no CPU path)");
  if (const int rc = rccl_load()) return rc;
  HIPCHK(hipGetDevice(&g_comm_device));  // the device selected with oh_set_device
  ncclUniqueId u;
  memcpy(u.internal, id, OH_COMM_ID_BYTES);
  HIPCHK(hipStreamCreate(&g_comm_stream));
  // (oh_chain-sized: the broadcast of the constants lands here first and is validated before a handle adopts it)
  if (hipMalloc((void**)&g_comm_scratch, 2 * sizeof(double) + sizeof(oh_chain)) != hipSuccess) {
    hipStreamDestroy(g_comm_stream);
    g_comm_stream = nullptr;
    return fail(OH_ERR_HIP, "oh_comm_init: scratch allocation failed");
  }
  const ncclResult_t ir = g_rccl.CommInitRank(&g_comm, world, u, rank);
  if (ir != ncclSuccess) {  // a retry must not leak the stream and the scratch
    g_comm = nullptr;
    hipFree(g_comm_scratch);
    g_comm_scratch = nullptr;
    hipStreamDestroy(g_comm_stream);
    g_comm_stream = nullptr;
    return rccl_fail("ncclCommInitRank", ir);
  }
  g_comm_rank = rank;
  g_comm_world = world;
  return OH_OK;
}

extern "C" int oh_comm_destroy(void) {
  if (!g_comm) return OH_OK;
  hipSetDevice(g_comm_device);
  hipStreamSynchronize(g_comm_stream);
  g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  hipFree(g_comm_scratch);
  g_comm_scratch = nullptr;
  hipStreamDestroy(g_comm_stream);
  g_comm_stream = nullptr;
  g_comm_rank = -1;
  g_comm_world = 0;
  return OH_OK;
}

static int comm_allreduce(double* value, ncclRedOp_t op, const char* who) {
  if (!value) return fail(OH_ERR_INVALID, std::string(who) + ": null");
  if (!g_comm) return fail(OH_ERR_STATE, std::string(who) + ": call oh_comm_init first");
  HIPCHK(hipSetDevice(g_comm_device));
  HIPCHK(hipMemcpyAsync(g_comm_scratch, value, sizeof(double), hipMemcpyHostToDevice, g_comm_stream));
  RCCLCHK(g_rccl.AllReduce(g_comm_scratch, g_comm_scratch + 1, 1, ncclFloat64, op, g_comm, g_comm_stream));
  HIPCHK(hipMemcpyAsync(value, g_comm_scratch + 1, sizeof(double), hipMemcpyDeviceToHost, g_comm_stream));
  HIPCHK(hipStreamSynchronize(g_comm_stream));
  return OH_OK;
}
extern "C" int oh_comm_allreduce_max(double* value) { return comm_allreduce(value, ncclMax, "oh_comm_allreduce_max"); }
extern "C" int oh_comm_allreduce_sum(double* value) { return comm_allreduce(value, ncclSum, "oh_comm_allreduce_sum"); }
extern "C" int oh_comm_barrier(void) {
  double one = 1.0;
  return comm_allreduce(&one, ncclSum, "oh_comm_barrier");
}

extern "C" int oh_comm_broadcast_constants(oh_handle* h, int root) {
  if (!h) return fail(OH_ERR_INVALID, "oh_comm_broadcast_constants: null handle");
  if (!g_comm) return fail(OH_ERR_STATE, "oh_comm_broadcast_constants: call oh_comm_init first");
  if (root < 0 || root >= g_comm_world) return fail(OH_ERR_INVALID, "oh_comm_broadcast_constants: bad root");
  if (!h->d_chain) return fail(OH_ERR_STATE, "oh_comm_broadcast_constants: this handle takes no kinematic constants");
  if (g_comm_rank == root && !h->have_chain) return fail(OH_ERR_STATE, "oh_comm_broadcast_constants: the root must call oh_set_constants first");
  HIPCHK(hipSetDevice(h->device));
  // one ncclBroadcast of the oh_chain block (2952 B) on the handle's stream: out of the root's constants buffer, into a scratch block on the
  // other ranks -- a chain this handle rejects (wrong ndof, unsupported joint) must not have replaced its constants already
  unsigned char* const land = (unsigned char*)(g_comm_scratch + 2);
  RCCLCHK(g_rccl.Broadcast(h->d_chain, g_comm_rank == root ? (void*)h->d_chain : (void*)land, sizeof(oh_chain), ncclUint8, root, g_comm, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (g_comm_rank != root) {
    oh_chain tmp;
    HIPCHK(hipMemcpy(&tmp, land, sizeof(oh_chain), hipMemcpyDeviceToHost));
    if (const int rc = validate_chain(h, tmp)) return rc;
    HIPCHK(hipMemcpy(h->d_chain, land, sizeof(oh_chain), hipMemcpyDeviceToDevice));
    adopt_chain(h, tmp);
  }
  return OH_OK;
}

// The optional gather of SURVEY 8(e): every rank contributes `bytes` bytes of a device buffer (objectives, statuses, or whole solutions of its shard) and
// receives all ranks' blocks in rank order -- one ncclAllGather over xGMI on the communicator's stream, after the solves; never part of the data path.
extern "C" int oh_comm_allgather(const void* d_send, void* d_recv, size_t bytes) {
  if (!d_send || !d_recv || bytes == 0) return fail(OH_ERR_INVALID, "oh_comm_allgather: null buffer or zero size");
  if (!g_comm) return fail(OH_ERR_STATE, "oh_comm_allgather: call oh_comm_init first");
  if (!g_rccl.AllGather) return fail(OH_ERR_HIP, "oh_comm_allgather: librccl lacks ncclAllGather");
  HIPCHK(hipSetDevice(g_comm_device));
  RCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, g_comm, g_comm_stream));
  HIPCHK(hipStreamSynchronize(g_comm_stream));
  return OH_OK;
}

// world size and rank as RCCL sees them (a harness prints them to prove the communicator spans the job)
extern "C" int oh_comm_info(int* rank, int* world) {
  if (!g_comm) return fail(OH_ERR_STATE, "oh_comm_info: call oh_comm_init first");
  int r = -1, w = 0;
  if (g_rccl.CommUserRank) RCCLCHK(g_rccl.CommUserRank(g_comm, &r)); else r = g_comm_rank;
  if (g_rccl.CommCount) RCCLCHK(g_rccl.CommCount(g_comm, &w)); else w = g_comm_world;
  if (rank) *rank = r;
  if (world) *world = w;
  return OH_OK;
}

extern "C" int oh_get_constants(oh_handle* h, oh_chain* out) {
  if (!h || !out) return fail(OH_ERR_INVALID, "oh_get_constants: null argument");
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_get_constants: no constants set");
  *out = h->chain_host;
  return OH_OK;
}

static bool spec_applies(const oh_handle* h) {
  return h->desc.kind == OH_PROBLEM_FIGURE_EIGHT && h->have_chain && h->desc.lock_orientation && !h->have_guards && !h->chain_host.has_lead;
}
// handles with joint / joint-velocity limits only: their persistent kernel (k_tail_vel) comes out of the same module; the batched guarded
// kernels stay generic
static bool spec_tail_vel_applies(const oh_handle* h) {
  return h->desc.kind == OH_PROBLEM_FIGURE_EIGHT && h->have_chain && h->desc.lock_orientation && h->have_guards && !h->chain_host.has_lead &&
         (h->guards.limits || h->guards.vel_limits) && h->guards.n_links == 0 && h->tail_vel;
}
static int specialize_fk(oh_handle* h) {
  if (h->fk_spec) return OH_OK;
  std::string err;
  const FkSpec* sp = nullptr;
  if (oh_jit_fkjac(h->chain_host, &sp, &err)) {
    h->fk_spec_failed = true;
    return fail(OH_ERR_HIP, "oh_specialize: " + err);
  }
  h->fk_spec = sp;
  return OH_OK;
}
extern "C" int oh_specialize(oh_handle* h) {
  if (!h) return fail(OH_ERR_INVALID, "oh_specialize: null handle");
  if (!h->have_chain) return fail(OH_ERR_STATE, "oh_specialize: call oh_set_constants first");
  HIPCHK(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  int rc = specialize_fk(h);  // K1: every handle with constants
  if (rc == OH_OK && (spec_applies(h) || spec_tail_vel_applies(h)) && !h->spec) {
    std::string err;
    const FigSpec* sp = nullptr;
    if (oh_jit_figure8(h->chain_host, h->desc.ndof, &sp, &err)) {
      h->spec_failed = true;
      rc = fail(OH_ERR_HIP, "oh_specialize: " + err);
    } else h->spec = sp;
  }
  h->spec_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
extern "C" int oh_specialize_compile(const oh_chain* chain, double* info2) {
  if (!chain) return fail(OH_ERR_INVALID, "oh_specialize_compile: null chain");
  if (chain->ndof < 4 || chain->ndof > OH_MAX_CHAIN || chain->n_chain != chain->ndof || chain->has_lead)
    return fail(OH_ERR_INVALID, "oh_specialize_compile: the chain must cover every model joint in order (4..16 joints, no lead joint)");
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<char> code;
  bool from_disk = false;
  std::string err;
  bool fd2 = false;
  if (oh_jit_compile_cached(oh_jit_figure8_source(*chain, chain->ndof), &code, &from_disk, &err) || oh_jit_compile_cached(oh_jit_fkjac_source(*chain), &code, &fd2, &err))
    return fail(OH_ERR_HIP, "oh_specialize_compile: " + err);
  from_disk = from_disk && fd2;
  if (info2) {
    info2[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info2[1] = from_disk ? 1.0 : 0.0;
  }
  return OH_OK;
}
extern "C" int oh_specialize_info(oh_handle* h, double* info4) {
  if (!h || !info4) return fail(OH_ERR_INVALID, "oh_specialize_info: null argument");
  info4[0] = h->spec ? 1.0 : 0.0;
  info4[1] = h->fk_spec ? 1.0 : 0.0;
  info4[2] = h->spec_seconds;
  info4[3] = ((h->spec && h->spec->from_disk) || (!h->spec && h->fk_spec && h->fk_spec->from_disk)) ? 1.0 : 0.0;
  return OH_OK;
}
// How the handle's last solve was (or its next one will be) scheduled, by name: "fuse_couple" (1: coupling folded into evaluation and sweep, no
// k_couple launch), "tail_threshold", "specialized"; tape handles: "tape_wave" (0: thread per instance, 1 / 2: wavefront per instance with the (s, y)
// pairs in global memory / in LDS), "tape_levels", "tape_passes" (dependency levels and 64-instruction passes of one evaluation).
extern "C" int oh_get_flag(oh_handle* h, const char* name, int* value) {
  if (!h || !name || !value) return fail(OH_ERR_INVALID, "oh_get_flag: null argument");
  const std::string n(name);
  if (n == "fuse_couple") {
    fill_params(h);
    *value = h->P.zc;
  } else if (n == "tail_threshold") *value = h->tail_threshold;
  else if (n == "specialized") *value = h->spec ? 1 : 0;
  else if (n == "tape_wave") *value = h->tape_wave.ready ? (h->tape_wave.hist_lds ? 2 : 1) : 0;
  else if (n == "tape_regs_lds") *value = h->tape_wave.ready && h->tape_wave.reg_lds ? 1 : 0;
  else if (n == "tape_metric") *value = (h->TP.h0 && h->TP.lbfgs > 0) ? 1 : 0;
  else if (n == "tape_levels") *value = h->tape_wave.n_levels;
  else if (n == "tape_passes") *value = h->tape_wave.n_fw_pass + h->tape_wave.n_rv_pass;
  else return fail(OH_ERR_INVALID, "oh_get_flag: unknown flag " + n);
  return OH_OK;
}

extern "C" int oh_kernel_info(const char* kernel, int* out5);
extern "C" int oh_kernel_info_handle(oh_handle* h, const char* kernel, int* out5) {
  if (!h || !kernel || !out5) return fail(OH_ERR_INVALID, "oh_kernel_info_handle: null argument");
  OhKernelInfo k{};
  if ((h->spec && oh_spec_kernel_info(*h->spec, kernel, &k)) || (h->fk_spec && !strcmp(kernel, "k_fk_jac") && oh_spec_fk_kernel_info(*h->fk_spec, &k))) {
    out5[0] = k.vgprs; out5[1] = k.scratch_bytes; out5[2] = k.lds_bytes; out5[3] = k.block; out5[4] = k.blocks_per_cu;
    return OH_OK;
  }
  return oh_kernel_info(kernel, out5);
}

extern "C" int oh_kernel_info(const char* kernel, int* out5) {
  if (!kernel || !out5) return fail(OH_ERR_INVALID, "oh_kernel_info: null argument");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return fail(OH_ERR_HIP, "oh_kernel_info: no HIP device available (this library has no CPU path)");
  OhKernelInfo k{};
  if (!oh_kernel_info_figure8(kernel, &k) && !oh_kernel_info_fkjac(kernel, &k) && !oh_kernel_info_torque(kernel, &k))
    return fail(OH_ERR_INVALID, std::string("oh_kernel_info: unknown kernel or attribute query failed: ") + kernel);
  out5[0] = k.vgprs; out5[1] = k.scratch_bytes; out5[2] = k.lds_bytes; out5[3] = k.block; out5[4] = k.blocks_per_cu;
  return OH_OK;
}
