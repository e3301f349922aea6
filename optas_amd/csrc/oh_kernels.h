// Shared host/device declarations between oh_kernels.hip (device code) and oh_api.hip (C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "optas_hip.h"
#include "oh_types.h"

// Code-object facts of a kernel by name (hipFuncGetAttributes + occupancy query): what bench.py reports as roofline.occupancy.
struct OhKernelInfo {
  int vgprs, scratch_bytes, lds_bytes, block, blocks_per_cu;
};
bool oh_kernel_info_figure8(const char* name, OhKernelInfo* out);
bool oh_kernel_info_fkjac(const char* name, OhKernelInfo* out);
bool oh_kernel_info_torque(const char* name, OhKernelInfo* out);

bool oh_launch_rnea_jac(hipStream_t s, const oh_dynamics* d_dyn, int nbodies, int n, const double* q, const double* qd, const double* qdd, double* J);
bool oh_launch_rnea(hipStream_t s, const oh_dynamics* d_dyn, int nbodies, int n, const double* q, const double* qd, const double* qdd, double* tau);
void oh_launch_fk_jac(hipStream_t s, bool soa, const oh_chain* d_chain, int n_chain, int ndof, int n, const double* q, double* pose, double* J);
bool oh_launch_setup(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const double* x0, const double* p);
bool oh_launch_eval(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot, int part = 0);
bool oh_launch_carry(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot);
bool oh_launch_eval_lead(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_couple(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_couple_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
bool oh_launch_step(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_eval_free(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_couple_free(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_couple_free_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
bool oh_launch_step_free(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot, bool pcr = false);  // pcr: one block per instance
// Scheduling choices of the launchers that are not part of a kernel's parameter block.  oh_api.hip fills this from the handle's options
// (oh_set_option) at the start of every call; until round 4 the launchers read OH_* environment variables themselves.
struct OhLaunchOpts {
  int free_bb = 1;         // position-tracking family, 7 joints, <= free_pcr_max instances: twisted factorisation (k_step_free_bb); 0: the cyclic-reduction kernels
  int free_cp_max = 512;   // ... cyclic reduction with eight lanes per knot up to this many instances (horizons <= 64 knots)
  int pm_wave_max = 20480; // point mass: a wavefront per plant up to this many plants
  int qp_mode = -1;        // dense QP: -1 automatic, 0 / 1 / 2 force a work-set placement
  int tape_lds_max = 1 << 30;  // generated tape evaluators: the solver's work set in LDS up to this many instances (0: never)
  int tape_wave_nt = 256;  // wavefront tape evaluator: threads per instance (256 or 64)
  int tape_wave_regs = -1; // ... register file: -1 chosen per launch, 0 global memory, 1 LDS
  int tape_wave_hist = -1; // ... quasi-Newton pairs: -1 in LDS when they fit, 0 global memory
};
OhLaunchOpts& oh_launch_opts();  // of the calling thread (a handle is not thread-safe; the options of the handle in the call)
void oh_launch_move_rows(hipStream_t s, void* arr, void* scr, int rows, int Bp, int B, int Bnew, const int* newidx, bool is_int);
void oh_launch_move_rows_live(hipStream_t s, double* a0, double* a1, double* scr, int rows, int Bp, int B, int Bnew, const int* newidx, const int* cur, const int* curn);
bool oh_launch_setup_guards(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, const double* p);
void oh_launch_guard_infeasible(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const double* p, int B, double* kkt, int* status);
bool oh_launch_eval_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
bool oh_launch_free_persist(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB);  // whole solve, one block per instance
bool oh_launch_step_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot,
                            bool pcr = false);
bool oh_launch_eval_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot, int part);
bool oh_launch_step_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
void oh_launch_guard_emit(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int NV, int only_done);
void oh_launch_guard_compact(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int NV, int phase, int Bnew);
bool oh_launch_tail(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_tail_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
bool oh_launch_finalize(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int only_done, double* x, double* f, double* kkt,
                        int* iters, int* status, int parts = 3);  // parts: 1 the solution x (LDS transpose), 2 scalars + multipliers, 3 both
void oh_launch_scan_running(hipStream_t s, const FigBuffers& D, int sort);
bool oh_launch_compact(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int phase, int Bnew, int slot);

// ---- OH_PROBLEM_POINT_MASS_MPC ---------------------------------------------------------------------------
struct PmParams {
  int T;
  double dt, w_acc, ylim, vlim, safe_sq, tol;
  int max_iter;
  int final_only;  // 1: tracking cost on the last knot only (point_mass_planner.py:43), 0: on every knot (point_mass_mpc.py:131)
  double w_vel;    // weight of sum ||dy_t||^2 (point_mass_planner.py:45-47; 0 in the MPC script)
  int fix_vf;      // 1: terminal row dy_{T-1} = 0 (point_mass_planner.py:34-35)
};
struct PmBuffers {
  int B, Bp;
  double *a, *X, *s, *lam, *K, *kk, *dX, *da;  // [rows][Bp], rows: 2(T-1), 4T, 9T, 9T, 8(T-1), 2(T-1), 4T, 2(T-1)
};
void oh_launch_pm_solve(hipStream_t s, const PmParams& P, const PmBuffers& D, const double* x0, const double* p, double* x, double* f, double* kkt,
                        int* iters, int* status);

void oh_launch_pm_tick_params(hipStream_t s, int B, int T, int tick, int advance, double ramp, const double* state, const double* obs_table, double* p);
void oh_launch_pm_advance(hipStream_t s, int B, int T, int advance, const double* x, double* state_next);

// ---- OH_PROBLEM_TORQUE_MPC (BASELINE configs[4]: RNEA dynamics rows) -------------------------------
// Layouts are unit-contiguous ([instance][knot][...]): the evaluation kernel works with one lane per (instance, knot,
// joint) and the Riccati kernel with 16 lanes per instance, so a wavefront reads whole stage records.
#define TQ_XS 24    // per knot: q (N) at 0, dq (N) at 8, ddq (N) at 16
// per knot stage record: H packed lower (3N)(3N+1)/2 at 0 | g_f (3N) at 231: gradient of the cost | f 252, barrier sum B 253, relaxed rows 254, viol 255 |
// tau (N) at 256 | max lam s 263 | g_b (3N) at 272: gradient of the barrier per unit mu_b (the stage gradient is g_f + mu_b g_b) | d tau / d z (N x 3N
// row-major) at 296: the step kernel's fraction-to-the-boundary rule needs the linearised rows
#define TQ_SD 448
#define TQ_SD_GB 272
#define TQ_SD_J 296
#define TQ_LAM 32   // per knot: multipliers of tau - lo >= 0 (N), then of up - tau >= 0 (N); at 16: of dq - dq_lo >= 0 (N), then of dq_up - dq >= 0 (N)
#define TQ_HC 232   // per knot: the packed lower triangle of the 3N x 3N curvature term (231 entries at N = 7)
#define TQ_GN 128   // per knot: two doubles per lane of k_tq_step (lane 8 r + c: K_q[r][c], K_dq[r][c]; c = 7: k[r], 0)
struct TqParams {
  int T, N, max_iter;
  double dt, w_path, w_vel, w_tau, tol, tol_compl, mu_b0, mu0;
  double theta;      // rows below delta = theta mu_b continue the logarithm by its second-order Taylor polynomial (relaxed barrier)
  double mu_dec;                        // factor on the Levenberg-Marquardt damping after a step whose gain ratio exceeded 0.9 (option tq_mu_dec)
  double kappa_eps, kappa_mu, theta_mu;  // barrier update (Waechter & Biegler 2006, eq. 7): mu_b <- max(mu_min, min(kappa_mu mu_b, mu_b^theta_mu)) once stat <= kappa_eps mu_b
  double curv_from;  // exact Lagrangian curvature in the stage blocks once the reduced gradient is below this ...
  double curv_late;  // ... or below this after curv_after barrier updates (a Gauss-Newton iteration that stalls just above curv_from late in the solve)
  int curv_after;
  int stall_max;     // watchdog: this many steps at one barrier parameter without reaching its test send the instance back to 100 mu_b
  double tau_ftb;    // fraction to the boundary: a step leaves every slack (and multiplier) at least 1 - tau_ftb of itself
  int max_back;      // quarterings of a boundary-shortened step before the damping is raised instead
  int curv_lag;      // exact curvature is evaluated afresh at most every (curv_lag + 1)-th evaluation of an instance, the stored term added in between (option tq_curv_lag; 0: always afresh)
  int ls_curv;       // 1: a rejected Newton step (exact curvature) is quartered like a boundary-shortened one before the damping is raised (option tq_ls_curv)
  double tau_lo[OH_MAX_CHAIN], tau_up[OH_MAX_CHAIN];
  double dq_lo[OH_MAX_CHAIN], dq_up[OH_MAX_CHAIN];  // joint-velocity rows on the velocity states (vel != 0)
  int vel;
  int jac_closed_form;  // d tau / dz in closed form (rnea_idsva) -- the dynamics tables describe a rigid-body chain; 0: dual numbers through the recursion
  int nx, np;
};
struct TqBuffers {
  int B;
  const oh_chain* chain;
  const oh_dynamics* dyn;
  double* xs;      // [2][B][T][TQ_XS]
  double* st;      // [2][B][T][TQ_SD]
  double* lam;     // [2][B][T][TQ_LAM]  (slot = the point they were updated at: a rejected trial leaves the accepted point's multipliers alone)
  double* gains;   // [B][T][TQ_GN]
  double* goal;    // [B][T][4]
  double* hc;      // [cap][T][TQ_HC]  the curvature term k_tq_curv last computed for a knot (entries it never writes stay 0)
  // [B]: merit and cost of the accepted point, its barrier sum, Levenberg-Marquardt damping and its growth factor, barrier parameter, reduced gradient,
  // scale of the feed-forward of the pending trial, q_u^T k and |dx|^2 of the unit step (predicted decrease of a scaled step), violation
  double *f_cur, *f_true, *bsum, *mu, *nun, *mub, *stat, *alpha, *qk, *ndx, *viol;
  int* curv_age;   // [B] evaluations since the stored curvature term was computed (-1: none stored)
  int *cur, *first, *curv, *status, *iters, *rejected, *n_barrier, *nrel, *n_back, *stall;  // [B]
  int* n_running;  // [1]
  int* list;       // [B] instances still running when the list was last rebuilt (kernels walk this list: finished instances cost nothing)
  int* n_list;     // [1]
  int n_run;       // length of the list the launches below cover
};
bool oh_launch_rnea_hess(hipStream_t s, const oh_dynamics* d_dyn, int nbodies, int n, const double* q, const double* qd, const double* qdd, const double* c, double* H);
void oh_launch_tq_list(hipStream_t s, const TqBuffers& D);
bool oh_launch_tq_setup(hipStream_t s, const TqParams& P, const TqBuffers& D, const double* x0, const double* p);
bool oh_launch_tq_eval(hipStream_t s, const TqParams& P, const TqBuffers& D);
bool oh_launch_tq_step(hipStream_t s, const TqParams& P, const TqBuffers& D);
bool oh_launch_tq_finalize(hipStream_t s, const TqParams& P, const TqBuffers& D, double* x, double* f, double* kkt, int* iters, int* status, double* mult);
void oh_launch_tq_tick_params(hipStream_t s, int B, int T, int N, int first_row, int n_rows, const double* state, const double* goal_table, double* p);
void oh_launch_tq_shift_seed(hipStream_t s, int B, int T, int N, int advance, const double* x_prev, double* x_seed);
void oh_launch_tq_advance(hipStream_t s, int B, int T, int N, int advance, const double* x, double* state_next, double* tau0);

// ---- OH_PROBLEM_IK -----------------------------------------------------------------------------------------
struct IkParams {
  int ndof, max_iter;
  double w, tol, tol_feas, rho0;
  double lo[OH_MAX_CHAIN], up[OH_MAX_CHAIN];
};
bool oh_launch_ik_solve(hipStream_t s, const oh_chain* d_chain, const IkParams& P, int B, const double* x0, const double* p, double* x, double* f,
                        double* kkt, int* iters, int* status, double* mult);

// ---- OH_PROBLEM_QP -----------------------------------------------------------------------------------------
struct QpParams {
  int n, m, me, np, nwork, max_iter;
  double tol;
};
void oh_launch_qp_solve(hipStream_t s, const QpParams& Q, int B, int Bp, const double* x0, const double* p, double* work, double* x, double* f, double* kkt,
                        int* iters, int* status, double* mult);  // work: [Q.nwork][Bp] (used when the work set of a block does not fit LDS)

// ---- OH_PROBLEM_TAPE ---------------------------------------------------------------------------------------
#define OH_TAPE_ST_CONVERGED OH_STATUS_CONVERGED
#define OH_TAPE_ST_MAX_ITER OH_STATUS_MAX_ITER
#define OH_TAPE_ST_NUMERICAL OH_STATUS_NUMERICAL
#include "oh_tape_solver.h"  // TapeParams, TapeWork and the solver shared by the interpreter and the generated code
// QP data read off a tape on the device (oh_qp_set_tape): val = [T.len][Bp] work, rows_out = [B][Q.np], f0 = [B]
void oh_launch_qp_assemble(hipStream_t s, const QpParams& Q, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows,
                           const int* xdep, int n_xdep, int B, int Bp, const double* p_raw, double* val, double* rows_out, double* f0);  // xdep: x-dependent instructions
void oh_launch_qp_add_constant(hipStream_t s, int B, double* f, const double* f0);
struct TapeJit {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  hipFunction_t fn_lds = nullptr;  // the same kernel with the solver's work arrays in LDS (small batches)
};
// wavefront-per-instance evaluator of trajectory-sized tapes (oh_tape_wave.hip): the level schedule built when the handle is created
struct TapeWave {
  bool ready = false, hist_lds = false, reg_lds = true;  // placement of the last launch: the quasi-Newton pairs / the tape's registers in LDS or in global memory
  bool reg_lds_fits = false, hist_lds_by[2] = {false, false};  // by placement of the registers: [0] global memory, [1] LDS
  size_t lds_bytes_by[2] = {0, 0};
  int reg_choice = -1;  // OH_TAPE_WAVE_REGS: -1 by batch size, 0 global, 1 lds
  int nt = 256;  // threads per instance
  int n_reg = 0, n_fw_pass = 0, n_rv_pass = 0, n_cst = 0, n_par = 0, n_seed = 0, n_seed_rows = 0, seed_cost = -1, n_small = 0, n_levels = 0;
  size_t lds_bytes = 0;
  int4 *d_fw = nullptr, *d_rv = nullptr;
  int *d_cons = nullptr, *d_cst_reg = nullptr, *d_par_reg = nullptr, *d_par_k = nullptr, *d_small = nullptr;
  double *d_cst_val = nullptr, *d_hist = nullptr, *d_regs = nullptr;
  int hist_cap = 0, regs_cap = 0;
};
int oh_tape_wave_build(const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows, size_t lds_limit, TapeWave* out,
                       std::string* err);
void oh_tape_wave_release(TapeWave* w);
hipError_t oh_launch_tape_wave(hipStream_t s, TapeWave& W, const TapeParams& T, int B, const double* x0, const double* p, double* x, double* f, double* kkt,
                               int* iters, int* status, double* mult);
size_t oh_tape_work_rows(const TapeParams& T, bool jit);
void oh_launch_tape_probe(hipStream_t s, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows, int B, int Bp,
                          const double* x, const double* p, double* work, int n_regs, const int* regs, double* val, const double* seeds, double* adj, double* grad);
void oh_launch_tape_solve(hipStream_t s, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows, int B, int Bp,
                          const double* x0, const double* p, double* work, double* x, double* f, double* kkt, int* iters, int* status, double* mult);
std::string oh_tape_jit_source(const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows);
int oh_tape_jit_compile(const std::string& src, std::vector<char>* code, std::string* err);  // hiprtc for gfx950; needs no device
int oh_tape_jit_load(const std::vector<char>& code, TapeJit* out, std::string* err);
void oh_tape_jit_forget(const std::string& src);  // drop a cached object that did not load (disk and memory)
void oh_tape_jit_release(TapeJit* j);
hipError_t oh_launch_tape_jit(hipStream_t s, const TapeJit& j, TapeParams T, int B, int Bp, const double* x0, const double* p, double* work, double* x, double* f,
                              double* kkt, int* iters, int* status, double* mult);
