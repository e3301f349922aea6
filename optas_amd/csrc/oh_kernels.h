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
bool oh_launch_setup_guards(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, const double* p);
bool oh_launch_eval_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
bool oh_launch_step_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot,
                            bool pcr = false);
bool oh_launch_eval_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot, int part);
bool oh_launch_step_locked_guarded(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
void oh_launch_guard_emit(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int NV, int only_done);
void oh_launch_guard_compact(hipStream_t s, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int NV, int phase, int Bnew);
bool oh_launch_tail(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int slot);
bool oh_launch_tail_vel(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, const GuardParams& GP, const GuardBuffers& GB, int slot);
bool oh_launch_finalize(hipStream_t s, int n, const FigParams& P, const FigBuffers& D, int only_done, double* x, double* f, double* kkt,
                        int* iters, int* status);
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
// tangent direction) and the Riccati kernel with 16 lanes per instance, so a wavefront reads whole stage records.
#define TQ_XS 24    // per knot: q (N) at 0, dq (N) at 8, ddq (N) at 16
#define TQ_SD 272   // per knot stage record: H packed lower (3N)(3N+1)/2 at 0 | g (3N) at 231 | phi 252, phi_true 253, meas 254, viol 255 | tau (N) at 256 | compl 263
#define TQ_LAM 32   // per knot: multipliers of tau - lo >= 0 (N), then of up - tau >= 0 (N); at 16: of dq - dq_lo >= 0 (N), then of dq_up - dq >= 0 (N)
#define TQ_GN 112   // per knot: gains K (column c of 2N: N values at c N), feed-forward k at 2N N
struct TqParams {
  int T, N, max_iter;
  double dt, w_path, w_vel, w_tau, tol, tol_feas, rho0, mu0;
  double tau_lo[OH_MAX_CHAIN], tau_up[OH_MAX_CHAIN];
  double dq_lo[OH_MAX_CHAIN], dq_up[OH_MAX_CHAIN];  // joint-velocity rows on the velocity states (vel != 0)
  int vel;
  int nx, np;
  int aa_m;         // Anderson acceleration of the Gauss-Newton iteration: history depth (0 off, <= 3), see k_tq_step
  double aa_from;   // ... once the reduced gradient is below this
};
#define TQ_HS 16    // per knot and history entry: control u (N) at 0, Gauss-Newton step du (N) at 8
struct TqBuffers {
  int B;
  const oh_chain* chain;
  const oh_dynamics* dyn;
  double* xs;      // [2][B][T][TQ_XS]
  double* st;      // [2][B][T][TQ_SD]
  double* lam;     // [B][T][TQ_LAM]
  double* gains;   // [B][T][TQ_GN]
  double* goal;    // [B][T][4]
  double *f_cur, *f_true, *pred, *mu, *nun, *rho, *rho_next, *omega, *meas_prev, *meas, *stat;  // [B]
  int *cur, *first, *outer, *status, *iters, *rejected, *n_outer;                                // [B]
  int* n_running;  // [1]
  double* hist;    // [B][4][T][TQ_HS] ring of the last accepted control sequences and the steps taken from them (Anderson history)
  int* hcnt;       // [B] entries appended since the history was last dropped
  int* aa;         // [B] 1: the pending trial is the extrapolated point
  int* list;       // [B] instances still running when the list was last rebuilt (kernels walk this list: finished instances cost nothing)
  int* n_list;     // [1]
  int n_run;       // length of the list the launches below cover
};
void oh_launch_tq_list(hipStream_t s, const TqBuffers& D);
bool oh_launch_tq_setup(hipStream_t s, const TqParams& P, const TqBuffers& D, const double* x0, const double* p);
bool oh_launch_tq_eval(hipStream_t s, const TqParams& P, const TqBuffers& D);
bool oh_launch_tq_step(hipStream_t s, const TqParams& P, const TqBuffers& D);
bool oh_launch_tq_finalize(hipStream_t s, const TqParams& P, const TqBuffers& D, double* x, double* f, double* kkt, int* iters, int* status, double* mult);

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
size_t oh_tape_work_rows(const TapeParams& T, bool jit);
void oh_launch_tape_solve(hipStream_t s, const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows, int B, int Bp,
                          const double* x0, const double* p, double* work, double* x, double* f, double* kkt, int* iters, int* status, double* mult);
std::string oh_tape_jit_source(const TapeParams& T, const int* op, const int* a, const int* b, const double* c, const int* rows);
int oh_tape_jit_compile(const std::string& src, std::vector<char>* code, std::string* err);  // hiprtc for gfx950; needs no device
int oh_tape_jit_load(const std::vector<char>& code, TapeJit* out, std::string* err);
void oh_tape_jit_release(TapeJit* j);
hipError_t oh_launch_tape_jit(hipStream_t s, const TapeJit& j, TapeParams T, int B, int Bp, const double* x0, const double* p, double* work, double* x, double* f,
                              double* kkt, int* iters, int* status, double* mult);
