// Plain structs the kernels of the figure-eight / position-tracking families take by value (scalars -> SGPRs, SoA device pointers).
// Device-visible and free of host headers: the kernels that are specialised at run time (oh_jit.hip) compile this file with hiprtc.
#pragma once
#include "optas_hip.h"

// Scalar parameters of the figure-eight family (passed by value to every kernel -> SGPRs).
struct FigParams {
  int T;
  int t0;            // first free knot: 2 when q_0 and dq_0 are fixed (q_1 = q_0), 1 when only q_0 is fixed
  int lock;          // 1: orientation rows R(q_t) = R(qc) present (null-space dimension N-3), 0: position-only tracking
  int path_in_frame; // 1: path_t = p(qc) + R(qc) local_t, 0: path_t = p(qc) + local_t
  int nx;            // ndof*T + ndof*(T-1)
  double dt;
  double w_path;
  double kappa;      // w_vel / dt^2 : weight of ||q_{t+1}-q_t||^2 after eliminating dq
  double tol;
  double tol_feas;
  double tol_retract;
  double tol_retract_min;  // end-game retraction tolerance follows 1e-2 pred down to this (= tol_retract: off; 1e-13 on handles with inequality rows)
  double feas_accept;
  int max_retract;
  int max_iter;
  int hessian;
  double hyb_switch; // OH_HESSIAN_HYBRID: exact curvature once stat <= hyb_switch
  double mu0;
  double relax;      // over-relaxation of Gauss-Newton steps in the crawl phase of OH_HESSIAN_HYBRID (1: off), see step_instance
  int relax_from;    // ... from this step count on
  int al_fuse;       // position-tracking family with inequality rows: the launch that decides on a multiplier update also takes the pending step (1; 0: stays put)
  double settle_k;   // the retraction skips the kinematics pass that would only confirm a Newton step dq when settle_k ||dq||_1^2 <= tolerance (1: the rigorous bound)
  const double* local_path;  // device, [T][3]
  int np;            // row stride of the parameter matrix p (ndof, or ndof + guard parameters)
  int zc;            // 1: coupling folded into evaluation and sweep (eval_unit<.., ZC>, step_instance_zc): Gfull[] holds G of every evaluated point
  int zc_free;       // 1: position-tracking family without velocity rows: coupling folded into the evaluation (couple_inline_free), no k_couple_free launch
  int inst_major;    // 1: position-tracking family, stage blocks Dr / gt instance-major [b][t][.] (every launch gives an instance its own block; oh_free.hip DRX / GTX)
};

// Inequality rows of the position-tracking family (oh_guards): constants and per-instance state.
struct GuardParams {
  int limits, n_links, n_obs, NC;  // NC = 2 N limits + n_links n_obs rows per knot
  int link_joint[OH_MAX_SPHERE_LINKS];
  double link_off[OH_MAX_SPHERE_LINKS][3];
  double lo[OH_MAX_CHAIN], up[OH_MAX_CHAIN];
  double rho0;
  int vel;                                  // joint-velocity rows present (orientation-locked family)
  double vlo[OH_MAX_CHAIN], vup[OH_MAX_CHAIN];
  double vscale;                            // their penalty is rho * vscale (dt^2 / 40: Gauss-Newton weight w_path / 4 at rho0 = 10 w_path)
};
struct GuardBuffers {
  double* lam;        // [T][NC][Bp]  multipliers of the last outer update
  double* par;        // [n_links + 4 n_obs][Bp]  link radii, then x, y, z, r of each obstacle
  double* psi[2];     // [slot][T][Bp] augmented-Lagrangian part of phi
  double* rho;        // [Bp] penalty the stored stage data was evaluated with
  double* rho_next;   // [Bp] penalty after the pending outer update
  double* omega;      // [Bp] inner tolerance on the reduced gradient
  double* meas_prev;  // [Bp] |min(g, lam/rho)|_inf at the previous outer update
  int* outer;         // [Bp] 1: the next evaluation first refreshes the multipliers (outer iteration)
  int* n_outer;       // [Bp]
  double* mcv[2];     // [slot][T][Bp] per-knot |min(g, lam/rho)|_inf (orientation-locked family; D.cv holds the orientation residual there)
  double* meas;       // [Bp] its maximum over the knots of the accepted point
  double* lamv;       // [T][2N][Bp] multipliers of the velocity rows; row block t = interval (t-1, t) = dq_{t-1}: [dq - vlo (N); vup - dq (N)]
  double* lam_out;    // [T][NC][Bp] multipliers of finished instances at their ORIGINAL index (instances move when the batch is compacted)
  double* lamv_out;   // [T][2N][Bp] same for the velocity rows
  double* scr;        // [T (NC + 2N) + n_par + 8][Bp] scratch of the compaction (k_guard_gather / k_guard_scatter)
  // line search along a rejected step (OH_LS_MAX): shortenings so far, and g^T z and g^T z + mu z^T z of the step as it was solved for
  // (predicted decrease of s z: -s gd + s^2 q / 2); reset whenever a point is accepted, so they need not move with a compacted instance
  int* ls_count;      // [Bp]
  double* ls_gd;      // [Bp]
  double* ls_q;       // [Bp]
};
// Handles with inequality rows: a rejected trial is followed by up to OH_LS_MAX shorter steps along the same direction (factor OH_LS_SHRINK each)
// before the Levenberg-Marquardt damping is raised.  What rejects a step of these problems is a row that is inactive at the accepted point
// and violated at the trial: the model cannot know it, and damping the whole step to ~1e3 and easing it back costs a dozen steps
// (oracle/guarded.py, oracle/structured.py; numpy port, config 4 synthetic: 30.0 -> 22.0 steps per arm, slowest arm 70 -> 57 launches).
#define OH_LS_MAX 3
#define OH_LS_SHRINK 0.3

// Device buffers of one handle (SoA, instance index fastest; Bp = B rounded up to 64).
struct FigBuffers {
  int B, Bp;
  const oh_chain* chain;  // device copy of the kinematic constants
  double* q[2];           // [slot][T][N][Bp]      knots: current / trial
  double* q_spare[2];     // [T][N][Bp] x 2     where a compaction lays the survivors' knots down; swapped with q[] afterwards (no copy back)
  double* G_spare;        // [T][N][Bp]         same for the Lagrangian gradient of the accepted point
  double* Z[2];           // [slot][T][3N-3][Bp]   Householder vectors of the null-space basis of the orientation rows (Z is rebuilt from them)
  double* Dr[2];          // [slot][T][NZ(NZ+1)/2][Bp] reduced Hessian block Z^T W Z (packed lower)
  double* g[2];           // [slot][T][N][Bp]      tracking gradient
  double* phi[2];         // [slot][T][Bp]         tracking cost
  double* cv[2];          // [slot][T][Bp]         |c|_inf after retraction
  double* Gfull[2];       // [slot][T][N][Bp]      Lagrangian gradient G_t (exact-Hessian mode: multiplier estimate)
  double* mdl[2];         // [slot][T][3+3NZ][Bp]  end-effector position e_t and Jp_t Z_t: the linear model the next retraction targets
  double* E[2];           // [slot][T][NZ*NZ][Bp]  coupling blocks -2 kappa Z_t^T Z_{t+1}
  double* gt[2];          // [slot][T][NZ][Bp]     reduced gradient Z_t^T G_t
  double* merit[2];       // [slot][T][Bp]         phi_t + kappa ||q_t - q_{t-1}||^2
  double* zstep;          // [T][NZ][Bp]           reduced step of the pending trial
  double* Kmat;           // [T][NZ*NZ][Bp]        Riccati gains
  double* kvec;           // [T][NZ][Bp]
  double* ref;            // [12][Bp]              p(qc), R(qc)
  double* fconst;         // [Bp]
  double* f_cur;          // [Bp]
  double* pred;           // [Bp]
  double* mu;             // [Bp]  Levenberg-Marquardt damping
  double* nun;            // [Bp]  Nielsen growth factor
  double* stat;           // [Bp]
  double* feas;           // [Bp]
  double* fpsi;           // [Bp] augmented-Lagrangian part of f_cur (nullptr without inequality rows)
  double* lead;           // [T][Bp] angle of the parameterised lead joint at every knot (nullptr: chain has none)
  int* cur;               // [Bp] slot holding the accepted point
  int* first;             // [Bp]
  int* scan_blk;          // [8][1024] per-block class counts / offsets of the compaction scan
  int* skip;              // [Bp] 1: last trial rejected -> sit the next launch out (keeps the slot parity uniform)
  int* polish;            // [Bp] 1: the next evaluation re-retracts the accepted point itself (zero step, floor tolerance) and is accepted as is;
                          //      2: (k_step_zc, deferred refactorisation) it lays the accepted point down unchanged, accepted as is, swept at the raised damping
  int* stale;             // [Bp] 1: the accepted point's stage data was left behind by a compaction (k_carry_*); a rejected trial restarts
  int* status;            // [Bp] -1 running, else OH_STATUS_*
  int* iters;             // [Bp]
  int* orig;              // [Bp] original instance index (instances are compacted as the batch drains)
  int* newidx;            // [Bp] scratch of the compaction scan
  int* n_running;         // [1] instances still running after the last k_step
  int* n_new;             // [1] result of the compaction scan
  double* lam_h;          // [B][T][4] multipliers of the quaternion rows, reference form (original order)
  unsigned long long* work;  // [1] sum over k_step launches of running instances
  // deferred refactorisation (k_step_zc, round 6): instances whose older Lagrangian gradient k_defer_copy puts back after the sweep, by launch parity
  int* n_defer;           // [2]
  int* defer_list;        // [2][Bp]
};

