/*
 * optas_hip.h -- C ABI of liboptas_hip.so, the MI355X (gfx950) batched NLP solver backend that drops
 * in behind optas.solver.Solver (reference: optas/solver.py:61-314, IPOPT path :321-419).
 *
 * The reference has no native boundary at all: Solver._solve() (solver.py:386-398) calls CasADi's SWIG
 * object which interprets SX tapes and runs IPOPT/MUMPS on one problem instance.  This header is the
 * boundary a maintainer binds with ctypes: plain pointers and sizes, int return
 * codes, no C++ exceptions, no torch types.  Each entry point names the reference interface it replaces.
 *
 * Conventions
 *   - all floating point data is IEEE double (CasADi DM/SX are double; solver.py:76,79);
 *   - host-facing arrays use the reference layouts: x[b][.] is SXContainer.vec() order
 *     (sx_container.py:83-89), i.e. for the figure-eight family x = [vec(Q ndof x T); vec(dQ ndof x (T-1))],
 *     x[ndof*t + j] = q_j(t); p[b][.] is parameters.vec() order (qc for the figure-eight family);
 *   - the caller owns every buffer it passes; the library only borrows it for the duration of the call;
 *     device scratch is owned by the handle;
 *   - a handle is bound to the HIP device that was current in oh_create and to one HIP stream; it is
 *     not thread-safe; use one handle per GPU (one process per GPU);
 *   - return value 0 = OH_OK; otherwise oh_last_error() describes the failure (thread-local string).
 */
#ifndef OPTAS_HIP_H
#define OPTAS_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes layout or an entry point changes meaning; oh_abi_version() returns the value the library was
   built with, and a host binding refuses a library that answers otherwise (a binding that misreads a descriptor fails silently).
   5: round 5 -- OH_STATUS_INFEASIBLE / OH_STATUS_ACCEPTABLE, oh_set_option / oh_get_option, oh_tq_rollout, tape opcodes 25-26.
   6: round 5 -- oh_tape_set_metric.
   7: round 6 -- OH_MAX_T 256, oh_comm_allgather, options tol / pipe / pipe_chunk, OH_STATUS_INFEASIBLE from the point-mass iteration, oh_solve in chunks on two lanes. */
#define OH_ABI_VERSION 7

#define OH_MAX_CHAIN 16 /* actuated joints on one root->link chain */
#define OH_MAX_T 256    /* horizon knots (128 until round 6; the persistent kernels take horizons of up to 64 / 128 free knots, longer ones run in batched launches) */

enum {
  OH_OK = 0,
  OH_ERR_INVALID = 1,  /* bad argument / descriptor */
  OH_ERR_HIP = 2,      /* HIP runtime error (no device, OOM, launch failure) */
  OH_ERR_STATE = 3     /* call order (e.g. solve before set_constants) */
};

/* per-instance solver status written to status[b] (Solver.did_solve(), solver.py:407-412: status==0) */
enum {
  OH_STATUS_CONVERGED = 0,
  OH_STATUS_MAX_ITER = 1,
  OH_STATUS_NUMERICAL = 2,
  /* the instance has no feasible point: an inequality row that depends on the parameters alone (a limit or sphere-clearance row of a knot that
     the equality rows pin to qc) is negative beyond tol_feas.  IPOPT can only report such a problem (solver.py:407-412 -> did_solve() False,
     :133-134 raises under error_on_fail).  x is the optimum with those rows left out, kkt[1] includes their violation. */
  OH_STATUS_INFEASIBLE = 3,
  /* torque-MPC family: the acceptable level (IPOPT's Solved_To_Acceptable_Level, which the reference also counts as success, solver.py:407-412):
     stall_max steps at the floor of the barrier parameter with the gradient within 10 tol.  did_solve() is True; kkt[0] holds the value. */
  OH_STATUS_ACCEPTABLE = 4
};

/* problem families that have been lowered to kernels */
enum {
  /* no optimisation problem: the handle only serves oh_fk_jac* (any ndof <= OH_MAX_CHAIN);
     T, dt, weights and local_path of the descriptor are ignored */
  OH_PROBLEM_KINEMATICS = 0,
  /* example/figure_eight_plan.py:16-113: joint position/velocity trajectory of one
     serial chain; a = [qc-q0; 0-dq0; Euler integration]; h = quat_c - quat(q_t);
     f = w_path*sum||path_t - p(q_t)||^2 + w_vel*sum||dQ||^2, path_t = p(qc) + R(qc) local_path[t]. */
  OH_PROBLEM_FIGURE_EIGHT = 1,
  /* example/point_mass_mpc.py:88-154 Controller: planar point mass, receding-horizon tick with
     box limits on position/velocity and one moving circular obstacle; created with oh_create_pointmass.
     x = [vec(Y 2xT); vec(dY 2xT)] (nx = 4T), p = [curr(2); dcurr(2); vec(goal 2xT); vec(obs 2xT)] (np = 4+4T). */
  OH_PROBLEM_POINT_MASS_MPC = 2,
  /* example/example.py:13-60 (BASELINE configs[0]): one configuration q of a serial chain,
     f = w ||q - q_nominal||^2, h = p_goal - p_link(q) (builder.py:354), k = [q - lo; up - q] (builder.py:471-509);
     created with oh_create_ik.  x = q (nx = ndof), p = [q_nominal(ndof); p_goal(3)] (np = ndof + 3). */
  OH_PROBLEM_IK = 3,
  /* the QuadraticCost{Unconstrained, LinearConstraints} classes with small dense data (optimization.py:312-388; what the reference hands to
     OSQP / CVXOPT / qpOASES, solver.py:421-584): min x^T P x + q^T x s.t. M x + c >= 0, A x + b = 0; created with oh_create_qp.
     x (nx = n), and one parameter row per instance p = [P (n*n row-major); q (n); M (m*n); c (m); A (me*n); b (me)]. */
  OH_PROBLEM_QP = 4,
  /* any other small dense problem: the host compiles its expression trees into one scalar instruction tape (the counterpart of the CasADi SX
     tape the reference's back-ends interpret, optimization.py:8-24) and the GPU interprets it; created with oh_create_tape.
     x (nx), p (np) in vec() order. */
  OH_PROBLEM_TAPE = 5,
  /* BASELINE configs[4]: joint-space torque MPC whose equality rows are the inverse dynamics RobotModel.rnea
     (models.py:1731-1884).  Written with the reference's builder as
       RobotModel(time_derivs=[0,1,2]) + TaskModel("tau", ndof, dlim={0: [lo, up]}), OptimizationBuilder(T, derivs_align=True)  (builder.py:14-99)
       fix_configuration(q, qc), fix_configuration(dq, dqc), integrate_model_states(.., 1, dt), integrate_model_states(.., 2, dt)  (:419-469,525-539)
       add_equality_constraint(lhs=rnea(Q, dQ, ddQ), rhs=TAU)  ->  h = TAU - rnea                                                  (:337-360)
       enforce_model_limits("tau")                             ->  k = [TAU - lo; up - TAU]                                        (:471-509)
       f = w_path sumsqr(p_link(Q) - goal) + w_vel sumsqr(dQ) + w_tau sumsqr(TAU).
     x = [vec(Q); vec(dQ); vec(ddQ); vec(TAU)] (nx = 4 ndof T), p = [qc (ndof); dqc (ndof); vec(goal 3 x T)] (np = 2 ndof + 3 T);
     created with oh_create_torque; needs oh_set_constants (chain of the tracked link) and oh_set_dynamics before the first solve. */
  OH_PROBLEM_TORQUE_MPC = 6
};

enum {
  OH_HESSIAN_GAUSS_NEWTON = 0, /* 2 w Jp^T Jp */
  OH_HESSIAN_EXACT = 1,        /* + exact curvature of tracking cost and of the orientation constraint */
  OH_HESSIAN_HYBRID = 2        /* Gauss-Newton far from a stationary point, exact once the reduced gradient of the accepted
                                  point is below 1e-5 * w_path (large-residual problems: GN alone converges only linearly) */
};

/*
 * Kinematic constants of one root->link chain with fixed joints folded into the next actuated joint
 * (what RobotModel.get_global_link_transform, models.py:826-868, multiplies joint by joint):
 *   T <- T * [R0_k p0_k] * Rot(axis_k, q[qidx_k])   (revolute/continuous, jtype 0)
 *   T <- T * [R0_k p0_k] * Trans(axis_k * q[qidx_k]) (prismatic, jtype 1)
 *   T_link = T * [R_tool p_tool]
 * quat0_k / quat_tool are the same fixed rotations as xyzw quaternions accumulated with the
 * reference's own product (models.py:1049-1088, spatialmath.py:298-349) so that oh_fk_jac reproduces the
 * reference's quaternion *including its sign*.  This block (sizeof(oh_chain) = 2952 bytes) is what is
 * broadcast once over RCCL/xGMI in multi-GPU runs.
 */
typedef struct oh_chain {
  int ndof;    /* actuated joints of the robot model = rows of q per knot (RobotModel.ndof, models.py:414) */
  int n_chain; /* actuated joints on this chain, in chain order */
  int jtype[OH_MAX_CHAIN];
  int qidx[OH_MAX_CHAIN]; /* actuated-joint index (models.py:661-667) */
  /* structure hints the host derives from exact comparisons (kernels take cheaper, wave-uniform paths):
     axcode: 0 general axis, +-1/+-2/+-3 = axis is exactly +-x/+-y/+-z;  r0ident: 1 if R0 is exactly I */
  int axcode[OH_MAX_CHAIN];
  int r0ident[OH_MAX_CHAIN];
  double R0[OH_MAX_CHAIN][9]; /* row-major */
  double p0[OH_MAX_CHAIN][3];
  double axis[OH_MAX_CHAIN][3]; /* unit (models.py:653-659) */
  double quat0[OH_MAX_CHAIN][4];
  double R_tool[9];
  double p_tool[3];
  double quat_tool[4];
  /* One parameterised joint ahead of the chain (RobotModel(param_joints=[...]), models.py:286-321; example/figure_eight_plan_6dof.py:30-34):
     T <- [lead_R0 lead_p0] * Rot(lead_axis, theta) before the first chain joint, theta = the parameter "{name}/q/p" of the knot.  The
     chain itself then lists the optimised joints only (ndof = their number).  Solver families only; oh_fk_jac ignores it.
     With has_lead the parameter row of OH_PROBLEM_FIGURE_EIGHT is p = [qc of the optimised joints (ndof); theta of qc (1); theta_t (T)]. */
  int has_lead;
  int lead_axcode;
  double lead_R0[9];
  double lead_p0[3];
  double lead_axis[3];
} oh_chain;

/*
 * Inverse-dynamics constants in the selection RobotModel.rnea makes (models.py:1742-1784): body i = child of
 * the i-th joint of get_chain(root, last link) with the first (fixed) joint dropped; masses / centres of mass /
 * inertias of the links that carry <inertial> with the first one dropped; the last body is rigidly attached
 * (models.py:1820-1832).  tau has n-1 entries.
 */
#define OH_MAX_BODIES 10
typedef struct oh_dynamics {
  int n;    /* bodies = len(joints_list_r) */
  int ndof; /* n - 1 */
  double R0[OH_MAX_BODIES][9];  /* rpy2r(joint origin rpy), row-major */
  double xyz[OH_MAX_BODIES][3]; /* joint origin */
  double axis[OH_MAX_BODIES][3];
  double mass[OH_MAX_BODIES];
  double com[OH_MAX_BODIES][3];
  double inertia[OH_MAX_BODIES][9];
  double vd0[3]; /* linear acceleration of the base = -gravity = (0, 0, 9.81) (models.py:1789-1801) */
} oh_dynamics;

typedef struct oh_problem_desc {
  int kind;     /* OH_PROBLEM_* */
  int T;        /* knots (OptimizationBuilder(T=...), builder.py:14-42) */
  int ndof;     /* must equal chain.ndof */
  double dt;    /* Euler step of integrate_model_states (builder.py:419-469) */
  double w_path; /* 1000.0 in figure_eight_plan.py:99 */
  double w_vel;  /* 0.01   in figure_eight_plan.py:103 */
  const double* local_path; /* T x 3 row-major, path in the end-effector frame at qc (:90-96) */
  int lock_orientation;     /* 1: h = quat_c - quat(q_t) rows present (:105-107); 0: position-only tracking (dual_arm.py) */
  int fix_dq0;              /* 1: fix_configuration(time_deriv=1) present, dq_0 = 0 (:65-67); 0: dq_0 free (dual_arm.py:41-42) */
  int path_in_frame;        /* 1: path_t = p(qc) + R(qc) local_path[t] (:94-96); 0: path_t = p(qc) + local_path[t] (dual_arm.py:82-113) */
  /* solver options (the reference passes an options dict to nlpsol, solver.py:333,382) */
  int max_iter;       /* <=0: default 200 */
  double tol;         /* KKT stationarity (inf-norm of the reduced gradient); <=0: default 1e-6 */
  double tol_feas;    /* constraint violation; <=0: default 1e-9 */
  int hessian;        /* OH_HESSIAN_* */
  double mu0;         /* initial Levenberg-Marquardt damping; <0: default */
} oh_problem_desc;

typedef struct oh_pointmass_desc {
  int T;          /* knots (20 in the script) */
  double dt;      /* 0.05 */
  double w_acc;   /* weight of sum ||(dy_{t+1}-dy_t)/dt||^2 : 0.0025 / T (point_mass_mpc.py:133-136) */
  double ylim;    /* position box, 1.5 (:96,110) */
  double vlim;    /* velocity box, 1.0 (:96,111) */
  double safe;    /* obstacle radius + point-mass radius = 0.2 + 0.1 (:91,94,123) */
  int max_iter;   /* <= 0: 100 */
  double tol;     /* KKT tolerance (stationarity, feasibility, complementarity); <= 0: 1e-8 */
  /* planner variant (example/point_mass_planner.py:17-55), all 0 for the MPC tick: */
  int track_final_only;   /* 1: sumsqr(goal - y) on the last knot only (:43) */
  double w_vel;           /* weight of sum ||dy_t||^2 (:45-47), 0.01 / T in the script */
  int fix_final_velocity; /* 1: equality row dy_{T-1} = 0 (:34-35) */
} oh_pointmass_desc;

/*
 * Inequality rows of the trajectory families (OH_PROBLEM_FIGURE_EIGHT, with or without lock_orientation), set with
 * oh_set_guards before the first solve:
 *   limits:  q_t - q_lo >= 0, q_up - q_t >= 0 at every knot   (enforce_model_limits, builder.py:471-509, rows "_l", "_r")
 *   spheres: ||c_l(q_t) - o_j||^2 - (r_l + r_j)^2 >= 0 for every sphere link l and obstacle j
 *            (sphere_collision_avoidance_constraints, builder.py:366-417; c_l = origin of link l in the root frame)
 * Row order per knot: [q - lo (ndof); up - q (ndof); spheres link-major, obstacle-minor; dq - dq_lo (ndof); dq_up - dq (ndof)]; knots t < t0
 * are constants.
 * With guards the parameter row of an instance is
 *   p = [qc (ndof); link radii (n_links); for each obstacle: position (3), radius (1)],  np = ndof + n_links + 4 n_obstacles
 * (the order in which the reference creates these parameters, builder.py:391-405).
 */
#define OH_MAX_SPHERE_LINKS 8
#define OH_MAX_OBSTACLES 16
typedef struct oh_guards {
  int limits;                               /* 1: joint-limit rows present */
  double q_lo[OH_MAX_CHAIN];
  double q_up[OH_MAX_CHAIN];
  int n_links;                              /* sphere links, 0: no sphere rows */
  int link_joint[OH_MAX_SPHERE_LINKS];      /* chain index of the last actuated joint before the link (>= 0) */
  double link_offset[OH_MAX_SPHERE_LINKS][3]; /* link origin in the frame that follows that joint's motion */
  int n_obstacles;
  double rho0;                              /* initial augmented-Lagrangian penalty; <= 0: 10 * w_path */
  /* joint-velocity limits: enforce_model_limits(name, time_deriv=1) (builder.py:471-509), rows dq_t - dq_lo >= 0, dq_up - dq_t >= 0 on
     dq_t = (q_{t+1} - q_t) / dt, t = 0..T-2; both trajectory families (position tracking since round 3; the torque-MPC family has its own
     fields, oh_torque_desc.dq_lo / dq_up, because the velocities are states there).  They couple neighbouring knots exactly
     like the velocity cost: while row j of dq_t is active it adds rho_v / dt^2 to the weight 2 kappa of (q_{t+1,j} - q_{t,j})^2 in the
     block-tridiagonal model.  oh_get_multipliers appends the 2 ndof multipliers of dq_t to knot t's rows (knot T-1: zeros). */
  int vel_limits;
  double dq_lo[OH_MAX_CHAIN];
  double dq_up[OH_MAX_CHAIN];
} oh_guards;

typedef struct oh_ik_desc {
  int ndof;         /* 6 or 7; the chain must cover every model joint in order */
  double w_nominal; /* weight of ||q - q_nominal||^2, 1.0 in example.py:30 */
  double q_lo[OH_MAX_CHAIN]; /* enforce_model_limits (example.py:33; RobotModel limits, models.py:332-368) */
  double q_up[OH_MAX_CHAIN];
  int max_iter;     /* kinematics evaluations per instance; <= 0: 200 */
  double tol;       /* KKT stationarity (projected gradient of the Lagrangian, inf-norm); <= 0: 1e-6 */
  double tol_feas;  /* ||p_goal - p_link(q)||_inf; <= 0: 1e-9 */
  double rho0;      /* initial augmented-Lagrangian penalty; <= 0: 100 w */
} oh_ik_desc;

typedef struct oh_torque_desc {
  int T;          /* knots, 2..OH_MAX_T */
  int ndof;       /* 2 .. 7: every joint of the chain is actuated and the inverse-dynamics tables have ndof + 1 bodies */
  double dt;      /* Euler step of both integrate_model_states calls */
  double w_path;  /* weight of sum ||p_link(q_t) - goal_t||^2 */
  double w_vel;   /* weight of sum ||dq_t||^2 (>= 0) */
  double w_tau;   /* weight of sum ||tau_t||^2 (> 0: it is what makes the stage Hessian in ddq positive definite) */
  double tau_lo[OH_MAX_CHAIN]; /* effort limits: TaskModel dlim[0] (models.py:79-214), rows "_l" / "_r" of enforce_model_limits */
  double tau_up[OH_MAX_CHAIN];
  int max_iter;    /* evaluations after the first; <= 0: 300 */
  double tol;      /* |gradient of the rolled-out Lagrangian w.r.t. ddq|_inf (multipliers lam = mu_b / s of the inequality rows); <= 0: 1e-6.
                      OH_STATUS_CONVERGED also covers the acceptable level (IPOPT's acceptable_tol in spirit): 25 steps at the floor of the barrier
                      parameter with the gradient within 10 tol -- the arithmetic floor when an active row has a slack of ~1e-8; kkt[0] reports the value */
  double tol_compl; /* complementarity lam_i s_i = mu_b of every inequality row at the returned point (IPOPT's tol plays this part,
                       solver.py:355-398); <= 0: 1e-8.  The rows themselves hold strictly: the iterates are interior */
  double mu_barrier0; /* initial barrier parameter; <= 0: 0.1 (IPOPT's mu_init) */
  double mu0;      /* initial Levenberg-Marquardt damping of the state part of the step; < 0: 0 */
  int vel_limits;  /* != 0: joint-velocity limits on the velocity states: enforce_model_limits(name, time_deriv=1) (builder.py:471-509), rows */
  double dq_lo[OH_MAX_CHAIN]; /* dq_t - dq_lo >= 0, dq_up - dq_t >= 0 at every knot (dq_lo < dq_up) */
  double dq_up[OH_MAX_CHAIN];
} oh_torque_desc;

#define OH_QP_MAX_N 32
#define OH_QP_MAX_M 256
#define OH_QP_MAX_ME 32
typedef struct oh_qp_desc {
  int n;        /* decision variables, <= OH_QP_MAX_N */
  int m;        /* rows of M x + c >= 0, <= OH_QP_MAX_M */
  int me;       /* rows of A x + b = 0, <= OH_QP_MAX_ME */
  int max_iter; /* <= 0: 100 */
  double tol;   /* KKT tolerance (stationarity, feasibility, complementarity); <= 0: 1e-9 */
} oh_qp_desc;

#define OH_TAPE_MAX_N 4096
#define OH_TAPE_MAX_LEN (1 << 18)
/* Instruction i writes register i.  op: 0 CONST c | 1 X a | 2 P a | 3 ADD a b | 4 SUB a b | 5 MUL a b | 6 DIV a b | 7 NEG a | 8 SIN a | 9 COS a |
   10 ATAN2 a b | 11 SQRT a | 12 SQR a | 13 ASIN a | 14 FABS a | 15 FMIN a b | 16 FMAX a b | 17 LT a b | 18 LE a b | 19 EQ a b | 20 NE a b |
   21 NOT a | 22 AND a b | 23 OR a b (comparisons / logic: 1.0 or 0.0, zero derivative) | 24 IFZ a b (casadi's if_else_zero: b where a != 0, else 0;
   derivative conventions of 13-16 and 24 are casadi's, casadi/core/calculus.hpp) | 25 EXP a | 26 LOG a (round 5: what user costs written with `from casadi import *`,
   optas/__init__.py:2, add to the core's own set -- the host expresses pow, tanh, sinh, cosh, acos, atan, asinh, acosh, atanh, log1p, expm1 and sign through these).  rows: registers of the constraint rows, the n_ineq rows that must be >= 0 first, then the n_eq rows
   that must vanish (the rows of v = [k; g; a; -a; h; -h] without the mirrored ones, optimization.py:27-51). */
typedef struct oh_tape_desc {
  int nx, np;       /* nx <= OH_TAPE_MAX_N */
  int len;          /* instructions, <= OH_TAPE_MAX_LEN */
  const int* op;
  const int* a;
  const int* b;
  const double* c;
  int out_cost;     /* register holding f */
  int n_ineq, n_eq;
  const int* rows;  /* [n_ineq + n_eq] */
  int max_iter;     /* tape evaluations per instance; <= 0: 2000 */
  double tol;       /* |grad of the Lagrangian|_inf; <= 0: 1e-6 */
  double tol_feas;  /* row violation / complementarity measure; <= 0: 1e-9 */
  double rho0;      /* initial penalty; <= 0: 10 */
  int jit;          /* != 0: generate straight-line HIP code from the tape and compile it with hiprtc when the handle is created (registers in
                       VGPRs); 0: interpret the instruction arrays (registers in HBM/L2; no set-up cost, far slower per evaluation).  Ignored beyond 48
                       variables when the tape's live registers fit the LDS: those handles run one block of wavefronts per instance over the
                       dependency levels of the tape (registers in LDS, no generated code; oh_get_flag "tape_wave") */
  int no_wave;      /* != 0: never that evaluator (option "tape_wave" = 0) */
  int lbfgs;        /* quasi-Newton matrix: 0 by size (dense inverse BFGS up to 48 variables, 12 limited-memory pairs beyond), m > 0: m pairs, < 0: dense
                       (option "tape_lbfgs") */
} oh_tape_desc;

typedef struct oh_handle oh_handle;

/* Replaces Solver.__init__ + CasADiSolver.setup (solver.py:64-88,333-384): allocates the handle,
   creates its stream.  The descriptor (and local_path) is copied. */
int oh_create(const oh_problem_desc* desc, oh_handle** out);

/* Same for OH_PROBLEM_POINT_MASS_MPC (no kinematic constants needed; solve with oh_solve / oh_solve_device). */
int oh_create_pointmass(const oh_pointmass_desc* desc, oh_handle** out);

/* Same for OH_PROBLEM_QP (no kinematic constants). oh_get_multipliers returns [B][m + me] = (lam >= 0 of the M rows, nu of the A rows). */
int oh_create_qp(const oh_qp_desc* desc, oh_handle** out);

/* OH_PROBLEM_QP whose data is a function of the problem's parameters (the reference's QuadraticCost* classes hold P, q, M, c, A, b as
   cs.Functions of p, optimization.py:219-260, and evaluate them before every solve, solver.py:453-467,539-551).  Attaches the problem's
   instruction tape (cost register quadratic in x; n_ineq = m rows affine in x that must be >= 0, then n_eq = me rows that must vanish;
   max_iter / tol / jit of the descriptor are ignored).  From then on the p of oh_solve / oh_solve_device is [B][tape.np] parameter
   vectors: one thread per instance reads [P | q | M | c | A | b] off the tape on the device (values at 0, +-e_i, e_i + e_j: exact for
   these classes) before the solve, and f includes the cost's constant term f(0, p). */
int oh_qp_set_tape(oh_handle* h, const oh_tape_desc* tape);

/* Same for OH_PROBLEM_TAPE: the tape is copied to the device.  oh_get_multipliers returns [B][n_ineq + n_eq] (lam >= 0 of the >= rows, signed mu
   of the = rows in L = f - lam^T g - mu^T c). */
int oh_create_tape(const oh_tape_desc* desc, oh_handle** out);

/* One forward sweep -- and, with seeds, one reverse sweep -- of an OH_PROBLEM_TAPE handle's tape at given points, on the device: what the reference gets
   by calling the cs.Functions of an Optimization and their AD derivatives (optimization.py:8-24).  x [B][nx], p [B][np]; regs [n_regs] registers;
   val [B][n_regs] their values.  seeds NULL, or [B][1 + n_ineq + n_eq] weights of (cost, rows in the tape's order): then adj [B][n_regs] is the
   derivative of that combination with respect to each register and grad [B][nx] its gradient (either may be NULL).  Host buffers.  The host side uses
   it to read back variables it has eliminated from a tape and the multipliers of the rows that went with them (optas_amd/tape.py). */
int oh_tape_probe(oh_handle* h, int B, const double* x, const double* p, int n_regs, const int* regs, double* val, const double* seeds, double* adj,
                  double* grad);

/* Initial metric of an OH_PROBLEM_TAPE handle's limited-memory quasi-Newton iteration: H0 [nx][nx], symmetric positive definite, host memory (copied);
   NULL takes it away again.  The two-loop recursion then starts from r = H0 q instead of the identity scaled by the newest pair.  The reference hands
   IPOPT the exact Hessian of the Lagrangian (optimization.py:8-24, solver.py:355-384); this is the part of it that is known before the first solve:
   the host passes the inverse of the constant block of the cost's Hessian (optas_amd/tape.py:quadratic_cost_metric -- the sum-of-squares terms of a
   trajectory cost), and the (s, y) pairs are left with the curvature of the rows.  Ignored by the dense form (nx <= 48).  oh_get_flag "tape_metric". */
int oh_tape_set_metric(oh_handle* h, const double* H0);

/* What oh_create_tape does for desc->jit != 0 before it touches a device (CasADi's "jit" option, solver.py:333-384 passes it through): generate
   the kernel source of this tape and compile it for gfx950.  Needs no GPU.  source (optional, source_cap bytes) receives the generated
   text, *source_len its full length, *code_bytes the size of the code object. */
int oh_tape_compile(const oh_tape_desc* desc, size_t* code_bytes, char* source, size_t source_cap, size_t* source_len);

/* Same for OH_PROBLEM_TORQUE_MPC.  oh_get_multipliers returns [B][T][2 ndof] = multipliers >= 0 of (TAU - lo, up - TAU) per knot; the
   multipliers of the other rows follow from these and the solution (h: nu_t = 2 w_tau tau_t - lam_lo + lam_up, a: the costates). */
int oh_create_torque(const oh_torque_desc* desc, oh_handle** out);

/* Same for OH_PROBLEM_IK; needs oh_set_constants before the first solve. */
int oh_create_ik(const oh_ik_desc* desc, oh_handle** out);

/* Kinematic constants from host memory / from device memory (the latter after an RCCL broadcast). */
int oh_set_constants(oh_handle* h, const oh_chain* chain);
int oh_set_constants_device(oh_handle* h, const void* d_chain, size_t nbytes);

/*
 * Multi-GPU (one process per GPU).  MPC instances are independent: the host shards them over the ranks and nothing is exchanged on
 * the data path.  The single collective of a job is the broadcast of the URDF-derived constants (the reference has no
 * counterpart -- it is one process with one problem, solver.py:64-88).  The library owns the RCCL communicator:
 *   rank 0:  oh_comm_unique_id(id)  -> ship the 128 bytes to the other ranks (file, socket, environment: the host's business)
 *   all:     oh_set_device(local_rank); oh_comm_init(rank, world, id)
 *   root:    oh_set_constants(h, chain);  all: oh_comm_broadcast_constants(h, root)   (ncclBroadcast over xGMI, in place in the handle)
 * oh_comm_barrier / oh_comm_allreduce_{max,sum} (one host double, in place) are what a timing harness needs around the solves.
 * librccl is opened with dlopen on first use; processes that never create a communicator do not load it.
 */
#define OH_COMM_ID_BYTES 128
int oh_comm_unique_id(char* id /* [OH_COMM_ID_BYTES] */);
int oh_comm_init(int rank, int world, const char* id /* [OH_COMM_ID_BYTES] */);
int oh_comm_broadcast_constants(oh_handle* h, int root);
int oh_comm_barrier(void);
int oh_comm_allreduce_max(double* value);
int oh_comm_allreduce_sum(double* value);
/* Optional gather of results (SURVEY 8(e) "optional ncclAllGather / host gather of x*"): `bytes` bytes of every rank's device buffer d_send land in d_recv
   (world x bytes, rank order) on every rank.  After the solves, never between iterations; the caller synchronises the solves first. */
int oh_comm_allgather(const void* d_send, void* d_recv, size_t bytes);
int oh_comm_destroy(void);
/* Rank and world size as RCCL itself reports them for this process's communicator (ncclCommUserRank / ncclCommCount): what a harness prints to
   show that the communicator spans the job. */
int oh_comm_info(int* rank, int* world);
/* The constants a handle holds (after oh_set_constants* or oh_comm_broadcast_constants). */
int oh_get_constants(oh_handle* h, oh_chain* out);

/* Inequality rows for the trajectory families (see oh_guards). */
int oh_set_guards(oh_handle* h, const oh_guards* guards);

/* Replaces B sequential calls of Solver.reset_initial_seed + reset_parameters + _solve
   (solver.py:103-116,386-398).  Host buffers:
     x0 [B][nx], p [B][np]  in;  x [B][nx], f [B], kkt [B][3] = (stationarity, feasibility,
     complementarity), iters [B], status [B] out (any output pointer may be NULL).
   nx = ndof*T + ndof*(T-1), np = ndof for OH_PROBLEM_FIGURE_EIGHT; see the OH_PROBLEM_* comments for the others.
   One call of the orientation-locked family takes at most about 2^32 / (8 T (ndof-3)^2) instances (a stage array is addressed with 32-bit
   offsets; the row pad of large batches counts): oh_max_batch says exactly how many; larger batches return OH_ERR_INVALID and are to be
   split by the caller. */
int oh_solve(oh_handle* h, int B, const double* x0, const double* p, double* x, double* f, double* kkt,
             int* iters, int* status);

/* Scheduling facts of a handle by name, for harnesses that report what ran: "fuse_couple" (1: the orientation-locked family's iteration is
   k_retract + k_evalb_zc + k_step_zc, the neighbour coupling folded in; 0: k_couple runs as a launch of its own), "tail_threshold",
   "specialized"; OH_PROBLEM_TAPE handles: "tape_wave" (0: one thread per instance; 1 / 2: one block of wavefronts per instance, the quasi-Newton
   pairs in global memory / in LDS), "tape_regs_lds" (1: the tape's registers of the last launch in LDS, 0: in global memory -- batches beyond 512 instances and tapes that do
   not fit), "tape_levels" and "tape_passes" (dependency levels of the tape; instruction passes of one evaluation). */
int oh_get_flag(oh_handle* h, const char* name, int* value);

/*
 * Options of ONE handle by name (round 5).  The reference passes an options dict through to its back-end (solver.py:333-384: nlpsol's `opts`);
 * these are the library's counterpart for scheduling and experiment knobs -- until round 4 they were OH_* environment variables read inside the
 * library, so two handles of one process could not differ.  Unknown names return OH_ERR_INVALID.  Scheduling options change WHEN work is done and by
 * which kernel, never what is computed, except where noted:
 *   batch_invariant (0)   1: no restarts, no persistent tail kernel: every instance runs the batched launches to the end, and its answer is a
 *                         function of the instance alone, bit for bit, whatever batch it is part of (default path: the same optimum for all but a
 *                         handful of a 262 144 batch, bit-identical only where the same kernels ran).  The batch is still compacted,
 *                         but only by moving a survivor with everything it owns once invariant_compact_frac (0.65) of the batch is left
 *                         (0: never; invariant_move_slim / invariant_move_live (1): arrays no kernel carries across launches, and the slot of the
 *                         judged trial, stay behind), and a large batch is still solved in parts on two streams (invariant_split, 1).
 *                         Costs 1.3 x device time at 262 144 instances (3.0 x without the moving compaction).
 *   tail_threshold (16384), tail_vel (1), tail_vel_threshold, compaction (1), compact_frac (0.97), compact_sort (1), compact_carry (1),
 *   sparse_check_below (2048), check_every (1), fuse_couple (1), lg_split (1), row_pad (13)            -- figure-eight family scheduling
 *   streams (2), split_min (65536): a batch of the plain orientation-locked family of at least split_min instances is solved in `streams` contiguous parts,
 *       each on a HIP stream and a host thread of its own (results at every index = the part solved as a batch of its own); 1: one stream
 *       (the torque-MPC family likewise from tq_split_min (1024) instances on: its answers do not depend on the batch, so the split is invisible)
 *       (the position-tracking family from free_split_min (256) instances on)
 *   free_pcr_max (1536), free_bb (1), free_cp_max (512), free_persist (-1 auto / 0 / 1)                  -- position-tracking family sweeps
 *   specialize (2 = auto, 0 never, 1 at the first call)                                                 -- run-time specialisation (oh_specialize)
 *   hyb_switch (1e-5, x w_path), relax (1.5), relax_from (4), retract_min (1e-13), settle_k (1)          -- algorithm constants (change the iterates)
 *   tol (0 = the descriptor's): stopping tolerance on the reduced gradient of a trajectory handle, changeable between solves
 *   pm_wave_max (20480), qp_mode (-1), tape_lds_max                                                     -- point-mass / QP / tape launch shapes
 *   tape_wave (1), tape_lbfgs (-1 = by size), tape_wave_nt (256), tape_wave_regs (-1), tape_wave_hist (-1) -- tape evaluator (rebuilt when set)
 *   tq_check (4), tq_rebuild (0.9), tq_stall (25), tq_curv_after (3), tq_curv_from (0.1), tq_ftb (0.995), tq_theta_mu (1.35), tq_kappa_mu (0.4),
 *   tq_kappa_eps (10), tq_curv_late (1), tq_max_back (3), tq_mu_dec (1/3; warm ticks of oh_tq_rollout: tq_mu_dec_warm, 0.1), tq_ls_curv (1), tq_curv_lag (3: the exact-curvature term is computed at every 4th evaluation of an instance and reused in between; 0: always),
 *   tq_jac_dual (0)                                                                                     -- torque-MPC family
 * The one environment hook left: OH_DEBUG_OPTIONS="name=value,name=value" is applied to every handle when it is created (A/B tooling).
 */
int oh_set_option(oh_handle* h, const char* name, double value);
int oh_get_option(oh_handle* h, const char* name, double* value);

/* Largest B one oh_solve / oh_solve_device call of this handle takes (*out = 0: the library sets no bound of its own). */
int oh_max_batch(oh_handle* h, int* out);

/* Same with buffers already resident in HBM (what bench.py times).  Synchronous on return. */
int oh_solve_device(oh_handle* h, int B, const void* d_x0, const void* d_p, void* d_x, void* d_f, void* d_kkt,
                    void* d_iters, void* d_status);

/* Closed-loop receding horizon for OH_PROBLEM_POINT_MASS_MPC, resident on the device: replaces the main loop
   of example/point_mass_mpc.py (:293-306) around Controller.next_state (:156-161) for B plants at once.  Per tick k:
     p_k = [curr; dcurr; goal; obs], goal[:, i] = curr + ramp * i, obs[:, i] = obs_table[k * advance + i];
     seed = previous solution (:157-158); solve; the plant takes the plan's state at knot `advance` (plan(advance * dt), :160-161).
   Host buffers: state0 [B][4] = (y, dy); obs_table [n_ticks * advance + T][2]; out: states [n_ticks + 1][B][4] (states[0] = state0),
   f, iters, status [n_ticks][B] (any may be NULL).  Nothing crosses PCIe between the first and the last tick. */
int oh_pm_rollout(oh_handle* h, int B, int n_ticks, int advance, double ramp, const double* state0, const double* obs_table, double* states,
                  double* f, int* iters, int* status);

/* The same loop for OH_PROBLEM_TORQUE_MPC (round 5; BASELINE configs[4] is an MPC: "solves per second" in steady state are warm-started ticks).
   Per tick k:  p_k = [q_k; dq_k; rows k * advance .. k * advance + T - 1 of the plant's goal table];
     seed = the accelerations of the previous plan shifted by `advance` knots, the last one repeated (the reference's pattern, point_mass_mpc.py:157-158:
     seed from the previous solution; tick 0: zero accelerations = the cold solve), barrier parameter of a warm tick mu_warm (<= 0: 1e-6; the cold
     default 0.1 = IPOPT's mu_init would first walk the iterate back to the centre of the feasible set);
     solve; the plant follows the plan for `advance` knots: (q, dq) <- the plan's state at knot `advance` -- the Euler roll-out of its accelerations,
     with the torques of the inverse-dynamics rows, tau_t = rnea(q_t, dq_t, ddq_t).
   Host buffers: state0 [B][2 ndof] = (q, dq); goal_table [B][n_ticks * advance + T][3]; out (any may be NULL): states [n_ticks + 1][B][2 ndof]
   (states[0] = state0), tau0 [n_ticks][B][ndof] (the torque applied at each tick: knot 0 of its plan), f, iters, status [n_ticks][B].
   Nothing crosses PCIe between the first and the last tick; oh_get_timing: [4] device ms of the solves, [5] launches, [6] instance-launches. */
int oh_tq_rollout(oh_handle* h, int B, int n_ticks, int advance, double mu_warm, const double* state0, const double* goal_table, double* states,
                  double* tau0, double* f, int* iters, int* status);

/* Multipliers of the last oh_solve/oh_solve_device in the reference's form: lam_h [B][4*T] for the rows
   h = quat_c - quat(q_t) (signed mu = lam+ - lam- of the (h,-h) pair, optimization.py:47-51). Host buffer.
   OH_PROBLEM_IK: lam_h [B][3 + 2*ndof] = (mu of h = p_goal - p_link(q) (3), multipliers of q - lo >= 0 (ndof),
   multipliers of up - q >= 0 (ndof)).
   Handles with oh_set_guards: lam_h [B][T][NC], NC = 2 ndof limits + n_links n_obstacles + 2 ndof velocity limits, row order
   of oh_guards (multipliers >= 0 of the g >= 0 rows; knots t < t0 carry zeros). */
int oh_get_multipliers(oh_handle* h, int B, double* lam_h);

/* Replaces RobotModel.get_global_link_{position,quaternion,geometric_jacobian}_function(link, n=N)
   (models.py:935-947,1090-1106,1266-1281): q [N][ndof] -> pose [N][7] = (p xyz, quat xyzw),
   J [N][6][ndof] row-major (rows 0-2 linear, 3-5 angular; models.py:1239,1246).  pose or J may be NULL. */
int oh_fk_jac(oh_handle* h, int N, const double* q, double* pose, double* J);
int oh_fk_jac_device(oh_handle* h, int N, const void* d_q, void* d_pose, void* d_J);

/* Replaces RobotModel.rnea(q, qd, qdd) (models.py:1731-1884), batched: q, qd, qdd [N][ndof] -> tau [N][ndof]
   (Craig's recursive Newton-Euler exactly as the reference writes it).  Needs oh_set_dynamics. */
int oh_set_dynamics(oh_handle* h, const oh_dynamics* dyn);
int oh_rnea(oh_handle* h, int N, const double* q, const double* qd, const double* qdd, double* tau);
int oh_rnea_device(oh_handle* h, int N, const void* d_q, const void* d_qd, const void* d_qdd, void* d_tau);
/* Its Jacobian, what the reference gets from casadi.jacobian of the same graph (optimization.py:8-24; the dh of a dynamics row
   h = TAU - rnea(Q, dQ, ddQ)): J [N][ndof][3 ndof] row-major = d tau / d (q, qd, qdd), exact (the recursion run on dual numbers). */
int oh_rnea_jac(oh_handle* h, int N, const double* q, const double* qd, const double* qdd, double* J);
/* Its second derivatives contracted with a multiplier vector, what the reference gets as ddh / the Lagrangian Hessian of the dynamics rows by AD of
   the same graph (optimization.py:8-24, 262-290): H [N][3 ndof][3 ndof] row-major = sum_i c_i d^2 tau_i / d (q, qd, qdd)^2 for c [N][ndof], exact
   (hand-written adjoint of the recursion, run on dual numbers; the ddq-ddq block is zero: the torques are linear in the accelerations). */
int oh_rnea_hess(oh_handle* h, int N, const double* q, const double* qd, const double* qdd, const double* c, double* H);

/* Structure-of-arrays variant used inside the solver and for roofline measurement:
   q [ndof][N], pose [7][N], J [6*ndof][N] (unit index fastest => fully coalesced). */
int oh_fk_jac_soa_device(oh_handle* h, int N, const void* d_q, void* d_pose, void* d_J);

/* Timing of the last oh_solve*: HIP-event milliseconds accumulated per kernel on the handle's stream.
   out[0]=eval kernel total ms, out[1]=eval launches, out[2]=step kernel total ms, out[3]=step launches,
   out[4]=whole solve ms, out[5]=SQP iterations launched (pairs), out[6]=sum over launches of the number
   of instances still running (a launch touches only those: work actually done), out[7]=batch compactions,
   out[8]=couple kernel total ms, out[9]=rejected steps (summed over instances), out[10]=iterations run
   inside the persistent tail kernel (summed over instances).
   out[0..3] need oh_set_profiling(h,1) (one hipEventRecord after every kernel). */
int oh_set_profiling(oh_handle* h, int enable);
int oh_get_timing(oh_handle* h, double* out11);

/* Run-time specialisation.  The reference's own speed comes from code generated for one problem: CasADi turns the robot model into a
   straight-line SX program with the URDF constants folded in (models.py:826-868 builds the chain walk symbolically, solver.py:333-398
   hands the resulting functions to the back-end).  The equivalent here: the kernels that walk the chain in their inner loops are
   compiled once more with hiprtc behind a constexpr copy of this handle's oh_chain -- K1 (oh_fk_jac*) for every handle with constants,
   and the evaluation kernels k_retract / k_evalb / k_tail of OH_PROBLEM_FIGURE_EIGHT with lock_orientation (no lead joint, no guards).
   oh_specialize compiles (or fetches from the process / disk cache: $OPTAS_HIP_CACHE, default ~/.cache/optas_hip, empty = none) and
   loads them now; without the call the library does it by itself at the first solve of >= 4096 instances / the first oh_fk_jac* of
   >= 65536 units (option specialize = 0: never, 1: at the first call of any size).  Until then, and if compilation is unavailable (then
   oh_specialize returns OH_ERR_HIP and oh_last_error says why), the generic kernels run: same text, same results up to the rounding
   of folded constants.  info4: [0] 1 if the specialised solver kernels are loaded, [1] 1 if the specialised K1 is, [2] seconds the last
   oh_specialize of this handle took, [3] 1 if the code object came from the disk cache. */
enum { OH_SPECIALIZE_NEVER = 0, OH_SPECIALIZE_ALWAYS = 1, OH_SPECIALIZE_AUTO = 2 };
int oh_specialize(oh_handle* h);
/* Compilation only (hiprtc needs no device): fills the disk cache for a chain ahead of time, e.g. in a build step.  info2: [0] seconds,
   [1] 1 if the code object was already in the cache. */
int oh_specialize_compile(const oh_chain* chain, double* info2);
int oh_specialize_info(oh_handle* h, double* info4);
/* Code-object facts like oh_kernel_info, of the kernels this handle would launch (the specialised ones once loaded). */
int oh_kernel_info_handle(oh_handle* h, const char* kernel, int* out5);

/* Thin device-memory helpers so a ctypes host needs no other GPU runtime binding. */
int oh_device_count(int* n);
int oh_set_device(int index); /* hipSetDevice: call before oh_create in one-process-per-GPU launches */
int oh_device_malloc(void** ptr, size_t nbytes);
int oh_device_free(void* ptr);
int oh_memcpy_h2d(void* dst, const void* src, size_t nbytes);
int oh_memcpy_d2h(void* dst, const void* src, size_t nbytes);
int oh_device_synchronize(void);
/* HIP-event timing of an arbitrary region on the handle's stream (for bench.py's roofline object). */
int oh_event_timer_start(oh_handle* h);
int oh_event_timer_stop(oh_handle* h, double* ms);

/* Code-object facts of a kernel of this library, read from the loaded module (hipFuncGetAttributes, occupancy query): name in
   {k_retract, k_evalb, k_couple, k_step, k_tail, k_fk_jac, k_tq_eval, k_tq_step} (the ndof-7 instantiations);
   out5 = {registers per lane (VGPR + AGPR), scratch bytes per lane, LDS bytes per block, block size, resident blocks per CU}.
   Waves per SIMD = blocks per CU x block size / 64 / 4.  What bench.py reports as roofline.occupancy. */
int oh_kernel_info(const char* kernel, int* out5);

const char* oh_last_error(void);
const char* oh_version(void);
int oh_abi_version(void); /* OH_ABI_VERSION of the build */
void oh_destroy(oh_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* OPTAS_HIP_H */
