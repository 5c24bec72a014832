/*
 * cddp_hip.h -- C-ABI of the MI355X-native batched CLDDP / IPDDP solver core.
 *
 * This is the drop-in boundary for the hot path of astomodynamics/cddp-cpp:
 * everything cddp::ISolverAlgorithm::{initialize,solve} does for the "CLDDP"
 * and "IPDDP" solvers (reference include/cddp-cpp/cddp_core/cddp_core.hpp:186-210,
 * src/cddp_core/cddp_solver_base.cpp:29-186, clddp_solver.cpp, ipddp_solver.cpp,
 * boxqp.cpp), executed for a whole BATCH of independent trajectories on one GPU.
 *
 * Conventions (nothing like this exists in the reference):
 *   - extern "C", plain pointers and sizes, no exceptions / STL / Eigen / torch.
 *   - every entry point returns 0 on success, <0 on error;
 *     cddp_hip_last_error() returns a thread-local message.
 *   - all matrices are ROW-MAJOR doubles; trajectories handed over the boundary
 *     are batch-major: X[b][t][i], U[b][t][j], K[b][t][j][i].
 *   - host buffers are caller-owned, device buffers are library-owned (unless a
 *     *_device entry point says the pointer is a device pointer).
 *   - a handle is bound to (device, stream) and is NOT thread-safe; distinct
 *     handles are independent.
 *
 * The reference-side binding a maintainer adds is shown in INTEGRATION.md.
 */
#ifndef CDDP_HIP_H
#define CDDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: cddp_hip_options gained max_cpu_time (shifts cddp_hip_problem's tail), cddp_hip_stats gained rollout_steps (80 -> 88 bytes).
 * A host built against version 1 is refused by cddp_hip_create instead of being read at shifted offsets.
 * 3: cddp_hip_options gained the LogDDP barrier fields, cddp_hip_plugin gained constraint_hessians.
 * 4: cddp_hip_options gained the MSIPDDP multi-shooting fields. */
#define CDDP_HIP_ABI_VERSION 5
#define CDDP_HIP_MAX_MODEL_PARAMS 24
#define CDDP_HIP_NAME_LEN 48
#define CDDP_HIP_MAX_ALPHAS 32

/* ---- enumerations -------------------------------------------------------- */

/* Built-in plants with device-side dynamics + derivatives
 * (reference src/dynamics_model/{pendulum,cartpole,unicycle,lti_system,quadrotor,
 *  manipulator}.cpp).  model_params layout is documented per entry. */
enum cddp_hip_model {
  CDDP_HIP_MODEL_PENDULUM = 0,   /* params: length, mass, damping, gravity                     */
  CDDP_HIP_MODEL_CARTPOLE = 1,   /* params: cart_mass, pole_mass, pole_length, gravity, damping */
  CDDP_HIP_MODEL_UNICYCLE = 2,   /* params: (none)                                              */
  CDDP_HIP_MODEL_LTI = 3,        /* discrete x+ = A x + B u; A,B via lti_A / lti_B              */
  CDDP_HIP_MODEL_QUADROTOR = 4,  /* nx=13 quaternion; params: mass, arm, Ixx,Iyy,Izz, gravity   */
  CDDP_HIP_MODEL_MANIPULATOR = 5,/* 3-DOF, central-FD Jacobians h=2e-5; params: (none)          */
  CDDP_HIP_MODEL_QUADROTOR_EULER12 = 6, /* SYNTHETIC nx=12 (BASELINE config 4 shape)            */
  CDDP_HIP_MODEL_MANIPULATOR7 = 7,      /* SYNTHETIC nx=14/nu=7 (BASELINE config 5 shape)       */
  CDDP_HIP_MODEL_BICYCLE = 8,    /* kinematic bicycle [x,y,theta,v] / [a,delta] (bicycle.cpp); params: wheelbase            */
  CDDP_HIP_MODEL_CAR = 9,        /* DISCRETE car [x,y,theta,v] / [delta,a] (car.cpp:24-60, h = dt); params: wheelbase      */
  CDDP_HIP_MODEL_HCW = 10        /* Hill-Clohessy-Wiltshire relative motion [x,y,z,vx,vy,vz] / [Fx,Fy,Fz] (spacecraft_linear.cpp:24-120);
                                    params: mean_motion, mass                                                                  */
};

/* reference src/cddp_core/dynamical_system.cpp:28-83 */
enum cddp_hip_integrator {
  CDDP_HIP_EULER = 0, CDDP_HIP_HEUN = 1, CDDP_HIP_RK3 = 2, CDDP_HIP_RK4 = 3
};

/* Which reference solver core is replaced (cddp_core.cpp:213-233). */
enum cddp_hip_solver {
  CDDP_HIP_SOLVER_CLDDP = 0, CDDP_HIP_SOLVER_IPDDP = 1,
  CDDP_HIP_SOLVER_LOGDDP = 2,  /* logddp_solver.cpp:43-707 + barrier.hpp:37-296: single-shooting relaxed-log-barrier DDP.  Device-resident
                                  through cddp_hip_create / cddp_hip_solve for the built-in plants (round 4; register-resident up to nx = 8,
                                  csrc/kernels_logddp.hpp: every path constraint of the problem enters the barrier; options logddp_*,
                                  filter_*, regularisation and line-search fields; terminal constraints are ignored as the reference's
                                  LogDDP ignores them; result.barrier_mu = mu, result.inf_pr = the violation of the last resetFilter);
                                  scratch-backed above; full DDP only for plants with explicit Hessian tensors); user plug-ins:
                                  cddp_hip_plugin_solve (host loop + stack-fed GPU sweeps) */
  CDDP_HIP_SOLVER_MSIPDDP = 3  /* msipddp_solver.cpp:33-1930: multiple-shooting interior-point DDP (costates, dynamics defects at segment
                                  boundaries, three gap-closing rollout rules, multi-point filter).  Device-resident through cddp_hip_create /
                                  cddp_hip_solve for the built-in plants with nx <= 8 and no terminal set (round 4, csrc/kernels_msipddp.hpp;
                                  options msipddp_*, ipddp_slack / dual init scales, barrier_*, filter_*, regularisation and line-search
                                  fields; warm_start with a state guess in cddp_hip_set_initial is the multiple-shooting start: the guess
                                  is NOT rolled out; result.barrier_mu = mu; cddp_hip_get_duals returns its slacks / duals); user
                                  plug-ins, nx > 8 and terminal sets: cddp_hip_plugin_solve (host loop + stack-fed GPU sweeps).  Path
                                  constraints with nu > 1 and nx != nu are refused by both routes: the reference adds an (nx x nu) product
                                  to its (nu x nx) block Q_ux there (msipddp_solver.cpp:1398), which is only defined for nu = 1 (same
                                  linear layout) or nx = nu.  The unconstrained branch keeps the reference's per-step factor cache
                                  (msipddp_solver.cpp:1169-1185) for the lifetime of the handle */
};

/* Path-constraint kinds (reference include/cddp-cpp/cddp_core/constraint.hpp:144-404). */
enum cddp_hip_constraint_kind {
  CDDP_HIP_CON_CONTROL_BOX = 0, /* BoxConstraint<Control>: g=[-u;u]*s, upper=[-lb;ub]*s  */
  CDDP_HIP_CON_STATE_BOX = 1,   /* BoxConstraint<State>                                   */
  CDDP_HIP_CON_BALL = 2,        /* BallConstraint: g=-s*|x[:d]-c|^2, upper=-s*r^2         */
  CDDP_HIP_CON_LINEAR = 3,      /* LinearConstraint: g=A x, upper=b                       */
  CDDP_HIP_CON_SOC = 4,         /* SecondOrderConeConstraint (constraint.hpp:626-800): g = cos(fov) sqrt(|x[:3]-o|^2+eps) - (x[:3]-o).axis;
                                   center = origin o (3), lower = UNIT opening direction (3), radius = cos(fov), scale = eps, dim = 3 */
  CDDP_HIP_CON_THRUST = 5,      /* ThrustMagnitudeConstraint (:802-927): g = [min-|u|, |u|-max]; lower[0] = min, radius = max, scale = eps, dim = nu */
  CDDP_HIP_CON_MAX_THRUST = 6   /* MaxThrustMagnitudeConstraint (:929-1048): g = |u|-max; radius = max, scale = eps, dim = nu */
};

/* reference include/cddp-cpp/cddp_core/terminal_constraint.hpp:62-263 */
enum cddp_hip_terminal_kind { CDDP_HIP_TERM_EQUALITY = 0, CDDP_HIP_TERM_INEQUALITY = 1 };

/* status_message strings of the reference (cddp_solver_base.cpp:69,82,202,210;
 * clddp_solver.cpp:209,270,274; ipddp_solver.cpp:941,1968,1985,2070). */
enum cddp_hip_status {
  CDDP_HIP_STATUS_RUNNING = 0,
  CDDP_HIP_STATUS_OPTIMAL = 1,            /* "OptimalSolutionFound"                       */
  CDDP_HIP_STATUS_ACCEPTABLE = 2,         /* "AcceptableSolutionFound"                    */
  CDDP_HIP_STATUS_MAX_ITERATIONS = 3,     /* "MaxIterationsReached"                       */
  CDDP_HIP_STATUS_REG_LIMIT = 4,          /* "RegularizationLimitReached_NotConverged"    */
  CDDP_HIP_STATUS_MAX_CPU_TIME = 5,       /* "MaxCpuTimeReached"                          */
  CDDP_HIP_STATUS_REG_LIMIT_CONVERGED = 6 /* "RegularizationLimitReached_Converged" (LogDDP, logddp_solver.cpp:216-222) */
};

/* Line-search selection rule (cddp_solver_base.cpp:255-263 vs :264-314). */
enum cddp_hip_linesearch_rule {
  CDDP_HIP_LS_FIRST_SUCCESS = 0, /* enable_parallel=false: first successful alpha wins      */
  CDDP_HIP_LS_BEST_MERIT = 1     /* enable_parallel=true : lowest merit among successes     */
};

enum cddp_hip_barrier_strategy { CDDP_HIP_BARRIER_ADAPTIVE = 0, CDDP_HIP_BARRIER_MONOTONIC = 1,
                                 CDDP_HIP_BARRIER_IPOPT = 2 };

/* ---- option POD (field-for-field twin of cddp::CDDPOptions, options.hpp:41-251) */
typedef struct cddp_hip_options {
  double tolerance;              /* 1e-5 */
  double acceptable_tolerance;   /* 1e-6 */
  int32_t max_iterations;        /* 1    */
  int32_t use_ilqr;              /* 1; 0 = full DDP (second-order dynamics terms, ipddp_solver.cpp:1070-1082, 1160-1178,
                                  * 1396-1408): pendulum, cart-pole, unicycle, LTI; refused for the other plants        */
  int32_t enable_parallel;       /* 0 -> CDDP_HIP_LS_FIRST_SUCCESS, 1 -> BEST_MERIT        */
  int32_t return_iteration_info; /* 0 */
  int32_t warm_start;            /* 0 */
  int32_t _pad0;
  double termination_scaling_max_factor; /* 100 */
  /* LineSearchOptions */
  int32_t ls_max_iterations;     /* 11 */
  int32_t _pad1;
  double ls_initial_step_size;   /* 1.0 */
  double ls_min_step_size;       /* 1e-8 */
  double ls_step_reduction_factor; /* 0.5 */
  /* RegularizationOptions */
  double reg_initial_value;      /* 1e-6 */
  double reg_update_factor;      /* 10 */
  double reg_max_value;          /* 1e7 */
  double reg_min_value;          /* 1e-10 */
  /* BoxQPOptions (boxqp.hpp:30-41) */
  int32_t boxqp_max_iterations;  /* 100 */
  int32_t _pad2;
  double boxqp_min_gradient_norm;        /* 1e-8 */
  double boxqp_min_relative_improvement; /* 1e-8 */
  double boxqp_step_decrease_factor;     /* 0.6 */
  double boxqp_min_step_size;            /* 1e-22 */
  double boxqp_armijo_constant;          /* 0.1 */
  /* SolverSpecificFilterOptions */
  double filter_merit_acceptance_threshold;     /* 1e-6 */
  double filter_violation_acceptance_threshold; /* 1e-6 */
  double filter_max_violation_threshold;        /* 1e4 */
  double filter_min_violation_for_armijo_check; /* 1e-7 */
  double filter_armijo_constant;                /* 1e-4 */
  /* IPDDPAlgorithmOptions */
  double ipddp_dual_var_init_scale;   /* 0.1 */
  double ipddp_slack_var_init_scale;  /* 1e-2 */
  double ipddp_barrier_tol_mult;      /* 0.1 */
  double ipddp_barrier_update_dual_weight; /* 0.01 */
  double ipddp_mu_kappa_epsilon;      /* 10 */
  int32_t ipddp_check_state_stationarity; /* 0 */
  int32_t ipddp_theta_norm_l2;        /* 0 = "l1" */
  int32_t ipddp_max_filter_size;      /* 5 */
  int32_t ipddp_warmstart_repair;     /* 0 */
  double ipddp_theta_0_floor;         /* 1.0 */
  double ipddp_warmstart_s_min;       /* 1e-4 */
  double ipddp_warmstart_y_min;       /* 1e-4 */
  double ipddp_warmstart_interior_factor; /* 1.1 */
  double ipddp_jacobian_regularization_value;    /* 1e-8 */
  double ipddp_jacobian_regularization_exponent; /* 0.25 */
  /* SolverSpecificBarrierOptions */
  double barrier_mu_initial;          /* 1.0 */
  double barrier_mu_min_value;        /* 1e-10 */
  double barrier_mu_update_factor;    /* 0.5 */
  double barrier_mu_update_power;     /* 1.2 */
  double barrier_min_fraction_to_boundary; /* 0.99 */
  int32_t barrier_strategy;           /* CDDP_HIP_BARRIER_ADAPTIVE */
  int32_t _pad3;
  /* CDDPOptions::max_cpu_time [s], 0 = unlimited (options.hpp:212; checked at the top of every iteration,
   * cddp_solver_base.cpp:77-90).  One clock for the batch: when it expires, every trajectory still running
   * terminates with "MaxCpuTimeReached" and iterations = the iteration the check fired in. */
  double max_cpu_time;                /* 0 */
  /* LogBarrierOptions (options.hpp:135-143; LogDDP only): log_barrier.barrier.{mu_initial, mu_min_value, mu_update_factor} and
   * log_barrier.relaxed_log_barrier_delta */
  double logddp_mu_initial;           /* 1.0 */
  double logddp_mu_min_value;         /* 1e-10 */
  double logddp_mu_update_factor;     /* 0.5 */
  double logddp_relaxed_delta;        /* 1e-10 */
  /* MultiShootingOptions (options.hpp:120-130; MSIPDDP only).  With solver = MSIPDDP the InteriorPointOptions half of options.msipddp
   * (dual_var_init_scale, slack_var_init_scale, barrier.*) travels in ipddp_dual_var_init_scale / ipddp_slack_var_init_scale /
   * barrier_* above -- the mirrors copy options.msipddp there instead of options.ipddp. */
  double msipddp_costate_var_init_scale;  /* 1e-6 */
  int32_t msipddp_segment_length;         /* 5 */
  int32_t msipddp_rollout_type;           /* 0 = "nonlinear", 2 = "hybrid", 1 = any other string (plain rollout at the boundary) */
  int32_t msipddp_use_controlled_rollout; /* 0 */
  int32_t _pad4;
} cddp_hip_options;

/* Fill *opt with the reference defaults (options.hpp in-class initialisers). */
void cddp_hip_default_options(cddp_hip_options *opt);

/* ---- constraint descriptors --------------------------------------------- */

/* One entry of CDDP::path_constraint_set_ (std::map<std::string, unique_ptr<Constraint>>,
 * cddp_core.hpp:422).  The library sorts entries by `name` (lexicographic, as std::map
 * iterates) to reproduce the reference's dual stacking order (ipddp_solver.cpp:1371-1384).
 * CLDDP only honours a CONTROL_BOX whose name is exactly "ControlConstraint"
 * (clddp_solver.cpp:85-86). */
typedef struct cddp_hip_constraint {
  char name[CDDP_HIP_NAME_LEN];
  int32_t kind;        /* cddp_hip_constraint_kind */
  int32_t dim;         /* box: #variables (dual dim = 2*dim); ball: center size; linear: #rows */
  const double *lower; /* box: dim */
  const double *upper; /* box: dim */
  const double *center;/* ball: dim */
  const double *A;     /* linear: dim x nx row-major */
  const double *b;     /* linear: dim */
  double radius;       /* ball */
  double scale;        /* scale_factor (box, ball); 1.0 default */
} cddp_hip_constraint;

/* One entry of CDDP::terminal_constraint_set_. */
typedef struct cddp_hip_terminal_constraint {
  char name[CDDP_HIP_NAME_LEN];
  int32_t kind;         /* cddp_hip_terminal_kind */
  int32_t dim;          /* equality: nx; inequality: #rows of A_N */
  const double *target; /* equality: nx   (h = x_N - target)   */
  const double *A;      /* inequality: dim x nx (g = A x_N - b) */
  const double *b;      /* inequality: dim */
} cddp_hip_terminal_constraint;

/* ---- problem descriptor -------------------------------------------------- */

/* Everything cddp::CDDP holds for one problem (cddp_core.hpp:215-423), as a POD.
 * All trajectories of a batch share this descriptor; they differ in x0 / U0 only. */
typedef struct cddp_hip_problem {
  int32_t abi_version;  /* CDDP_HIP_ABI_VERSION */
  int32_t solver;       /* cddp_hip_solver */
  int32_t model;        /* cddp_hip_model */
  int32_t integrator;   /* cddp_hip_integrator */
  int32_t nx, nu, horizon;
  int32_t _pad0;
  double dt;
  double model_params[CDDP_HIP_MAX_MODEL_PARAMS];
  const double *lti_A;  /* CDDP_HIP_MODEL_LTI: nx*nx (discrete A) */
  const double *lti_B;  /* CDDP_HIP_MODEL_LTI: nx*nu (discrete B) */
  /* QuadraticObjective(Q, R, Qf, x_ref, reference_states, dt) -- objective.cpp:30-65.
   * Q and R are given UNSCALED; the library multiplies them by dt as the ctor does. */
  const double *Q;      /* nx*nx */
  const double *R;      /* nu*nu */
  const double *Qf;     /* nx*nx */
  const double *x_ref;  /* nx    */
  const double *x_ref_traj; /* (horizon+1)*nx per-step references or NULL (objective.cpp:83-88) */
  int32_t n_constraints;
  int32_t n_terminal;
  const cddp_hip_constraint *constraints;
  const cddp_hip_terminal_constraint *terminal;
  cddp_hip_options options;
} cddp_hip_problem;

/* ---- per-trajectory result record --------------------------------------- */

/* Twin of cddp::CDDPSolution scalars (cddp_core.hpp:54-76) plus work counters used
 * for the roofline accounting. 96 bytes. */
typedef struct cddp_hip_result {
  double final_objective;
  double merit_function;
  double inf_pr, inf_du, inf_comp;
  double barrier_mu;
  double regularization;
  double alpha_pr, alpha_du;
  double step_norm;
  int32_t iterations;
  int32_t status;          /* cddp_hip_status */
  int32_t n_backward;      /* backward sweeps executed (incl. regularisation retries) */
  int32_t n_forward;       /* forward rollouts required by the sequential rule        */
} cddp_hip_result;

/* The 16-byte record that is all-gathered across GPUs. */
typedef struct cddp_hip_gather_record {
  double final_objective;
  int32_t iterations;
  int32_t status;
} cddp_hip_gather_record;

/* Result of one line-search trial (twin of ForwardPassResult scalars, cddp_core.hpp:105-145). */
typedef struct cddp_hip_trial {
  double alpha, alpha_pr, alpha_du;
  double cost, merit_function, theta, inf_pr, inf_comp;
  int32_t success;
  int32_t _pad;
} cddp_hip_trial;

/* Timing / work summary of one cddp_hip_solve call. */
typedef struct cddp_hip_stats {
  double solve_ms;          /* hipEvent time of the device-resident loop           */
  double backward_ms;       /* sum of K1+K2 kernel time (hipEvent); 0 unless the timing detail covers it */
  double forward_ms;        /* sum of K4 (rollout) kernel time; 0 unless covered   */
  double update_ms;         /* sum of K4b (costate) + K5 kernel time; 0 unless covered */
  int64_t sweeps;           /* sum over trajectories of n_backward                 */
  int64_t rollouts;         /* sum over trajectories of n_forward                  */
  int64_t rollouts_launched;/* rollouts actually executed (speculative alphas too) */
  int64_t traj_iterations;  /* sum over trajectories of iterations                 */
  int32_t outer_iterations; /* host loop trips                                     */
  int32_t n_converged;      /* status OPTIMAL or ACCEPTABLE                        */
  int32_t kernel_launches;
  int32_t timing_detail;    /* CDDP_HIP_TIMING_* the class times were taken with   */
  int64_t rollout_steps;    /* time steps the credited rollouts (`rollouts`) actually traversed: a trial the
                             * reference abandons at its first fraction-to-boundary violation
                             * (ipddp_solver.cpp:1632-1645) counts the steps completed before it, not N */
} cddp_hip_stats;

/* Which kernel classes cddp_hip_solve brackets with hipEvents when a stats block is requested.  An event costs
 * about 5 us of queue time, i.e. bracketing every class of every iteration adds ~5 % to a C2 solve; the default
 * brackets the rollout launches only (2 events per iteration). */
enum {
  CDDP_HIP_TIMING_ROLLOUT = 0,   /* forward_ms only                       */
  CDDP_HIP_TIMING_ALL = 1,       /* backward_ms, forward_ms, update_ms    */
  CDDP_HIP_TIMING_SWEEP = 2      /* backward_ms only                      */
};

typedef struct cddp_hip_handle cddp_hip_handle;

/* ---- entry points -------------------------------------------------------- */

int cddp_hip_abi_version(void);
/* 0: the reference's plants evaluate sin / cos with the device libm (lib/libcddp_hip.so, the product build); 1: with the
 * branch-free routine of csrc/dev_trig.hpp (lib/libcddp_hip_sharedtrig.so, the parity build a host selects with
 * CDDP_HIP_TRIG=shared): same C-ABI, same kernels; a CPU checker can run that routine too, which makes plants whose accept /
 * reject decisions hinge on the last bit of a sine bit-comparable (tests/test_shared_trig_parity.py). */
int cddp_hip_trig_shared(void);
const char *cddp_hip_last_error(void);
/* Number of visible HIP devices (0 when the runtime finds none). */
int cddp_hip_device_count(void);
const char *cddp_hip_status_string(int status);

/* Build line-search ladder exactly as detail::buildLineSearchAlphas
 * (cddp_context_utils.cpp:37-57). Returns the number of alphas written (<= cap). */
int cddp_hip_build_alphas(const cddp_hip_options *opt, double *alphas, int cap);

/* Create a solver instance for `batch` independent trajectories of `problem` on `device`.
 * Replaces CDDP::createSolver + ISolverAlgorithm construction (cddp_core.cpp:213-241).
 * Fails (returns <0) when no GPU/HIP runtime is available -- there is no CPU fallback. */
int cddp_hip_create(const cddp_hip_problem *problem, int batch, int device,
                    cddp_hip_handle **out);
int cddp_hip_destroy(cddp_hip_handle *h);

/* Run all subsequent work of this handle on an existing hipStream_t (e.g. torch's). */
int cddp_hip_set_stream(cddp_hip_handle *h, void *hip_stream);

/* Select the class-timing detail (CDDP_HIP_TIMING_*) of later cddp_hip_solve calls that pass a stats block. */
int cddp_hip_set_timing_detail(cddp_hip_handle *h, int detail);

/* CDDP::setInitialState / setInitialTrajectory for the whole batch (cddp_core.cpp:66-140).
 * x0: batch*nx. U0: batch*N*nu or NULL (zeros). X0: batch*(N+1)*nx or NULL (x0 replicated,
 * as cddp::example::makeInitialTrajectory does). CLDDP linearises X0 as given
 * (clddp_solver.cpp:68-74); IPDDP re-rolls X from U (ipddp_solver.cpp:868-874). */
int cddp_hip_set_initial(cddp_hip_handle *h, const double *x0, const double *U0,
                         const double *X0);

/* ISolverAlgorithm::initialize for the batch (clddp_solver.cpp:28-75, ipddp_solver.cpp:644-914).
 * The handle is the solver object: with options.warm_start the first call takes the reference's "warm start
 * with provided trajectory" branch (ipddp_solver.cpp:733-816; CLDDP falls back to its cold start), every later
 * call the "existing solver state" branch (ipddp_solver.cpp:675-731, clddp_solver.cpp:51-60): gains, slack /
 * dual / costate / terminal variables, regularisation and step lengths persist on the device, X is re-rolled
 * out from the current (or newly supplied) controls.  Without warm_start every call is a cold start from the
 * trajectory of the last cddp_hip_set_initial. */
int cddp_hip_initialize(cddp_hip_handle *h);

/* CDDP::setOptions on a live handle (e.g. to switch warm_start on between two solves).  The line-search
 * ladder size is fixed at create time. */
int cddp_hip_set_options(cddp_hip_handle *h, const cddp_hip_options *options);

/* CDDP::setInitialState for the batch: x0[b][i] only; controls, duals and gains on the device are kept
 * (the MPC restart of SURVEY.md 8(f1)).  cddp_hip_set_initial additionally replaces the trajectory
 * (CDDP::setInitialTrajectory). */
int cddp_hip_set_initial_state(cddp_hip_handle *h, const double *x0);

/* Forget the solver state of the handle -- gains, slack / dual / costate / terminal variables, regularisation -- as if the reference's
 * solver OBJECT were constructed anew, without re-allocating anything: with options.warm_start the next initialize / solve then takes the
 * reference's "warm start with provided trajectory" branch (ipddp_solver.cpp:733-816, clddp_solver.cpp:35-60: the trajectory of the last
 * cddp_hip_set_initial, barrier parameter from its largest violation, duals re-initialised) instead of "existing solver state"
 * (:653-731).  The MPC caller that builds a fresh problem per step (examples/ipddp_mpcc_rc.py:649-705, test_ipddp_solver.cpp's warm-start
 * tests) is cddp_hip_forget_solver_state + cddp_hip_set_initial(x0, shifted U, shifted X) + cddp_hip_solve on one long-lived handle. */
int cddp_hip_forget_solver_state(cddp_hip_handle *h);

/* Overwrite the path slack / dual variables S[b][t][m], Y[b][t][m] (either may be NULL) and the terminal
 * slack / dual / multipliers S_T[b][mT], Y_T[b][mT], Lambda_T[b][pT] of an initialised handle: the
 * reference's IPDDPSolverTestAccess::setPathInterior / setTerminalInterior / setTerminalEqualityMultiplier,
 * and the hook for callers that shift duals between MPC solves. */
int cddp_hip_set_duals(cddp_hip_handle *h, const double *S, const double *Y);
int cddp_hip_set_terminal(cddp_hip_handle *h, const double *S_T, const double *Y_T, const double *Lambda_T);
/* Overwrite the per-trajectory barrier parameter mu_[b] (IPDDP; mu_ of ipddp_solver.hpp) and / or the
 * regularisation context.regularization_[b] of an initialised handle (either may be NULL): with
 * cddp_hip_set_initial + cddp_hip_set_duals this installs an arbitrary iterate (X, U, S, Y, mu, reg), e.g. a late
 * iterate of another solve, for one cddp_hip_backward / cddp_hip_forward -- the role of the reference's
 * IPDDPSolverTestAccess (tests/cddp_core/test_ipddp_solver.cpp:30-135). */
int cddp_hip_set_barrier_state(cddp_hip_handle *h, const double *mu, const double *reg);

/* One backwardPass for every trajectory (clddp_solver.cpp:79-204 / ipddp_solver.cpp:960-1569),
 * including the "retry with larger regularisation" loop of cddp_solver_base.cpp:93-111.
 * ok[b] (optional, batch ints) receives 1 when the sweep succeeded. */
int cddp_hip_backward(cddp_hip_handle *h, int32_t *ok);

/* forwardPass for every trajectory and every given alpha
 * (clddp_solver.cpp:215-262 / ipddp_solver.cpp:1571-1876).  trials: batch*n_alphas records,
 * trials[b*n_alphas + a]. Does not commit anything. */
int cddp_hip_forward(cddp_hip_handle *h, const double *alphas, int n_alphas,
                     cddp_hip_trial *trials);

/* ISolverAlgorithm::solve for the batch: cddp_solver_base.cpp:29-186 as a per-trajectory
 * device state machine.  Calls cddp_hip_initialize first.  stats may be NULL. */
int cddp_hip_solve(cddp_hip_handle *h, cddp_hip_stats *stats);

/* ---- getters (host buffers, batch-major) -------------------------------- */
int cddp_hip_get_results(cddp_hip_handle *h, cddp_hip_result *results /* batch */);
/* Head of the plan (round 4): u_0[batch][nu] and x_1[batch][nx] of every trajectory's current iterate -- what a receding-horizon
 * caller applies / re-starts from (examples/ipddp_mpcc_rc.py:649-705 reads sol.control_trajectory[0]); one small gather + one
 * copy instead of the whole (N + 1) x nx x batch trajectory.  Either pointer may be NULL. */
int cddp_hip_get_plan_head(cddp_hip_handle *h, double *u0, double *x1);
int cddp_hip_get_trajectory(cddp_hip_handle *h, double *X /* B*(N+1)*nx */, double *U /* B*N*nu */);
/* feedback_gains K_u (B*N*nu*nx) and feed-forward k_u (B*N*nu); either may be NULL. */
int cddp_hip_get_gains(cddp_hip_handle *h, double *K, double *k);
/* Value-function expansion along the horizon: Vx B*(N+1)*nx, Vxx B*(N+1)*nx*nx
 * (k_lambda_/K_lambda_ of ipddp_solver.cpp:1050-1104; V_x/V_xx of clddp_solver.cpp:188-192). */
int cddp_hip_get_value(cddp_hip_handle *h, double *Vx, double *Vxx);
/* The dynamics linearisation of the last backward pass, i.e. what precomputeDynamicsDerivatives leaves in F_x_ / F_u_
 * (cddp_solver_base.cpp:319-394): A[b][t] = I + dt f_x(x_t, u_t) (nx x nx, row-major), B[b][t] = dt f_u (nx x nu).  Either may be NULL. */
int cddp_hip_get_linearization(cddp_hip_handle *h, double *A /* B*N*nx*nx */, double *Bm /* B*N*nx*nu */);
/* Slack / dual / constraint residual trajectories, B*N*m each (m = total path dual dim). */
int cddp_hip_get_duals(cddp_hip_handle *h, double *S, double *Y, double *G);
/* Costate trajectory of the current iterate: Lambda_[t] = Lambda_[t] + alpha_pr k_lambda[t] + K_lambda[t] dx_t of the accepted forward pass
 * (ipddp_solver.cpp:1613-1616, 1660-1663: N + 1 rows; msipddp_solver.cpp:1466-1467, 1639-1641: N rows).  *rows receives the row count,
 * Lambda (B * rows * nx, may be NULL to query the count) the rows.  IPDDP and MSIPDDP handles only (added in round 5; ABI version unchanged). */
int cddp_hip_get_costates(cddp_hip_handle *h, double *Lambda, int32_t *rows);
/* Terminal-constraint state (IPDDP): stacked terminal-inequality slack / dual / residual
 * S_T, Y_T, G_T (B*mT each) and terminal-equality multipliers Lambda_T (B*pT); any may be NULL.
 * dims[0] = mT, dims[1] = pT (S_T_, Y_T_, G_T_, Lambda_T_eq_ of ipddp_solver.hpp). */
int cddp_hip_get_terminal(cddp_hip_handle *h, double *S_T, double *Y_T, double *G_T, double *Lambda_T, int32_t *dims);
/* Scalars of the last backward pass: dV (B*2), and per-trajectory regularisation (B). */
int cddp_hip_get_backward_scalars(cddp_hip_handle *h, double *dV, double *reg);
/* CDDPSolution::History for the first `hist_batch` trajectories (requires
 * options.return_iteration_info): hist[b][it][9] = {objective, merit, alpha_pr, alpha_du,
 * inf_du, inf_pr, inf_comp, mu, regularization}; counts[b] = entries; the `it` extent of the block is
 * cddp_hip_history_capacity(h) rows. */
int cddp_hip_get_history(cddp_hip_handle *h, int hist_batch, double *hist, int32_t *counts);
/* Rows per trajectory of the history block above: max_iterations + 1 of the options the handle was CREATED with
 * (cddp_hip_set_options may lower max_iterations afterwards; the row stride does not follow). */
int cddp_hip_history_capacity(cddp_hip_handle *h);

/* Write the 16-byte gather records of this handle's batch into a DEVICE buffer
 * (batch * sizeof(cddp_hip_gather_record)), on the handle's stream: the send buffer of the
 * single RCCL all-gather of SURVEY.md section 8(e). */
int cddp_hip_write_gather_records_device(cddp_hip_handle *h, void *device_ptr);

/* ---- multi-GPU: the single collective of the path --------------------------------------
 * One process (or host thread) per GPU, each with its own handle over a contiguous block of the global batch; nothing
 * is exchanged during a solve.  After it, ONE RCCL all-gather collects every trajectory's 16-byte record.
 * The communicator is the caller's ncclComm_t (passed as void*), or one made with the helpers below (RCCL is loaded
 * lazily with dlopen: the solver core has no link-time dependency on it). */
#define CDDP_HIP_COMM_ID_BYTES 128
/* ncclGetUniqueId: call on ONE rank, hand the 128 bytes to the others by any means (MPI, TCP store, file). */
int cddp_hip_comm_unique_id(char *id_out /* CDDP_HIP_COMM_ID_BYTES */);
/* ncclCommInitRank on `device` (collective over all ranks). */
int cddp_hip_comm_init(const char *id_in, int world, int rank, int device, void **comm_out);
int cddp_hip_comm_destroy(void *comm);
/* ncclCommCount / ncclCommUserRank of the communicator: the number of ranks RCCL itself sees, and this process's rank (round 4) */
int cddp_hip_comm_info(void *comm, int *count_out, int *rank_out);
/* All-gather the records of this rank's batch into recv_device (DEVICE buffer of world * shard_capacity records,
 * rank r's block at r * shard_capacity).  shard_capacity >= every rank's batch: with an uneven block partition the
 * ranks pad to the largest shard; padding records read status = iterations = -1.  comm == NULL is valid for
 * world == 1 (device copy).  Runs on the stream given to cddp_hip_set_stream (else the handle's own) and returns when
 * the gathered buffer is complete. */
int cddp_hip_allgather_results(cddp_hip_handle *h, void *comm /* ncclComm_t */, int world, int shard_capacity,
                               void *recv_device);

/* Total path dual dimension m and terminal-equality dimension p of the handle's problem. */
int cddp_hip_dual_dim(cddp_hip_handle *h);
int cddp_hip_batch(cddp_hip_handle *h);
/* Number of tile groups the handle's batch is cut into (whole 64-trajectory tiles per group; each group has its own
 * device buffers and stream and cddp_hip_solve keeps all of them in flight).  Results do not depend on it.
 * Environment CDDP_HIP_GROUPS=n pins it at create time (1 = one stream). */
int cddp_hip_num_groups(cddp_hip_handle *h);
/* How many of those groups cddp_hip_solve keeps in flight AT ONCE (round 5).  1: the groups (chunks of a large batch) are solved one
 * after the other.  2: the static CU partition of IPDDP / CLDDP batches of >= 32 tiles -- two groups, each with every kernel of its
 * iterations on its own symmetric half of the chip (streams made with hipExtStreamCreateWithCUMask), so one "launch" of a kernel
 * class in the timing model is the two concurrent half-batch launches.  No reference counterpart (the reference solves one problem). */
int cddp_hip_concurrency(cddp_hip_handle *h);

/* ---- stack-fed mode (host plugins) ---------------------------------------
 * Arbitrary DynamicalSystem / Objective / Constraint subclasses (e.g. the user-defined QuadraticScalarSystem of
 * tests/cddp_core/test_ipddp_solver.cpp:291-346) cannot run on the GPU.  In stack-fed mode the caller evaluates them
 * on the host and hands over the (N x batch) derivative stacks the reference precomputes per backward pass
 * (cddp_solver_base.cpp:319-394, ipddp_solver.cpp:2145-2250); the GPU runs the backward pass on them.  The forward
 * rollout needs the plugin's f(x, u) and therefore stays with the host.
 * A stack handle owns its device buffers: upload the stacks of an iterate once, sweep (one launch), read the results.
 * Layout: batch-major host arrays, fx[b][t][i][j] = A_t = I + dt*f_x, fu[b][t][i][j] = B_t = dt*f_u,
 * lx[b][t][i], lu[b][t][j], lxx[b][t][i][i'], luu[b][t][j][j'], lux[b][t][j][i]; VxN[b][i], VxxN[b][i][i'] = terminal cost
 * gradient / Hessian; path constraints stacked in std::map (name) order: y, s, g [b][t][r] (g = evaluate - upper bound),
 * Gx[b][t][r][i], Gu[b][t][r][j]. */
typedef struct cddp_hip_stack_handle cddp_hip_stack_handle;
enum cddp_hip_stacks_branch {
  CDDP_HIP_STACKS_CLDDP = 0,      /* clddp_solver.cpp:79-204 without control bounds                    */
  CDDP_HIP_STACKS_IPDDP = 1,      /* ipddp_solver.cpp:1048-1118 (no constraints)                       */
  CDDP_HIP_STACKS_IPDDP_PATH = 2, /* ipddp_solver.cpp:1355-1568 (path constraints; handle with m > 0)  */
  CDDP_HIP_STACKS_MSIPDDP = 4,    /* msipddp_solver.cpp:1112-1208 (no constraints): IPDDP recursion + defect stack (cddp_hip_set_defect_stack) */
  CDDP_HIP_STACKS_MSIPDDP_PATH = 5, /* msipddp_solver.cpp:1222-1420 (path constraints; handle with m > 0, defect stack): the condensation with
                                     plain y / s ratios (no slack floor, no clipping), Q_ux updated as :1398 writes it -- nu = 1 or nx = nu
                                     only --, no linear-policy rollout / step caps */
  CDDP_HIP_STACKS_IPDDP_TERM_EQ = 6, /* ipddp_solver.cpp:1120-1353, 413-639: the reduced LQR with terminal-equality rows.  The stacks carry the LQ
                                   * model the reference builds at :1143-1245 (fx = A, fu = B, lx = q, lu = r, lxx = Q, luu = R WITHOUT the
                                   * regularisation, lux = M as nx x nu, VxN = q_N, VxxN = Q_N; path constraints condensed by the caller) and
                                   * cddp_hip_set_terminal_equality the dense H_T, b_T = -h_T, previous multipliers (handle with m = 0) */
  CDDP_HIP_STACKS_LOGDDP = 3      /* logddp_solver.cpp:470-575: the caller folds the relaxed log barrier's gradients / Hessians (barrier.hpp:95-262)
                                     into lx, lu, lxx, luu, lux; handle with m = 0                     */
};
/* ABI 5: cddp_hip_stacks_backward copies a cddp_hip_options from a caller pointer, so the handle is created against a stated ABI
 * version and options size; C / C++ callers use the macro below, which passes the values of the header they compile against;
 * other bindings pass their own.  A mismatch is refused. */
int cddp_hip_stacks_create_abi(int abi_version, int options_bytes, int device, int batch, int nx, int nu,
                               int m /* total path dual dim, 0 = none */, int horizon, cddp_hip_stack_handle **out);
#define cddp_hip_stacks_create(device, batch, nx, nu, m, horizon, out) \
  cddp_hip_stacks_create_abi(CDDP_HIP_ABI_VERSION, (int)sizeof(cddp_hip_options), (device), (batch), (nx), (nu), (m), (horizon), (out))
int cddp_hip_stacks_destroy(cddp_hip_stack_handle *h);
/* Upload the dynamics / cost stacks of the current iterate.  The first call must supply all of them; later calls may
 * pass NULL for stacks that did not change (e.g. constant cost Hessians). */
int cddp_hip_set_stacks(cddp_hip_stack_handle *h, const double *fx, const double *fu, const double *lx, const double *lu,
                        const double *lxx, const double *luu, const double *lux, const double *VxN, const double *VxxN);
/* Multiple-shooting defects d[b][t] = f(x_t, u_t) - x_{t+1} (nx each) for CDDP_HIP_STACKS_MSIPDDP: Q_x and Q_u are formed with
 * V_x + V_xx d_t (msipddp_solver.cpp:1144-1145).  The caller derives k_lambda[t] = -lambda_t + V_x(t+1) + V_xx(t+1) d_t and
 * K_lambda[t] = V_xx(t+1) (:1192-1194) from the returned value stacks.  NULL drops the stack. */
int cddp_hip_set_defect_stack(cddp_hip_stack_handle *h, const double *defects);
/* Control limits of the CLDDP branch (the constraint named "ControlConstraint", clddp_solver.cpp:85-86, 147-178): every step
 * solves the BoxQP  min 0.5 k^T Q_uu_reg k + Q_u^T k,  lower - u_t <= k <= upper - u_t  (boxqp.cpp:25-250, parameters from
 * options.boxqp_*), warm-started with the k_t of the handle's previous sweep, and the feedback gain lives on the free
 * directions.  lower / upper: nu each; U: the current controls, batch-major [b][t][nu].  The first call supplies all three;
 * later calls may pass NULL bounds to keep them (U changes every iteration); three NULLs remove the box. */
int cddp_hip_set_control_box(cddp_hip_stack_handle *h, const double *lower, const double *upper, const double *U);
/* Full DDP (options.use_ilqr = false) for host plug-ins: the dynamics Hessian tensors of the current iterate, as the reference
 * keeps them in F_xx_ / F_uu_ / F_ux_ (cddp_solver_base.cpp:346-356) but ALREADY multiplied by dt -- Fxx[b][t][i] (nx x nx),
 * Fuu[b][t][i] (nu x nu), Fux[b][t][i] (nu x nx), i = output row of f.  The IPDDP and LogDDP branches then add V_x(i) times
 * them to Q_xx, Q_uu, Q_ux (ipddp_solver.cpp:1070-1082, 1396-1408; logddp_solver.cpp:505-515); the reference's CLDDP backward pass
 * has no such terms and cddp_hip_stacks_backward refuses that combination.  Three NULLs return to Gauss-Newton. */
int cddp_hip_set_hessian_stacks(cddp_hip_stack_handle *h, const double *Fxx, const double *Fuu, const double *Fux);
/* MSIPDDP's per-step factor cache (msipddp_solver.cpp:1169-1185, unconstrained branch): the reference keeps one LDLT of
 * sym(Q_uu) + reg I per step and refactors a step only while its cached factor is invalid, so from the second sweep of a solve on
 * every step is solved with the matrix of its FIRST sweep.  enable != 0 allocates (first call) and CLEARS the cache: call it at
 * the start of each solve; the CDDP_HIP_STACKS_MSIPDDP branch then factors the cached matrix of a step where there is one and
 * caches the one it factored where there was none.  enable == 0 (the default of a new handle): every sweep factors its own. */
int cddp_hip_stacks_factor_cache(cddp_hip_stack_handle *h, int enable);
/* Upload the condensation inputs of the path-constrained branch (same NULL rule). */
int cddp_hip_set_constraint_stacks(cddp_hip_stack_handle *h, const double *y, const double *s, const double *g,
                                   const double *Gx, const double *Gu);
/* One backwardPass on the uploaded stacks.  reg[b] = context.regularization_ per trajectory; mu[b] = barrier parameter
 * (path branch; NULL otherwise); retry != 0 adds the "increase regularisation and retry" loop of
 * cddp_solver_base.cpp:93-111 (options->reg_update_factor / reg_max_value).  ok[b] (optional) = 1 where the sweep succeeded. */
int cddp_hip_stacks_backward(cddp_hip_stack_handle *h, int branch, const cddp_hip_options *options, const double *reg,
                             const double *mu, int retry, int32_t *ok);
/* hipEvent time of the last sweep launch [ms]. */
double cddp_hip_stacks_last_kernel_ms(cddp_hip_stack_handle *h);
/* Form of the last sweep launch: 0 = one lane per trajectory, 1 = lane-cooperative (sixteen lanes per trajectory, step state in
 * LDS: the default for nx > 8; CDDP_HIP_STACKS_SWEEP=lane|coop overrides where both forms are instantiated).  Bitwise the same
 * results either way. */
int cddp_hip_stacks_last_sweep_form(cddp_hip_stack_handle *h);
/* K (B*N*nu*nx), k (B*N*nu), Vx (B*(N+1)*nx), Vxx (B*(N+1)*nx*nx), dV (B*2); any may be NULL. */
int cddp_hip_stacks_get_gains(cddp_hip_stack_handle *h, double *K, double *k, double *Vx, double *Vxx, double *dV);
/* Path branch: k_y, k_s (B*N*m), K_y, K_s (B*N*m*nx) and the linear-policy rollout dX (B*(N+1)*nx), ipddp_solver.cpp:1458-1520. */
int cddp_hip_stacks_get_constraint_gains(cddp_hip_stack_handle *h, double *k_y, double *K_y, double *k_s, double *K_s, double *dX);
/* Per-trajectory scalars of the sweep (B each, any may be NULL): regularisation used, inf_du, inf_pr, inf_comp, step_norm
 * and computeMaxStepSizes' (alpha_pr_max, alpha_du_max) (ipddp_solver.cpp:2939-2988; 1 without path constraints). */
int cddp_hip_stacks_get_scalars(cddp_hip_stack_handle *h, double *reg, double *inf_du, double *inf_pr, double *inf_comp,
                                double *step_norm, double *alpha_pr_max, double *alpha_du_max);
/* Terminal-equality branch (CDDP_HIP_STACKS_IPDDP_TERM_EQ; ipddp_solver.cpp:478-639): H_T [B][pT][nx] = the stacked terminal-equality
 * Jacobian (dense rows), b_T [B][pT] = -h_T(x_N), lambda_prev [B][pT] = Lambda_T_eq_ of the iterate, reg_floor [B] = max(1e-10,
 * options.ipddp.jacobian_regularization_value * pow(max(mu, 0), jacobian_regularization_exponent)) evaluated by the caller (:577-578: the
 * plug-in route keeps the host's elementary functions).  1 <= pT <= 8; handle created with m = 0. */
int cddp_hip_set_terminal_equality(cddp_hip_stack_handle *h, int pT, const double *H_T, const double *b_T, const double *lambda_prev,
                                   const double *reg_floor);
/* Results of the terminal-equality sweep: lambda_delta = dLambda_T_eq_ (B*pT) and the linear-policy rollout dX (B*(N+1)*nx) of the
 * recombined gains (:1252-1268); K, k, Vx (= k_lambda), Vxx (= K_lambda) come through cddp_hip_stacks_get_gains, the regularisation used,
 * inf_du and step_norm through cddp_hip_stacks_get_scalars.  Either may be NULL. */
int cddp_hip_stacks_get_terminal(cddp_hip_stack_handle *h, double *lambda_delta, double *dX);
/* One-shot unconstrained form (create + upload + one launch + download + destroy): Gauss-Newton sweep of
 * ipddp_solver.cpp:1048-1118 when reg_in_value != 0, clddp_solver.cpp:79-204 without bounds otherwise, scalar `reg`. */
int cddp_hip_backward_stacks(int device, int batch, int nx, int nu, int horizon,
                             const double *fx, const double *fu, const double *lx,
                             const double *lu, const double *lxx, const double *luu,
                             const double *lux, const double *VxN, const double *VxxN,
                             double reg, int reg_in_value, double *K, double *k, double *Vx,
                             double *Vxx, double *dV, int32_t *ok, double *kernel_ms);

/* ---- host plug-in solve (north_star: "keeps cddp-cpp's DynamicsModel / Constraint / Objective plugin surface") ----------------
 * CDDP::solve() for problems whose DynamicalSystem / Objective / Constraint objects are arbitrary HOST subclasses (the reference's
 * QuadraticScalarSystem of tests/cddp_core/test_ipddp_solver.cpp:291-346, Python plug-ins of python/tests/test_custom_dynamics.py,
 * NonlinearObjective subclasses ...).  The callbacks are the reference's virtual functions, flattened to plain pointers:
 *   discrete_dynamics            DynamicalSystem::getDiscreteDynamics(x, u, t)                       (dynamical_system.hpp)
 *   jacobians                    getStateJacobian / getControlJacobian: CONTINUOUS-time f_x (nx x nx), f_u (nx x nu), row-major;
 *                                the library forms A = I + dt f_x, B = dt f_u (cddp_solver_base.cpp:340-344)
 *   hessians                     getStateHessian / getControlHessian / getCrossHessian (needed iff options.use_ilqr == 0):
 *                                fxx[i] (nx x nx), fuu[i] (nu x nu), fux[i] (nu x nx), i = output row of f; or NULL
 *   running_cost, terminal_cost  Objective::running_cost(x, u, index), terminal_cost(x_N)           (objective.hpp)
 *   running_cost_derivatives     l_x (nx), l_u (nu), l_xx (nx x nx), l_uu (nu x nu), l_ux (nu x nx)
 *   terminal_cost_derivatives    getFinalCostGradient (nx), getFinalCostHessian (nx x nx)
 *   constraints                  the path-constraint set, stacked in std::map (name) order: g = evaluate(x, u) - getUpperBound()
 *                                (m rows), getStateJacobian (m x nx), getControlJacobian (m x nu); gx / gu may be NULL when only g
 *                                is wanted.  n_constraints objects of constraint_dims[] rows each (the reference sums the
 *                                barrier / violation terms constraint by constraint, ipddp_solver.cpp:2778-2937).
 *   control_lower / control_upper  CLDDP only: the bounds of the constraint literally named "ControlConstraint"
 *                                (clddp_solver.cpp:85-86, 147-178, 227-228) or NULL; CLDDP ignores every other constraint.
 * The GPU runs the backward pass of the whole batch (stack-fed sweeps above); the forward pass needs the plug-in's f(x, u) and runs
 * on the host (csrc/plugin_solve.hip).  All trajectories of the batch share the plug-in; they differ in x0 / U0 / X0.  Callbacks are
 * called from the calling thread only unless cddp_hip_plugin_set_host_threads says otherwise.  Terminal constraints: cddp_hip_plugin_solve_terminal
 * below.  Warm starts: a stateless call can only take the "provided trajectory" branch; the terminal entry point implements it.
 * solver = CDDP_HIP_SOLVER_LOGDDP runs the reference's LogDDP (logddp_solver.cpp:43-707: single shooting, relaxed log barrier of every
 * path constraint folded into the cost derivatives on the host, Riccati sweep of the batch on the GPU, filter line search on the
 * host); it takes the same callbacks, plus constraint_hessians when a constraint has curvature. */
/* Host evaluation of a BUILT-IN plant (the kernels' own model source compiled for the host, csrc/host_models.cpp): the reference's
 * DynamicalSystem::getDiscreteDynamics (x_next), getStateJacobian / getControlJacobian (continuous-time f_x nx*nx, f_u nx*nu,
 * row-major) and getStateHessian / getControlHessian / getCrossHessian (fxx[i] nx*nx, fuu[i] nu*nu, fux[i] nu*nx per state row i;
 * the three come together) for one (x, u).  Any output may be NULL.  This is what lets a built-in plant be paired with a user
 * Objective / Constraint in the plug-in solve below (the reference's car-parking test has exactly that shape).  model_params:
 * CDDP_HIP_MAX_MODEL_PARAMS doubles as in cddp_hip_problem.  LTI plants are not served here (x+ = A x + B u is the caller's). */
int cddp_hip_model_eval(int model /* cddp_hip_model */, int integrator /* cddp_hip_integrator */, double dt, const double *model_params,
                        int nx, int nu, const double *x, const double *u, double *x_next, double *fx, double *fu,
                        double *fxx, double *fuu, double *fux);

#define CDDP_HIP_PLUGIN_MAX_CONSTRAINTS 8
typedef struct cddp_hip_plugin {
  /* ABI 5: the entry point copies a cddp_hip_options from a caller pointer, so the caller states which layout it was built against
   * (cddp_hip_create has checked cddp_hip_problem::abi_version since round 2; this struct and the stack handles did not).  Set
   * abi_version = CDDP_HIP_ABI_VERSION and options_bytes = sizeof(cddp_hip_options); a mismatch is refused. */
  int32_t abi_version, options_bytes;
  /* Optional (may be NULL): a word the caller sets non-zero to stop the solve -- e.g. a binding whose callback raised an exception.
   * Checked once per outer iteration and after every batch of callbacks; cddp_hip_plugin_solve then returns -50. */
  const volatile int32_t *abort_flag;
  void *user;
  int32_t nx, nu;
  int32_t n_constraints;
  int32_t constraint_dims[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS];
  void (*discrete_dynamics)(void *user, const double *x, const double *u, double time, double *x_next);
  void (*jacobians)(void *user, const double *x, const double *u, double time, double *fx, double *fu);
  void (*hessians)(void *user, const double *x, const double *u, double time, double *fxx, double *fuu, double *fux);
  double (*running_cost)(void *user, const double *x, const double *u, int index);
  double (*terminal_cost)(void *user, const double *x);
  void (*running_cost_derivatives)(void *user, const double *x, const double *u, int index, double *lx, double *lu, double *lxx, double *luu, double *lux);
  void (*terminal_cost_derivatives)(void *user, const double *x, double *lx, double *lxx);
  void (*constraints)(void *user, const double *x, const double *u, int index, double *g, double *gx, double *gu);
  const double *control_lower, *control_upper;
  /* LogDDP only (may be NULL): second derivatives of the m stacked constraint rows, gxx[r] (nx x nx), guu[r] (nu x nu), gux[r] (nu x nx),
   * Constraint::getHessians; rows of constraints that provide none (or throw std::logic_error in the reference) stay zero.  They
   * enter the relaxed log barrier's Hessian (barrier.hpp:137-213). */
  void (*constraint_hessians)(void *user, const double *x, const double *u, int index, double *gxx, double *guu, double *gux);
} cddp_hip_plugin;
/* ISolverAlgorithm::initialize + solve for `batch` trajectories of a host plug-in problem.  x0: batch*nx; U0: batch*N*nu or NULL
 * (zeros); X0: batch*(N+1)*nx or NULL (x0 replicated).  results: batch records; X (batch*(N+1)*nx), U (batch*N*nu), K
 * (batch*N*nu*nx, feedback gains of the last sweep) may be NULL. */
/* Terminal constraints of a plug-in problem (round 6; CDDP::addTerminalConstraint, cddp_core.cpp:173-179).  Only IPDDP reads them
 * (ipddp_solver.cpp:84-215): objects in std::map (name) order, each either a terminal EQUALITY (TerminalEqualityConstraint: residual
 * h(x_N), rows stacked into H_T / Lambda_T_eq) or a terminal INEQUALITY (TerminalInequalityConstraint: residual g_T(x_N) <= 0 with slack /
 * dual S_T, Y_T).  evaluate returns the residual rows of ALL objects stacked in that order (sum of dims) and, when rx != NULL, their state
 * Jacobian rows (sum of dims x nx, row-major).  At most 8 equality rows in total. */
typedef struct cddp_hip_plugin_terminal {
  int32_t n_terminal;
  int32_t dims[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS];
  int32_t equality[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS];   /* 1 = equality, 0 = inequality */
  void (*evaluate)(void *user, const double *x_terminal, double *r, double *rx);   /* user = cddp_hip_plugin::user */
} cddp_hip_plugin_terminal;
/* cddp_hip_plugin_solve with a terminal set.  The Riccati work stays on the GPU: with terminal-equality rows the reduced LQR of
 * ipddp_solver.cpp:478-639, 1120-1353 (CDDP_HIP_STACKS_IPDDP_TERM_EQ; the host condenses the path constraints into its LQ stacks as :1143-1245
 * does), otherwise the ordinary stack-fed sweeps on the terminal value with the terminal-inequality barrier terms (:1000-1031).
 * options.warm_start: the "warm start with provided trajectory" initialisation (:733-816) from (x0, U0).  terminal_out (optional):
 * per trajectory [S_T (inequality rows) | Y_T | Lambda_T_eq (equality rows)].  terminal == NULL or n_terminal == 0: cddp_hip_plugin_solve. */
int cddp_hip_plugin_solve_terminal(const cddp_hip_plugin *plugin, const cddp_hip_plugin_terminal *terminal, int solver /* cddp_hip_solver */,
                                   int horizon, double dt, const cddp_hip_options *options, int device, int batch, const double *x0,
                                   const double *U0, const double *X0, cddp_hip_result *results, double *X, double *U, double *K,
                                   double *terminal_out);
/* Host threads that run the per-trajectory host work of cddp_hip_plugin_solve's IPDDP / CLDDP loop (derivative fill, forward passes, updates;
 * the reference fans its own line search out with std::async, cddp_solver_base.cpp:264-314).  Default 1: callbacks on the calling thread
 * only.  n > 1 (0 = one per hardware thread): the plug-in's callbacks are called CONCURRENTLY for different trajectories -- for thread-safe
 * plug-ins.  Process-wide; CDDP_HIP_PLUGIN_THREADS is the default when this was never called.  Results do not depend on the count. */
int cddp_hip_plugin_set_host_threads(int n);
/* Time split of this process's LAST IPDDP / CLDDP cddp_hip_plugin_solve (any output may be NULL): wall ms of the whole call, of its GPU
 * sections (stack upload + sweep launch + gain download), the sweeps' kernel ms (hipEvents), batch sweeps launched, host threads used. */
int cddp_hip_plugin_last_stats(double *total_ms, double *gpu_section_ms, double *kernel_ms, int *sweeps, int *threads);
int cddp_hip_plugin_solve(const cddp_hip_plugin *plugin, int solver /* cddp_hip_solver */, int horizon, double dt,
                          const cddp_hip_options *options, int device, int batch, const double *x0, const double *U0, const double *X0,
                          cddp_hip_result *results, double *X, double *U, double *K);

#ifdef __cplusplus
}
#endif
#endif /* CDDP_HIP_H */
