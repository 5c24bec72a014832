// C-ABI of the MI355X-native batched CLDDP / IPDDP solver core (include/cddp_hip.h).
// Host side only: descriptor flattening, device-buffer ownership, kernel sequencing
// (the device-resident per-trajectory state machine lives in kernels.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>

#include "launch.hpp"

// 1 when a[0] > a[1] > ... > a[n - 1] (NaN-free): the ladders cddp_hip_build_alphas makes; DevBuf::ladder_sorted
static int ladder_strictly_decreasing(const double *a, int n) {
  for (int i = 1; i < n; ++i) if (!(a[i] < a[i - 1])) return 0;
  return n > 0 ? 1 : 0;
}

using namespace cddp_dev;

namespace cddp_dev {
// gather records for the RCCL all-gather
static __global__ void k_gather_records(DevBuf d, cddp_hip_gather_record *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  cddp_hip_gather_record r;
  r.final_objective = d.cost[b]; r.iterations = d.iter[b]; r.status = d.status[b];
  out[b] = r;
}

// head of the plan: u_0 and x_1 of every trajectory's CURRENT iterate (batch-major), what a receding-horizon caller reads back
static __global__ void k_gather_plan_head(DevBuf d, int nx, int nu, double *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX, *Uc = d.U + (size_t)cur * d.planeU;
  double *o = out + (size_t)b * (nu + nx);
  for (int i = 0; i < nu; ++i) o[i] = Uc[((((size_t)0 * d.NB + (size_t)(b >> 6)) * nu + i) * 64) + (size_t)(b & 63)];
  for (int i = 0; i < nx; ++i) o[nu + i] = Xc[((((size_t)1 * d.NB + (size_t)(b >> 6)) * nx + i) * 64) + (size_t)(b & 63)];
}

// CDDPOptions::max_cpu_time expired (cddp_solver_base.cpp:77-90): the check sits after ++iter and before the
// backward pass, so every trajectory still running reports the iteration the check fired in.
static __global__ void k_mark_cpu_time(DevBuf d) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B || d.phase[b] == PH_DONE) return;
  d.iter[b] += 1; d.status[b] = CDDP_HIP_STATUS_MAX_CPU_TIME; d.phase[b] = PH_DONE;
}

}  // namespace cddp_dev

int cddp_host_model_eval(int model, int integrator, double dt, const double *params, int nx, int nu, const double *x, const double *u, double *x_next,
                         double *fx, double *fu, double *fxx, double *fuu, double *fux, std::string &err);   // host_models.cpp

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace
extern "C" int cddp_hip_internal_set_error(int code, const char *msg) { g_err = msg ? msg : ""; return code; }   // comm.hip
extern "C" int cddp_hip_internal_allgather(const void *send, void *recv, size_t bytes_per_rank, void *comm, void *stream);
namespace {
#define HIPCHK(expr)                                                                          \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess)                                           \
      return fail(-10, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

const std::vector<KernelSet> &registry() {
  static std::vector<KernelSet> v = [] {
    std::vector<KernelSet> r;
    register_pendulum(r); register_cartpole(r); register_unicycle(r); register_lti(r);
    register_quadrotor(r); register_quad12(r); register_manipulator(r); register_manip7(r); register_terminal(r); register_statebox(r);
    register_vehicles(r);
    return r;
  }();
  return v;
}

}  // namespace

// ---- CU-partitioned streams (round 5) --------------------------------------------------------------------------
// CDDP_HIP_CUMASK = "<spec>[|<spec> ...]", one spec per tile group (the last one repeats), spec = comma-separated "X=lo-hi" with
//   C = the group's main stream (derivative fill, condensation, post, costate, update), F = its rollout stream, W = its serial-sweep stream
// and [lo, hi) a range of CU-mask bits.  On gfx942 / gfx950 the KFD maps mask bit i to XCC i % 8, then shader engine, then CU
// (kfd_mqd_manager.c::mqd_symmetrically_map_cu_mask), so a contiguous bit range is a symmetric slice of every XCD.  A class without
// a range runs on the main stream (F, W) or on an unmasked stream (C).  Kernels of one group on different streams are ordered by
// events (SolveRun::enqueue_iteration, launch.hpp::sweep_hop_in / out); nothing else changes, so results are bitwise those of one
// stream (tests/test_determinism.py).
//   "X=x<digits>" instead of a range: the whole of the named XCDs (bits i with i % 8 among the digits).
struct CuSpec { int lo[3] = {-1, -1, -1}, hi[3] = {-1, -1, -1}; unsigned xcd[3] = {0, 0, 0}; };   // index 0 = C, 1 = F, 2 = W; xcd: bit k = XCD k (0 = a range)
struct CuPlan {
  hipStream_t fwd = nullptr, sweep = nullptr;   // nullptr: the main stream
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // main->fwd, fwd->main (stage 1, stage 2); main->sweep, sweep->main
  SweepHop hop{nullptr, nullptr, nullptr};
};

struct Inner {
  ProblemDev P;
  DevBuf d;
  const KernelSet *ks = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  CuPlan cu;
  std::vector<void *> allocs;
  ProblemDev *dP = nullptr;
  double *d_xref_traj = nullptr;
  unsigned long long *d_launched = nullptr;
  double *d_Xinit = nullptr, *d_Uinit = nullptr;   // initial trajectory kept on the device so solve() is repeatable
  bool have_initial = false;
  bool initialized = false;
  bool has_state = false;        // gains / duals of an earlier initialize() or solve() live on the device (warm start)
  bool initial_dirty = false;    // set_initial() since the last initialize(): the caller supplied a new trajectory
  bool ms_cache_started = false; // MSIPDDP: the per-step factor cache has been cleared once (it lives as long as the handle)
  size_t bytes = 0;
  int timing_detail = CDDP_HIP_TIMING_ROLLOUT;   // which kernel classes cddp_hip_solve brackets with events
  std::vector<hipEvent_t> ev_pool;               // reused across solves (creating an event per mark costs host time)
  hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_poll = nullptr, ev_poll2 = nullptr;
  double *d_head = nullptr, *h_head = nullptr; size_t head_cap = 0;   // cddp_hip_get_plan_head staging (device, pinned host)
  int *h_poll = nullptr;                         // pinned host words of the solve loop's polls, two slots of kPollWords: [0] running count, [1..] alpha histogram
  std::map<unsigned long long, hipGraphExec_t> graphs;   // CDDP_HIP_GRAPH=1: captured iteration windows by (ladder shape, length, last flag)
  std::map<unsigned long long, int> graph_launches;
  int last_t4 = 0;               // layout of the A / B stacks the last derivative fill wrote (launch.hpp::t4_layout at that launch)
  unsigned env_sig = 0;          // kernel-selection environment the cached graphs were captured under (CDDP_HIP_SWEEP / T4 / K4_NA / COOP_H)
};

namespace {

template <class T>
int dalloc(Inner *h, T **p, size_t n) {
  if (n == 0) n = 1;
  void *q = nullptr;
  HIPCHK(hipMalloc(&q, n * sizeof(T)));
  HIPCHK(hipMemsetAsync(q, 0, n * sizeof(T), h->stream));
  h->allocs.push_back(q);
  h->bytes += n * sizeof(T);
  *p = (T *)q;
  return 0;
}

int pool_put(ProblemDev &P, int &top, const double *src, int n) {
  if (top + n > kPool) return -1;
  int off = top;
  for (int i = 0; i < n; ++i) P.pool[off + i] = src[i];
  top += n;
  return off;
}

// plants with device-side Hessian tensors (dev_models.hpp: Model::kHasHess) -- full DDP (options.use_ilqr = 0) needs them
bool model_has_hessians(int model) {
  return model == CDDP_HIP_MODEL_PENDULUM || model == CDDP_HIP_MODEL_CARTPOLE || model == CDDP_HIP_MODEL_UNICYCLE || model == CDDP_HIP_MODEL_LTI ||
         model == CDDP_HIP_MODEL_BICYCLE || model == CDDP_HIP_MODEL_CAR || model == CDDP_HIP_MODEL_MANIPULATOR || model == CDDP_HIP_MODEL_HCW ||
         model == CDDP_HIP_MODEL_QUADROTOR || model == CDDP_HIP_MODEL_QUADROTOR_EULER12 || model == CDDP_HIP_MODEL_MANIPULATOR7;   // round 4: blocked second-order duals
}

// cddp_hip_problem -> ProblemDev (constraints sorted by name as std::map iterates)
int flatten(const cddp_hip_problem *p, ProblemDev &P) {
  std::memset(&P, 0, sizeof(P));
  if (p->abi_version != CDDP_HIP_ABI_VERSION) return fail(-2, "ABI version mismatch: got %d want %d", p->abi_version, CDDP_HIP_ABI_VERSION);
  if (p->nx <= 0 || p->nu <= 0 || p->horizon <= 0 || !(p->dt > 0)) return fail(-2, "bad dimensions nx=%d nu=%d N=%d dt=%g", p->nx, p->nu, p->horizon, p->dt);
  if (!p->Q || !p->R || !p->Qf || !p->x_ref) return fail(-2, "objective matrices (Q, R, Qf, x_ref) must be set before solving");
  // (LogDDP and MSIPDDP, round 4: resident for the built-in plants -- kernels_logddp.hpp, kernels_msipddp.hpp)
  if (p->solver != CDDP_HIP_SOLVER_CLDDP && p->solver != CDDP_HIP_SOLVER_IPDDP && p->solver != CDDP_HIP_SOLVER_LOGDDP && p->solver != CDDP_HIP_SOLVER_MSIPDDP)
    return fail(-2, "UnknownSolver - No solver registered for id %d", p->solver);
  if (p->solver == CDDP_HIP_SOLVER_MSIPDDP && p->n_constraints > 0 && !(p->nu == 1 || p->nx == p->nu))
    return fail(-3, "MSIPDDP with path constraints is only defined for nu = 1 or nx = nu: the reference adds an (nx x nu) product to its (nu x nx) block Q_ux (msipddp_solver.cpp:1398)");
  if (p->solver == CDDP_HIP_SOLVER_LOGDDP && !(p->options.logddp_relaxed_delta > 0.0))
    return fail(-2, "Relaxation delta must be positive.");   // barrier.hpp:49-51
  // the device-resident retry loops (cddp_solver_base.cpp:93-111) terminate because the regularisation grows: a factor <= 1
  // is an endless loop on the reference's host and would be a wedged queue here
  if (!(p->options.reg_update_factor > 1.0) || !(p->options.reg_max_value > 0.0))
    return fail(-2, "regularization.update_factor must be > 1 and max_value > 0 (got %g, %g)", p->options.reg_update_factor, p->options.reg_max_value);
  if (!p->options.use_ilqr && !model_has_hessians(p->model))
    return fail(-3, "use_ilqr=false needs the plant's Hessian tensors, which model id %d does not have", p->model);
  P.solver = p->solver; P.model = p->model; P.integrator = p->integrator;
  P.nx = p->nx; P.nu = p->nu; P.N = p->horizon; P.dt = p->dt; P.opt = p->options;
  P.ls_rule = p->options.enable_parallel ? CDDP_HIP_LS_BEST_MERIT : CDDP_HIP_LS_FIRST_SUCCESS;
  for (int i = 0; i < CDDP_HIP_MAX_MODEL_PARAMS; ++i) P.mp[i] = p->model_params[i];
  if (p->model == CDDP_HIP_MODEL_CAR) P.mp[1] = p->dt;   // the car is a discrete plant: its step uses the timestep (car.cpp:24-60)
  if (p->model == CDDP_HIP_MODEL_LTI) {
    if (!p->lti_A || !p->lti_B) return fail(-2, "LTI model needs lti_A and lti_B");
    if (p->nx * p->nx + p->nx * p->nu + 1 > 32) return fail(-3, "LTI dims too large for the device parameter block");
    for (int i = 0; i < p->nx * p->nx; ++i) P.mp[i] = p->lti_A[i];
    for (int i = 0; i < p->nx * p->nu; ++i) P.mp[p->nx * p->nx + i] = p->lti_B[i];
    P.mp[p->nx * p->nx + p->nx * p->nu] = p->dt;
  }
  int top = 0;
  {
    std::vector<double> q((size_t)p->nx * p->nx), r((size_t)p->nu * p->nu);
    for (size_t i = 0; i < q.size(); ++i) q[i] = p->Q[i] * p->dt;   // objective.cpp:38-39
    for (size_t i = 0; i < r.size(); ++i) r[i] = p->R[i] * p->dt;
    P.off_Qdt = pool_put(P, top, q.data(), (int)q.size());
    P.off_Rdt = pool_put(P, top, r.data(), (int)r.size());
    P.off_Qf = pool_put(P, top, p->Qf, p->nx * p->nx);
    P.off_xref = pool_put(P, top, p->x_ref, p->nx);
    if (P.off_Qdt < 0 || P.off_Rdt < 0 || P.off_Qf < 0 || P.off_xref < 0) return fail(-3, "constant pool overflow");
  }
  P.has_xref_traj = p->x_ref_traj ? 1 : 0;
  if (p->n_constraints > kMaxCons) return fail(-3, "too many path constraints (%d > %d)", p->n_constraints, kMaxCons);
  std::vector<int> order(p->n_constraints);
  for (int i = 0; i < p->n_constraints; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::strcmp(p->constraints[a].name, p->constraints[b].name) < 0; });
  int off = 0;
  P.clddp_box = -1;
  for (int k = 0; k < p->n_constraints; ++k) {
    const cddp_hip_constraint &c = p->constraints[order[k]];
    ConDev &cd = P.cons[k];
    cd.kind = c.kind; cd.dim = c.dim; cd.scale = c.scale; cd.radius = c.radius; cd.offset = off;
    cd.off_lower = cd.off_upper = cd.off_center = cd.off_A = cd.off_b = -1;
    switch (c.kind) {
      case CDDP_HIP_CON_CONTROL_BOX:
      case CDDP_HIP_CON_STATE_BOX:
        if (!c.lower || !c.upper) return fail(-2, "Cannot add null constraint.");
        if (c.dim != (c.kind == CDDP_HIP_CON_CONTROL_BOX ? p->nu : p->nx)) return fail(-2, "box constraint '%s' dimension mismatch", c.name);
        cd.dual_dim = 2 * c.dim;
        cd.off_lower = pool_put(P, top, c.lower, c.dim); cd.off_upper = pool_put(P, top, c.upper, c.dim);
        if (c.kind == CDDP_HIP_CON_CONTROL_BOX && std::strcmp(c.name, "ControlConstraint") == 0) P.clddp_box = k;  // clddp_solver.cpp:85-86
        break;
      case CDDP_HIP_CON_BALL:
        if (!c.center) return fail(-2, "Cannot add null constraint.");
        cd.dual_dim = 1; cd.off_center = pool_put(P, top, c.center, c.dim); break;
      case CDDP_HIP_CON_LINEAR:
        if (!c.A || !c.b) return fail(-2, "Cannot add null constraint.");
        cd.dual_dim = c.dim; cd.off_A = pool_put(P, top, c.A, c.dim * p->nx); cd.off_b = pool_put(P, top, c.b, c.dim); break;
      case CDDP_HIP_CON_SOC: {   // constraint.hpp:626-668: state dimension >= 3, the opening direction stored as a unit vector
        if (!c.center || !c.lower) return fail(-2, "Cannot add null constraint.");
        if (p->nx < 3 || c.dim != 3) return fail(-2, "SecondOrderConeConstraint: State dimension must be at least 3.");
        if (!(c.scale > 0.0)) return fail(-2, "SecondOrderConeConstraint: Regularization epsilon must be positive.");
        const double n2 = c.lower[0] * c.lower[0] + c.lower[1] * c.lower[1] + c.lower[2] * c.lower[2];
        if (n2 == 0.0) return fail(-2, "SecondOrderConeConstraint: Opening direction cannot be zero vector.");
        if (std::fabs(std::sqrt(n2) - 1.0) > 1e-6) return fail(-2, "SecondOrderConeConstraint: pass the opening direction normalised (the reference's constructor normalises it)");
        cd.dual_dim = 1; cd.off_center = pool_put(P, top, c.center, 3); cd.off_lower = pool_put(P, top, c.lower, 3); break;
      }
      case CDDP_HIP_CON_THRUST:
      case CDDP_HIP_CON_MAX_THRUST: {   // constraint.hpp:802-838, 929-953
        if (c.dim != p->nu) return fail(-2, "thrust-magnitude constraint '%s' dimension mismatch", c.name);
        if (!(c.scale > 0.0)) return fail(-2, "ThrustMagnitudeConstraint: epsilon must be positive.");
        const double mn = (c.kind == CDDP_HIP_CON_THRUST && c.lower) ? c.lower[0] : 0.0;
        if (c.kind == CDDP_HIP_CON_THRUST && !c.lower) return fail(-2, "Cannot add null constraint.");
        if (mn < 0.0) return fail(-2, "ThrustMagnitudeConstraint: min_thrust_norm must be non-negative.");
        if (c.radius < mn) return fail(-2, "ThrustMagnitudeConstraint: max_thrust_norm must be greater than or equal to min_thrust_norm.");
        cd.dual_dim = c.kind == CDDP_HIP_CON_THRUST ? 2 : 1; cd.off_lower = pool_put(P, top, &mn, 1); break;
      }
      default: return fail(-2, "unknown constraint kind %d", c.kind);
    }
    if (top > kPool || (cd.off_lower < 0 && cd.off_center < 0 && cd.off_A < 0)) return fail(-3, "constant pool overflow");
    off += cd.dual_dim;
  }
  P.n_cons = p->n_constraints; P.m = off;
  // terminal constraints, in std::map (name) order; inequality rows and equality rows are stacked separately
  if (p->n_terminal > kMaxTerms) return fail(-3, "too many terminal constraints (%d > %d)", p->n_terminal, kMaxTerms);
  if (p->n_terminal > 0 && p->solver != CDDP_HIP_SOLVER_IPDDP) { /* CLDDP ignores the terminal set, as the reference does */ }
  {
    std::vector<int> tord(p->n_terminal);
    for (int i = 0; i < p->n_terminal; ++i) tord[i] = i;
    std::stable_sort(tord.begin(), tord.end(), [&](int a, int b) { return std::strcmp(p->terminal[a].name, p->terminal[b].name) < 0; });
    int offI = 0, offE = 0;
    for (int k = 0; k < p->n_terminal && p->solver == CDDP_HIP_SOLVER_IPDDP; ++k) {
      const cddp_hip_terminal_constraint &c = p->terminal[tord[k]];
      TermDev &td = P.terms[k];
      td.kind = c.kind; td.dim = c.dim; td.off_target = td.off_A = td.off_b = -1;
      if (c.kind == CDDP_HIP_TERM_EQUALITY) {
        if (!c.target) return fail(-2, "Cannot add null constraint.");
        if (c.dim > p->nx) return fail(-2, "TerminalEqualityConstraint: final_state dimension mismatch.");
        td.offset = offE; offE += c.dim; td.off_target = pool_put(P, top, c.target, c.dim);
        if (td.off_target < 0) return fail(-3, "constant pool overflow");
      } else if (c.kind == CDDP_HIP_TERM_INEQUALITY) {
        if (!c.A || !c.b) return fail(-2, "Cannot add null constraint.");
        td.offset = offI; offI += c.dim; td.off_A = pool_put(P, top, c.A, c.dim * p->nx); td.off_b = pool_put(P, top, c.b, c.dim);
        if (td.off_A < 0 || td.off_b < 0) return fail(-3, "constant pool overflow");
      } else {
        return fail(-2, "IPDDP: terminal constraint '%s' has unsupported type. Supported terminal constraints are "
                        "TerminalEqualityConstraint and TerminalInequalityConstraint.", c.name);   // ipddp_solver.cpp:58-67
      }
      P.n_term = k + 1;
    }
    P.mT = offI; P.pT = offE;
    if (P.mT > 8 || P.pT > 16) return fail(-3, "terminal constraint dimension too large (ineq %d > 8 or eq %d > 16)", P.mT, P.pT);
  }
  P.n_alphas = cddp_hip_build_alphas(&p->options, P.alphas, CDDP_HIP_MAX_ALPHAS);
  if (P.n_alphas <= 0) return fail(-2, "empty line-search ladder");
  return 0;
}

int restore_initial(Inner *h) {
  if (!h->have_initial) return fail(-1, "cddp_hip_set_initial must be called before initialize/solve");
  HIPCHK(hipMemcpyAsync(h->d.X, h->d_Xinit, h->d.planeX * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d.U, h->d_Uinit, h->d.planeU * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

int free_all(Inner *h) {
  for (void *q : h->allocs) hipFree(q);
  h->allocs.clear();
  return 0;
}

}  // namespace

extern "C" {

void cddp_hip_default_options(cddp_hip_options *o) {
  std::memset(o, 0, sizeof(*o));
  o->tolerance = 1e-5; o->acceptable_tolerance = 1e-6; o->max_iterations = 1; o->use_ilqr = 1;
  o->termination_scaling_max_factor = 100.0;
  o->ls_max_iterations = 11; o->ls_initial_step_size = 1.0; o->ls_min_step_size = 1e-8; o->ls_step_reduction_factor = 0.5;
  o->reg_initial_value = 1e-6; o->reg_update_factor = 10.0; o->reg_max_value = 1e7; o->reg_min_value = 1e-10;
  o->boxqp_max_iterations = 100; o->boxqp_min_gradient_norm = 1e-8; o->boxqp_min_relative_improvement = 1e-8;
  o->boxqp_step_decrease_factor = 0.6; o->boxqp_min_step_size = 1e-22; o->boxqp_armijo_constant = 0.1;
  o->filter_merit_acceptance_threshold = 1e-6; o->filter_violation_acceptance_threshold = 1e-6;
  o->filter_max_violation_threshold = 1e4; o->filter_min_violation_for_armijo_check = 1e-7; o->filter_armijo_constant = 1e-4;
  o->ipddp_dual_var_init_scale = 0.1; o->ipddp_slack_var_init_scale = 1e-2; o->ipddp_barrier_tol_mult = 0.1;
  o->ipddp_barrier_update_dual_weight = 0.01; o->ipddp_mu_kappa_epsilon = 10.0; o->ipddp_max_filter_size = 5;
  o->ipddp_theta_0_floor = 1.0; o->ipddp_warmstart_s_min = 1e-4; o->ipddp_warmstart_y_min = 1e-4;
  o->ipddp_warmstart_interior_factor = 1.1; o->ipddp_jacobian_regularization_value = 1e-8;
  o->ipddp_jacobian_regularization_exponent = 0.25;
  o->barrier_mu_initial = 1.0; o->barrier_mu_min_value = 1e-10; o->barrier_mu_update_factor = 0.5;
  o->barrier_mu_update_power = 1.2; o->barrier_min_fraction_to_boundary = 0.99; o->barrier_strategy = CDDP_HIP_BARRIER_ADAPTIVE;
  o->max_cpu_time = 0.0;
  o->logddp_mu_initial = 1.0; o->logddp_mu_min_value = 1e-10; o->logddp_mu_update_factor = 0.5; o->logddp_relaxed_delta = 1e-10;
  o->msipddp_costate_var_init_scale = 1e-6; o->msipddp_segment_length = 5; o->msipddp_rollout_type = 0; o->msipddp_use_controlled_rollout = 0; o->_pad4 = 0;
}

int cddp_hip_abi_version(void) { return CDDP_HIP_ABI_VERSION; }

int cddp_hip_model_eval(int model, int integrator, double dt, const double *model_params, int nx, int nu, const double *x, const double *u,
                        double *x_next, double *fx, double *fu, double *fxx, double *fuu, double *fux) {
  if (!model_params || !x || !u) return fail(-2, "cddp_hip_model_eval: null argument");
  std::string err;
  const int rc = cddp_host_model_eval(model, integrator, dt, model_params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
  return rc == 0 ? 0 : fail(rc, "cddp_hip_model_eval: %s", err.c_str());
}
const char *cddp_hip_last_error(void) { return g_err.c_str(); }

int cddp_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *cddp_hip_status_string(int status) {
  switch (status) {
    case CDDP_HIP_STATUS_RUNNING: return "Running";
    case CDDP_HIP_STATUS_OPTIMAL: return "OptimalSolutionFound";
    case CDDP_HIP_STATUS_ACCEPTABLE: return "AcceptableSolutionFound";
    case CDDP_HIP_STATUS_MAX_ITERATIONS: return "MaxIterationsReached";
    case CDDP_HIP_STATUS_REG_LIMIT: return "RegularizationLimitReached_NotConverged";
    case CDDP_HIP_STATUS_MAX_CPU_TIME: return "MaxCpuTimeReached";
    case CDDP_HIP_STATUS_REG_LIMIT_CONVERGED: return "RegularizationLimitReached_Converged";
  }
  return "Unknown";
}

// detail::buildLineSearchAlphas (reference cddp_context_utils.cpp:37-57)
int cddp_hip_build_alphas(const cddp_hip_options *opt, double *alphas, int cap) {
  int n = 0;
  double cur = opt->ls_initial_step_size;
  for (int i = 0; i < opt->ls_max_iterations; ++i) {
    if (n < cap) alphas[n] = cur;
    ++n;
    cur *= opt->ls_step_reduction_factor;
    if (cur < opt->ls_min_step_size && i < opt->ls_max_iterations - 1) {
      if (n < cap) alphas[n] = opt->ls_min_step_size;
      ++n;
      break;
    }
  }
  if (n == 0) { if (cap > 0) alphas[0] = opt->ls_initial_step_size; n = 1; }
  return n > cap ? cap : n;
}


// one "X=lo-hi,..." spec
static bool parse_cu_spec(const std::string &t, CuSpec *o) {
  size_t i = 0;
  while (i < t.size()) {
    size_t j = t.find(',', i); if (j == std::string::npos) j = t.size();
    const std::string f = t.substr(i, j - i);
    i = j + 1;
    if (f.empty()) continue;
    int lo = 0, hi = 0; char c = 0;
    unsigned xm = 0;
    if (f.size() >= 4 && f[1] == '=' && f[2] == 'x') {
      c = f[0];
      for (size_t q = 3; q < f.size(); ++q) { if (f[q] < '0' || f[q] > '7') return false; xm |= 1u << (f[q] - '0'); }
      lo = 0; hi = 1024;
    } else if (std::sscanf(f.c_str(), " %c=%d-%d", &c, &lo, &hi) != 3 || lo < 0 || hi <= lo || hi > 1024) return false;
    const int k = (c == 'C') ? 0 : (c == 'F') ? 1 : (c == 'W') ? 2 : -1;
    if (k < 0) return false;
    o->lo[k] = lo; o->hi[k] = hi; o->xcd[k] = xm;
  }
  return true;
}
// the spec of group `gi` from CDDP_HIP_CUMASK (false: no partition requested / malformed -> plain streams)
static bool cu_spec_for_group(int gi, CuSpec *o) {
  const char *e = std::getenv("CDDP_HIP_CUMASK");
  if (!e || !e[0]) return false;
  std::vector<std::string> parts;
  { std::string t(e); size_t i = 0; for (;;) { size_t j = t.find('|', i); parts.push_back(t.substr(i, j == std::string::npos ? j : j - i)); if (j == std::string::npos) break; i = j + 1; } }
  const std::string &t = parts[std::min((size_t)gi, parts.size() - 1)];
  CuSpec sp;
  if (!parse_cu_spec(t, &sp)) { std::fprintf(stderr, "[cddp_hip] CDDP_HIP_CUMASK: cannot parse '%s' -- ignored\n", t.c_str()); return false; }
  *o = sp;
  return true;
}
static int device_cu_count(int device) {
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
  return ncu;
}
// A mask that selects no CU of THIS device (a spec written for 256 CUs on a 128-CU partition), or a runtime that refuses the masked
// stream, must not fail the handle: the group then runs on a plain stream (same results -- the partition is placement only).
static hipError_t make_stream(hipStream_t *s, int device, int lo, int hi, unsigned xcd = 0) {
  if (lo < 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  const int ncu = device_cu_count(device);
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask((size_t)words, 0u);
  int bits = 0;
  for (int b = lo; b < hi && b < ncu; ++b) if (!xcd || ((xcd >> (b % 8)) & 1u)) { mask[(size_t)b >> 5] |= 1u << (b & 31); ++bits; }
  if (bits == 0) {
    std::fprintf(stderr, "[cddp_hip] CU mask [%d, %d) selects none of the device's %d CUs: plain stream instead\n", lo, hi, ncu);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  }
  const hipError_t e = hipExtStreamCreateWithCUMask(s, (uint32_t)words, mask.data());
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  std::fprintf(stderr, "[cddp_hip] hipExtStreamCreateWithCUMask([%d, %d) of %d CUs): %s -- plain stream instead\n", lo, hi, ncu, hipGetErrorString(e));
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
static int in_destroy(Inner *h);

static int in_create(const cddp_hip_problem *problem, int batch, int device, Inner **out, const CuSpec *cu = nullptr) {
  if (!problem || !out) return fail(-1, "null argument");
  if (batch <= 0) return fail(-1, "batch must be positive");
  int ndev = cddp_hip_device_count();
  if (ndev <= 0) return fail(-20, "no HIP device available: the cddp_hip solver core has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(-1, "device %d out of range (%d devices)", device, ndev);
  Inner *h = new Inner();
  int rc = flatten(problem, h->P);
  if (rc) { delete h; return rc; }
  for (const KernelSet &k : registry()) if (k.matches(h->P)) { h->ks = &k; break; }
  if (!h->ks) {
    const int m = h->P.m, nc = h->P.n_cons;
    delete h;
    return fail(-4, "no kernel instantiation for model=%d nx=%d nu=%d with this constraint layout (m=%d, %d constraints)",
                problem->model, problem->nx, problem->nu, m, nc);
  }
  if (h->P.solver == CDDP_HIP_SOLVER_LOGDDP && !h->ks->has_logddp) {
    delete h;
    return fail(-4, "LogDDP has no resident kernels for this layout (model=%d, nx=%d): use cddp_hip_plugin_solve", problem->model, problem->nx);
  }
  if (h->P.solver == CDDP_HIP_SOLVER_MSIPDDP && !h->ks->has_msipddp) {
    delete h;
    return fail(-4, "MSIPDDP has no resident kernels for this layout (model=%d, nx=%d; terminal sets and nx > 8 are served by cddp_hip_plugin_solve)", problem->model, problem->nx);
  }
  if (h->P.solver == CDDP_HIP_SOLVER_MSIPDDP && !h->P.opt.use_ilqr) {
    bool curved = false;
    for (int c = 0; c < h->P.n_cons; ++c) { const int kd = h->P.cons[c].kind; if (kd != CDDP_HIP_CON_CONTROL_BOX && kd != CDDP_HIP_CON_STATE_BOX && kd != CDDP_HIP_CON_LINEAR) curved = true; }
    if (curved || !h->ks->logddp_ddp) {
      delete h;
      return fail(-3, "MSIPDDP with use_ilqr=false is resident for plants with explicit Hessian tensors and constraint rows without curvature: use cddp_hip_plugin_solve");
    }
  }
  if (h->P.solver == CDDP_HIP_SOLVER_LOGDDP && !h->P.opt.use_ilqr && !h->ks->logddp_ddp) {
    delete h;
    return fail(-3, "LogDDP with use_ilqr=false needs the plant's explicit Hessian tensors, which model id %d keeps only in the blocked dual form: use cddp_hip_plugin_solve", problem->model);
  }
  h->device = device;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { delete h; return fail(-10, "hipSetDevice: %s", hipGetErrorString(e)); }
  e = make_stream(&h->stream, device, cu ? cu->lo[0] : -1, cu ? cu->hi[0] : -1, cu ? cu->xcd[0] : 0);
  if (e != hipSuccess) { delete h; return fail(-10, "hipStreamCreate: %s", hipGetErrorString(e)); }
  h->own_stream = true;
  if (cu && (cu->lo[1] >= 0 || cu->lo[2] >= 0)) {
    if (cu->lo[1] >= 0) e = make_stream(&h->cu.fwd, device, cu->lo[1], cu->hi[1], cu->xcd[1]);
    if (e == hipSuccess && cu->lo[2] >= 0) e = make_stream(&h->cu.sweep, device, cu->lo[2], cu->hi[2], cu->xcd[2]);
    for (int k = 0; k < 6 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&h->cu.ev[k], hipEventDisableTiming);
    if (e != hipSuccess) { const int rc_ = fail(-10, "CU-masked stream: %s", hipGetErrorString(e)); in_destroy(h); return rc_; }   // (streams / events created so far go with it)
    h->cu.hop = SweepHop{h->cu.sweep, h->cu.ev[4], h->cu.ev[5]};
  }

  const ProblemDev &P = h->P;
  DevBuf &d = h->d;
  std::memset(&d, 0, sizeof(d));
  const int B = batch, Bp = (batch + 63) / 64 * 64, N = P.N, nx = P.nx, nu = P.nu, m = P.m;
  d.B = B; d.Bp = Bp; d.NB = Bp / 64; d.N = N; d.n_alphas = P.n_alphas; d.n_slots = P.n_alphas + 1;
  d.ladder_sorted = ladder_strictly_decreasing(P.alphas, P.n_alphas);
  d.ddp = P.opt.use_ilqr ? 0 : 1;
  { const char *e = std::getenv("CDDP_HIP_TEST_FAIL_COSTATE"); d.fail_costate_mask = e ? std::atoi(e) : 0;   // test hook, DevBuf::fail_costate_mask
    if (d.fail_costate_mask) std::fprintf(stderr, "[cddp_hip] TEST HOOK active: CDDP_HIP_TEST_FAIL_COSTATE=%d discards line-search trials -- results are not the solver's\n", d.fail_costate_mask); }
  { const char *e = std::getenv("CDDP_HIP_XCD_MAP"); d.xcd_map = (e && e[0] == '0') ? 0 : 1; }
  d.t4 = 0;   // set per launch by launch.hpp::derivs / backward (kernels.hpp::GT)
  d.hist_batch = P.opt.return_iteration_info ? std::min(B, 64) : 0;
  d.hist_cap = std::max(P.opt.max_iterations, 0) + 1;
  d.planeX = (size_t)(N + 1) * nx * Bp; d.planeU = (size_t)N * nu * Bp; d.planeM = (size_t)N * (m > 0 ? m : 0) * Bp;
  const bool ip = (P.solver == CDDP_HIP_SOLVER_IPDDP);
  d.lg = (P.solver == CDDP_HIP_SOLVER_LOGDDP) ? 1 : 0;
  const bool ms = (P.solver == CDDP_HIP_SOLVER_MSIPDDP);
  d.ms = ms ? 1 : 0;
  d.filt_cap = ms ? std::max(P.opt.max_iterations, 0) + 2 : kFilterCap;   // (a negative max_iterations is a legal no-op solve, SolveRun::begin)
#define DA(ptr, n) do { int rc_ = dalloc(h, &(ptr), (size_t)(n)); if (rc_) { in_destroy(h); return rc_; } } while (0)
  DA(d.X, d.planeX * d.n_slots); DA(d.U, d.planeU * d.n_slots);
  if (ip) { DA(d.S, d.planeM * d.n_slots); DA(d.Y, d.planeM * d.n_slots); DA(d.G, d.planeM * d.n_slots); DA(d.Lam, d.planeX * d.n_slots); }
  if (ms) {   // kernels_msipddp.hpp: slack / dual / constraint / costate / dynamics-value planes per slot, costate gains, factor cache
    if (m > 0) { DA(d.S, d.planeM * d.n_slots); DA(d.Y, d.planeM * d.n_slots); DA(d.G, d.planeM * d.n_slots);
                 DA(d.ks, (size_t)N * m * Bp); DA(d.ky, (size_t)N * m * Bp); DA(d.Ks, (size_t)N * m * nx * Bp); DA(d.Ky, (size_t)N * m * nx * Bp); }
    DA(d.Lam, d.planeX * d.n_slots); DA(d.F, d.planeX * d.n_slots);
    DA(d.kl, (size_t)N * nx * Bp);
    if (m == 0) DA(d.fac, (size_t)N * (nu * nu + nu + 1) * Bp);
    if (m > 0) { DA(d.ms_res, (size_t)3 * Bp); DA(d.ev, (size_t)d.n_slots * N * P.n_cons * Bp); }   // parked log-barrier sums per slot (k_update_msipddp replays them)
    if (m > 0 && h->ks->ms_cst_size > 0) DA(d.cst, (size_t)N * h->ks->ms_cst_size * Bp);   // V-independent terms of the split sweep (k_ms_condense)
  }
  DA(d.A, (size_t)N * nx * nx * Bp); DA(d.Bm, (size_t)N * nx * nu * Bp);
  DA(d.K, (size_t)N * nu * nx * Bp); DA(d.k, (size_t)N * nu * Bp);
  if (ip && nx > 8) DA(d.Kt, (size_t)N * (nu * nx + nu) * Bp);   // G = 16 sweeps (launch.hpp::t4_layout)
  DA(d.Vx, (size_t)(N + 1) * nx * Bp); DA(d.Vxx, (size_t)(N + 1) * nx * nx * Bp);
  if (ip && h->ks->cst_size > 0) { DA(d.cst, (size_t)N * h->ks->cst_size * Bp); DA(d.dX, (size_t)N * nx * Bp); DA(d.ys, (size_t)N * (m > 0 ? m : 1) * Bp); }
  if (ip && m > 0) { DA(d.ks, (size_t)N * m * Bp); DA(d.ky, (size_t)N * m * Bp); DA(d.Ks, (size_t)N * m * nx * Bp); DA(d.Ky, (size_t)N * m * nx * Bp); }
  double **scal[] = {&d.cost, &d.merit, &d.inf_pr, &d.inf_du, &d.inf_comp, &d.step_norm, &d.alpha_pr, &d.alpha_du, &d.reg, &d.mu,
                     &d.dV0, &d.dV1, &d.phi, &d.theta, &d.filter_theta, &d.apr_max, &d.adu_max};
  for (double **sp : scal) DA(*sp, Bp);
  DA(d.filt, (size_t)2 * d.filt_cap * Bp);
  int **iscal[] = {&d.filt_n, &d.iter, &d.status, &d.phase, &d.cur, &d.n_bwd, &d.n_fwd, &d.bwd_ok};
  for (int **sp : iscal) DA(*sp, Bp);
  double **tr[] = {&d.t_cost, &d.t_merit, &d.t_theta, &d.t_inf_pr, &d.t_inf_comp, &d.t_apr, &d.t_adu, &d.t_ysmin, &d.t_ysmax};
  for (double **sp : tr) DA(*sp, (size_t)d.n_alphas * Bp);
  DA(d.sink, kSinkDoubles);
  DA(d.t_success, (size_t)d.n_alphas * Bp);
  DA(d.t_steps, (size_t)d.n_alphas * Bp); DA(d.n_fwd_steps, Bp); DA(d.cand, Bp);
  if (ip && P.n_cons > 0) DA(d.ev, (size_t)d.n_alphas * N * 2 * P.n_cons * Bp);
  if (d.lg && P.n_cons > 0) DA(d.ev, (size_t)d.n_slots * N * P.n_cons * Bp);   // parked barrier sums per trial slot (kernels_logddp.hpp)
  DA(d.hist, (size_t)std::max(1, d.hist_batch) * d.hist_cap * kHistCols);
  DA(d.hist_n, std::max(1, d.hist_batch));
  if (ip && P.n_term > 0) {
    DA(d.ST, (size_t)8 * Bp); DA(d.YT, (size_t)8 * Bp); DA(d.GT, (size_t)8 * Bp); DA(d.dST, (size_t)8 * Bp); DA(d.dYT, (size_t)8 * Bp);
    DA(d.LamT, (size_t)16 * Bp); DA(d.dLamT, (size_t)16 * Bp);
    DA(d.STt, (size_t)d.n_alphas * 8 * Bp); DA(d.YTt, (size_t)d.n_alphas * 8 * Bp); DA(d.GTt, (size_t)d.n_alphas * 8 * Bp);
    DA(d.LamTt, (size_t)d.n_alphas * 16 * Bp);
    if (P.pT > 0) {
      // cooperative reduced-LQR sweep (kernels_te.hpp): one lane per gradient variant, variant stride 16
      const bool te_coop = h->ks->te_rec_size > 0 && P.mT == 0 && P.pT + 1 <= h->ks->te_group && P.pT <= P.nx;   // TeCfg::PMAX
      const size_t nv = te_coop ? 16 : (size_t)(P.pT + 1);
      DA(d.te_k, nv * N * nu * Bp); DA(d.te_p, nv * (N + 1) * nx * Bp);
      if (te_coop) {
        DA(d.te_cst, (size_t)N * h->ks->te_rec_size * Bp); DA(d.te_cnt, Bp);
        if (!d.dX) DA(d.dX, (size_t)N * nx * Bp);
        if (!d.ys) DA(d.ys, (size_t)N * (m > 0 ? m : 1) * Bp);
      }
    }
  }
  DA(d.n_active, 1);
  DA(d.win_hist, CDDP_HIP_MAX_ALPHAS + 1);
  DA(h->d_launched, 1);
  DA(h->dP, 1);
  DA(h->d_Xinit, d.planeX); DA(h->d_Uinit, d.planeU);
  if (problem->x_ref_traj) {
    DA(h->d_xref_traj, (size_t)(N + 1) * nx);
    hipMemcpyAsync(h->d_xref_traj, problem->x_ref_traj, sizeof(double) * (N + 1) * nx, hipMemcpyHostToDevice, h->stream);
    d.xref_traj = h->d_xref_traj;
  }
#undef DA
  hipMemcpyAsync(h->dP, &h->P, sizeof(ProblemDev), hipMemcpyHostToDevice, h->stream);
  d.P = h->dP;
  d.launched = h->d_launched;
  e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { const int rc_ = fail(-10, "device initialisation failed: %s", hipGetErrorString(e)); in_destroy(h); return rc_; }
  *out = h;
  return 0;
}

static int in_destroy(Inner *h) {
  if (h) { for (auto &kv : h->graphs) hipGraphExecDestroy(kv.second); h->graphs.clear(); }
  if (!h) return 0;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->cu.fwd) hipStreamSynchronize(h->cu.fwd);
  if (h->cu.sweep) hipStreamSynchronize(h->cu.sweep);
  free_all(h);
  for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
  if (h->ev_begin) { hipEventDestroy(h->ev_begin); hipEventDestroy(h->ev_end); }
  if (h->ev_poll) hipEventDestroy(h->ev_poll);
  if (h->ev_poll2) hipEventDestroy(h->ev_poll2);
  if (h->h_poll) hipHostFree(h->h_poll);
  if (h->d_head) hipFree(h->d_head);
  if (h->h_head) hipHostFree(h->h_head);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  if (h->cu.fwd) hipStreamDestroy(h->cu.fwd);
  if (h->cu.sweep) hipStreamDestroy(h->cu.sweep);
  for (hipEvent_t e : h->cu.ev) if (e) hipEventDestroy(e);
  delete h;
  return 0;
}

static int in_set_timing_detail(Inner *h, int detail) {
  if (!h) return fail(-1, "null handle");
  if (detail != CDDP_HIP_TIMING_ROLLOUT && detail != CDDP_HIP_TIMING_ALL && detail != CDDP_HIP_TIMING_SWEEP)
    return fail(-2, "unknown timing detail %d", detail);
  h->timing_detail = detail;
  return 0;
}

static int in_set_stream(Inner *h, void *hip_stream) {
  if (!h) return fail(-1, "null handle");
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  h->stream = (hipStream_t)hip_stream;
  h->own_stream = false;
  return 0;
}

static int in_dual_dim(Inner *h) { return h ? h->P.m : -1; }
static int in_batch(Inner *h) { return h ? h->d.B : -1; }

// host batch-major [b][t][e]  <->  device batch-minor [t][e][b]
// wave-tiled stack index (dev_types.hpp): element e of step t of trajectory b
static inline size_t tix(int t, int E, int e, int b, int Bp) { return ((((size_t)t * (Bp / 64) + (size_t)(b >> 6)) * E + e) * 64) + (size_t)(b & 63); }
static void to_soa(const double *src, double *dst, int B, int Bp, int T, int E) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int e = 0; e < E; ++e) dst[tix(t, E, e, b, Bp)] = src[((size_t)b * T + t) * E + e];
}
static void from_soa(const double *src, double *dst, int B, int Bp, int T, int E) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int e = 0; e < E; ++e) dst[((size_t)b * T + t) * E + e] = src[tix(t, E, e, b, Bp)];
}

// the same, from the sub-tile-minor layout of the G = 16 sweeps' input stacks (kernels.hpp::GT)
static void from_t4(const double *src, double *dst, int B, int Bp, int T, int E) {
  const size_t NB16 = (size_t)(Bp / 64) * 16;
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int e = 0; e < E; ++e) dst[((size_t)b * T + t) * E + e] = src[((((size_t)t * NB16 + (size_t)(b >> 2)) * (size_t)E + (size_t)e) * 4) + (size_t)(b & 3)];
}

static int in_set_initial(Inner *h, const double *x0, const double *U0, const double *X0) {
  if (!h || !x0) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(h->device));
  const DevBuf &d = h->d;
  const int B = d.B, Bp = d.Bp, N = d.N, nx = h->P.nx, nu = h->P.nu;
  std::vector<double> hx((size_t)(N + 1) * nx * Bp, 0.0), hu((size_t)N * nu * Bp, 0.0);
  if (X0) to_soa(X0, hx.data(), B, Bp, N + 1, nx);
  else
    for (int b = 0; b < B; ++b)
      for (int t = 0; t <= N; ++t)
        for (int e = 0; e < nx; ++e) hx[tix(t, nx, e, b, Bp)] = x0[(size_t)b * nx + e];
  for (int b = 0; b < B; ++b)
    for (int e = 0; e < nx; ++e) hx[tix(0, nx, e, b, Bp)] = x0[(size_t)b * nx + e];   // X_[0] = initial_state (cddp_core.cpp:294)
  if (U0) to_soa(U0, hu.data(), B, Bp, N, nu);
  HIPCHK(hipMemcpyAsync(h->d_Xinit, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_Uinit, hu.data(), hu.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_initial = true;
  h->initialized = false;
  h->initial_dirty = true;
  return 0;
}

// ISolverAlgorithm::initialize on this handle (the handle IS the solver object).  Cold: the initial trajectory is
// restored and everything re-initialised.  options.warm_start: "provided trajectory" on a fresh handle, "existing
// solver state" afterwards -- the live slack / dual / costate rows are staged into slot 0 (and X, U too unless the
// caller supplied a new trajectory since), see k_init.
static int run_initialize(Inner *h) {
  const bool ip = (h->P.solver == CDDP_HIP_SOLVER_IPDDP) || (h->P.solver == CDDP_HIP_SOLVER_MSIPDDP);   // slack / dual / costate rows staged for a warm re-solve
  int mode = kInitCold;
  if (h->P.opt.warm_start) mode = h->has_state ? kInitWarmExisting : kInitWarmProvided;
  if (mode == kInitWarmExisting) {
    h->ks->stage(h->d, h->initial_dirty ? 0 : 1, ip ? 1 : 0, h->stream);
    if (h->initial_dirty) { int rc = restore_initial(h); if (rc) return rc; }
  } else {
    int rc = restore_initial(h); if (rc) return rc;
  }
  if (h->d.ms) {   // the reference's factor cache belongs to the solver object, not to one solve: invalid only before the handle's first initialize
    DevBuf dd = h->d;
    dd.ms_fresh = h->ms_cache_started ? 0 : 1;
    h->ks->init(dd, mode, h->stream);
    h->ms_cache_started = true;
  } else
  h->ks->init(h->d, mode, h->stream);
  h->has_state = true;
  h->initial_dirty = false;
  h->initialized = true;
  return 0;
}

static int in_initialize(Inner *h) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  { int rc = run_initialize(h); if (rc) return rc; }
  HIPCHK(hipGetLastError());
  return 0;
}

static int in_set_options(Inner *h, const cddp_hip_options *opt) {
  if (!h || !opt) return fail(-1, "null argument");
  if (!(opt->reg_update_factor > 1.0) || !(opt->reg_max_value > 0.0))
    return fail(-2, "regularization.update_factor must be > 1 and max_value > 0 (got %g, %g)", opt->reg_update_factor, opt->reg_max_value);
  if (!opt->use_ilqr && !model_has_hessians(h->P.model))
    return fail(-3, "use_ilqr=false needs the plant's Hessian tensors, which model id %d does not have", h->P.model);
  HIPCHK(hipSetDevice(h->device));
  double al[CDDP_HIP_MAX_ALPHAS];
  const int na = cddp_hip_build_alphas(opt, al, CDDP_HIP_MAX_ALPHAS);
  if (na != h->P.n_alphas) return fail(-3, "the line-search ladder size is fixed at create time (%d alphas, new options give %d)", h->P.n_alphas, na);
  if (h->d.ms && opt->max_iterations + 2 > h->d.filt_cap)
    return fail(-3, "max_iterations cannot grow beyond %d on an MSIPDDP handle (filter capacity fixed by cddp_hip_create)", h->d.filt_cap - 2);
  if (opt->max_iterations + 1 > h->d.hist_cap && h->P.opt.return_iteration_info)
    return fail(-3, "max_iterations cannot grow beyond %d on a handle created with return_iteration_info (history capacity)", h->d.hist_cap - 1);
  if ((opt->return_iteration_info != 0) != (h->P.opt.return_iteration_info != 0))
    return fail(-3, "return_iteration_info is fixed at create time (the history buffers are sized by cddp_hip_create)");
  h->P.opt = *opt;
  h->d.ddp = opt->use_ilqr ? 0 : 1;
  for (int i = 0; i < na; ++i) h->P.alphas[i] = al[i];
  h->d.ladder_sorted = ladder_strictly_decreasing(al, na);
  h->P.ls_rule = opt->enable_parallel ? CDDP_HIP_LS_BEST_MERIT : CDDP_HIP_LS_FIRST_SUCCESS;
  HIPCHK(hipMemcpyAsync(h->dP, &h->P, sizeof(ProblemDev), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

static int in_set_initial_state(Inner *h, const double *x0) {
  if (!h || !x0) return fail(-1, "null argument");
  if (!h->have_initial) return fail(-1, "cddp_hip_set_initial must be called once before cddp_hip_set_initial_state");
  HIPCHK(hipSetDevice(h->device));
  const DevBuf &d = h->d;
  const int B = d.B, Bp = d.Bp, nx = h->P.nx;
  // the t = 0 record of every tile is the first NB * nx * 64 doubles of an X plane
  std::vector<double> row((size_t)d.NB * nx * 64, 0.0);
  for (int b = 0; b < B; ++b)
    for (int e = 0; e < nx; ++e) row[tix(0, nx, e, b, Bp)] = x0[(size_t)b * nx + e];
  for (int sl = 0; sl < d.n_slots; ++sl)
    HIPCHK(hipMemcpyAsync(d.X + (size_t)sl * d.planeX, row.data(), row.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_Xinit, row.data(), row.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->initialized = false;
  return 0;
}

static int in_set_duals(Inner *h, const double *S, const double *Y) {
  if (!h) return fail(-1, "null handle");
  if (h->P.solver != CDDP_HIP_SOLVER_IPDDP || h->P.m <= 0) return fail(-1, "cddp_hip_set_duals: the problem has no path duals");
  if (!h->has_state) return fail(-1, "cddp_hip_set_duals needs an initialised handle (call cddp_hip_initialize or cddp_hip_solve first)");
  HIPCHK(hipSetDevice(h->device));
  const DevBuf &d = h->d;
  std::vector<double> buf(d.planeM);
  const double *src[2] = {S, Y};
  double *dst[2] = {d.S, d.Y};
  for (int k = 0; k < 2; ++k) {
    if (!src[k]) continue;
    std::fill(buf.begin(), buf.end(), 0.0);
    to_soa(src[k], buf.data(), d.B, d.Bp, d.N, h->P.m);
    for (int sl = 0; sl < d.n_slots; ++sl)   // the live slot differs per trajectory: every slot gets the rows
      HIPCHK(hipMemcpyAsync(dst[k] + (size_t)sl * d.planeM, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

static int in_set_barrier_state(Inner *h, const double *mu, const double *reg) {
  if (!h) return fail(-1, "null handle");
  if (!h->has_state) return fail(-1, "cddp_hip_set_barrier_state needs an initialised handle (call cddp_hip_initialize or cddp_hip_solve first)");
  HIPCHK(hipSetDevice(h->device));
  const DevBuf &d = h->d;
  if (mu) {
    if (h->P.solver != CDDP_HIP_SOLVER_IPDDP) return fail(-1, "cddp_hip_set_barrier_state: mu is an IPDDP quantity");
    for (int b = 0; b < d.B; ++b) if (!(mu[b] > 0.0)) return fail(-2, "barrier parameter of trajectory %d must be positive (got %g)", b, mu[b]);
    HIPCHK(hipMemcpyAsync(d.mu, mu, sizeof(double) * d.B, hipMemcpyHostToDevice, h->stream));
  }
  if (reg) {
    for (int b = 0; b < d.B; ++b) if (!(reg[b] >= 0.0)) return fail(-2, "regularisation of trajectory %d must be non-negative (got %g)", b, reg[b]);
    HIPCHK(hipMemcpyAsync(d.reg, reg, sizeof(double) * d.B, hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

static int in_set_terminal(Inner *h, const double *S_T, const double *Y_T, const double *Lambda_T) {
  if (!h) return fail(-1, "null handle");
  if (h->P.n_term <= 0) return fail(-1, "cddp_hip_set_terminal: the problem has no terminal constraints");
  if (!h->has_state) return fail(-1, "cddp_hip_set_terminal needs an initialised handle");
  HIPCHK(hipSetDevice(h->device));
  const DevBuf &d = h->d;
  struct { const double *host; double *dev; int n; } items[] = {{S_T, d.ST, h->P.mT}, {Y_T, d.YT, h->P.mT}, {Lambda_T, d.LamT, h->P.pT}};
  for (auto &it : items) {
    if (!it.host || it.n <= 0) continue;
    std::vector<double> buf((size_t)it.n * d.Bp, 0.0);
    for (int b = 0; b < d.B; ++b) for (int i = 0; i < it.n; ++i) buf[(size_t)i * d.Bp + b] = it.host[(size_t)b * it.n + i];
    HIPCHK(hipMemcpyAsync(it.dev, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

static int in_backward(Inner *h, int32_t *ok) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialized) { int rc = in_initialize(h); if (rc) return rc; }
  h->last_t4 = h->ks->t4_layout(h->d);
  h->ks->derivs(h->d, 1, h->stream);
  h->ks->backward(h->d, h->P.solver, 1, 0, h->stream);
  HIPCHK(hipGetLastError());
  if (ok) HIPCHK(hipMemcpyAsync(ok, h->d.bwd_ok, sizeof(int) * h->d.B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

static int in_forward(Inner *h, const double *alphas, int n_alphas, cddp_hip_trial *trials) {
  if (!h || !alphas || !trials) return fail(-1, "null argument");
  if (n_alphas <= 0 || n_alphas > h->d.n_alphas) return fail(-1, "n_alphas must be in [1, %d] (the handle's ladder size)", h->d.n_alphas);
  HIPCHK(hipSetDevice(h->device));
  // temporarily install the caller's alphas in the device problem block
  ProblemDev tmp = h->P;
  for (int i = 0; i < n_alphas; ++i) tmp.alphas[i] = alphas[i];
  HIPCHK(hipMemcpyAsync(h->dP, &tmp, sizeof(ProblemDev), hipMemcpyHostToDevice, h->stream));
  DevBuf dcall = h->d;   // (the caller's ladder may be any set of step sizes)
  dcall.ladder_sorted = ladder_strictly_decreasing(tmp.alphas, h->d.n_alphas);
  h->ks->forward(dcall, h->P.solver, 0, n_alphas, PH_FWD1, 1, 0, h->stream);
  h->ks->costate(dcall, h->P.solver, 0, n_alphas, PH_FWD1, 1, 0, h->stream);
  HIPCHK(hipGetLastError());
  const DevBuf &d = h->d;
  const size_t n = (size_t)n_alphas * d.Bp;
  std::vector<double> c(n), mf(n), th(n), ipr(n), ic(n), ap(n), ad(n);
  std::vector<int> su(n);
  HIPCHK(hipMemcpyAsync(c.data(), d.t_cost, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(mf.data(), d.t_merit, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(th.data(), d.t_theta, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(ipr.data(), d.t_inf_pr, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(ic.data(), d.t_inf_comp, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(ap.data(), d.t_apr, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(ad.data(), d.t_adu, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(su.data(), d.t_success, n * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->dP, &h->P, sizeof(ProblemDev), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < d.B; ++b)
    for (int a = 0; a < n_alphas; ++a) {
      cddp_hip_trial &t = trials[(size_t)b * n_alphas + a];
      const size_t i = (size_t)a * d.Bp + b;
      t.alpha = alphas[a]; t.alpha_pr = ap[i]; t.alpha_du = ad[i]; t.cost = c[i]; t.merit_function = mf[i];
      t.theta = th[i]; t.inf_pr = ipr[i]; t.inf_comp = ic[i]; t.success = (su[i] == 1) ? 1 : 0; t._pad = 0;   // 2 = non-finite costate (k_costate)
    }
  return 0;
}

// ---- ISolverAlgorithm::solve (cddp_solver_base.cpp:29-186) for one tile group -------------------------------
// The host loop of a group is resumable: `advance` enqueues iterations up to and including the next one that needs the
// "anything still running?" poll and records an event behind the poll's device-to-host copies; `complete_poll` waits for
// that event and digests the poll.  cddp_hip_solve interleaves the groups of a handle this way, so the kernels of several
// groups are in flight at once (each on its own stream) and the phases of different groups overlap on the chip.
struct SolveRun {
  Inner *h = nullptr;
  bool want_stats = false;
  int conc = 1;                     // groups solved concurrently (the ladder heuristics count the chip's wavefronts)
  int detail = -1;
  std::vector<int> ev_slot;         // ev_slot[6 * (iteration - 1) + point] = index into the pool, or -1
  size_t ev_used = 0;
  int launches = 0, outer = 0, it = 0, max_it = 0, na = 0;
  bool two_stage_marks = false, first_rule = true, pinned = false, one_stage = true, done = false, cpu_time_hit = false, use_graph = false;
  int k1 = 1, k_cap = 1, k_cap2 = 1;
  long waves_all = 0, per_alpha_waves = 1, two_stage_max_waves = 768;
  std::vector<int> hist_now, hist_prev;
  // Polls are double-buffered (round 5): behind a poll the host enqueues run_ahead() more iterations (default 1; none behind the two early polls that settle the ladder shape) BEFORE it waits for
  // the poll's words, so the group's queue does not drain while the host wakes up and enqueues (the poll only decides the ladder
  // shape and whether anything is still running; surplus iterations after the last trajectory finished are launches whose every
  // lane exits on its phase check).  A whole window ahead (4) lags the ladder adaptation of converging solves by a window and costs
  // C3 more than the drain (profiles/r05_cumask.md); CDDP_HIP_RUNAHEAD=0 restores the drain-at-every-poll loop of rounds 1 - 4.
  static constexpr int kPollWords = CDDP_HIP_MAX_ALPHAS + 2;
  hipEvent_t poll_evs[2] = {nullptr, nullptr};
  int win_enq = 0, win_dig = 0, win_it[2] = {0, 0};
  int outstanding() const { return win_enq - win_dig; }
  static int run_ahead() { static const int v = [] { const char *e = std::getenv("CDDP_HIP_RUNAHEAD"); const int n = e ? std::atoi(e) : 1; return (n >= 0 && n <= 16) ? n : 1; }(); return v; }
  bool ran_ahead = false;
  int post_poll(hipStream_t s) {
    const int slot = win_enq & 1;
    HIPCHK(hipMemcpyAsync(h->h_poll + slot * kPollWords, h->d.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h->h_poll + slot * kPollWords + 1, h->d.win_hist, sizeof(int) * (na + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(poll_evs[slot], s));
    win_it[slot] = it;
    ++win_enq;
    return 0;
  }
  std::chrono::steady_clock::time_point wall0;
  // ping-pong of two half-batch groups (cddp_hip_solve): this group's rollout launches wait for the other group's last rollout and
  // are followed by an event the other group waits for -- the two groups' rollouts never share the chip, each runs beside the other
  // group's sweep / update kernels
  hipEvent_t fwd_wait = nullptr, fwd_done = nullptr;

  void mark(int point) {
    if (detail < 0) return;
    const bool want = detail == CDDP_HIP_TIMING_ALL ||
                      (detail == CDDP_HIP_TIMING_ROLLOUT && (point == 1 || point == 2 || (two_stage_marks && (point == 3 || point == 4)))) ||
                      (detail == CDDP_HIP_TIMING_SWEEP && (point == 0 || point == 1));
    if (!want) return;
    // class-timing events: no system-scope fence at the record (hipEventDisableSystemFence) -- a default event releases to system
    // scope, i.e. a cache write-back between two kernels of the chain; only the timestamps are read, after the solve's final sync.
    // CDDP_HIP_EVENT_FENCE=1 restores the default flags (A/B: profiles/r04_graph_ab.md)
    if (ev_used == h->ev_pool.size()) {
      hipEvent_t e;
      static const bool fence = [] { const char *v = std::getenv("CDDP_HIP_EVENT_FENCE"); return v && v[0] == '1'; }();
      if ((fence ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableSystemFence)) != hipSuccess) return;
      h->ev_pool.push_back(e);
    }
    const size_t slot = (size_t)6 * (size_t)(outer - 1) + (size_t)point;
    if (ev_slot.size() <= slot) ev_slot.resize(slot + 1, -1);
    hipEventRecord(h->ev_pool[ev_used], h->stream);
    ev_slot[slot] = (int)ev_used++;
  }

  // Ladder shape.  one_stage: all n_alpha trials of every trajectory in ONE launch.  Otherwise stage 1 evaluates the
  // first k1 alphas for every trajectory and stage 2 the rest, only for the trajectories none of the first k1 worked
  // for (a launch whose workgroups exit at once when there is no such trajectory).  The rollout is a latency chain:
  // up to one wavefront per SIMD (1024) extra alphas in a launch cost no wall time, beyond that the chains slow
  // each other down -- and a stage-2 launch costs a full chain as soon as ONE trajectory needs it.  The best shape
  // therefore depends on how many alphas the problem at hand needs (measured on MI355X, ms per solve at B = 4096 /
  // 8192: cart-pole, 5.3 alphas per iteration on average: one launch 48.7 / 98, k1 = 4: 54.6 / 89.5; unicycle, 1.4
  // alphas: one launch 96 / 152, k1 = 4: 66.8 / 86.0, k1 = 1: - / 124), so it follows the accepted-alpha histogram
  // K5 keeps (read with the "anything still running" poll): k1 = the smallest count that satisfies all but about one
  // trajectory per four iterations, capped at one wavefront per SIMD; when nearly the whole ladder is needed, all of it
  // in one launch if that fits two wavefronts per SIMD, else as many alphas as do (B = 16384 cart-pole: 210 vs 243 ms).
  // The selected trials do not depend on the shape (tests/test_gpu_parity.py::test_two_stage_ladder_selects_the_same_trials);
  // CDDP_HIP_LS_STAGES=1|2 and CDDP_HIP_LS_FIRST=k pin it.
  void adapt_ladder(int window_iters) {
    if (pinned) return;
    long total = 0;
    std::vector<long> hd(na + 1);
    for (int a = 0; a <= na; ++a) { hd[a] = (long)hist_now[a] - (long)hist_prev[a]; total += hd[a]; }
    hist_prev = hist_now;
    if (total <= 0) return;
    const long allow = std::max(1L, (long)window_iters / 4);   // trajectory-iterations left to stage 2 per window
    int kq = na;                                               // alphas needed to satisfy all but `allow`
    long tail = hd[na];
    for (int k = na - 1; k >= 1; --k) {                        // tail(k) = # not satisfied by the first k alphas
      tail += hd[k];
      if (tail <= allow) kq = k; else break;
    }
    if (kq >= na - 1) {
      if (waves_all <= 2048) { one_stage = true; }
      else { one_stage = false; k1 = k_cap2; }
    } else {
      one_stage = false; k1 = std::max(1, std::min(kq + ladder_margin(), k_cap));   // + 1: the histogram drifts between polls (CDDP_HIP_LS_MARGIN)
      // A chip that the whole ladder fills at most twice (one-stage territory) gains from a short first stage only what the
      // smaller launch saves over the rollout's latency floor (C2: 291 -> ~200 us at five step sizes), and loses a full second
      // rollout + costate + update (~235 us) whenever ONE trajectory walks past k1: kept only for a first stage of at most 768
      // waves (CDDP_HIP_LS_TWO_MAX_WAVES; profiles/r03_ladder_sweep.md: 512 / 640 / 768 / 1024 -> 47.3 / 46.8 / 46.1 / 46.4 ms
      // per C2 solve on one box, differences at the box-to-box noise level).
      if (waves_all <= 2048 && (long)k1 * per_alpha_waves > two_stage_max_waves) one_stage = true;
      // A ladder that fits the chip once (at most one wavefront per SIMD) and light per-step work (nx <= 8): a first stage of more than
      // half of the ladder saves next to nothing over the whole ladder, while ONE trajectory past k1 costs a second full chain -- and
      // the histogram of such solves drifts towards smaller steps from window to window (resident LogDDP, cart-pole: k1 = 6 ... 10 of 11
      // during iterations 4 - 15; 34.3 -> 32.7 ms per solve with the whole ladder at once, profiles/r04_ladder_small.md).  Heavier plants keep the
      // short first stage (C4 share, 704 wavefronts: 955 ms adaptive against 985 ms with the whole ladder at once).  "Half" became "a third" after a sweep (ladder_frac).
      // (the two-role rollouts of LogDDP / MSIPDDP, round 5, put two wavefronts per (tile, alpha) in the launch: the same ladders, twice the count)
      const long small_cap = (h->P.solver == CDDP_HIP_SOLVER_LOGDDP || h->P.solver == CDDP_HIP_SOLVER_MSIPDDP) ? 2048 : 1024;
      if (waves_all <= small_cap && h->P.nx <= 8 && ladder_frac() * k1 > na) one_stage = true;
    }
    if (std::getenv("CDDP_HIP_DEBUG_LADDER")) {
      std::fprintf(stderr, "[ladder] it=%d total=%ld kq=%d -> %s k1=%d hist:", outer, total, kq, one_stage ? "one" : "two", k1);
      for (int a = 0; a <= na; ++a) std::fprintf(stderr, " %ld", hd[a]);
      std::fprintf(stderr, "\n");
    }
  }

  int begin(Inner *h_, bool stats_, int conc_) {
    h = h_; want_stats = stats_; conc = std::max(1, conc_);
    const ProblemDev &P = h->P;
    const DevBuf &d = h->d;
    hipStream_t s = h->stream;
    const KernelSet *ks = h->ks;
    max_it = P.opt.max_iterations;
    first_rule = (P.ls_rule == CDDP_HIP_LS_FIRST_SUCCESS);
    na = d.n_alphas;
    // Class timing: up to six mark points per outer iteration (0 start, 1 after the sweep, 2 after rollout stage 1,
    // 3 after update 1, 4 after rollout stage 2, 5 after update 2).  Every event costs ~5 us of queue time, so only
    // the points the selected detail needs are recorded (cddp_hip_set_timing_detail): 2 per iteration by default.
    { const char *e = std::getenv("CDDP_HIP_GRAPH"); use_graph = e && e[0] == '1'; }
    if (use_graph && conc > 1) {   // captured kernel nodes do not carry a stream's CU mask: the static partition is lost in replay (experiment switch; results unchanged)
      static bool said = false;
      if (!said) { said = true; std::fprintf(stderr, "[cddp_hip] CDDP_HIP_GRAPH=1: graph replay does not keep the CU-masked streams' partition; the tile groups share the whole chip\n"); }
    }
    detail = (want_stats && !use_graph) ? h->timing_detail : -1;
    if (!h->ev_begin) { HIPCHK(hipEventCreate(&h->ev_begin)); HIPCHK(hipEventCreate(&h->ev_end)); }
    if (!h->ev_poll) { HIPCHK(hipEventCreateWithFlags(&h->ev_poll, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_poll2, hipEventDisableTiming)); }
    poll_evs[0] = h->ev_poll; poll_evs[1] = h->ev_poll2;
    win_enq = win_dig = 0; ran_ahead = false;
    if (!h->h_poll) HIPCHK(hipHostMalloc((void **)&h->h_poll, sizeof(int) * 2 * kPollWords));
    HIPCHK(hipMemsetAsync(h->d_launched, 0, sizeof(unsigned long long), s));
    HIPCHK(hipEventRecord(h->ev_begin, s));
    { int rc = run_initialize(h); if (rc) return rc; }
    launches = 1; outer = 0; it = 0; done = false;
    *h->h_poll = d.B;
    // Speculative line search: when batch x n_alphas wavefronts still underfill the chip (256 CUs x 4 SIMDs),
    // evaluating the whole ladder in ONE launch costs no extra wall time and removes one rollout latency
    // per iteration; the first-success rule is then applied to the recorded trials, so results are unchanged.
    // (the two-role rollout of the path-constrained layouts runs two wavefronts per tile and alpha; with several tile
    // groups in flight the chip is shared: the group's wavefront count is scaled by the number of groups)
    waves_all = (long)conc * (long)((d.B + 63) / 64) * na * (((P.solver == CDDP_HIP_SOLVER_IPDDP && (ks->cst_size > 0 || (d.te_cst && P.m > 0))) || P.solver == CDDP_HIP_SOLVER_MSIPDDP || P.solver == CDDP_HIP_SOLVER_LOGDDP) ? 2 : 1);   // (layouts with two consumer waves, KernelSet::k4_waves = 3, keep the count the ladder rules were tuned with: their third wave is light)
    // CDDP_HIP_LS_STAGES=2 forces the two-stage ladder (alpha_0 first, the rest only for trajectories that need it)
    // regardless of the fill heuristic -- same selected trials; used by the tests to cover both launch shapes.
    const char *ls_env = std::getenv("CDDP_HIP_LS_STAGES");
    const bool force_two = ls_env && ls_env[0] == '2', force_one = ls_env && ls_env[0] == '1';
    const long per_alpha = std::max(1L, waves_all / std::max(1, na));
    per_alpha_waves = per_alpha;
    { const char *e = std::getenv("CDDP_HIP_LS_TWO_MAX_WAVES"); two_stage_max_waves = e ? std::atol(e) : 768; }
    k_cap = (int)std::max(1L, std::min((long)na - 1, 1024 / per_alpha));    // one wavefront per SIMD
    k_cap2 = (int)std::max(1L, std::min((long)na - 1, 2048 / per_alpha));   // two (a ladder that is needed almost whole)
    const char *kf_env = std::getenv("CDDP_HIP_LS_FIRST");
    const int k_forced = kf_env ? std::atoi(kf_env) : 0;
    pinned = !first_rule || na == 1 || force_one || force_two || (k_forced >= 1 && k_forced < na);
    one_stage = !first_rule || na == 1 || force_one || (waves_all <= 2048 && !force_two);
    k1 = force_two ? 1 : k_cap2;
    if (k_forced >= 1 && k_forced < na && !force_one && first_rule) { one_stage = false; k1 = k_forced; }
    hist_now.assign(na + 1, 0); hist_prev.assign(na + 1, 0);
    HIPCHK(hipMemsetAsync(d.win_hist, 0, sizeof(int) * (CDDP_HIP_MAX_ALPHAS + 1), s));
    wall0 = std::chrono::steady_clock::now();
    if (max_it <= 0) { ks->update(d, 2, 0, 1, 1, s); ++launches; done = true; }
    return 0;
  }

  // The kernels of ONE outer iteration (K1 .. K5 in the current ladder shape), enqueued on the group's stream.
  void enqueue_iteration(int last) {
    const ProblemDev &P = h->P;
    const DevBuf &d = h->d;
    hipStream_t s = h->stream;
    const KernelSet *ks = h->ks;
    // CU-partitioned streams (CuPlan): the rollout on its own stream, ordered against the main stream by two events per launch
    hipStream_t sf = h->cu.fwd ? h->cu.fwd : s;
    auto to_fwd = [&](int k) { if (sf != s) { order_check(hipEventRecord(h->cu.ev[k], s)); order_check(hipStreamWaitEvent(sf, h->cu.ev[k], 0)); } };
    auto from_fwd = [&](int k) { if (sf != s) { order_check(hipEventRecord(h->cu.ev[k], sf)); order_check(hipStreamWaitEvent(s, h->cu.ev[k], 0)); } };
    two_stage_marks = !one_stage;
    mark(0);
    h->last_t4 = ks->t4_layout(d);   // the layout this fill writes (cddp_hip_get_linearization reads it back with the same rule's answer)
    ks->derivs(d, 0, s);
    tl_sweep_hop = h->cu.sweep ? &h->cu.hop : nullptr;
    ks->backward(d, P.solver, 0, 1, s);
    tl_sweep_hop = nullptr;
    mark(1);
    to_fwd(0);
    if (fwd_wait) hipStreamWaitEvent(sf, fwd_wait, 0);
    if (one_stage) {
      ks->forward(d, P.solver, 0, na, PH_FWD1, 0, first_rule ? 1 : 0, sf);
      if (fwd_done) hipEventRecord(fwd_done, sf);
      from_fwd(1);
      mark(2);
      ks->costate(d, P.solver, 0, na, PH_FWD1, 0, first_rule ? 1 : 2, s);   // best-merit rule: the candidate winner's costate only (k_costate)
      ks->update(d, 1, na, last, 1, s);
      mark(3);
      launches += 4;
    } else {
      ks->forward(d, P.solver, 0, k1, PH_FWD1, 0, 1, sf);
      from_fwd(1);
      mark(2);
      ks->costate(d, P.solver, 0, k1, PH_FWD1, 0, 1, s);
      ks->update(d, 1, k1, last, 0, s);
      mark(3);
      to_fwd(2);
      ks->forward(d, P.solver, k1, na - k1, PH_FWD2, 0, 1, sf);
      if (fwd_done) hipEventRecord(fwd_done, sf);
      from_fwd(3);
      mark(4);
      ks->costate(d, P.solver, k1, na - k1, PH_FWD2, 0, 1, s);
      ks->update(d, 2, na, last, 1, s);
      mark(5);
      launches += 6;
    }
  }
  static int ladder_frac() { static const int v = [] { const char *e = std::getenv("CDDP_HIP_LS_SMALL_FRAC"); const int n = e ? std::atoi(e) : 3; return n >= 1 ? n : 3; }(); return v; }   // "more than 1 / frac of the ladder" in the rule above (CDDP_HIP_LS_SMALL_FRAC; 2 / 3 / 4 / 11 -> C2-CLDDP 30.55 / 30.33 / 30.29 / 30.3 ms, pendulum MSIPDDP 15.2 / 15.0 / 15.05 / 17.3)
  static int ladder_margin() { static const int v = [] { const char *e = std::getenv("CDDP_HIP_LS_MARGIN"); const int n = e ? std::atoi(e) : 1; return (n >= 0 && n <= 8) ? n : 1; }(); return v; }
  static int poll_every() {   // CDDP_HIP_POLL_EVERY=n (experiment): iterations between two "anything still running?" polls (default 4)
    static const int v = [] { const char *e = std::getenv("CDDP_HIP_POLL_EVERY"); const int n = e ? std::atoi(e) : 4; return n >= 1 ? n : 4; }();
    return v;
  }
  static bool polled_iteration(int it, int max_it, bool pinned) { return it % poll_every() == 0 || it == max_it || (it <= 2 && !pinned); }

  // CDDP_HIP_GRAPH=1 (experiment): the iterations between two polls are captured ONCE per (ladder shape, window
  // length, last-iteration flag) into a hipGraph and replayed with one hipGraphLaunch -- every kernel argument of an iteration is
  // a function of exactly those.  Measured on MI355X (profiles/r04_graph_ab.md): the gaps between dependent kernels are device-side
  // (barrier packet + cache invalidate), not submission latency -- the host already runs four iterations ahead of the device --
  // so replaying changes nothing measurable; class timing (hipEvents inside the window) is unavailable in this mode.  Off by default.
  int advance_graph() {
    const hipStream_t s = h->stream;
    int w = 0;
    while (it + w < max_it) { ++w; if (polled_iteration(it + w, max_it, pinned)) break; }
    if (w == 0) { done = true; return 0; }
    const int last = (it + w == max_it) ? 1 : 0;
    const unsigned long long key = ((unsigned long long)(one_stage ? 0 : k1) << 16) | ((unsigned long long)w << 4) | (unsigned long long)last;
    {   // a captured window bakes in the kernels launch.hpp selected from the environment: drop the cache when that selection changed
      unsigned sig = 2166136261u;
      for (const char *v : {"CDDP_HIP_SWEEP", "CDDP_HIP_T4", "CDDP_HIP_K4_NA", "CDDP_HIP_COOP_H", "CDDP_HIP_COOP_W", "CDDP_HIP_MS_ROLLOUT", "CDDP_HIP_LG_ROLLOUT",
                            "CDDP_HIP_K4_CONSUMERS", "CDDP_HIP_SWEEP_ROLES"}) {   // every variable launch.hpp reads per launch
        const char *e = std::getenv(v);
        for (const char *c = e ? e : "-"; *c; ++c) sig = (sig ^ (unsigned char)*c) * 16777619u;
        sig = (sig ^ 0xffu) * 16777619u;
      }
      if (sig != h->env_sig) { for (auto &kv : h->graphs) hipGraphExecDestroy(kv.second); h->graphs.clear(); h->graph_launches.clear(); h->env_sig = sig; }
    }
    auto f = h->graphs.find(key);
    if (f == h->graphs.end()) {
      hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
      const int launches0 = launches;
      HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int j = 1; j <= w; ++j) enqueue_iteration((j == w) ? last : 0);
      HIPCHK(hipStreamEndCapture(s, &g));
      HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipGraphDestroy(g);
      launches = launches0;
      f = h->graphs.emplace(key, ge).first;
      h->graph_launches.emplace(key, 0);
    }
    { const int l0 = launches; for (int j = 0; j < w; ++j) launches += one_stage ? 4 : 6; (void)l0; }
    HIPCHK(hipGraphLaunch(f->second, s));
    it += w; outer += w;
    { int rc = post_poll(s); if (rc) return rc; }
    return 1;
  }

  // enqueue iterations up to (and including) the next polled one
  int advance(int max_new = 1 << 30) {   // returns 1: a poll is pending, 2: stopped at the cap (no poll), 0: nothing more to enqueue
    if (done) return 0;
    int enq = 0;
    const ProblemDev &P = h->P;
    const DevBuf &d = h->d;
    hipStream_t s = h->stream;
    if (use_graph && !(P.opt.max_cpu_time > 0.0)) return advance_graph();
    constexpr int kPollEvery = 4;
    while (it < max_it) {
      ++it;
      if (P.opt.max_cpu_time > 0.0) {   // cddp_solver_base.cpp:77-90 (host clock, like the reference; the queue is drained first)
        HIPCHK(hipStreamSynchronize(s));
        // whole elapsed milliseconds against max_cpu_time * 1000, as the reference's duration_cast<milliseconds> compares them
        const double el_ms = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count();
        if (el_ms > P.opt.max_cpu_time * 1000.0) {
          hipLaunchKernelGGL(k_mark_cpu_time, dim3((d.B + 255) / 256), dim3(256), 0, s, d);
          ++launches; done = true; cpu_time_hit = true;
          return 0;
        }
      }
      ++outer;
      const int last = (it == max_it) ? 1 : 0;
      enqueue_iteration(last);
      // The "anything still running?" poll drains the group's queue (host round trip + an empty pipeline for the next
      // launches), so it is made every kPollEvery iterations; the up-to-3 surplus iterations after the last
      // trajectory finished are launches whose every lane exits on its phase check.
      if (polled_iteration(it, max_it, pinned)) {   // every kPollEvery iterations, the last one, and two early polls: the ladder statistics settle the shape
        { int rc = post_poll(s); if (rc) return rc; }
        return 1;   // a poll is pending
      }
      if (++enq >= max_new) return 2;
    }
    done = true;
    return 0;
  }

  int complete_poll() {   // digests the OLDEST outstanding poll
    const int kPollEvery = poll_every();
    if (outstanding() <= 0) return 0;
    const int slot = win_dig & 1;
    HIPCHK(hipEventSynchronize(poll_evs[slot]));
    ++win_dig;
    const int *w = h->h_poll + slot * kPollWords;
    const int pit = win_it[slot];
    if (w[0] == 0 || pit >= max_it) { done = true; win_dig = win_enq; return 0; }   // (later polls, if any, are dropped: finish() drains the stream)
    for (int a = 0; a <= na; ++a) hist_now[a] = w[1 + a];
    adapt_ladder(pit <= 2 ? 1 : (pit == kPollEvery ? 2 : kPollEvery));
    return 0;
  }

  int finish(cddp_hip_stats *stats) {
    const DevBuf &d = h->d;
    hipStream_t s = h->stream;
    HIPCHK(hipEventRecord(h->ev_end, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (tl_order_error != hipSuccess) {   // a cross-stream ordering call of this solve failed (launch.hpp::order_check): the results may come from racing kernels
      const hipError_t oe = tl_order_error; tl_order_error = hipSuccess;
      return fail(-10, "stream ordering (event record / wait) failed during the solve: %s", hipGetErrorString(oe));
    }
    if (!stats) return 0;
    std::memset(stats, 0, sizeof(*stats));
    float ms = 0;
    hipEventElapsedTime(&ms, h->ev_begin, h->ev_end);
    stats->solve_ms = ms;
    auto span = [&](size_t it0, int pa, int pb) -> double {   // elapsed between two mark points of one iteration
      const size_t ia = it0 * 6 + (size_t)pa, ib = it0 * 6 + (size_t)pb;
      if (ib >= ev_slot.size() || ev_slot[ia] < 0 || ev_slot[ib] < 0) return 0.0;
      float v = 0;
      hipEventElapsedTime(&v, h->ev_pool[ev_slot[ia]], h->ev_pool[ev_slot[ib]]);
      return v;
    };
    for (size_t it0 = 0; it0 * 6 < ev_slot.size(); ++it0) {
      stats->backward_ms += span(it0, 0, 1);                     // derivs + sweep
      stats->forward_ms += span(it0, 1, 2) + span(it0, 3, 4);    // rollout stage 1 (+ stage 2)
      stats->update_ms += span(it0, 2, 3) + span(it0, 4, 5);     // costate + update
    }
    // (two-stage iterations record the point between update 1 and rollout 2 in every mode; only the classes the
    //  detail names are reported)
    if (detail != CDDP_HIP_TIMING_ALL) stats->update_ms = 0.0;
    if (detail == CDDP_HIP_TIMING_ROLLOUT) stats->backward_ms = 0.0;
    if (detail == CDDP_HIP_TIMING_SWEEP) stats->forward_ms = 0.0;
    stats->timing_detail = detail;
    std::vector<int> nb(d.B), nf(d.B), itv(d.B), stv(d.B), nst(d.B);
    HIPCHK(hipMemcpy(nst.data(), d.n_fwd_steps, sizeof(int) * d.B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nb.data(), d.n_bwd, sizeof(int) * d.B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nf.data(), d.n_fwd, sizeof(int) * d.B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(itv.data(), d.iter, sizeof(int) * d.B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(stv.data(), d.status, sizeof(int) * d.B, hipMemcpyDeviceToHost));
    for (int b = 0; b < d.B; ++b) {
      stats->sweeps += nb[b]; stats->rollouts += nf[b]; stats->traj_iterations += itv[b]; stats->rollout_steps += nst[b];
      if (stv[b] == CDDP_HIP_STATUS_OPTIMAL || stv[b] == CDDP_HIP_STATUS_ACCEPTABLE) stats->n_converged++;
    }
    unsigned long long nl = 0;
    HIPCHK(hipMemcpy(&nl, h->d_launched, sizeof(nl), hipMemcpyDeviceToHost));
    stats->rollouts_launched = (int64_t)nl;
    stats->outer_iterations = outer; stats->kernel_launches = launches;
    return 0;
  }
};

// ---- getters -----------------------------------------------------------------------------
static int fetch(Inner *h, const double *dev, size_t n, std::vector<double> &host) {
  host.resize(n);
  HIPCHK(hipMemcpy(host.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

static int in_get_results(Inner *h, cddp_hip_result *r) {
  if (!h || !r) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  const int B = d.B;
  std::vector<double> v[10];
  const double *src[10] = {d.cost, d.merit, d.inf_pr, d.inf_du, d.inf_comp, d.mu, d.reg, d.alpha_pr, d.alpha_du, d.step_norm};
  for (int i = 0; i < 10; ++i) { int rc = fetch(h, src[i], B, v[i]); if (rc) return rc; }
  std::vector<int> it(B), st(B), nb(B), nf(B);
  HIPCHK(hipMemcpy(it.data(), d.iter, sizeof(int) * B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(st.data(), d.status, sizeof(int) * B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(nb.data(), d.n_bwd, sizeof(int) * B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(nf.data(), d.n_fwd, sizeof(int) * B, hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b) {
    r[b].final_objective = v[0][b]; r[b].merit_function = v[1][b]; r[b].inf_pr = v[2][b]; r[b].inf_du = v[3][b];
    r[b].inf_comp = v[4][b]; r[b].barrier_mu = v[5][b]; r[b].regularization = v[6][b]; r[b].alpha_pr = v[7][b];
    r[b].alpha_du = v[8][b]; r[b].step_norm = v[9][b]; r[b].iterations = it[b]; r[b].status = st[b];
    r[b].n_backward = nb[b]; r[b].n_forward = nf[b];
  }
  return 0;
}

// copies the CURRENT slot of every trajectory of a slotted array
static int fetch_current(Inner *h, const double *base, size_t plane, int T, int E, double *out) {
  const DevBuf &d = h->d;
  std::vector<int> cur(d.B);
  HIPCHK(hipMemcpy(cur.data(), d.cur, sizeof(int) * d.B, hipMemcpyDeviceToHost));
  std::vector<char> need(d.n_slots, 0);
  for (int b = 0; b < d.B; ++b) need[cur[b]] = 1;
  std::vector<double> buf;
  for (int s = 0; s < d.n_slots; ++s) {
    if (!need[s]) continue;
    int rc = fetch(h, base + (size_t)s * plane, plane, buf);
    if (rc) return rc;
    for (int b = 0; b < d.B; ++b) {
      if (cur[b] != s) continue;
      for (int t = 0; t < T; ++t)
        for (int e = 0; e < E; ++e) out[((size_t)b * T + t) * E + e] = buf[tix(t, E, e, b, d.Bp)];
    }
  }
  return 0;
}

static int in_get_trajectory(Inner *h, double *X, double *U) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  if (X) { int rc = fetch_current(h, d.X, d.planeX, d.N + 1, h->P.nx, X); if (rc) return rc; }
  if (U) { int rc = fetch_current(h, d.U, d.planeU, d.N, h->P.nu, U); if (rc) return rc; }
  return 0;
}

static int in_get_gains(Inner *h, double *K, double *k) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  std::vector<double> buf;
  if (K) { int rc = fetch(h, d.K, (size_t)d.N * h->P.nu * h->P.nx * d.Bp, buf); if (rc) return rc; from_soa(buf.data(), K, d.B, d.Bp, d.N, h->P.nu * h->P.nx); }
  if (k) { int rc = fetch(h, d.k, (size_t)d.N * h->P.nu * d.Bp, buf); if (rc) return rc; from_soa(buf.data(), k, d.B, d.Bp, d.N, h->P.nu); }
  return 0;
}

static int in_get_value(Inner *h, double *Vx, double *Vxx) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  std::vector<double> buf;
  if (Vx) { int rc = fetch(h, d.Vx, (size_t)(d.N + 1) * h->P.nx * d.Bp, buf); if (rc) return rc; from_soa(buf.data(), Vx, d.B, d.Bp, d.N + 1, h->P.nx); }
  if (Vxx) { int rc = fetch(h, d.Vxx, (size_t)(d.N + 1) * h->P.nx * h->P.nx * d.Bp, buf); if (rc) return rc; from_soa(buf.data(), Vxx, d.B, d.Bp, d.N + 1, h->P.nx * h->P.nx); }
  return 0;
}

static int in_get_linearization(Inner *h, double *A, double *Bm) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  std::vector<double> buf;
  // the layout the LAST derivative fill wrote (recorded at that launch: the environment may have been switched since)
  const bool t4 = h->P.solver == CDDP_HIP_SOLVER_IPDDP && h->last_t4 != 0;
  auto conv = t4 ? &from_t4 : &from_soa;
  if (A) { int rc = fetch(h, d.A, (size_t)d.N * h->P.nx * h->P.nx * d.Bp, buf); if (rc) return rc; conv(buf.data(), A, d.B, d.Bp, d.N, h->P.nx * h->P.nx); }
  if (Bm) { int rc = fetch(h, d.Bm, (size_t)d.N * h->P.nx * h->P.nu * d.Bp, buf); if (rc) return rc; conv(buf.data(), Bm, d.B, d.Bp, d.N, h->P.nx * h->P.nu); }
  return 0;
}

static int in_get_duals(Inner *h, double *S, double *Y, double *G) {
  if (!h) return fail(-1, "null handle");
  if ((h->P.solver != CDDP_HIP_SOLVER_IPDDP && h->P.solver != CDDP_HIP_SOLVER_MSIPDDP) || h->P.m == 0) return fail(-1, "no slack/dual trajectories for this problem");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  if (S) { int rc = fetch_current(h, d.S, d.planeM, d.N, h->P.m, S); if (rc) return rc; }
  if (Y) { int rc = fetch_current(h, d.Y, d.planeM, d.N, h->P.m, Y); if (rc) return rc; }
  if (G) { int rc = fetch_current(h, d.G, d.planeM, d.N, h->P.m, G); if (rc) return rc; }
  return 0;
}

// Costate trajectory of the current iterate (Lambda_ of ipddp_solver.hpp / msipddp_solver.hpp): IPDDP N + 1 rows (k_costate / k_costate_one write
// the accepted trial's), MSIPDDP N rows (k_rows_msipddp)
static int costate_rows(const Inner *h) { return h->P.solver == CDDP_HIP_SOLVER_IPDDP ? h->d.N + 1 : h->d.N; }
static int in_get_costates(Inner *h, double *Lambda, int32_t *rows) {
  if (!h) return fail(-1, "null handle");
  if (h->P.solver != CDDP_HIP_SOLVER_IPDDP && h->P.solver != CDDP_HIP_SOLVER_MSIPDDP) return fail(-1, "no costate trajectory for this solver (IPDDP and MSIPDDP carry one)");
  if (rows) *rows = costate_rows(h);
  if (!Lambda) return 0;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return fetch_current(h, h->d.Lam, h->d.planeX, costate_rows(h), h->P.nx, Lambda);
}

static int in_get_terminal(Inner *h, double *S_T, double *Y_T, double *G_T, double *Lambda_T, int32_t *dims) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  const int mT = h->P.mT, pT = h->P.pT;
  if (dims) { dims[0] = mT; dims[1] = pT; }
  std::vector<double> buf;
  struct { const double *dev; double *host; int n; } items[] = {{d.ST, S_T, mT}, {d.YT, Y_T, mT}, {d.GT, G_T, mT}, {d.LamT, Lambda_T, pT}};
  for (auto &it : items) {
    if (!it.host || it.n == 0 || !it.dev) continue;
    int rc = fetch(h, it.dev, (size_t)it.n * d.Bp, buf); if (rc) return rc;
    for (int b = 0; b < d.B; ++b) for (int i = 0; i < it.n; ++i) it.host[(size_t)b * it.n + i] = buf[(size_t)i * d.Bp + b];
  }
  return 0;
}

static int in_get_backward_scalars(Inner *h, double *dV, double *reg) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const DevBuf &d = h->d;
  std::vector<double> a, b2;
  if (dV) {
    int rc = fetch(h, d.dV0, d.B, a); if (rc) return rc;
    rc = fetch(h, d.dV1, d.B, b2); if (rc) return rc;
    for (int b = 0; b < d.B; ++b) { dV[2 * b] = a[b]; dV[2 * b + 1] = b2[b]; }
  }
  if (reg) { int rc = fetch(h, d.reg, d.B, a); if (rc) return rc; for (int b = 0; b < d.B; ++b) reg[b] = a[b]; }
  return 0;
}

static int in_history_capacity(Inner *h) { return h ? h->d.hist_cap : -1; }

static int in_get_history(Inner *h, int hist_batch, double *hist, int32_t *counts) {
  if (!h || !hist || !counts) return fail(-1, "null argument");
  const DevBuf &d = h->d;
  if (hist_batch > d.hist_batch) return fail(-1, "history kept for %d trajectories only (options.return_iteration_info=%d)", d.hist_batch, h->P.opt.return_iteration_info);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(hist, d.hist, sizeof(double) * (size_t)hist_batch * d.hist_cap * kHistCols, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(counts, d.hist_n, sizeof(int) * hist_batch, hipMemcpyDeviceToHost));
  return 0;
}

static int in_get_plan_head(Inner *h, double *u0, double *x1) {
  if (!h) return fail(-1, "null handle");
  if (!h->has_state) return fail(-1, "cddp_hip_get_plan_head needs a solved or initialised handle");
  HIPCHK(hipSetDevice(h->device));
  const int B = h->d.B, nx = h->P.nx, nu = h->P.nu;
  const size_t n = (size_t)B * (nx + nu);
  if (h->head_cap < n) {
    if (h->d_head) hipFree(h->d_head);
    if (h->h_head) hipHostFree(h->h_head);
    h->d_head = nullptr; h->h_head = nullptr; h->head_cap = 0;
    HIPCHK(hipMalloc((void **)&h->d_head, n * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&h->h_head, n * sizeof(double)));
    h->head_cap = n;
  }
  hipLaunchKernelGGL(k_gather_plan_head, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->d, nx, nu, h->d_head);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h->h_head, h->d_head, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < B; ++b) {
    const double *r = h->h_head + (size_t)b * (nx + nu);
    if (u0) for (int i = 0; i < nu; ++i) u0[(size_t)b * nu + i] = r[i];
    if (x1) for (int i = 0; i < nx; ++i) x1[(size_t)b * nx + i] = r[nu + i];
  }
  return 0;
}

static int in_write_gather_records_device(Inner *h, void *device_ptr) {
  if (!h || !device_ptr) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_gather_records, dim3((h->d.B + 255) / 256), dim3(256), 0, h->stream, h->d, (cddp_hip_gather_record *)device_ptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}


// ================================================================================================================
// Public handle: the batch is cut into tile GROUPS (whole 64-trajectory tiles), each an independent Inner solver
// with its own device buffers, stream, counters and ladder statistics.  Trajectories are independent problems, so a
// trajectory's result does not depend on the group it lands in (tests/test_full_size.py: full batch == sub-batches,
// bit for bit); cddp_hip_solve keeps the kernels of all groups in flight at once, which lets the short, latency-bound
// sweep of one group run under the rollout of another instead of leaving three quarters of the SIMDs idle.
// CDDP_HIP_GROUPS=n pins the group count (1 = the single-stream path of round 1).
// ================================================================================================================
}  // extern "C" (reopened below)

struct cddp_hip_handle {
  std::vector<Inner *> g;
  std::vector<int> b0;       // first trajectory of each group
  int B = 0, device = 0;
  int conc = 1;              // groups in flight at once in cddp_hip_solve (pick_groups)
  bool pingpong = false;     // two groups, rollouts alternating (CDDP_HIP_PINGPONG=1 with CDDP_HIP_GROUPS=2)
  hipEvent_t ev_pp[2] = {nullptr, nullptr};
  int nx = 0, nu = 0, N = 0, m = 0, mT = 0, pT = 0;
  hipStream_t user_stream = nullptr;   // cddp_hip_set_stream: work is ordered after / before this stream's work
  bool have_user_stream = false;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void *d_send = nullptr;              // send buffer of cddp_hip_allgather_results (shard_capacity records)
  int send_cap = 0;
};

namespace {

// How the batch is cut.  Returns the group count; *conc = how many groups cddp_hip_solve keeps in flight at once.
//  * CDDP_HIP_GROUPS=n  (experiments, tests): n groups, all in flight concurrently (rounds 2-3).
//  * otherwise (round 4): CHUNKS.  A batch whose rollout launch would put more than ~2.75 wavefronts on every SIMD is solved as
//    several groups ONE AFTER THE OTHER.  Measured on MI355X at C2 (profiles/r03_batch_curve.md): 88.9 k trajectories/s at
//    B = 4096, 89.6 k at 8192, then 77.5 k at 16384 and 67.6 k at 32768 -- the rollout class grows super-linearly (51 -> 134 ->
//    333 ms) once the launch needs more than two residency rounds and the per-solve state (~0.1 MB per trajectory) is several
//    times the 256 MB Infinity Cache.  A trajectory's result does not depend on its group (bitwise, tests/test_full_size.py,
//    tests/test_determinism.py), so oversubscription costs nothing: the large batch runs at the throughput of its best chunk size
//    (profiles/r04_batch_curve.md).  CDDP_HIP_CHUNK=<trajectories> overrides the chunk size (0 = never chunk).
//  * round 5: STATIC CU PARTITION.  An IPDDP / CLDDP batch (or each chunk of a large one) of at least 32 tiles is cut into TWO groups that
//    are solved concurrently, each with every kernel of its iterations on its own symmetric half of the chip (streams created with
//    hipExtStreamCreateWithCUMask; mask bits [0, 128) and [128, 256) = half of the CUs of every XCD, see CuSpec).  Two independent
//    half-chips at half the batch run 3 - 6 % faster than the whole chip at the whole batch (profiles/r05_cumask.md: C2 45.6 -> 42.8 ms,
//    C3 81.5 -> 78.9, C4 share 955 -> 916, C5 share 1041 -> 992; CLDDP + 1.4 %, LogDDP 0, MSIPDDP - 4 %: not partitioned), while the
//    same two groups WITHOUT masks lose 3 - 5 % (their wavefronts share SIMDs) and any split that moves one kernel class to another
//    stream pays ~20 us of cross-stream events per iteration.  *part = slices (1 = no masks); CDDP_HIP_PARTITION=n overrides (1 = off).
int pick_groups(int batch, int n_alphas, int solver, int *conc, int *part) {
  const int tiles = (batch + 63) / 64;
  const char *e = std::getenv("CDDP_HIP_GROUPS");
  int n = e ? std::atoi(e) : 0;
  *part = 1;
  if (n > 0) { n = std::max(1, std::min(n, tiles)); *conc = n; return n; }
  *conc = 1;
  int chunk_tiles = std::max(16, 2816 / (2 * std::max(1, n_alphas)));     // 128 tiles (8192 trajectories) at 11 step sizes
  bool chunking = true;
  if (const char *c = std::getenv("CDDP_HIP_CHUNK")) {
    const int v = std::atoi(c);
    if (v <= 0) chunking = false; else chunk_tiles = std::max(1, (v + 63) / 64);
  }
  const int chunks = chunking ? std::max(1, (tiles + chunk_tiles - 1) / chunk_tiles) : 1;
  int np = (solver == CDDP_HIP_SOLVER_IPDDP || solver == CDDP_HIP_SOLVER_CLDDP) ? 2 : 1;
  if (const char *c = std::getenv("CDDP_HIP_PARTITION")) { const int v = std::atoi(c); if (v >= 1 && v <= 16 && 256 % v == 0) np = v; }
  if (std::getenv("CDDP_HIP_CUMASK")) np = 1;                              // an explicit plan (experiments) replaces the default one
  if (np > 1 && tiles / chunks >= 16 * np) { *conc = np; *part = np; return chunks * np; }
  return chunks;
}

// ordering against a caller-supplied stream: fork = group streams wait for the user's stream, join = the reverse
int fork_from_user(cddp_hip_handle *h) {
  if (!h->have_user_stream) return 0;
  if (!h->ev_fork) { HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming)); }
  HIPCHK(hipEventRecord(h->ev_fork, h->user_stream));
  for (Inner *q : h->g) if (q->stream != h->user_stream) HIPCHK(hipStreamWaitEvent(q->stream, h->ev_fork, 0));
  return 0;
}
int join_to_user(cddp_hip_handle *h) {
  if (!h->have_user_stream) return 0;
  for (Inner *q : h->g) {
    if (q->stream == h->user_stream) continue;
    HIPCHK(hipEventRecord(h->ev_join, q->stream));
    HIPCHK(hipStreamWaitEvent(h->user_stream, h->ev_join, 0));
  }
  return 0;
}

}  // namespace

extern "C" {

int cddp_hip_create(const cddp_hip_problem *problem, int batch, int device, cddp_hip_handle **out) {
  if (!problem || !out) return fail(-1, "null argument");
  if (batch <= 0) return fail(-1, "batch must be positive");
  int conc = 1, part = 1;
  double ladder[CDDP_HIP_MAX_ALPHAS];
  const int n_ladder = cddp_hip_build_alphas(&problem->options, ladder, CDDP_HIP_MAX_ALPHAS);   // the real ladder length (it can stop early at ls_min_step_size)
  const int tiles = (batch + 63) / 64, ng = pick_groups(batch, n_ladder, problem->solver, &conc, &part);
  cddp_hip_handle *h = new cddp_hip_handle();
  h->B = batch; h->device = device; h->conc = conc;
  { const char *e = std::getenv("CDDP_HIP_PINGPONG"); h->pingpong = e && e[0] == '1'; }
  int t0 = 0;
  for (int k = 0; k < ng; ++k) {
    const int nt = tiles / ng + (k < tiles % ng ? 1 : 0);          // whole tiles per group, sizes differ by at most one tile
    const int first = t0 * 64, last = std::min(batch, (t0 + nt) * 64);
    t0 += nt;
    if (last <= first) continue;
    Inner *q = nullptr;
    CuSpec cus;
    bool have_cu = cu_spec_for_group(k, &cus);
    if (!have_cu && part > 1) {
      // slice k of `part` equal slices of the CU-mask bits of THIS device (bit i = XCC i % 8, so a multiple of 8 bits is the same number of
      // CUs on every XCD); a device or partition mode too small for two CUs per XCD and slice runs unpartitioned
      const int w = device_cu_count(device) / part / 8 * 8;
      if (w >= 16) { cus.lo[0] = (k % part) * w; cus.hi[0] = cus.lo[0] + w; have_cu = true; }
    }
    int rc = in_create(problem, last - first, device, &q, have_cu ? &cus : nullptr);
    if (rc) { for (Inner *p : h->g) in_destroy(p); delete h; return rc; }
    h->g.push_back(q); h->b0.push_back(first);
  }
  const ProblemDev &P = h->g[0]->P;
  h->nx = P.nx; h->nu = P.nu; h->N = P.N; h->m = P.m; h->mT = P.mT; h->pT = P.pT;
  *out = h;
  return 0;
}

int cddp_hip_destroy(cddp_hip_handle *h) {
  if (!h) return 0;
  for (Inner *q : h->g) in_destroy(q);
  if (h->ev_fork) { hipEventDestroy(h->ev_fork); hipEventDestroy(h->ev_join); }
  for (int k = 0; k < 2; ++k) if (h->ev_pp[k]) hipEventDestroy(h->ev_pp[k]);
  if (h->d_send) hipFree(h->d_send);
  delete h;
  return 0;
}

int cddp_hip_num_groups(cddp_hip_handle *h) { return h ? (int)h->g.size() : -1; }
int cddp_hip_concurrency(cddp_hip_handle *h) { return h ? std::max(1, std::min(h->conc, (int)h->g.size())) : -1; }

int cddp_hip_set_timing_detail(cddp_hip_handle *h, int detail) {
  if (!h) return fail(-1, "null handle");
  for (Inner *q : h->g) { int rc = in_set_timing_detail(q, detail); if (rc) return rc; }
  return 0;
}

int cddp_hip_set_stream(cddp_hip_handle *h, void *hip_stream) {
  if (!h) return fail(-1, "null handle");
  h->user_stream = (hipStream_t)hip_stream; h->have_user_stream = true;
  if (h->g.size() == 1) return in_set_stream(h->g[0], hip_stream);   // one group: it simply runs on the caller's stream
  return 0;   // several groups keep their own streams; every entry point forks from / joins to the caller's stream
}

int cddp_hip_dual_dim(cddp_hip_handle *h) { return h ? h->m : -1; }
int cddp_hip_batch(cddp_hip_handle *h) { return h ? h->B : -1; }

#define OFF(ptr, stride) ((ptr) ? (ptr) + (size_t)h->b0[gi_] * (size_t)(stride) : nullptr)
#define FOR_GROUPS(call) do { if (!h) return fail(-1, "null handle"); { int rc_ = fork_from_user(h); if (rc_) return rc_; } \
    for (size_t gi_ = 0; gi_ < h->g.size(); ++gi_) { Inner *q = h->g[gi_]; (void)q; int rc_ = (call); if (rc_) return rc_; } \
    return join_to_user(h); } while (0)

int cddp_hip_set_initial(cddp_hip_handle *h, const double *x0, const double *U0, const double *X0) {
  if (!x0) return fail(-1, "null argument");
  FOR_GROUPS(in_set_initial(q, OFF(x0, h->nx), OFF(U0, h->N * h->nu), OFF(X0, (h->N + 1) * h->nx)));
}
int cddp_hip_initialize(cddp_hip_handle *h) { FOR_GROUPS(in_initialize(q)); }
int cddp_hip_set_options(cddp_hip_handle *h, const cddp_hip_options *opt) { FOR_GROUPS(in_set_options(q, opt)); }
int cddp_hip_set_initial_state(cddp_hip_handle *h, const double *x0) {
  if (!x0) return fail(-1, "null argument");
  FOR_GROUPS(in_set_initial_state(q, OFF(x0, h->nx)));
}
int cddp_hip_forget_solver_state(cddp_hip_handle *h) {
  if (!h) return fail(-1, "null handle");
  for (Inner *q : h->g) { q->has_state = false; q->initialized = false; }   // (run_initialize: warm_start now means "provided trajectory"; the MSIPDDP factor cache stays with the handle as documented)
  return 0;
}
int cddp_hip_set_duals(cddp_hip_handle *h, const double *S, const double *Y) { FOR_GROUPS(in_set_duals(q, OFF(S, h->N * h->m), OFF(Y, h->N * h->m))); }
int cddp_hip_set_barrier_state(cddp_hip_handle *h, const double *mu, const double *reg) { FOR_GROUPS(in_set_barrier_state(q, OFF(mu, 1), OFF(reg, 1))); }
int cddp_hip_set_terminal(cddp_hip_handle *h, const double *S_T, const double *Y_T, const double *Lambda_T) {
  FOR_GROUPS(in_set_terminal(q, OFF(S_T, h->mT), OFF(Y_T, h->mT), OFF(Lambda_T, h->pT)));
}
int cddp_hip_backward(cddp_hip_handle *h, int32_t *ok) { FOR_GROUPS(in_backward(q, OFF(ok, 1))); }
int cddp_hip_forward(cddp_hip_handle *h, const double *alphas, int n_alphas, cddp_hip_trial *trials) {
  if (!alphas || !trials) return fail(-1, "null argument");
  FOR_GROUPS(in_forward(q, alphas, n_alphas, OFF(trials, n_alphas)));
}
int cddp_hip_get_results(cddp_hip_handle *h, cddp_hip_result *r) { if (!r) return fail(-1, "null argument"); FOR_GROUPS(in_get_results(q, OFF(r, 1))); }
int cddp_hip_get_plan_head(cddp_hip_handle *h, double *u0, double *x1) { FOR_GROUPS(in_get_plan_head(q, OFF(u0, h->nu), OFF(x1, h->nx))); }
int cddp_hip_get_trajectory(cddp_hip_handle *h, double *X, double *U) { FOR_GROUPS(in_get_trajectory(q, OFF(X, (h->N + 1) * h->nx), OFF(U, h->N * h->nu))); }
int cddp_hip_get_gains(cddp_hip_handle *h, double *K, double *k) { FOR_GROUPS(in_get_gains(q, OFF(K, h->N * h->nu * h->nx), OFF(k, h->N * h->nu))); }
int cddp_hip_get_value(cddp_hip_handle *h, double *Vx, double *Vxx) { FOR_GROUPS(in_get_value(q, OFF(Vx, (h->N + 1) * h->nx), OFF(Vxx, (h->N + 1) * h->nx * h->nx))); }
int cddp_hip_get_linearization(cddp_hip_handle *h, double *A, double *Bm) { FOR_GROUPS(in_get_linearization(q, OFF(A, h->N * h->nx * h->nx), OFF(Bm, h->N * h->nx * h->nu))); }
int cddp_hip_get_duals(cddp_hip_handle *h, double *S, double *Y, double *G) {
  FOR_GROUPS(in_get_duals(q, OFF(S, h->N * h->m), OFF(Y, h->N * h->m), OFF(G, h->N * h->m)));
}
int cddp_hip_get_costates(cddp_hip_handle *h, double *Lambda, int32_t *rows) {
  if (!h) return fail(-1, "null handle");
  const int r = (h->g.empty() || !h->g[0]) ? 0 : costate_rows(h->g[0]);
  FOR_GROUPS(in_get_costates(q, OFF(Lambda, r * h->nx), rows));
}
int cddp_hip_get_terminal(cddp_hip_handle *h, double *S_T, double *Y_T, double *G_T, double *Lambda_T, int32_t *dims) {
  FOR_GROUPS(in_get_terminal(q, OFF(S_T, h->mT), OFF(Y_T, h->mT), OFF(G_T, h->mT), OFF(Lambda_T, h->pT), dims));
}
int cddp_hip_get_backward_scalars(cddp_hip_handle *h, double *dV, double *reg) { FOR_GROUPS(in_get_backward_scalars(q, OFF(dV, 2), OFF(reg, 1))); }
int cddp_hip_history_capacity(cddp_hip_handle *h) { return h ? in_history_capacity(h->g[0]) : -1; }
int cddp_hip_get_history(cddp_hip_handle *h, int hist_batch, double *hist, int32_t *counts) {
  if (!h) return fail(-1, "null handle");
  // the history of the first min(batch, 64) trajectories is kept: they all live in group 0 (groups are whole tiles)
  return in_get_history(h->g[0], hist_batch, hist, counts);
}
int cddp_hip_write_gather_records_device(cddp_hip_handle *h, void *device_ptr) {
  if (!device_ptr) return fail(-1, "null argument");
  FOR_GROUPS(in_write_gather_records_device(q, (char *)device_ptr + (size_t)h->b0[gi_] * sizeof(cddp_hip_gather_record)));
}
#undef FOR_GROUPS
#undef OFF

// The single collective of the path: all-gather of the 16-byte result records over RCCL.
int cddp_hip_allgather_results(cddp_hip_handle *h, void *comm, int world, int shard_capacity, void *recv_device) {
  if (!h || !recv_device) return fail(-1, "null argument");
  if (world <= 0) return fail(-1, "world must be positive");
  if (shard_capacity < h->B) return fail(-1, "shard_capacity %d is smaller than this rank's batch %d", shard_capacity, h->B);
  if (!comm && world != 1) return fail(-1, "a NULL communicator is only valid for world == 1 (got %d)", world);
  HIPCHK(hipSetDevice(h->device));
  if (h->send_cap < shard_capacity) {
    if (h->d_send) { hipFree(h->d_send); h->d_send = nullptr; h->send_cap = 0; }
    HIPCHK(hipMalloc(&h->d_send, (size_t)shard_capacity * sizeof(cddp_hip_gather_record)));
    h->send_cap = shard_capacity;
  }
  hipStream_t cs = h->have_user_stream ? h->user_stream : h->g[0]->stream;   // the collective's stream
  // padding records of an uneven block partition: every byte 0xFF -> status = iterations = -1, cost = NaN
  if (shard_capacity > h->B)
    HIPCHK(hipMemsetAsync((char *)h->d_send + (size_t)h->B * sizeof(cddp_hip_gather_record), 0xFF,
                          (size_t)(shard_capacity - h->B) * sizeof(cddp_hip_gather_record), cs));
  for (size_t k = 0; k < h->g.size(); ++k) {   // (in_write_gather_records_device synchronises the group's stream)
    int rc = in_write_gather_records_device(h->g[k], (char *)h->d_send + (size_t)h->b0[k] * sizeof(cddp_hip_gather_record));
    if (rc) return rc;
  }
  const size_t bytes = (size_t)shard_capacity * sizeof(cddp_hip_gather_record);
  if (!comm) HIPCHK(hipMemcpyAsync(recv_device, h->d_send, bytes, hipMemcpyDeviceToDevice, cs));
  else { int rc = cddp_hip_internal_allgather(h->d_send, recv_device, bytes, comm, (void *)cs); if (rc) return rc; }
  HIPCHK(hipStreamSynchronize(cs));
  return 0;
}

int cddp_hip_solve(cddp_hip_handle *h, cddp_hip_stats *stats) {
  if (!h) return fail(-1, "null handle");
  HIPCHK(hipSetDevice(h->device));
  { int rc = fork_from_user(h); if (rc) return rc; }
  const int ng = (int)h->g.size();
  const auto solve_t0 = std::chrono::steady_clock::now();
  std::vector<SolveRun> run(ng);
  // `conc` groups are in flight at a time (all of them with CDDP_HIP_GROUPS; one with the default chunking of a large batch).
  // The whole-handle time is the span from group 0's begin to an end event group 0's stream records after it has waited for
  // every other group's end.
  const int conc = std::max(1, std::min(h->conc, ng));
  if (h->pingpong && ng == 2) {
    // two half-batch groups in lockstep, one iteration each in turn; rollouts serialised A1 B1 A2 B2 ... by events (SolveRun::fwd_wait)
    if (!h->ev_pp[0]) for (int k = 0; k < 2; ++k) HIPCHK(hipEventCreateWithFlags(&h->ev_pp[k], hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) { int rc = run[k].begin(h->g[k], stats != nullptr, 1); if (rc) return rc; run[k].fwd_wait = h->ev_pp[1 - k]; run[k].fwd_done = h->ev_pp[k]; }
    for (;;) {
      bool any = false;
      for (int k = 0; k < 2; ++k) {
        if (run[k].done || run[k].outstanding() > 0) continue;
        int rc = run[k].advance(1); if (rc < 0) return rc;
        any = any || rc != 0;
      }
      for (int k = 0; k < 2; ++k) if (run[k].outstanding() > 0) { int rc = run[k].complete_poll(); if (rc) return rc; any = true; }
      if (!any) break;
    }
  } else
  for (int base = 0; base < ng; base += conc) {
    const int top = std::min(ng, base + conc);
    for (int k = base; k < top; ++k) { int rc = run[k].begin(h->g[k], stats != nullptr, conc); if (rc) return rc; run[k].wall0 = solve_t0; }   // ONE max_cpu_time clock per solve, shared by successive chunks
    // each pass, per group: no poll outstanding -> enqueue iterations up to the next polled one; one outstanding and not yet run
    // ahead -> enqueue run_ahead() more iterations behind it; otherwise digest the oldest poll.  So a group's queue holds work while
    // the host waits for a poll, and the host never blocks on one group while another has nothing queued
    for (;;) {
      bool any = false;
      for (int k = base; k < top; ++k) {
        SolveRun &r = run[k];
        const int ra = (h->g[k]->P.opt.max_cpu_time > 0.0 || r.use_graph || r.it < SolveRun::poll_every()) ? 0 : SolveRun::run_ahead();
        if (!r.done && r.outstanding() == 0) { int rc = r.advance(); if (rc < 0) return rc; r.ran_ahead = false; any = true; }
        else if (!r.done && r.outstanding() == 1 && ra > 0 && !r.ran_ahead) { int rc = r.advance(ra); if (rc < 0) return rc; r.ran_ahead = true; any = true; }
        else if (r.outstanding() > 0) { int rc = r.complete_poll(); if (rc) return rc; r.ran_ahead = r.outstanding() > 0; any = true; }
      }
      if (!any) break;
    }
  }
  // whole-handle device time: group 0's stream waits for the other groups' end events, then stamps the end
  std::vector<cddp_hip_stats> gs(ng);
  for (int k = 1; k < ng; ++k) {
    HIPCHK(hipEventRecord(h->g[k]->ev_end, h->g[k]->stream));
    HIPCHK(hipStreamWaitEvent(h->g[0]->stream, h->g[k]->ev_end, 0));
  }
  for (int k = 0; k < ng; ++k) { int rc = run[k].finish(stats ? &gs[k] : nullptr); if (rc) return rc; }
  { int rc = join_to_user(h); if (rc) return rc; }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    float span_ms = 0;
    double first_begin_off = 0.0;
    // span = end(group 0, which waited for everybody) - earliest begin; the begins are enqueued back to back, so group
    // 0's begin is the earliest up to the few microseconds of its own enqueue
    hipEventElapsedTime(&span_ms, h->g[0]->ev_begin, h->g[0]->ev_end);
    (void)first_begin_off;
    stats->solve_ms = span_ms;
    for (int k = 0; k < ng; ++k) {
      // class times: groups that run concurrently overlap in wall time, so their per-class event spans are averaged over the
      // `conc` groups in flight together; successive chunks add up
      stats->backward_ms += gs[k].backward_ms / conc; stats->forward_ms += gs[k].forward_ms / conc; stats->update_ms += gs[k].update_ms / conc;
      stats->sweeps += gs[k].sweeps; stats->rollouts += gs[k].rollouts; stats->rollouts_launched += gs[k].rollouts_launched;
      stats->traj_iterations += gs[k].traj_iterations; stats->rollout_steps += gs[k].rollout_steps;
      stats->n_converged += gs[k].n_converged; stats->kernel_launches += gs[k].kernel_launches;
      stats->timing_detail = gs[k].timing_detail;
    }
    // outer iterations = rounds of (sweep, rollout, update) launches the handle went through one after the other: the longest group of
    // every concurrent set, summed over the successive sets (chunks) -- the divisor that matches the summed class times
    for (int base = 0; base < ng; base += conc) {
      int mx = 0;
      for (int k = base; k < std::min(ng, base + conc); ++k) mx = std::max(mx, gs[k].outer_iterations);
      stats->outer_iterations += mx;
    }
  }
  return 0;
}

}  // extern "C"
