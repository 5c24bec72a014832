// cddp_hip.hpp -- C++17 host-side mirror of cddp-cpp's plugin surface and CDDP::solve() API for the
// hot path, sitting directly on the C-ABI of include/cddp_hip.h (header-only; link libcddp_hip.so).
//
// Same names, argument meaning and error behaviour as the reference
// (include/cddp-cpp/cddp_core/{cddp_core,options,objective,constraint,terminal_constraint}.hpp,
//  include/cddp-cpp/dynamics_model/*.hpp), so tests read like the reference's own tests:
//
//   cddp::CDDP solver(x0, goal, horizon, dt, std::make_unique<cddp::Pendulum>(dt, 0.5, 1.0, 0.01, "euler"),
//                     std::make_unique<cddp::QuadraticObjective>(Q, R, Qf, goal, {}, dt), options);
//   solver.addPathConstraint("ControlConstraint", std::make_unique<cddp::ControlConstraint>(lo, up));
//   solver.setInitialTrajectory(X, U);
//   cddp::CDDPSolution sol = solver.solve(cddp::SolverType::IPDDP);      // runs on the MI355X
//   std::vector<cddp::CDDPSolution> sols = solver.solveBatch("IPDDP", x0s);   // NEW: batched API
//
// Differences that are forced by the boundary:
//   * Eigen is not a dependency: cddp::Vector / cddp::Matrix are minimal row-major containers; when
//     <Eigen/Dense> is available, ToVector()/ToMatrix() adapters accept Eigen types.
//   * DynamicalSystem / Objective / Constraint subclasses are DESCRIPTORS of the built-in plug-ins the
//     kernels implement on the device (enumerated by id); arbitrary host subclasses go through the
//     stack-fed entry point cddp_hip_backward_stacks (they cannot run on the GPU).
//   * "CLDDP" and "IPDDP" are served by HipBatchSolver through the same static registry the reference
//     uses (CDDP::registerSolver, cddp_core.cpp:578-595); other names return the reference's
//     "UnknownSolver - No solver registered for '<name>'" solution (cddp_core.cpp:243-265).
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cddp_hip.h"

namespace cddp {

using Vector = std::vector<double>;
struct Matrix {   // row-major dense matrix
  int rows = 0, cols = 0;
  std::vector<double> a;
  Matrix() {}
  Matrix(int r, int c, double v = 0.0) : rows(r), cols(c), a((size_t)r * c, v) {}
  static Matrix Zero(int r, int c) { return Matrix(r, c, 0.0); }
  static Matrix Identity(int n) { Matrix m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  double &operator()(int i, int j) { return a[(size_t)i * cols + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * cols + j]; }
  Matrix operator*(double s) const { Matrix m = *this; for (double &v : m.a) v *= s; return m; }
};
inline Matrix operator*(double s, const Matrix &m) { return m * s; }

// ---- options: nested structs exactly as options.hpp:41-251 ------------------------------------
enum class BarrierStrategy { ADAPTIVE, MONOTONIC, IPOPT };
struct LineSearchOptions { int max_iterations = 11; double initial_step_size = 1.0, min_step_size = 1e-8, step_reduction_factor = 0.5; };
struct RegularizationOptions { double initial_value = 1e-6, update_factor = 10.0, max_value = 1e7, min_value = 1e-10, step_initial_value = 1.0; };
struct BoxQPOptions { int max_iterations = 100; double min_gradient_norm = 1e-8, min_relative_improvement = 1e-8, step_decrease_factor = 0.6, min_step_size = 1e-22, armijo_constant = 0.1; bool verbose = false; };
struct SolverSpecificBarrierOptions { double mu_initial = 1.0, mu_min_value = 1e-10, mu_update_factor = 0.5, mu_update_power = 1.2, min_fraction_to_boundary = 0.99; BarrierStrategy strategy = BarrierStrategy::ADAPTIVE; };
struct SolverSpecificFilterOptions { double merit_acceptance_threshold = 1e-6, violation_acceptance_threshold = 1e-6, max_violation_threshold = 1e4, min_violation_for_armijo_check = 1e-7, armijo_constant = 1e-4; };
struct IPDDPAlgorithmOptions {
  double dual_var_init_scale = 1e-1, slack_var_init_scale = 1e-2, barrier_tol_mult = 0.1, barrier_update_dual_weight = 0.01, mu_kappa_epsilon = 10.0;
  bool check_state_stationarity = false; std::string theta_norm = "l1"; int max_filter_size = 5; double theta_0_floor = 1.0;
  bool warmstart_repair = false; double warmstart_s_min = 1e-4, warmstart_y_min = 1e-4, warmstart_interior_factor = 1.1, warmstart_reset_x0_threshold = -1.0;
  double jacobian_regularization_value = 1e-8, jacobian_regularization_exponent = 0.25;
  SolverSpecificBarrierOptions barrier;
};
struct CDDPOptions {
  double tolerance = 1e-5, acceptable_tolerance = 1e-6; int max_iterations = 1; double max_cpu_time = 0.0;
  bool verbose = true, debug = false, print_solver_header = true, print_solver_options = false, use_ilqr = true, enable_parallel = false;
  int num_threads = 1; bool return_iteration_info = false, warm_start = false; double termination_scaling_max_factor = 100.0;
  LineSearchOptions line_search; RegularizationOptions regularization; BoxQPOptions box_qp; SolverSpecificFilterOptions filter; IPDDPAlgorithmOptions ipddp;

  cddp_hip_options toPOD() const {
    cddp_hip_options o; cddp_hip_default_options(&o);
    o.tolerance = tolerance; o.acceptable_tolerance = acceptable_tolerance; o.max_iterations = max_iterations; o.max_cpu_time = max_cpu_time;
    o.use_ilqr = use_ilqr; o.enable_parallel = enable_parallel; o.return_iteration_info = return_iteration_info; o.warm_start = warm_start;
    o.termination_scaling_max_factor = termination_scaling_max_factor;
    o.ls_max_iterations = line_search.max_iterations; o.ls_initial_step_size = line_search.initial_step_size;
    o.ls_min_step_size = line_search.min_step_size; o.ls_step_reduction_factor = line_search.step_reduction_factor;
    o.reg_initial_value = regularization.initial_value; o.reg_update_factor = regularization.update_factor;
    o.reg_max_value = regularization.max_value; o.reg_min_value = regularization.min_value;
    o.boxqp_max_iterations = box_qp.max_iterations; o.boxqp_min_gradient_norm = box_qp.min_gradient_norm;
    o.boxqp_min_relative_improvement = box_qp.min_relative_improvement; o.boxqp_step_decrease_factor = box_qp.step_decrease_factor;
    o.boxqp_min_step_size = box_qp.min_step_size; o.boxqp_armijo_constant = box_qp.armijo_constant;
    o.filter_merit_acceptance_threshold = filter.merit_acceptance_threshold; o.filter_violation_acceptance_threshold = filter.violation_acceptance_threshold;
    o.filter_max_violation_threshold = filter.max_violation_threshold; o.filter_min_violation_for_armijo_check = filter.min_violation_for_armijo_check;
    o.filter_armijo_constant = filter.armijo_constant;
    o.ipddp_dual_var_init_scale = ipddp.dual_var_init_scale; o.ipddp_slack_var_init_scale = ipddp.slack_var_init_scale;
    o.ipddp_barrier_tol_mult = ipddp.barrier_tol_mult; o.ipddp_barrier_update_dual_weight = ipddp.barrier_update_dual_weight;
    o.ipddp_mu_kappa_epsilon = ipddp.mu_kappa_epsilon; o.ipddp_check_state_stationarity = ipddp.check_state_stationarity;
    o.ipddp_theta_norm_l2 = (ipddp.theta_norm == "l2"); o.ipddp_max_filter_size = ipddp.max_filter_size; o.ipddp_theta_0_floor = ipddp.theta_0_floor;
    o.ipddp_warmstart_repair = ipddp.warmstart_repair; o.ipddp_warmstart_s_min = ipddp.warmstart_s_min; o.ipddp_warmstart_y_min = ipddp.warmstart_y_min;
    o.ipddp_warmstart_interior_factor = ipddp.warmstart_interior_factor;
    o.ipddp_jacobian_regularization_value = ipddp.jacobian_regularization_value; o.ipddp_jacobian_regularization_exponent = ipddp.jacobian_regularization_exponent;
    o.barrier_mu_initial = ipddp.barrier.mu_initial; o.barrier_mu_min_value = ipddp.barrier.mu_min_value; o.barrier_mu_update_factor = ipddp.barrier.mu_update_factor;
    o.barrier_mu_update_power = ipddp.barrier.mu_update_power; o.barrier_min_fraction_to_boundary = ipddp.barrier.min_fraction_to_boundary;
    o.barrier_strategy = (int)ipddp.barrier.strategy;
    return o;
  }
};

// ---- plug-in descriptors --------------------------------------------------------------------
inline int integratorId(const std::string &s) {
  if (s == "euler") return CDDP_HIP_EULER; if (s == "heun") return CDDP_HIP_HEUN; if (s == "rk3") return CDDP_HIP_RK3; if (s == "rk4") return CDDP_HIP_RK4;
  return -1;   // reference prints "Integration type not supported!" and returns zeros (dynamical_system.cpp:79-82)
}
class DynamicalSystem {
 public:
  DynamicalSystem(int model, int nx, int nu, double timestep, std::string integration_type)
      : model_(model), state_dim_(nx), control_dim_(nu), timestep_(timestep), integration_type_(std::move(integration_type)) {}
  virtual ~DynamicalSystem() = default;
  int getStateDim() const { return state_dim_; }
  int getControlDim() const { return control_dim_; }
  double getTimestep() const { return timestep_; }
  const std::string &getIntegrationType() const { return integration_type_; }
  int modelId() const { return model_; }
  std::vector<double> params;         // cddp_hip_problem::model_params
  Matrix lti_A, lti_B;
 protected:
  int model_, state_dim_, control_dim_; double timestep_; std::string integration_type_;
};
struct Pendulum : DynamicalSystem {   // pendulum.hpp: (timestep, length, mass, damping, integration_type)
  Pendulum(double dt, double length = 1.0, double mass = 1.0, double damping = 0.0, std::string integ = "euler")
      : DynamicalSystem(CDDP_HIP_MODEL_PENDULUM, 2, 1, dt, integ) { params = {length, mass, damping, 9.81}; }
};
struct CartPole : DynamicalSystem {   // cartpole.hpp: (timestep, integration_type, cart_mass, pole_mass, pole_length, gravity, damping)
  CartPole(double dt, std::string integ = "rk4", double mc = 1.0, double mp = 0.2, double l = 0.5, double g = 9.81, double damping = 0.0)
      : DynamicalSystem(CDDP_HIP_MODEL_CARTPOLE, 4, 1, dt, integ) { params = {mc, mp, l, g, damping}; }
};
struct Unicycle : DynamicalSystem {
  Unicycle(double dt, std::string integ = "euler") : DynamicalSystem(CDDP_HIP_MODEL_UNICYCLE, 3, 2, dt, integ) {}
};
struct LTISystem : DynamicalSystem {  // lti_system.hpp: (A, B, timestep, integration_type)
  LTISystem(const Matrix &A, const Matrix &B, double dt, std::string integ = "euler")
      : DynamicalSystem(CDDP_HIP_MODEL_LTI, A.rows, B.cols, dt, integ) {
    if (A.rows != A.cols) throw std::invalid_argument("A matrix must be square");
    if (B.rows != A.rows) throw std::invalid_argument("B matrix must have same number of rows as A");
    lti_A = A; lti_B = B;
  }
};
struct Quadrotor : DynamicalSystem {  // quadrotor.hpp: (timestep, mass, inertia_matrix(diag), arm_length, integration_type)
  Quadrotor(double dt, double mass, const Matrix &inertia, double arm_length, std::string integ = "rk4")
      : DynamicalSystem(CDDP_HIP_MODEL_QUADROTOR, 13, 4, dt, integ) { params = {mass, arm_length, inertia(0, 0), inertia(1, 1), inertia(2, 2), 9.81}; }
};
struct Manipulator : DynamicalSystem {
  Manipulator(double dt, std::string integ = "rk4") : DynamicalSystem(CDDP_HIP_MODEL_MANIPULATOR, 6, 3, dt, integ) {}
};

class Objective { public: virtual ~Objective() = default; };
class QuadraticObjective : public Objective {   // objective.hpp: (Q, R, Qf, reference_state, reference_states, timestep)
 public:
  QuadraticObjective(const Matrix &Q, const Matrix &R, const Matrix &Qf, const Vector &reference_state,
                     const std::vector<Vector> &reference_states = {}, double timestep = 0.1)
      : Q_(Q), R_(R), Qf_(Qf), reference_state_(reference_state), reference_states_(reference_states), timestep_(timestep) {
    if (Q.rows != Q.cols) throw std::invalid_argument("Q matrix must be square");
    if (R.rows != R.cols) throw std::invalid_argument("R matrix must be square");
    if (Qf.rows != Qf.cols) throw std::invalid_argument("Qf matrix must be square");
    if (!reference_states_.empty()) {   // objective.cpp:55-63
      double n2 = 0; for (size_t i = 0; i < reference_state_.size(); ++i) { double d = reference_states_.back()[i] - reference_state_[i]; n2 += d * d; }
      if (std::sqrt(n2) > 1e-6) throw std::invalid_argument("Last reference state must be same as the reference state");
    }
  }
  Matrix Q_, R_, Qf_; Vector reference_state_; std::vector<Vector> reference_states_; double timestep_;
};

class Constraint {
 public:
  explicit Constraint(std::string name) : name_(std::move(name)) {}
  virtual ~Constraint() = default;
  const std::string &getName() const { return name_; }
  virtual int getDualDim() const = 0;
  virtual void fill(cddp_hip_constraint &c) const = 0;
 protected:
  std::string name_;
};
class ControlConstraint : public Constraint {   // BoxConstraint<Control>, constraint.hpp:144-251
 public:
  ControlConstraint(const Vector &lower, const Vector &upper, double scale = 1.0) : Constraint("ControlConstraint"), lower_(lower), upper_(upper), scale_(scale) {}
  int getDualDim() const override { return 2 * (int)upper_.size(); }
  void fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_CONTROL_BOX; c.dim = (int)upper_.size(); c.lower = lower_.data(); c.upper = upper_.data(); c.scale = scale_; }
  Vector lower_, upper_; double scale_;
};
class StateConstraint : public Constraint {
 public:
  StateConstraint(const Vector &lower, const Vector &upper, double scale = 1.0) : Constraint("StateConstraint"), lower_(lower), upper_(upper), scale_(scale) {}
  int getDualDim() const override { return 2 * (int)upper_.size(); }
  void fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_STATE_BOX; c.dim = (int)upper_.size(); c.lower = lower_.data(); c.upper = upper_.data(); c.scale = scale_; }
  Vector lower_, upper_; double scale_;
};
class BallConstraint : public Constraint {      // constraint.hpp:313-404
 public:
  BallConstraint(double radius, const Vector &center, double scale = 1.0) : Constraint("BallConstraint"), radius_(radius), center_(center), scale_(scale) {}
  int getDualDim() const override { return 1; }
  void fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_BALL; c.dim = (int)center_.size(); c.center = center_.data(); c.radius = radius_; c.scale = scale_; }
  double radius_; Vector center_; double scale_;
};
class LinearConstraint : public Constraint {    // constraint.hpp:253-311
 public:
  LinearConstraint(const Matrix &A, const Vector &b, double scale = 1.0) : Constraint("LinearConstraint"), A_(A), b_(b), scale_(scale) {}
  int getDualDim() const override { return (int)b_.size(); }
  void fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_LINEAR; c.dim = (int)b_.size(); c.A = A_.a.data(); c.b = b_.data(); c.scale = scale_; }
  Matrix A_; Vector b_; double scale_;
};
class TerminalConstraint {
 public:
  virtual ~TerminalConstraint() = default;
  virtual void fill(cddp_hip_terminal_constraint &c) const = 0;
};
class TerminalEqualityConstraint : public TerminalConstraint {   // terminal_constraint.hpp:62-158
 public:
  explicit TerminalEqualityConstraint(const Vector &target) : target_(target) {}
  void fill(cddp_hip_terminal_constraint &c) const override { c.kind = CDDP_HIP_TERM_EQUALITY; c.dim = (int)target_.size(); c.target = target_.data(); }
  Vector target_;
};
class TerminalInequalityConstraint : public TerminalConstraint { // terminal_constraint.hpp:160-263
 public:
  TerminalInequalityConstraint(const Matrix &A_N, const Vector &b_N) : A_(A_N), b_(b_N) {
    if (A_N.rows != (int)b_N.size()) throw std::invalid_argument("TerminalInequalityConstraint: A_N rows and b_N size mismatch.");
  }
  void fill(cddp_hip_terminal_constraint &c) const override { c.kind = CDDP_HIP_TERM_INEQUALITY; c.dim = (int)b_.size(); c.A = A_.a.data(); c.b = b_.data(); }
  Matrix A_; Vector b_;
};

// ---- solution / solver interface (cddp_core.hpp:54-210) ------------------------------------------
struct CDDPSolution {
  std::string solver_name; std::string status_message = "Running";
  int iterations_completed = 0; double solve_time_ms = 0.0, final_objective = 0.0, final_step_length = 0.0, final_regularization = 0.0;
  std::vector<double> time_points; std::vector<Vector> state_trajectory, control_trajectory; std::vector<Matrix> feedback_gains;
  double final_primal_infeasibility = 0.0, final_dual_infeasibility = 0.0, final_complementary_infeasibility = 0.0, final_barrier_mu = 0.0;
  struct History { std::vector<double> objective, merit_function, step_length_primal, step_length_dual, dual_infeasibility, primal_infeasibility, complementary_infeasibility, barrier_mu, regularization; } history;
};
enum class SolverType { CLDDP, LogDDP, IPDDP, MSIPDDP };
class CDDP;
class ISolverAlgorithm {
 public:
  virtual ~ISolverAlgorithm() = default;
  virtual void initialize(CDDP &context) = 0;
  virtual CDDPSolution solve(CDDP &context) = 0;
  virtual std::string getSolverName() const = 0;
};

class CDDP {
 public:
  CDDP(const Vector &initial_state, const Vector &reference_state, int horizon, double timestep,
       std::unique_ptr<DynamicalSystem> system = nullptr, std::unique_ptr<Objective> objective = nullptr,
       const CDDPOptions &options = CDDPOptions())
      : initial_state_(initial_state), reference_state_(reference_state), horizon_(horizon), timestep_(timestep),
        system_(std::move(system)), objective_(std::move(objective)), options_(options) {}

  void setDynamicalSystem(std::unique_ptr<DynamicalSystem> s) { system_ = std::move(s); }
  void setObjective(std::unique_ptr<Objective> o) { objective_ = std::move(o); }
  void setOptions(const CDDPOptions &o) { options_ = o; }
  void setInitialState(const Vector &x0) { initial_state_ = x0; }
  void setHorizon(int h) { horizon_ = h; }
  void setTimestep(double dt) { timestep_ = dt; }
  void setInitialTrajectory(const std::vector<Vector> &X, const std::vector<Vector> &U) { X_ = X; U_ = U; }
  void addPathConstraint(std::string name, std::unique_ptr<Constraint> c) {
    if (!c) throw std::runtime_error("Cannot add null constraint.");   // cddp_context_utils.cpp:82-84
    path_constraint_set_[name] = std::move(c);
  }
  bool removePathConstraint(const std::string &name) { return path_constraint_set_.erase(name) > 0; }
  void addTerminalConstraint(std::string name, std::unique_ptr<TerminalConstraint> c) {
    if (!c) throw std::runtime_error("Cannot add null constraint.");
    terminal_constraint_set_[name] = std::move(c);
  }
  int getTotalDualDim() const { int m = 0; for (auto &kv : path_constraint_set_) m += kv.second->getDualDim(); return m; }
  const CDDPOptions &getOptions() const { return options_; }
  int getHorizon() const { return horizon_; }
  double getTimestep() const { return timestep_; }
  const Vector &getInitialState() const { return initial_state_; }
  const DynamicalSystem &getSystem() const { return *system_; }

  // --- static solver registry (cddp_core.cpp:34-35, 578-595): consulted BEFORE the built-ins
  using Factory = std::function<std::unique_ptr<ISolverAlgorithm>()>;
  static void registerSolver(const std::string &name, Factory f) { registry()[name] = std::move(f); }
  static bool isSolverRegistered(const std::string &name) { return registry().count(name) > 0; }
  static std::vector<std::string> getRegisteredSolvers() { std::vector<std::string> v; for (auto &kv : registry()) v.push_back(kv.first); return v; }

  CDDPSolution solve(SolverType t) { return solve(std::string(t == SolverType::IPDDP ? "IPDDP" : t == SolverType::LogDDP ? "LogDDP" : t == SolverType::MSIPDDP ? "MSIPDDP" : "CLDDP")); }
  CDDPSolution solve(const std::string &solver_type);                                          // cddp_core.cpp:235-270
  std::vector<CDDPSolution> solveBatch(const std::string &solver_type, const std::vector<Vector> &x0s, int device = 0);

  // --- iterate state shared with solver strategies (public in the reference too, cddp_core.hpp:323-342)
  std::vector<Vector> X_, U_;
  double cost_ = 0, merit_function_ = 0, inf_pr_ = 0, inf_du_ = 0, inf_comp_ = 0, step_norm_ = 0, alpha_pr_ = 1.0, alpha_du_ = 0.0, regularization_ = 0;

  // flatten to the C-ABI descriptor (buffers stay owned by *this / scratch)
  struct Flat { cddp_hip_problem p; std::vector<cddp_hip_constraint> cons; std::vector<cddp_hip_terminal_constraint> terms; std::vector<double> xref_traj; };
  void flatten(int solver, Flat &f) const;
  void initializeProblemIfNecessary();

 private:
  static std::map<std::string, Factory> &registry() { static std::map<std::string, Factory> r; return r; }
  Vector initial_state_, reference_state_; int horizon_; double timestep_;
  std::unique_ptr<DynamicalSystem> system_; std::unique_ptr<Objective> objective_; CDDPOptions options_;
  std::map<std::string, std::unique_ptr<Constraint>> path_constraint_set_;          // std::map: name order == dual stacking order
  std::map<std::string, std::unique_ptr<TerminalConstraint>> terminal_constraint_set_;
};

// ---- the GPU solver strategy ----------------------------------------------------------------------
class HipBatchSolver : public ISolverAlgorithm {
 public:
  explicit HipBatchSolver(int solver_kind, int device = 0) : kind_(solver_kind), device_(device) {}
  ~HipBatchSolver() override { if (h_) cddp_hip_destroy(h_); }
  std::string getSolverName() const override { return kind_ == CDDP_HIP_SOLVER_IPDDP ? "IPDDP" : "CLDDP"; }
  // A second initialize() of the SAME solver object with options.warm_start keeps the device-resident solver state
  // (gains, slack / dual / costate variables): the reference's "existing solver state" branch
  // (clddp_solver.cpp:51-60, ipddp_solver.cpp:675-731).  CDDP::solve() creates a new solver per call, exactly as
  // the reference does (cddp_core.cpp:235-270), so through it a warm start is the "provided trajectory" branch.
  void initialize(CDDP &ctx) override {
    if (h_ && ctx.getOptions().warm_start && batch_ == 1 && nx_ == ctx.getSystem().getStateDim() && nu_ == ctx.getSystem().getControlDim() && N_ == ctx.getHorizon()) {
      ctx.initializeProblemIfNecessary();
      cddp_hip_options o = ctx.getOptions().toPOD();
      check(cddp_hip_set_options(h_, &o));
      upload(ctx, {ctx.getInitialState()});
      return;
    }
    create(ctx, {ctx.getInitialState()});
  }
  CDDPSolution solve(CDDP &ctx) override {
    std::vector<CDDPSolution> s = collect(ctx, 1);
    // leave the context updated as the reference solvers do (cddp_solver_base.cpp:161-171)
    ctx.X_ = s[0].state_trajectory; ctx.U_ = s[0].control_trajectory; ctx.cost_ = s[0].final_objective;
    ctx.alpha_pr_ = s[0].final_step_length; ctx.regularization_ = s[0].final_regularization;
    ctx.inf_pr_ = s[0].final_primal_infeasibility; ctx.inf_du_ = s[0].final_dual_infeasibility; ctx.inf_comp_ = s[0].final_complementary_infeasibility;
    return s[0];
  }
  std::vector<CDDPSolution> solveBatch(CDDP &ctx, const std::vector<Vector> &x0s) { create(ctx, x0s); return collect(ctx, (int)x0s.size()); }
  cddp_hip_stats stats{};

  static void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("cddp_hip: ") + cddp_hip_last_error()); }

 private:
  void create(CDDP &ctx, const std::vector<Vector> &x0s) {
    ctx.initializeProblemIfNecessary();
    if (h_) { cddp_hip_destroy(h_); h_ = nullptr; }
    CDDP::Flat f; ctx.flatten(kind_, f);
    const int B = (int)x0s.size(), nx = f.p.nx, nu = f.p.nu, N = f.p.horizon;
    check(cddp_hip_create(&f.p, B, device_, &h_));
    nx_ = nx; nu_ = nu; N_ = N; dt_ = f.p.dt; ret_hist_ = f.p.options.return_iteration_info; max_it_ = f.p.options.max_iterations; batch_ = B;
    upload(ctx, x0s);
  }
  void upload(CDDP &ctx, const std::vector<Vector> &x0s) {   // CDDP::setInitialState + setInitialTrajectory for the batch
    const int B = (int)x0s.size(), nx = nx_, nu = nu_, N = N_;
    std::vector<double> x0((size_t)B * nx), U0, X0;
    for (int b = 0; b < B; ++b) for (int i = 0; i < nx; ++i) x0[(size_t)b * nx + i] = x0s[b][i];
    if ((int)ctx.U_.size() == N) { U0.resize((size_t)B * N * nu); for (int b = 0; b < B; ++b) for (int t = 0; t < N; ++t) for (int i = 0; i < nu; ++i) U0[((size_t)b * N + t) * nu + i] = ctx.U_[t][i]; }
    if ((int)ctx.X_.size() == N + 1) { X0.resize((size_t)B * (N + 1) * nx); for (int b = 0; b < B; ++b) for (int t = 0; t <= N; ++t) for (int i = 0; i < nx; ++i) X0[((size_t)b * (N + 1) + t) * nx + i] = ctx.X_[t][i]; }
    check(cddp_hip_set_initial(h_, x0.data(), U0.empty() ? nullptr : U0.data(), X0.empty() ? nullptr : X0.data()));
  }
  std::vector<CDDPSolution> collect(CDDP &, int B) {
    check(cddp_hip_solve(h_, &stats));
    std::vector<cddp_hip_result> r(B);
    check(cddp_hip_get_results(h_, r.data()));
    std::vector<double> X((size_t)B * (N_ + 1) * nx_), U((size_t)B * N_ * nu_), K((size_t)B * N_ * nu_ * nx_);
    check(cddp_hip_get_trajectory(h_, X.data(), U.data()));
    check(cddp_hip_get_gains(h_, K.data(), nullptr));
    std::vector<double> hist; std::vector<int32_t> hn;
    const int HB = ret_hist_ ? std::min(B, 64) : 0;
    if (HB) { hist.resize((size_t)HB * (max_it_ + 1) * 9); hn.resize(HB); check(cddp_hip_get_history(h_, HB, hist.data(), hn.data())); }
    std::vector<CDDPSolution> out(B);
    for (int b = 0; b < B; ++b) {
      CDDPSolution &s = out[b];
      s.solver_name = getSolverName(); s.status_message = cddp_hip_status_string(r[b].status);
      s.iterations_completed = r[b].iterations; s.solve_time_ms = stats.solve_ms; s.final_objective = r[b].final_objective;
      s.final_step_length = r[b].alpha_pr; s.final_regularization = r[b].regularization;
      s.final_primal_infeasibility = r[b].inf_pr; s.final_dual_infeasibility = r[b].inf_du; s.final_complementary_infeasibility = r[b].inf_comp; s.final_barrier_mu = r[b].barrier_mu;
      for (int t = 0; t <= N_; ++t) { s.time_points.push_back(t * dt_); s.state_trajectory.emplace_back(X.begin() + ((size_t)b * (N_ + 1) + t) * nx_, X.begin() + ((size_t)b * (N_ + 1) + t + 1) * nx_); }
      for (int t = 0; t < N_; ++t) {
        s.control_trajectory.emplace_back(U.begin() + ((size_t)b * N_ + t) * nu_, U.begin() + ((size_t)b * N_ + t + 1) * nu_);
        Matrix Kt(nu_, nx_); std::memcpy(Kt.a.data(), K.data() + ((size_t)b * N_ + t) * nu_ * nx_, sizeof(double) * nu_ * nx_); s.feedback_gains.push_back(Kt);
      }
      if (b < HB) for (int i = 0; i < hn[b]; ++i) {
        const double *row = hist.data() + ((size_t)b * (max_it_ + 1) + i) * 9;
        s.history.objective.push_back(row[0]); s.history.merit_function.push_back(row[1]); s.history.step_length_primal.push_back(row[2]);
        s.history.step_length_dual.push_back(row[3]); s.history.dual_infeasibility.push_back(row[4]); s.history.primal_infeasibility.push_back(row[5]);
        s.history.complementary_infeasibility.push_back(row[6]); if (kind_ == CDDP_HIP_SOLVER_IPDDP) s.history.barrier_mu.push_back(row[7]); s.history.regularization.push_back(row[8]);
      }
    }
    return out;
  }
  int kind_, device_; cddp_hip_handle *h_ = nullptr; int nx_ = 0, nu_ = 0, N_ = 0, max_it_ = 0, batch_ = 0; double dt_ = 0; bool ret_hist_ = false;
};

// Register the GPU core under the reference's own solver names: a true drop-in (cddp_core.cpp:215-219).
inline void registerHipSolvers(int device = 0) {
  CDDP::registerSolver("IPDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_IPDDP, device); });
  CDDP::registerSolver("CLDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_CLDDP, device); });
  CDDP::registerSolver("CLCDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_CLDDP, device); });
}

inline void CDDP::initializeProblemIfNecessary() {   // cddp_core.cpp:272-306
  if (!system_) throw std::runtime_error("Dynamical system must be set before solving.");
  if (!objective_) throw std::runtime_error("Objective function must be set before solving.");
  const double inf = std::numeric_limits<double>::infinity();
  cost_ = merit_function_ = inf_pr_ = inf_du_ = inf_comp_ = inf;
  regularization_ = options_.regularization.initial_value;
}

inline void CDDP::flatten(int solver, Flat &f) const {
  std::memset(&f.p, 0, sizeof(f.p));
  const auto *qo = dynamic_cast<const QuadraticObjective *>(objective_.get());
  if (!qo) throw std::runtime_error("HipBatchSolver: only QuadraticObjective runs on the device (use the stack-fed entry point for other objectives)");
  cddp_hip_problem &p = f.p;
  p.abi_version = CDDP_HIP_ABI_VERSION; p.solver = solver; p.model = system_->modelId();
  p.integrator = integratorId(system_->getIntegrationType());
  if (p.integrator < 0) throw std::runtime_error("Integration type not supported!");
  p.nx = system_->getStateDim(); p.nu = system_->getControlDim(); p.horizon = horizon_; p.dt = timestep_;
  for (size_t i = 0; i < system_->params.size() && i < CDDP_HIP_MAX_MODEL_PARAMS; ++i) p.model_params[i] = system_->params[i];
  if (p.model == CDDP_HIP_MODEL_LTI) { p.lti_A = system_->lti_A.a.data(); p.lti_B = system_->lti_B.a.data(); }
  p.Q = qo->Q_.a.data(); p.R = qo->R_.a.data(); p.Qf = qo->Qf_.a.data(); p.x_ref = qo->reference_state_.data();
  if (!qo->reference_states_.empty()) { for (auto &v : qo->reference_states_) f.xref_traj.insert(f.xref_traj.end(), v.begin(), v.end()); p.x_ref_traj = f.xref_traj.data(); }
  for (auto &kv : path_constraint_set_) {
    cddp_hip_constraint c; std::memset(&c, 0, sizeof(c)); std::strncpy(c.name, kv.first.c_str(), CDDP_HIP_NAME_LEN - 1); c.scale = 1.0; kv.second->fill(c); f.cons.push_back(c);
  }
  for (auto &kv : terminal_constraint_set_) {
    cddp_hip_terminal_constraint c; std::memset(&c, 0, sizeof(c)); std::strncpy(c.name, kv.first.c_str(), CDDP_HIP_NAME_LEN - 1); kv.second->fill(c); f.terms.push_back(c);
  }
  p.n_constraints = (int)f.cons.size(); p.constraints = f.cons.empty() ? nullptr : f.cons.data();
  p.n_terminal = (int)f.terms.size(); p.terminal = f.terms.empty() ? nullptr : f.terms.data();
  p.options = options_.toPOD();
}

inline CDDPSolution CDDP::solve(const std::string &solver_type) {
  initializeProblemIfNecessary();
  std::unique_ptr<ISolverAlgorithm> s;
  auto it = registry().find(solver_type);
  if (it != registry().end()) s = it->second();
  if (!s) {   // cddp_core.cpp:243-265: no throw, status string carries the error
    CDDPSolution sol; sol.solver_name = solver_type;
    sol.status_message = "UnknownSolver - No solver registered for '" + solver_type + "'";
    sol.iterations_completed = 0; sol.solve_time_ms = 0.0; sol.final_objective = 0.0; sol.final_step_length = 1.0;
    return sol;
  }
  s->initialize(*this);
  return s->solve(*this);
}

inline std::vector<CDDPSolution> CDDP::solveBatch(const std::string &solver_type, const std::vector<Vector> &x0s, int device) {
  initializeProblemIfNecessary();
  const int kind = (solver_type == "IPDDP") ? CDDP_HIP_SOLVER_IPDDP : (solver_type == "CLDDP" || solver_type == "CLCDDP") ? CDDP_HIP_SOLVER_CLDDP : -1;
  if (kind < 0) throw std::runtime_error("UnknownSolver - No solver registered for '" + solver_type + "'");
  HipBatchSolver s(kind, device);
  return s.solveBatch(*this, x0s);
}

}  // namespace cddp
