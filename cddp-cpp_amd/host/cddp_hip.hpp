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
//   * The built-in DynamicalSystem / Objective / Constraint classes are DESCRIPTORS of plug-ins the kernels implement
//     on the device (enumerated by id).  A user subclass overrides the reference's virtual functions
//     (getDiscreteDynamics / getStateJacobian / running_cost / evaluate ...) and is served by the host plug-in solve
//     (cddp_hip_plugin_solve: batched backward passes on the GPU, forward passes through the virtuals on the host);
//     there is no autodiff here, so a plant must provide its Jacobians (and Hessians when use_ilqr is false).
//   * "CLDDP" and "IPDDP" are served by HipBatchSolver through the same static registry the reference
//     uses (CDDP::registerSolver, cddp_core.cpp:578-595); other names return the reference's
//     "UnknownSolver - No solver registered for '<name>'" solution (cddp_core.cpp:243-265).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <algorithm>
#include <functional>
#include <limits>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeinfo>
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
struct MSIPDDPAlgorithmOptions {   // options.hpp:110-130, 190: InteriorPointOptions + MultiShootingOptions
  double dual_var_init_scale = 1e-1, slack_var_init_scale = 1e-2; SolverSpecificBarrierOptions barrier;
  int segment_length = 5; std::string rollout_type = "nonlinear"; bool use_controlled_rollout = false; double costate_var_init_scale = 1e-6;
};
struct LogBarrierOptions { bool use_relaxed_log_barrier_penalty = false; double relaxed_log_barrier_delta = 1e-10; SolverSpecificBarrierOptions barrier; };   // options.hpp:135-143
struct CDDPOptions {
  double tolerance = 1e-5, acceptable_tolerance = 1e-6; int max_iterations = 1; double max_cpu_time = 0.0;
  bool verbose = true, debug = false, print_solver_header = true, print_solver_options = false, use_ilqr = true, enable_parallel = false;
  int num_threads = 1; bool return_iteration_info = false, warm_start = false; double termination_scaling_max_factor = 100.0;
  LineSearchOptions line_search; RegularizationOptions regularization; BoxQPOptions box_qp; SolverSpecificFilterOptions filter; IPDDPAlgorithmOptions ipddp;
  LogBarrierOptions log_barrier;
  MSIPDDPAlgorithmOptions msipddp;

  // msipddp = true: the InteriorPointOptions half of options.msipddp travels in the ipddp_* / barrier_* fields (include/cddp_hip.h)
  cddp_hip_options toPOD(bool for_msipddp = false) const {
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
    o.logddp_mu_initial = log_barrier.barrier.mu_initial; o.logddp_mu_min_value = log_barrier.barrier.mu_min_value;
    o.logddp_mu_update_factor = log_barrier.barrier.mu_update_factor; o.logddp_relaxed_delta = log_barrier.relaxed_log_barrier_delta;
    o.msipddp_costate_var_init_scale = msipddp.costate_var_init_scale; o.msipddp_segment_length = msipddp.segment_length;
    o.msipddp_rollout_type = msipddp.rollout_type == "nonlinear" ? 0 : msipddp.rollout_type == "hybrid" ? 2 : 1;
    o.msipddp_use_controlled_rollout = msipddp.use_controlled_rollout;
    if (for_msipddp) {
      o.ipddp_dual_var_init_scale = msipddp.dual_var_init_scale; o.ipddp_slack_var_init_scale = msipddp.slack_var_init_scale;
      o.barrier_mu_initial = msipddp.barrier.mu_initial; o.barrier_mu_min_value = msipddp.barrier.mu_min_value; o.barrier_mu_update_factor = msipddp.barrier.mu_update_factor;
      o.barrier_mu_update_power = msipddp.barrier.mu_update_power; o.barrier_min_fraction_to_boundary = msipddp.barrier.min_fraction_to_boundary;
      o.barrier_strategy = (int)msipddp.barrier.strategy;
    }
    return o;
  }
};

// ---- plug-in descriptors --------------------------------------------------------------------
inline int integratorId(const std::string &s) {
  if (s == "euler") return CDDP_HIP_EULER;
  if (s == "heun") return CDDP_HIP_HEUN;
  if (s == "rk3") return CDDP_HIP_RK3;
  if (s == "rk4") return CDDP_HIP_RK4;
  return -1;   // reference prints "Integration type not supported!" and returns zeros (dynamical_system.cpp:79-82)
}
class DynamicalSystem {
 public:
  DynamicalSystem(int model, int nx, int nu, double timestep, std::string integration_type)
      : model_(model), state_dim_(nx), control_dim_(nu), timestep_(timestep), integration_type_(std::move(integration_type)) {}
  // a user-defined (host) plant: override the virtuals below (dynamical_system.hpp:30-117)
  DynamicalSystem(int nx, int nu, double timestep, std::string integration_type)
      : DynamicalSystem(-1, nx, nu, timestep, std::move(integration_type)) {}
  virtual ~DynamicalSystem() = default;
  int getStateDim() const { return state_dim_; }
  int getControlDim() const { return control_dim_; }
  double getTimestep() const { return timestep_; }
  const std::string &getIntegrationType() const { return integration_type_; }
  int modelId() const { return model_; }
  bool isHostPlant() const { return model_ < 0; }
  std::vector<double> params;         // cddp_hip_problem::model_params
  Matrix lti_A, lti_B;

  virtual Vector getContinuousDynamics(const Vector &, const Vector &, double) const {
    throw std::runtime_error("getContinuousDynamics is not implemented for this DynamicalSystem");
  }
  virtual Vector getDiscreteDynamics(const Vector &x, const Vector &u, double t) const {   // dynamical_system.cpp:28-83
    if (!isHostPlant()) { Vector r(state_dim_); builtinEval(x, u, r.data(), nullptr, nullptr, nullptr, nullptr, nullptr); return r; }
    const double dt = timestep_; const int n = state_dim_;
    auto axpy = [n](const Vector &a, double s, const Vector &b) { Vector r(n); for (int i = 0; i < n; ++i) r[i] = a[i] + s * b[i]; return r; };
    const Vector k1 = getContinuousDynamics(x, u, t);
    Vector r(n);
    if (integration_type_ == "euler") return axpy(x, dt, k1);
    if (integration_type_ == "heun") {
      const Vector k2 = getContinuousDynamics(axpy(x, dt, k1), u, t + dt);
      for (int i = 0; i < n; ++i) r[i] = x[i] + (0.5 * dt) * (k1[i] + k2[i]);
      return r;
    }
    if (integration_type_ == "rk3") {
      const Vector k2 = getContinuousDynamics(axpy(x, 0.5 * dt, k1), u, t + 0.5 * dt);
      Vector x3(n); for (int i = 0; i < n; ++i) x3[i] = (x[i] - dt * k1[i]) + (2 * dt) * k2[i];
      const Vector k3 = getContinuousDynamics(x3, u, t + dt);
      for (int i = 0; i < n; ++i) r[i] = x[i] + (dt / 6) * ((k1[i] + 4.0 * k2[i]) + k3[i]);
      return r;
    }
    if (integration_type_ == "rk4") {
      const Vector k2 = getContinuousDynamics(axpy(x, 0.5 * dt, k1), u, t + 0.5 * dt);
      const Vector k3 = getContinuousDynamics(axpy(x, 0.5 * dt, k2), u, t + 0.5 * dt);
      const Vector k4 = getContinuousDynamics(axpy(x, dt, k3), u, t + dt);
      for (int i = 0; i < n; ++i) r[i] = x[i] + (dt / 6) * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
      return r;
    }
    throw std::runtime_error("Integration type not supported!");
  }
  // CONTINUOUS-time Jacobians / Hessians, as in the reference (the solver forms A = I + dt f_x, B = dt f_u)
  // (a built-in plant evaluates the kernels' model source compiled for the host: cddp_hip_model_eval)
  virtual Matrix getStateJacobian(const Vector &x, const Vector &u, double) const {
    if (isHostPlant()) noAutodiff("getStateJacobian");
    Matrix fx(state_dim_, state_dim_); builtinEval(x, u, nullptr, fx.a.data(), nullptr, nullptr, nullptr, nullptr); return fx;
  }
  virtual Matrix getControlJacobian(const Vector &x, const Vector &u, double) const {
    if (isHostPlant()) noAutodiff("getControlJacobian");
    Matrix fu(state_dim_, control_dim_); builtinEval(x, u, nullptr, nullptr, fu.a.data(), nullptr, nullptr, nullptr); return fu;
  }
  virtual std::vector<Matrix> getStateHessian(const Vector &x, const Vector &u, double) const { if (isHostPlant()) noAutodiff("getStateHessian"); return builtinHess(x, u, 0); }
  virtual std::vector<Matrix> getControlHessian(const Vector &x, const Vector &u, double) const { if (isHostPlant()) noAutodiff("getControlHessian"); return builtinHess(x, u, 1); }
  virtual std::vector<Matrix> getCrossHessian(const Vector &x, const Vector &u, double) const { if (isHostPlant()) noAutodiff("getCrossHessian"); return builtinHess(x, u, 2); }
 protected:
  void builtinEval(const Vector &x, const Vector &u, double *xn, double *fx, double *fu, double *fxx, double *fuu, double *fux) const {
    double mp[CDDP_HIP_MAX_MODEL_PARAMS] = {0};
    for (size_t i = 0; i < params.size() && i < CDDP_HIP_MAX_MODEL_PARAMS; ++i) mp[i] = params[i];
    if ((int)x.size() != state_dim_ || (int)u.size() != control_dim_) throw std::invalid_argument("DynamicalSystem: state / control size mismatch");
    int integ = -1;
    if (integration_type_ == "euler") integ = CDDP_HIP_EULER; else if (integration_type_ == "heun") integ = CDDP_HIP_HEUN;
    else if (integration_type_ == "rk3") integ = CDDP_HIP_RK3; else if (integration_type_ == "rk4") integ = CDDP_HIP_RK4;
    if (cddp_hip_model_eval(model_, integ, timestep_, mp, state_dim_, control_dim_, x.data(), u.data(), xn, fx, fu, fxx, fuu, fux) != 0)
      throw std::runtime_error(std::string("cddp_hip: ") + cddp_hip_last_error());
  }
  std::vector<Matrix> builtinHess(const Vector &x, const Vector &u, int which) const {
    const int nx = state_dim_, nu = control_dim_;
    std::vector<double> fxx((size_t)nx * nx * nx), fuu((size_t)nx * nu * nu), fux((size_t)nx * nu * nx);
    builtinEval(x, u, nullptr, nullptr, nullptr, fxx.data(), fuu.data(), fux.data());
    const int r = which == 0 ? nx : nu, c = which == 1 ? nu : nx;
    const std::vector<double> &src = which == 0 ? fxx : which == 1 ? fuu : fux;
    std::vector<Matrix> out(nx, Matrix(r, c));
    for (int i = 0; i < nx; ++i) std::memcpy(out[i].a.data(), src.data() + (size_t)i * r * c, sizeof(double) * r * c);
    return out;
  }
  static void noAutodiff(const char *what) {
    throw std::runtime_error(std::string("DynamicalSystem::") + what + ": autodiff is not available in the HIP mirror; override it in the subclass");
  }
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
  // host evaluation (lti_system.cpp:92-135), used when the LTI plant is paired with a host Objective / Constraint
  Vector getDiscreteDynamics(const Vector &x, const Vector &u, double) const override {
    Vector r(state_dim_, 0.0);
    for (int i = 0; i < state_dim_; ++i) { double a = 0.0, b = 0.0; for (int j = 0; j < state_dim_; ++j) a += lti_A(i, j) * x[j]; for (int j = 0; j < control_dim_; ++j) b += lti_B(i, j) * u[j]; r[i] = a + b; }
    return r;
  }
  Matrix getStateJacobian(const Vector &, const Vector &, double) const override {
    Matrix m = lti_A; for (int i = 0; i < state_dim_; ++i) m(i, i) -= 1.0; for (double &v : m.a) v /= timestep_; return m;
  }
  Matrix getControlJacobian(const Vector &, const Vector &, double) const override { Matrix m = lti_B; for (double &v : m.a) v /= timestep_; return m; }
  std::vector<Matrix> getStateHessian(const Vector &, const Vector &, double) const override { return std::vector<Matrix>(state_dim_, Matrix(state_dim_, state_dim_)); }
  std::vector<Matrix> getControlHessian(const Vector &, const Vector &, double) const override { return std::vector<Matrix>(state_dim_, Matrix(control_dim_, control_dim_)); }
  std::vector<Matrix> getCrossHessian(const Vector &, const Vector &, double) const override { return std::vector<Matrix>(state_dim_, Matrix(control_dim_, state_dim_)); }
};
struct Quadrotor : DynamicalSystem {  // quadrotor.hpp: (timestep, mass, inertia_matrix(diag), arm_length, integration_type)
  Quadrotor(double dt, double mass, const Matrix &inertia, double arm_length, std::string integ = "rk4")
      : DynamicalSystem(CDDP_HIP_MODEL_QUADROTOR, 13, 4, dt, integ) { params = {mass, arm_length, inertia(0, 0), inertia(1, 1), inertia(2, 2), 9.81}; }
};
struct HCW : DynamicalSystem {        // spacecraft_linear.hpp:33: (timestep, mean_motion, mass, integration_type); state [x, y, z, vx, vy, vz], control [Fx, Fy, Fz]
  HCW(double dt, double mean_motion, double mass, std::string integ = "euler") : DynamicalSystem(CDDP_HIP_MODEL_HCW, 6, 3, dt, integ) { params = {mean_motion, mass}; }
};
struct Bicycle : DynamicalSystem {    // bicycle.hpp: (timestep, wheelbase, integration_type); state [x, y, theta, v], control [a, delta]
  Bicycle(double dt, double wheelbase, std::string integ = "euler") : DynamicalSystem(CDDP_HIP_MODEL_BICYCLE, 4, 2, dt, integ) { params = {wheelbase}; }
};
struct Car : DynamicalSystem {        // car.hpp: (timestep, wheelbase, integration_type); a DISCRETE plant (car.cpp:24-60); control [delta, a]
  Car(double dt = 0.03, double wheelbase = 2.0, std::string integ = "euler") : DynamicalSystem(CDDP_HIP_MODEL_CAR, 4, 2, dt, integ) { params = {wheelbase}; }
};
struct Manipulator : DynamicalSystem {
  Manipulator(double dt, std::string integ = "rk4") : DynamicalSystem(CDDP_HIP_MODEL_MANIPULATOR, 6, 3, dt, integ) {}
};

class Objective {                       // objective.hpp:30-120
 public:
  virtual ~Objective() = default;
  virtual double running_cost(const Vector &x, const Vector &u, int index) const = 0;
  virtual double terminal_cost(const Vector &x) const = 0;
  virtual double evaluate(const std::vector<Vector> &X, const std::vector<Vector> &U) const {
    double J = 0.0; for (size_t t = 0; t < U.size(); ++t) J += running_cost(X[t], U[t], (int)t);
    return J + terminal_cost(X.back());
  }
  virtual Vector getRunningCostStateGradient(const Vector &x, const Vector &u, int index) const = 0;
  virtual Vector getRunningCostControlGradient(const Vector &x, const Vector &u, int index) const = 0;
  virtual Vector getFinalCostGradient(const Vector &x) const = 0;
  virtual Matrix getRunningCostStateHessian(const Vector &x, const Vector &u, int index) const = 0;
  virtual Matrix getRunningCostControlHessian(const Vector &x, const Vector &u, int index) const = 0;
  virtual Matrix getRunningCostCrossHessian(const Vector &x, const Vector &u, int index) const = 0;   // nu x nx
  virtual Matrix getFinalCostHessian(const Vector &x) const = 0;
  // objective.hpp:60-75: the context pushes its reference state / trajectory into the objective (cddp_core.cpp:56-60, 75-100, 115-124)
  virtual void setReferenceState(const Vector &) {}
  virtual void setReferenceStates(const std::vector<Vector> &) {}
  virtual Vector getReferenceState() const { return {}; }
};
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
  // host evaluation (objective.cpp:66-154): the constructor scales Q and R by the timestep
  double running_cost(const Vector &x, const Vector &u, int index) const override {
    const Vector &r = ref(index); const int nx = Q_.rows, nu = R_.rows;
    double qx = 0.0, ru = 0.0;
    for (int i = 0; i < nx; ++i) { double a = 0.0; for (int j = 0; j < nx; ++j) a += (Q_(i, j) * timestep_) * (x[j] - r[j]); qx += (x[i] - r[i]) * a; }
    for (int i = 0; i < nu; ++i) { double a = 0.0; for (int j = 0; j < nu; ++j) a += (R_(i, j) * timestep_) * u[j]; ru += u[i] * a; }
    return qx + ru;
  }
  double terminal_cost(const Vector &x) const override {
    const int nx = Qf_.rows; double q = 0.0;
    for (int i = 0; i < nx; ++i) { double a = 0.0; for (int j = 0; j < nx; ++j) a += Qf_(i, j) * (x[j] - reference_state_[j]); q += (x[i] - reference_state_[i]) * a; }
    return q;
  }
  Vector getRunningCostStateGradient(const Vector &x, const Vector &, int index) const override {
    const Vector &r = ref(index); Vector g(Q_.rows);
    for (int i = 0; i < Q_.rows; ++i) { double a = 0.0; for (int j = 0; j < Q_.rows; ++j) a += (Q_(i, j) * timestep_) * (x[j] - r[j]); g[i] = 2.0 * a; }
    return g;
  }
  Vector getRunningCostControlGradient(const Vector &, const Vector &u, int) const override {
    Vector g(R_.rows);
    for (int i = 0; i < R_.rows; ++i) { double a = 0.0; for (int j = 0; j < R_.rows; ++j) a += (R_(i, j) * timestep_) * u[j]; g[i] = 2.0 * a; }
    return g;
  }
  Vector getFinalCostGradient(const Vector &x) const override {
    Vector g(Qf_.rows);
    for (int i = 0; i < Qf_.rows; ++i) { double a = 0.0; for (int j = 0; j < Qf_.rows; ++j) a += Qf_(i, j) * (x[j] - reference_state_[j]); g[i] = 2.0 * a; }
    return g;
  }
  Matrix getRunningCostStateHessian(const Vector &, const Vector &, int) const override { return (Q_ * timestep_) * 2.0; }
  Matrix getRunningCostControlHessian(const Vector &, const Vector &, int) const override { return (R_ * timestep_) * 2.0; }
  Matrix getRunningCostCrossHessian(const Vector &, const Vector &, int) const override { return Matrix(R_.rows, Q_.rows); }
  Matrix getFinalCostHessian(const Vector &) const override { return Qf_ * 2.0; }
  void setReferenceState(const Vector &r) override { reference_state_ = r; }                      // objective.hpp:170-176
  void setReferenceStates(const std::vector<Vector> &rs) override { reference_states_ = rs; }
  Vector getReferenceState() const override { return reference_state_; }
  Matrix Q_, R_, Qf_; Vector reference_state_; std::vector<Vector> reference_states_; double timestep_;
 private:
  const Vector &ref(int index) const { return reference_states_.empty() ? reference_state_ : reference_states_[index]; }
};
// NonlinearObjective (objective.cpp:156-288): derivatives by central finite differences unless overridden
class NonlinearObjective : public Objective {
 public:
  explicit NonlinearObjective(double timestep = 0.1) : timestep_(timestep) {}
  double running_cost(const Vector &, const Vector &, int) const override { return 0.0; }
  double terminal_cost(const Vector &) const override { return 0.0; }
  Vector getRunningCostStateGradient(const Vector &x, const Vector &u, int k) const override { return fdGradient([&](const Vector &s) { return running_cost(s, u, k); }, x); }
  Vector getRunningCostControlGradient(const Vector &x, const Vector &u, int k) const override { return fdGradient([&](const Vector &c) { return running_cost(x, c, k); }, u); }
  Vector getFinalCostGradient(const Vector &x) const override { return fdGradient([&](const Vector &s) { return terminal_cost(s); }, x); }
  Matrix getRunningCostStateHessian(const Vector &x, const Vector &u, int k) const override { return fdHessian([&](const Vector &s) { return running_cost(s, u, k); }, x); }
  Matrix getRunningCostControlHessian(const Vector &x, const Vector &u, int k) const override { return fdHessian([&](const Vector &c) { return running_cost(x, c, k); }, u); }
  Matrix getFinalCostHessian(const Vector &x) const override { return fdHessian([&](const Vector &s) { return terminal_cost(s); }, x); }
  Matrix getRunningCostCrossHessian(const Vector &x, const Vector &u, int k) const override {   // objective.cpp:245-277: h = 2e-8
    const double h = 2e-8; const int nx = (int)x.size(), nu = (int)u.size();
    Matrix H(nu, nx); Vector xp = x, up = u;
    for (int i = 0; i < nu; ++i) for (int j = 0; j < nx; ++j) {
      up[i] = u[i] + h; xp[j] = x[j] + h; const double fpp = running_cost(xp, up, k);
      xp[j] = x[j] - h; const double fpm = running_cost(xp, up, k);
      up[i] = u[i] - h; const double fmm = running_cost(xp, up, k);
      xp[j] = x[j] + h; const double fmp = running_cost(xp, up, k);
      H(i, j) = (((fpp - fpm) - fmp) + fmm) / (4.0 * h * h); up[i] = u[i]; xp[j] = x[j];
    }
    return H;
  }
  template <class F> static Vector fdGradient(F f, const Vector &x, double h = 2e-5) {   // helper.hpp:34-55
    Vector g(x.size()), xp = x;
    for (size_t i = 0; i < x.size(); ++i) { xp[i] = x[i] + h; const double fp = f(xp); xp[i] = x[i] - h; const double fm = f(xp); g[i] = (fp - fm) / (2.0 * h); xp[i] = x[i]; }
    return g;
  }
  template <class F> static Matrix fdHessian(F f, const Vector &x, double h = 2e-5) {    // helper.hpp:158-179
    const int n = (int)x.size(); Matrix H(n, n); Vector xp = x;
    for (int i = 0; i < n; ++i) {
      xp[i] = x[i] + h; const Vector gp = fdGradient(f, xp, h); xp[i] = x[i] - h; const Vector gm = fdGradient(f, xp, h); xp[i] = x[i];
      for (int r = 0; r < n; ++r) H(r, i) = (gp[r] - gm[r]) / (2.0 * h);
    }
    return H;
  }
 protected:
  double timestep_;
};

class Constraint {                      // constraint.hpp:40-142
 public:
  explicit Constraint(std::string name) : name_(std::move(name)) {}
  virtual ~Constraint() = default;
  const std::string &getName() const { return name_; }
  virtual int getDualDim() const = 0;
  virtual Vector evaluate(const Vector &x, const Vector &u) const = 0;
  virtual Vector getUpperBound() const = 0;
  virtual Matrix getStateJacobian(const Vector &x, const Vector &u) const = 0;     // m x nx
  virtual Matrix getControlJacobian(const Vector &x, const Vector &u) const = 0;   // m x nu
  // second derivatives of the rows (constraint.hpp:86-120: zero matrices by default); `false` = "not provided" (the reference's
  // std::logic_error).  Used by LogDDP's relaxed log barrier only.  gxx[r] nx*nx, guu[r] nu*nu, gux[r] nu*nx, zero-filled by the caller.
  virtual bool getHessians(const Vector &, const Vector &, double * /*gxx*/, double * /*guu*/, double * /*gux*/) const { return true; }
  // device descriptor: true when a kernel implements this constraint; a user subclass keeps the default
  virtual bool fill(cddp_hip_constraint &) const { return false; }
 protected:
  std::string name_;
};
namespace detail {
inline Vector boxEval(const Vector &v, double s) { const size_t n = v.size(); Vector g(2 * n); for (size_t i = 0; i < n; ++i) { g[i] = -v[i] * s; g[n + i] = v[i] * s; } return g; }
inline Vector boxUpper(const Vector &lo, const Vector &up, double s) { const size_t n = up.size(); Vector g(2 * n); for (size_t i = 0; i < n; ++i) { g[i] = -lo[i] * s; g[n + i] = up[i] * s; } return g; }
inline Matrix boxJac(int n, double s) { Matrix J(2 * n, n); for (int i = 0; i < n; ++i) { J(i, i) = -s; J(n + i, i) = s; } return J; }
}  // namespace detail
class ControlConstraint : public Constraint {   // BoxConstraint<Control>, constraint.hpp:144-251
 public:
  ControlConstraint(const Vector &lower, const Vector &upper, double scale = 1.0) : Constraint("ControlConstraint"), lower_(lower), upper_(upper), scale_(scale) {}
  int getDualDim() const override { return 2 * (int)upper_.size(); }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_CONTROL_BOX; c.dim = (int)upper_.size(); c.lower = lower_.data(); c.upper = upper_.data(); c.scale = scale_; return true; }
  Vector evaluate(const Vector &, const Vector &u) const override { return detail::boxEval(u, scale_); }
  Vector getUpperBound() const override { return detail::boxUpper(lower_, upper_, scale_); }
  Matrix getStateJacobian(const Vector &x, const Vector &) const override { return Matrix(getDualDim(), (int)x.size()); }
  Matrix getControlJacobian(const Vector &, const Vector &) const override { return detail::boxJac((int)upper_.size(), scale_); }
  Vector lower_, upper_; double scale_;
};
class StateConstraint : public Constraint {
 public:
  StateConstraint(const Vector &lower, const Vector &upper, double scale = 1.0) : Constraint("StateConstraint"), lower_(lower), upper_(upper), scale_(scale) {}
  int getDualDim() const override { return 2 * (int)upper_.size(); }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_STATE_BOX; c.dim = (int)upper_.size(); c.lower = lower_.data(); c.upper = upper_.data(); c.scale = scale_; return true; }
  Vector evaluate(const Vector &x, const Vector &) const override { return detail::boxEval(x, scale_); }
  Vector getUpperBound() const override { return detail::boxUpper(lower_, upper_, scale_); }
  Matrix getStateJacobian(const Vector &, const Vector &) const override { return detail::boxJac((int)upper_.size(), scale_); }
  Matrix getControlJacobian(const Vector &, const Vector &u) const override { return Matrix(getDualDim(), (int)u.size()); }
  Vector lower_, upper_; double scale_;
};
class BallConstraint : public Constraint {      // constraint.hpp:313-404
 public:
  BallConstraint(double radius, const Vector &center, double scale = 1.0) : Constraint("BallConstraint"), radius_(radius), center_(center), scale_(scale) {}
  int getDualDim() const override { return 1; }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_BALL; c.dim = (int)center_.size(); c.center = center_.data(); c.radius = radius_; c.scale = scale_; return true; }
  Vector evaluate(const Vector &x, const Vector &) const override { double d2 = 0.0; for (size_t i = 0; i < center_.size(); ++i) { const double d = x[i] - center_[i]; d2 += d * d; } return {-scale_ * d2}; }
  Vector getUpperBound() const override { return {-scale_ * radius_ * radius_}; }
  Matrix getStateJacobian(const Vector &x, const Vector &) const override { Matrix J(1, (int)x.size()); for (size_t i = 0; i < center_.size(); ++i) J(0, (int)i) = -2.0 * scale_ * (x[i] - center_[i]); return J; }
  Matrix getControlJacobian(const Vector &, const Vector &u) const override { return Matrix(1, (int)u.size()); }
  bool getHessians(const Vector &x, const Vector &, double *gxx, double *, double *) const override {   // constraint.hpp:387-396
    const int nx = (int)x.size(); for (size_t i = 0; i < center_.size(); ++i) gxx[i * nx + i] = -2.0 * scale_; return true;
  }
  double radius_; Vector center_; double scale_;
};
class LinearConstraint : public Constraint {    // constraint.hpp:253-311
 public:
  LinearConstraint(const Matrix &A, const Vector &b, double scale = 1.0) : Constraint("LinearConstraint"), A_(A), b_(b), scale_(scale) {}
  int getDualDim() const override { return (int)b_.size(); }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_LINEAR; c.dim = (int)b_.size(); c.A = A_.a.data(); c.b = b_.data(); c.scale = scale_; return true; }
  Vector evaluate(const Vector &x, const Vector &) const override { Vector g(b_.size(), 0.0); for (int i = 0; i < A_.rows; ++i) for (int j = 0; j < A_.cols; ++j) g[i] += A_(i, j) * x[j]; return g; }
  Vector getUpperBound() const override { return b_; }
  Matrix getStateJacobian(const Vector &, const Vector &) const override { return A_; }
  Matrix getControlJacobian(const Vector &, const Vector &u) const override { return Matrix((int)b_.size(), (int)u.size()); }
  Matrix A_; Vector b_; double scale_;
};
class SecondOrderConeConstraint : public Constraint {   // constraint.hpp:626-800
 public:
  SecondOrderConeConstraint(const Vector &cone_origin, const Vector &opening_direction, double cone_angle_fov, double regularization_epsilon = 1e-6,
                            const std::string &name = "SecondOrderConeConstraint")
      : Constraint(name), origin_(cone_origin), axis_(opening_direction), cos_fov_(std::cos(cone_angle_fov)), epsilon_(regularization_epsilon) {
    if (cone_angle_fov < 0 || cone_angle_fov > 3.14159265358979323846) throw std::invalid_argument("SecondOrderConeConstraint: Cone angle must be between 0 and PI.");
    if (regularization_epsilon <= 0) throw std::invalid_argument("SecondOrderConeConstraint: Regularization epsilon must be positive.");
    if (origin_.size() != 3 || axis_.size() != 3) throw std::invalid_argument("SecondOrderConeConstraint: origin and direction are 3-vectors");
    const double n = std::sqrt(axis_[0] * axis_[0] + axis_[1] * axis_[1] + axis_[2] * axis_[2]);
    if (n == 0.0) throw std::invalid_argument("SecondOrderConeConstraint: Opening direction cannot be zero vector.");
    for (double &v : axis_) v /= n;
  }
  int getDualDim() const override { return 1; }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_SOC; c.dim = 3; c.center = origin_.data(); c.lower = axis_.data(); c.radius = cos_fov_; c.scale = epsilon_; return true; }
  Vector evaluate(const Vector &x, const Vector &) const override {
    if (x.size() < 3) throw std::invalid_argument("SecondOrderConeConstraint: State dimension must be at least 3.");
    const double v0 = x[0] - origin_[0], v1 = x[1] - origin_[1], v2 = x[2] - origin_[2];
    return {std::sqrt(((v0 * v0 + v1 * v1) + v2 * v2) + epsilon_) * cos_fov_ - ((v0 * axis_[0] + v1 * axis_[1]) + v2 * axis_[2])};
  }
  Vector getUpperBound() const override { return {0.0}; }
  Matrix getStateJacobian(const Vector &x, const Vector &) const override {
    Matrix J(1, (int)x.size());
    const double v[3] = {x[0] - origin_[0], x[1] - origin_[1], x[2] - origin_[2]};
    const double rn = std::sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + epsilon_);
    for (int i = 0; i < 3; ++i) J(0, i) = rn > 1e-9 ? cos_fov_ * (v[i] / rn) - axis_[i] : -axis_[i];
    return J;
  }
  Matrix getControlJacobian(const Vector &, const Vector &u) const override { return Matrix(1, (int)u.size()); }
  bool getHessians(const Vector &, const Vector &, double *, double *, double *) const override { return false; }   // constraint.hpp:772-786 throw
  Vector origin_, axis_; double cos_fov_, epsilon_;
};
class ThrustMagnitudeConstraint : public Constraint {   // constraint.hpp:802-927
 public:
  ThrustMagnitudeConstraint(double min_thrust_norm, double max_thrust_norm, double epsilon = 1e-6)
      : Constraint("ThrustMagnitudeConstraint"), min_{min_thrust_norm}, max_(max_thrust_norm), epsilon_(epsilon) {
    if (min_thrust_norm < 0.0) throw std::invalid_argument("ThrustMagnitudeConstraint: min_thrust_norm must be non-negative.");
    if (max_thrust_norm < min_thrust_norm) throw std::invalid_argument("ThrustMagnitudeConstraint: max_thrust_norm must be greater than or equal to min_thrust_norm.");
    if (epsilon <= 0.0) throw std::invalid_argument("ThrustMagnitudeConstraint: epsilon must be positive.");
  }
  int getDualDim() const override { return 2; }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_THRUST; c.dim = dim_; c.lower = min_.data(); c.radius = max_; c.scale = epsilon_; return dim_ > 0; }
  void setControlDim(int nu) { dim_ = nu; }   // the device descriptor carries the control dimension (CDDP::addPathConstraint sets it)
  Vector evaluate(const Vector &, const Vector &u) const override { double sq = 0; for (double v : u) sq += v * v; const double n = std::sqrt(sq); return {min_[0] - n, n - max_}; }
  Vector getUpperBound() const override { return {0.0, 0.0}; }
  Matrix getStateJacobian(const Vector &x, const Vector &) const override { return Matrix(2, (int)x.size()); }
  Matrix getControlJacobian(const Vector &, const Vector &u) const override {
    Matrix J(2, (int)u.size()); double sq = 0; for (double v : u) sq += v * v;
    const double rn = std::sqrt(sq + epsilon_);
    if (!(rn < epsilon_)) for (int i = 0; i < (int)u.size(); ++i) { J(0, i) = -(u[i] / rn); J(1, i) = u[i] / rn; }
    return J;
  }
  bool getHessians(const Vector &, const Vector &u, double *, double *guu, double *) const override {   // constraint.hpp:899-920: {-H, H}
    const int nu = (int)u.size(); double sq = 0; for (double v : u) sq += v * v;
    const double term = sq + epsilon_, den = std::pow(term, 1.5);
    if (den > std::numeric_limits<double>::min()) for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) {
      const double h = ((i == j ? term : 0.0) - u[i] * u[j]) / den; guu[i * nu + j] = -h; guu[nu * nu + i * nu + j] = h; }
    return true;
  }
  Vector min_; double max_, epsilon_; int dim_ = 0;
};
class MaxThrustMagnitudeConstraint : public Constraint {   // constraint.hpp:929-1048
 public:
  explicit MaxThrustMagnitudeConstraint(double max_thrust_norm, double epsilon = 1e-6) : Constraint("MaxThrustMagnitudeConstraint"), max_(max_thrust_norm), epsilon_(epsilon) {
    if (max_thrust_norm < 0.0) throw std::invalid_argument("MaxThrustMagnitudeConstraint: max_thrust_norm must be non-negative.");
    if (epsilon <= 0.0) throw std::invalid_argument("MaxThrustMagnitudeConstraint: epsilon must be positive.");
  }
  int getDualDim() const override { return 1; }
  bool fill(cddp_hip_constraint &c) const override { c.kind = CDDP_HIP_CON_MAX_THRUST; c.dim = dim_; c.radius = max_; c.scale = epsilon_; return dim_ > 0; }
  void setControlDim(int nu) { dim_ = nu; }
  Vector evaluate(const Vector &, const Vector &u) const override { double sq = 0; for (double v : u) sq += v * v; return {std::sqrt(sq) - max_}; }
  Vector getUpperBound() const override { return {0.0}; }
  Matrix getStateJacobian(const Vector &x, const Vector &) const override { return Matrix(1, (int)x.size()); }
  Matrix getControlJacobian(const Vector &, const Vector &u) const override {
    Matrix J(1, (int)u.size()); double sq = 0; for (double v : u) sq += v * v;
    const double rn = std::sqrt(sq + epsilon_);
    if (rn > std::numeric_limits<double>::min()) for (int i = 0; i < (int)u.size(); ++i) J(0, i) = u[i] / rn;
    return J;
  }
  bool getHessians(const Vector &, const Vector &u, double *, double *guu, double *) const override {   // constraint.hpp:1021-1042
    const int nu = (int)u.size(); double sq = 0; for (double v : u) sq += v * v;
    const double term = sq + epsilon_, den = std::pow(term, 1.5);
    if (den > std::numeric_limits<double>::min()) for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) guu[i * nu + j] = ((i == j ? term : 0.0) - u[i] * u[j]) / den;
    return true;
  }
  double max_, epsilon_; int dim_ = 0;
};
class TerminalConstraint {
 public:
  virtual ~TerminalConstraint() = default;
  virtual void fill(cddp_hip_terminal_constraint &c) const = 0;
  virtual int getDualDim() const { cddp_hip_terminal_constraint c; std::memset(&c, 0, sizeof(c)); fill(c); return c.dim; }
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
  // NEW (no reference counterpart): which path solved the problem -- "resident" (device-resident batch kernels, the library's shared
  // straight-line sin / cos / log) or "plugin" (host loop in the host libm + batched GPU backward passes); empty for UnknownSolver
  std::string route, arithmetic;
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
        system_(std::move(system)), objective_(std::move(objective)), options_(options) {
    alpha_pr_ = options.line_search.initial_step_size; regularization_ = options.regularization.initial_value;
    if (objective_ && !reference_state_.empty() && !isZero(reference_state_)) objective_->setReferenceState(reference_state_);   // cddp_core.cpp:56-60
  }

  // --- setters, as cddp_core.cpp:62-212 writes them
  void setDynamicalSystem(std::unique_ptr<DynamicalSystem> s) { system_ = std::move(s); initialized_ = false; }
  void setObjective(std::unique_ptr<Objective> o) {                       // :115-124
    objective_ = std::move(o);
    if (objective_ && !reference_states_.empty()) { objective_->setReferenceState(reference_state_); objective_->setReferenceStates(reference_states_); }
    else if (objective_ && !reference_state_.empty() && !isZero(reference_state_)) objective_->setReferenceState(reference_state_);
  }
  void setOptions(const CDDPOptions &o) { options_ = o; }
  void setInitialState(const Vector &x0) {                                // :68-76
    initial_state_ = x0;
    if (!X_.empty() && X_[0].size() == x0.size()) X_[0] = initial_state_;
  }
  void setReferenceState(const Vector &r) {                               // :78-86
    reference_state_ = r;
    if (objective_) objective_->setReferenceState(reference_state_);
    reference_states_.clear(); reference_states_.push_back(reference_state_);
  }
  void setReferenceStates(const std::vector<Vector> &rs) {                // :88-100
    reference_states_ = rs;
    if (!reference_states_.empty()) reference_state_ = reference_states_.back();
    if (objective_) {
      if (!reference_states_.empty()) objective_->setReferenceState(reference_state_);
      objective_->setReferenceStates(reference_states_);
    }
  }
  void setHorizon(int h) { horizon_ = h; initialized_ = false; }
  void setTimestep(double dt) { timestep_ = dt; }
  void setInitialTrajectory(const std::vector<Vector> &X, const std::vector<Vector> &U) {   // :126-141
    if ((int)X.size() != horizon_ + 1 || (int)U.size() != horizon_) std::cerr << "Warning: Provided initial trajectory dimensions do not match horizon." << std::endl;
    X_ = X; U_ = U;
    if (!X_.empty()) initial_state_ = X_[0];
  }
  void addPathConstraint(std::string name, std::unique_ptr<Constraint> c) {
    if (!c) throw std::runtime_error("Cannot add null constraint.");   // cddp_context_utils.cpp:82-84
    path_constraint_set_[name] = std::move(c); initialized_ = false;
  }
  bool removePathConstraint(const std::string &name) { const bool r = path_constraint_set_.erase(name) > 0; if (r) initialized_ = false; return r; }
  void addTerminalConstraint(std::string name, std::unique_ptr<TerminalConstraint> c) {
    if (!c) throw std::runtime_error("Cannot add null constraint.");
    terminal_constraint_set_[name] = std::move(c); initialized_ = false;
  }
  bool removeTerminalConstraint(const std::string &name) { const bool r = terminal_constraint_set_.erase(name) > 0; if (r) initialized_ = false; return r; }
  // total_dual_dim_ of cddp_core.cpp:161-198: path AND terminal entries, replacement subtracts the old entry
  int getTotalDualDim() const {
    int m = 0;
    for (auto &kv : path_constraint_set_) m += kv.second->getDualDim();
    for (auto &kv : terminal_constraint_set_) m += kv.second->getDualDim();
    return m;
  }
  int getStateDim() const { if (!system_) throw std::runtime_error("Dynamical system not set."); return system_->getStateDim(); }
  int getControlDim() const { if (!system_) throw std::runtime_error("Dynamical system not set."); return system_->getControlDim(); }
  const Vector &getReferenceState() const { return reference_state_; }
  const std::vector<Vector> &getReferenceStates() const { return reference_states_; }
  const CDDPOptions &getOptions() const { return options_; }
  int getHorizon() const { return horizon_; }
  double getTimestep() const { return timestep_; }
  const Vector &getInitialState() const { return initial_state_; }
  const DynamicalSystem &getSystem() const { return *system_; }
  const Objective &getObjective() const { return *objective_; }
  const std::map<std::string, std::unique_ptr<Constraint>> &getConstraintSet() const { return path_constraint_set_; }
  bool hasTerminalConstraints() const { return !terminal_constraint_set_.empty(); }
  const std::map<std::string, std::unique_ptr<TerminalConstraint>> &getTerminalConstraintSet() const { return terminal_constraint_set_; }   // cddp_core.hpp:305-319
  // true when some plug-in is a user subclass without a device kernel: the solve then goes through cddp_hip_plugin_solve
  bool needsHostPlugins() const {
    if (!system_ || !objective_) return false;
    if (system_->isHostPlant() || typeid(*objective_) != typeid(QuadraticObjective)) return true;
    cddp_hip_constraint c;
    for (auto &kv : path_constraint_set_) {
      if (auto *tm = dynamic_cast<ThrustMagnitudeConstraint *>(kv.second.get())) tm->setControlDim(system_->getControlDim());
      if (auto *mt = dynamic_cast<MaxThrustMagnitudeConstraint *>(kv.second.get())) mt->setControlDim(system_->getControlDim());
      if (!kv.second->fill(c)) return true;
    }
    return false;
  }

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
  int numPathConstraints() const { return (int)path_constraint_set_.size(); }
  int numTerminalConstraints() const { return (int)terminal_constraint_set_.size(); }
  void initializeProblemIfNecessary();

 private:
  static std::map<std::string, Factory> &registry() { static std::map<std::string, Factory> r; return r; }
  static bool isZero(const Vector &v) { for (double x : v) if (x != 0.0) return false; return true; }
  Vector initial_state_, reference_state_; std::vector<Vector> reference_states_; int horizon_; double timestep_;
  bool initialized_ = false;
  std::unique_ptr<DynamicalSystem> system_; std::unique_ptr<Objective> objective_; CDDPOptions options_;
  std::map<std::string, std::unique_ptr<Constraint>> path_constraint_set_;          // std::map: name order == dual stacking order
  std::map<std::string, std::unique_ptr<TerminalConstraint>> terminal_constraint_set_;
};

// ---- the GPU solver strategy ----------------------------------------------------------------------
class HipBatchSolver : public ISolverAlgorithm {
 public:
  explicit HipBatchSolver(int solver_kind, int device = 0) : kind_(solver_kind), device_(device) {}
  ~HipBatchSolver() override { if (h_) cddp_hip_destroy(h_); }
  std::string getSolverName() const override { return kind_ == CDDP_HIP_SOLVER_IPDDP ? "IPDDP" : kind_ == CDDP_HIP_SOLVER_LOGDDP ? "LogDDP" : kind_ == CDDP_HIP_SOLVER_MSIPDDP ? "MSIPDDP" : "CLDDP"; }
  // A second initialize() of the SAME solver object with options.warm_start keeps the device-resident solver state
  // (gains, slack / dual / costate variables): the reference's "existing solver state" branch
  // (clddp_solver.cpp:51-60, ipddp_solver.cpp:675-731).  CDDP::solve() creates a new solver per call, exactly as
  // the reference does (cddp_core.cpp:235-270), so through it a warm start is the "provided trajectory" branch.
  void initialize(CDDP &ctx) override {
    if (h_ && !ctx.needsHostPlugins() && ctx.getOptions().warm_start && batch_ == 1 && nx_ == ctx.getSystem().getStateDim() && nu_ == ctx.getSystem().getControlDim() && N_ == ctx.getHorizon()) {
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
  // NEW (no reference counterpart): one device-resident batch.
  std::vector<CDDPSolution> solveBatch(CDDP &ctx, const std::vector<Vector> &x0s) { create(ctx, x0s); return collect(ctx, (int)x0s.size()); }
  cddp_hip_stats stats{};
  // Route of LogDDP / MSIPDDP problems (round 5): Auto = an eligible problem (built-in plant with nx <= 8, built-in
  // objective / constraints; MSIPDDP also: no terminal set, and nu = 1 or nx = nu once a path constraint is present) runs on the resident
  // kernels (csrc/kernels_logddp.hpp, kernels_msipddp.hpp: the library's shared straight-line log / sin / cos) from solve() AND
  // solveBatch() -- one problem, one arithmetic, one iteration count whichever entry point is used; everything else takes the plug-in
  // route (host loop in the host libm + stack-fed GPU sweeps).  Plugin forces the latter (the reference's own arithmetic, one
  // trajectory at a time); the environment variable CDDP_HIP_F4_ROUTE=plugin|resident|auto sets the process-wide default.
  enum class Route { Auto, Plugin, Resident };
  Route route = defaultRoute();
  static Route defaultRoute() {
    const char *e = std::getenv("CDDP_HIP_F4_ROUTE");
    if (e && std::string(e) == "plugin") return Route::Plugin;
    if (e && std::string(e) == "resident") return Route::Resident;
    return Route::Auto;
  }

  static void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("cddp_hip: ") + cddp_hip_last_error()); }

 private:
  void create(CDDP &ctx, const std::vector<Vector> &x0s) {
    ctx.initializeProblemIfNecessary();
    if (h_) { cddp_hip_destroy(h_); h_ = nullptr; }
    const bool resident_logddp = kind_ == CDDP_HIP_SOLVER_LOGDDP && route != Route::Plugin && !ctx.needsHostPlugins() && ctx.getSystem().getStateDim() <= 8;
    // MSIPDDP batches: resident (csrc/kernels_msipddp.hpp) for a built-in plant with nx <= 8, no terminal set and -- once a path constraint
    // is present -- nu = 1 or nx = nu (the shapes msipddp_solver.cpp:1398 defines)
    const bool resident_msipddp = kind_ == CDDP_HIP_SOLVER_MSIPDDP && route != Route::Plugin && !ctx.needsHostPlugins() && ctx.getSystem().getStateDim() <= 8 &&
                                  ctx.numTerminalConstraints() == 0 &&
                                  (ctx.numPathConstraints() == 0 || ctx.getSystem().getControlDim() == 1 || ctx.getSystem().getStateDim() == ctx.getSystem().getControlDim());
    if (route == Route::Resident && ((kind_ == CDDP_HIP_SOLVER_LOGDDP && !resident_logddp) || (kind_ == CDDP_HIP_SOLVER_MSIPDDP && !resident_msipddp)))
      throw std::runtime_error("cddp_hip: Route::Resident requested, but this LogDDP / MSIPDDP problem has no resident kernels (built-in plant with nx <= 8 ...)");
    plugin_ = ctx.needsHostPlugins() || (kind_ == CDDP_HIP_SOLVER_LOGDDP && !resident_logddp) || (kind_ == CDDP_HIP_SOLVER_MSIPDDP && !resident_msipddp);   // host loop + stack-fed GPU sweeps
    if (plugin_) {
      const DynamicalSystem &sys = ctx.getSystem();
      nx_ = sys.getStateDim(); nu_ = sys.getControlDim(); N_ = ctx.getHorizon(); dt_ = ctx.getTimestep(); batch_ = (int)x0s.size(); ret_hist_ = false;
      x0s_ = x0s;
      return;
    }
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
  // ---- host plug-in route: the reference's virtual functions behind the flat callbacks of cddp_hip_plugin ----
  struct PluginCtx {
    const DynamicalSystem *sys; const Objective *obj; std::vector<const Constraint *> cons; int nx, nu, m;
    std::vector<cddp_hip_terminal_constraint> terms;   // terminal set in std::map order (round 6: cddp_hip_plugin_solve_terminal)
    std::exception_ptr error;   // a C++ exception must not unwind through the C library: parked here, rethrown after the call
    Vector x, u;
    void load(const double *xp, const double *up) { x.assign(xp, xp + nx); if (up) u.assign(up, up + nu); }
  };
  template <class F> static void guarded(PluginCtx *c, F f) { if (c->error) return; try { f(); } catch (...) { c->error = std::current_exception(); } }
  static void copyM(const Matrix &M, int r, int cc, double *out, const char *what) {
    if (M.rows != r || M.cols != cc) throw std::runtime_error(std::string(what) + ": unexpected shape");
    std::memcpy(out, M.a.data(), sizeof(double) * r * cc);
  }
  static void cbDyn(void *p, const double *x, const double *u, double t, double *xn) {
    auto *c = (PluginCtx *)p; std::fill(xn, xn + c->nx, std::numeric_limits<double>::quiet_NaN());
    guarded(c, [&] { c->load(x, u); Vector r = c->sys->getDiscreteDynamics(c->x, c->u, t); if ((int)r.size() != c->nx) throw std::runtime_error("getDiscreteDynamics: unexpected size"); std::copy(r.begin(), r.end(), xn); });
  }
  static void cbJac(void *p, const double *x, const double *u, double t, double *fx, double *fu) {
    auto *c = (PluginCtx *)p; std::fill(fx, fx + c->nx * c->nx, 0.0); std::fill(fu, fu + c->nx * c->nu, 0.0);
    guarded(c, [&] { c->load(x, u); copyM(c->sys->getStateJacobian(c->x, c->u, t), c->nx, c->nx, fx, "getStateJacobian"); copyM(c->sys->getControlJacobian(c->x, c->u, t), c->nx, c->nu, fu, "getControlJacobian"); });
  }
  static void cbHess(void *p, const double *x, const double *u, double t, double *fxx, double *fuu, double *fux) {
    auto *c = (PluginCtx *)p; const int nx = c->nx, nu = c->nu;
    std::fill(fxx, fxx + nx * nx * nx, 0.0); std::fill(fuu, fuu + nx * nu * nu, 0.0); std::fill(fux, fux + nx * nu * nx, 0.0);
    guarded(c, [&] {
      c->load(x, u);
      auto a = c->sys->getStateHessian(c->x, c->u, t); auto b = c->sys->getControlHessian(c->x, c->u, t); auto d = c->sys->getCrossHessian(c->x, c->u, t);
      if ((int)a.size() != nx || (int)b.size() != nx || (int)d.size() != nx) throw std::runtime_error("dynamics Hessians: one matrix per state row expected");
      for (int i = 0; i < nx; ++i) { copyM(a[i], nx, nx, fxx + i * nx * nx, "getStateHessian"); copyM(b[i], nu, nu, fuu + i * nu * nu, "getControlHessian"); copyM(d[i], nu, nx, fux + i * nu * nx, "getCrossHessian"); }
    });
  }
  static double cbRun(void *p, const double *x, const double *u, int k) {
    auto *c = (PluginCtx *)p; double v = std::numeric_limits<double>::quiet_NaN();
    guarded(c, [&] { c->load(x, u); v = c->obj->running_cost(c->x, c->u, k); }); return v;
  }
  static double cbTerm(void *p, const double *x) {
    auto *c = (PluginCtx *)p; double v = std::numeric_limits<double>::quiet_NaN();
    guarded(c, [&] { c->load(x, nullptr); v = c->obj->terminal_cost(c->x); }); return v;
  }
  static void cbRunD(void *p, const double *x, const double *u, int k, double *lx, double *lu, double *lxx, double *luu, double *lux) {
    auto *c = (PluginCtx *)p; const int nx = c->nx, nu = c->nu;
    std::fill(lx, lx + nx, 0.0); std::fill(lu, lu + nu, 0.0); std::fill(lxx, lxx + nx * nx, 0.0); std::fill(luu, luu + nu * nu, 0.0); std::fill(lux, lux + nu * nx, 0.0);
    guarded(c, [&] {
      c->load(x, u);
      Vector gx = c->obj->getRunningCostStateGradient(c->x, c->u, k), gu = c->obj->getRunningCostControlGradient(c->x, c->u, k);
      if ((int)gx.size() != nx || (int)gu.size() != nu) throw std::runtime_error("running-cost gradients: unexpected size");
      std::copy(gx.begin(), gx.end(), lx); std::copy(gu.begin(), gu.end(), lu);
      copyM(c->obj->getRunningCostStateHessian(c->x, c->u, k), nx, nx, lxx, "getRunningCostStateHessian");
      copyM(c->obj->getRunningCostControlHessian(c->x, c->u, k), nu, nu, luu, "getRunningCostControlHessian");
      copyM(c->obj->getRunningCostCrossHessian(c->x, c->u, k), nu, nx, lux, "getRunningCostCrossHessian");
    });
  }
  static void cbTermD(void *p, const double *x, double *lx, double *lxx) {
    auto *c = (PluginCtx *)p; std::fill(lx, lx + c->nx, 0.0); std::fill(lxx, lxx + c->nx * c->nx, 0.0);
    guarded(c, [&] { c->load(x, nullptr); Vector g = c->obj->getFinalCostGradient(c->x); if ((int)g.size() != c->nx) throw std::runtime_error("getFinalCostGradient: unexpected size");
                     std::copy(g.begin(), g.end(), lx); copyM(c->obj->getFinalCostHessian(c->x), c->nx, c->nx, lxx, "getFinalCostHessian"); });
  }
  static void cbCon(void *p, const double *x, const double *u, int, double *g, double *gx, double *gu) {
    auto *c = (PluginCtx *)p; std::fill(g, g + c->m, -1.0);
    if (gx) std::fill(gx, gx + c->m * c->nx, 0.0);
    if (gu) std::fill(gu, gu + c->m * c->nu, 0.0);
    guarded(c, [&] {
      c->load(x, u); int off = 0;
      for (const Constraint *k : c->cons) {
        const int d = k->getDualDim(); Vector e = k->evaluate(c->x, c->u), ub = k->getUpperBound();
        if ((int)e.size() != d || (int)ub.size() != d) throw std::runtime_error("constraint '" + k->getName() + "': evaluate / getUpperBound size differs from getDualDim");
        for (int i = 0; i < d; ++i) g[off + i] = e[i] - ub[i];
        if (gx) copyM(k->getStateJacobian(c->x, c->u), d, c->nx, gx + off * c->nx, "Constraint::getStateJacobian");
        if (gu) copyM(k->getControlJacobian(c->x, c->u), d, c->nu, gu + off * c->nu, "Constraint::getControlJacobian");
        off += d;
      }
    });
  }
  static void cbConHess(void *p, const double *x, const double *u, int, double *gxx, double *guu, double *gux) {
    auto *c = (PluginCtx *)p;
    guarded(c, [&] {
      c->load(x, u); int off = 0;
      for (const Constraint *k : c->cons) {
        const int d = k->getDualDim();
        double *pxx = gxx + (size_t)off * c->nx * c->nx, *puu = guu + (size_t)off * c->nu * c->nu, *pux = gux + (size_t)off * c->nu * c->nx;
        if (!k->getHessians(c->x, c->u, pxx, puu, pux)) {   // "not provided": the rows stay zero
          std::fill(pxx, pxx + (size_t)d * c->nx * c->nx, 0.0); std::fill(puu, puu + (size_t)d * c->nu * c->nu, 0.0); std::fill(pux, pux + (size_t)d * c->nu * c->nx, 0.0);
        }
        off += d;
      }
    });
  }
  // residual rows (and state-Jacobian rows) of the terminal set: h = x_N - target (identity rows), g_T = A_N x_N - b_N (rows of A_N)
  // (terminal_constraint.hpp:75-113, 180-219)
  static void cbTerminal(void *p, const double *xN, double *r, double *rx) {
    auto *c = (PluginCtx *)p; const int nx = c->nx;
    int row = 0;
    for (const cddp_hip_terminal_constraint &t : c->terms) {
      for (int i = 0; i < t.dim; ++i, ++row) {
        if (rx) std::fill(rx + (size_t)row * nx, rx + (size_t)(row + 1) * nx, 0.0);
        if (t.kind == CDDP_HIP_TERM_EQUALITY) { r[row] = xN[i] - t.target[i]; if (rx) rx[(size_t)row * nx + i] = 1.0; }
        else { double a = 0.0; for (int k = 0; k < nx; ++k) a += t.A[(size_t)i * nx + k] * xN[k]; r[row] = a - t.b[i]; if (rx) std::copy(t.A + (size_t)i * nx, t.A + (size_t)(i + 1) * nx, rx + (size_t)row * nx); }
      }
    }
  }
  std::vector<CDDPSolution> collectPlugin(CDDP &ctx, int B) {
    PluginCtx pc; pc.sys = &ctx.getSystem(); pc.obj = &ctx.getObjective(); pc.nx = nx_; pc.nu = nu_; pc.m = 0;
    cddp_hip_plugin pl; std::memset(&pl, 0, sizeof(pl));
    pl.abi_version = CDDP_HIP_ABI_VERSION; pl.options_bytes = (int)sizeof(cddp_hip_options);
    pl.user = &pc; pl.nx = nx_; pl.nu = nu_;
    const ControlConstraint *box = nullptr;
    if (kind_ == CDDP_HIP_SOLVER_IPDDP || kind_ == CDDP_HIP_SOLVER_LOGDDP || kind_ == CDDP_HIP_SOLVER_MSIPDDP) {
      for (auto &kv : ctx.getConstraintSet()) {   // std::map order == dual stacking order
        if ((int)pc.cons.size() == CDDP_HIP_PLUGIN_MAX_CONSTRAINTS) throw std::runtime_error("HipBatchSolver: too many path constraints for the plug-in solve");
        pl.constraint_dims[pc.cons.size()] = kv.second->getDualDim(); pc.m += kv.second->getDualDim(); pc.cons.push_back(kv.second.get());
      }
      pl.n_constraints = (int)pc.cons.size();
    } else {   // clddp_solver.cpp:85-86: only the constraint literally named "ControlConstraint"
      auto it = ctx.getConstraintSet().find("ControlConstraint");
      if (it != ctx.getConstraintSet().end()) box = dynamic_cast<const ControlConstraint *>(it->second.get());
      if (box) { pl.control_lower = box->lower_.data(); pl.control_upper = box->upper_.data(); }
    }
    pl.discrete_dynamics = cbDyn; pl.jacobians = cbJac; pl.hessians = ctx.getOptions().use_ilqr ? nullptr : cbHess;
    pl.running_cost = cbRun; pl.terminal_cost = cbTerm; pl.running_cost_derivatives = cbRunD; pl.terminal_cost_derivatives = cbTermD;
    pl.constraints = pc.cons.empty() ? nullptr : cbCon;
    pl.constraint_hessians = ((kind_ == CDDP_HIP_SOLVER_LOGDDP || (kind_ == CDDP_HIP_SOLVER_MSIPDDP && !ctx.getOptions().use_ilqr)) && !pc.cons.empty()) ? cbConHess : nullptr;
    const int nx = nx_, nu = nu_, N = N_;
    std::vector<double> x0((size_t)B * nx), U0, X0;
    for (int b = 0; b < B; ++b) for (int i = 0; i < nx; ++i) x0[(size_t)b * nx + i] = x0s_[b][i];
    if ((int)ctx.U_.size() == N) { U0.resize((size_t)B * N * nu); for (int b = 0; b < B; ++b) for (int t = 0; t < N; ++t) for (int i = 0; i < nu; ++i) U0[((size_t)b * N + t) * nu + i] = ctx.U_[t][i]; }
    if ((int)ctx.X_.size() == N + 1) { X0.resize((size_t)B * (N + 1) * nx); for (int b = 0; b < B; ++b) for (int t = 0; t <= N; ++t) for (int i = 0; i < nx; ++i) X0[((size_t)b * (N + 1) + t) * nx + i] = ctx.X_[t][i]; }
    std::vector<cddp_hip_result> r(B);
    std::vector<double> X((size_t)B * (N + 1) * nx), U((size_t)B * N * nu), K((size_t)B * N * nu * nx);
    cddp_hip_options o = ctx.getOptions().toPOD(kind_ == CDDP_HIP_SOLVER_MSIPDDP);
    cddp_hip_plugin_terminal tc; std::memset(&tc, 0, sizeof(tc));
    if (kind_ == CDDP_HIP_SOLVER_IPDDP && ctx.hasTerminalConstraints()) {   // only IPDDP reads the terminal set (ipddp_solver.cpp:84-215)
      for (auto &kv : ctx.getTerminalConstraintSet()) {
        if (tc.n_terminal == CDDP_HIP_PLUGIN_MAX_CONSTRAINTS) throw std::runtime_error("HipBatchSolver: too many terminal constraints for the plug-in solve");
        cddp_hip_terminal_constraint t; std::memset(&t, 0, sizeof(t)); kv.second->fill(t);
        if (t.kind == CDDP_HIP_TERM_EQUALITY && t.dim != nx) throw std::invalid_argument("TerminalEqualityConstraint: final_state dimension mismatch.");
        tc.dims[tc.n_terminal] = t.dim; tc.equality[tc.n_terminal] = (t.kind == CDDP_HIP_TERM_EQUALITY) ? 1 : 0; ++tc.n_terminal;
        pc.terms.push_back(t);
      }
      tc.evaluate = cbTerminal;
    }
    const int rc = tc.n_terminal > 0
        ? cddp_hip_plugin_solve_terminal(&pl, &tc, kind_, N, dt_, &o, device_, B, x0.data(), U0.empty() ? nullptr : U0.data(), X0.empty() ? nullptr : X0.data(), r.data(), X.data(), U.data(), K.data(), nullptr)
        : cddp_hip_plugin_solve(&pl, kind_, N, dt_, &o, device_, B, x0.data(), U0.empty() ? nullptr : U0.data(), X0.empty() ? nullptr : X0.data(), r.data(), X.data(), U.data(), K.data());
    if (pc.error) std::rethrow_exception(pc.error);
    check(rc);
    std::vector<CDDPSolution> out(B);
    for (int b = 0; b < B; ++b) {
      CDDPSolution &s = out[b];
      s.solver_name = getSolverName(); s.status_message = cddp_hip_status_string(r[b].status);
      s.route = "plugin"; s.arithmetic = "host libm (plug-in callbacks and outer loop on the host, batched GPU backward passes)";
      s.iterations_completed = r[b].iterations; s.final_objective = r[b].final_objective;
      s.final_step_length = r[b].alpha_pr; s.final_regularization = r[b].regularization;
      s.final_primal_infeasibility = r[b].inf_pr; s.final_dual_infeasibility = r[b].inf_du; s.final_complementary_infeasibility = r[b].inf_comp; s.final_barrier_mu = r[b].barrier_mu;
      for (int t = 0; t <= N; ++t) { s.time_points.push_back(t * dt_); s.state_trajectory.emplace_back(X.begin() + ((size_t)b * (N + 1) + t) * nx, X.begin() + ((size_t)b * (N + 1) + t + 1) * nx); }
      for (int t = 0; t < N; ++t) {
        s.control_trajectory.emplace_back(U.begin() + ((size_t)b * N + t) * nu, U.begin() + ((size_t)b * N + t + 1) * nu);
        Matrix Kt(nu, nx); std::memcpy(Kt.a.data(), K.data() + ((size_t)b * N + t) * nu * nx, sizeof(double) * nu * nx); s.feedback_gains.push_back(Kt);
      }
    }
    return out;
  }
  std::vector<CDDPSolution> collect(CDDP &ctx, int B) {
    if (plugin_) return collectPlugin(ctx, B);
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
      s.route = "resident"; s.arithmetic = "device, shared straight-line sin / cos / log (csrc/dev_trig.hpp)";
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
        s.history.complementary_infeasibility.push_back(row[6]); if (kind_ == CDDP_HIP_SOLVER_IPDDP || kind_ == CDDP_HIP_SOLVER_LOGDDP || kind_ == CDDP_HIP_SOLVER_MSIPDDP) s.history.barrier_mu.push_back(row[7]); s.history.regularization.push_back(row[8]);
      }
    }
    return out;
  }
  int kind_, device_; cddp_hip_handle *h_ = nullptr; int nx_ = 0, nu_ = 0, N_ = 0, max_it_ = 0, batch_ = 0; double dt_ = 0; bool ret_hist_ = false;
  bool plugin_ = false; std::vector<Vector> x0s_;
};

// Register the GPU core under the reference's own solver names: a true drop-in (cddp_core.cpp:215-219).
inline void registerHipSolvers(int device = 0) {
  CDDP::registerSolver("IPDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_IPDDP, device); });
  CDDP::registerSolver("CLDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_CLDDP, device); });
  CDDP::registerSolver("CLCDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_CLDDP, device); });
  CDDP::registerSolver("LogDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_LOGDDP, device); });
  CDDP::registerSolver("LOGDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_LOGDDP, device); });
  CDDP::registerSolver("MSIPDDP", [device] { return std::make_unique<HipBatchSolver>(CDDP_HIP_SOLVER_MSIPDDP, device); });
}

inline void CDDP::initializeProblemIfNecessary() {   // cddp_core.cpp:272-306
  if (initialized_) return;
  if (!system_) throw std::runtime_error("Dynamical system must be set before solving.");
  if (!objective_) throw std::runtime_error("Objective function must be set before solving.");
  const int nx = system_->getStateDim(), nu = system_->getControlDim();
  auto compatible = [](const std::vector<Vector> &tr, int n, int dim) {   // detail::hasCompatibleVectorLayout (cddp_context_utils.cpp:40-57)
    if ((int)tr.size() != n) return false;
    for (auto &v : tr) if ((int)v.size() != dim) return false;
    return true;
  };
  // (the reference keeps a compatible trajectory only under warm_start and otherwise re-shapes a stale one; a compatible
  //  trajectory a caller provided through setInitialTrajectory has the right shape already and is left alone by ensureTrajectoryShape)
  if (!compatible(X_, horizon_ + 1, nx)) X_.assign((size_t)horizon_ + 1, Vector(nx, 0.0));
  if (!compatible(U_, horizon_, nu)) U_.assign((size_t)horizon_, Vector(nu, 0.0));
  X_[0] = initial_state_;
  const double inf = std::numeric_limits<double>::infinity();
  cost_ = merit_function_ = inf_pr_ = inf_du_ = inf_comp_ = inf;
  regularization_ = options_.regularization.initial_value;
  initialized_ = true;
}

inline void CDDP::flatten(int solver, Flat &f) const {
  std::memset(&f.p, 0, sizeof(f.p));
  const auto *qo = dynamic_cast<const QuadraticObjective *>(objective_.get());
  if (!qo || system_->isHostPlant()) throw std::runtime_error("CDDP::flatten: host plug-ins have no device descriptor (they are served by cddp_hip_plugin_solve)");
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
    if (auto *tm = dynamic_cast<ThrustMagnitudeConstraint *>(kv.second.get())) tm->setControlDim(system_->getControlDim());
    if (auto *mt = dynamic_cast<MaxThrustMagnitudeConstraint *>(kv.second.get())) mt->setControlDim(system_->getControlDim());
    cddp_hip_constraint c; std::memset(&c, 0, sizeof(c)); std::strncpy(c.name, kv.first.c_str(), CDDP_HIP_NAME_LEN - 1); c.scale = 1.0; if (!kv.second->fill(c)) throw std::runtime_error("CDDP::flatten: constraint '" + kv.first + "' has no device descriptor"); f.cons.push_back(c);
  }
  for (auto &kv : terminal_constraint_set_) {
    cddp_hip_terminal_constraint c; std::memset(&c, 0, sizeof(c)); std::strncpy(c.name, kv.first.c_str(), CDDP_HIP_NAME_LEN - 1); kv.second->fill(c); f.terms.push_back(c);
  }
  p.n_constraints = (int)f.cons.size(); p.constraints = f.cons.empty() ? nullptr : f.cons.data();
  p.n_terminal = (int)f.terms.size(); p.terminal = f.terms.empty() ? nullptr : f.terms.data();
  p.options = options_.toPOD(solver == CDDP_HIP_SOLVER_MSIPDDP);
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
  const int kind = (solver_type == "IPDDP") ? CDDP_HIP_SOLVER_IPDDP : (solver_type == "CLDDP" || solver_type == "CLCDDP") ? CDDP_HIP_SOLVER_CLDDP
                   : (solver_type == "LogDDP" || solver_type == "LOGDDP") ? CDDP_HIP_SOLVER_LOGDDP : (solver_type == "MSIPDDP") ? CDDP_HIP_SOLVER_MSIPDDP : -1;
  if (kind < 0) throw std::runtime_error("UnknownSolver - No solver registered for '" + solver_type + "'");
  HipBatchSolver s(kind, device);
  return s.solveBatch(*this, x0s);
}

}  // namespace cddp
