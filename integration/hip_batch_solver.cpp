// Reference-side adapter (see hip_batch_solver.hpp): cddp::ISolverAlgorithm over the C-ABI of libcddp_hip.so, written against the REAL
// cddp-cpp headers.  Interfaces it implements / reads, by reference file:line:
//   ISolverAlgorithm {initialize, solve, getSolverName}        include/cddp-cpp/cddp_core/cddp_core.hpp:186-210
//   CDDP::registerSolver / createSolver (registry first)       cddp_core.hpp:305-319, src/cddp_core/cddp_core.cpp:213-233, 578-595
//   CDDP accessors, X_ / U_ / cost_ / inf_* left updated       cddp_core.hpp:223-245, 321-343; src/cddp_core/cddp_solver_base.cpp:161-171
//   CDDPOptions -> cddp_hip_options                            include/cddp-cpp/cddp_core/options.hpp:29-229
//   constraint set in std::map order                           cddp_core.hpp:237-245, 422; ipddp_solver.cpp:1371-1384
//   CLDDP honours only the constraint NAMED "ControlConstraint" clddp_solver.cpp:85-86 (the library applies the same rule to the descriptors)
//
// Routes.  "resident": plant, objective and every constraint can be described to the library as PODs -> cddp_hip_create / cddp_hip_solve,
// the whole batch device-resident.  "plugin": anything else (a user DynamicalSystem / Objective / Constraint subclass, or a built-in
// class whose parameters the reference keeps private without an accessor) -> cddp_hip_plugin_solve: the context's own virtual functions
// behind flat C callbacks, host forward passes, batched GPU backward passes.  Nothing is silently approximated: a problem neither route
// serves throws std::runtime_error with the library's message.
#include "hip_batch_solver.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <typeinfo>

#include "cddp_core/constraint.hpp"
#include "cddp_core/dynamical_system.hpp"
#include "cddp_core/objective.hpp"
#include "cddp_core/options.hpp"
#include "cddp_core/terminal_constraint.hpp"
#include "dynamics_model/bicycle.hpp"
#include "dynamics_model/car.hpp"
#include "dynamics_model/cartpole.hpp"
#include "dynamics_model/lti_system.hpp"
#include "dynamics_model/manipulator.hpp"
#include "dynamics_model/pendulum.hpp"
#include "dynamics_model/quadrotor.hpp"
#include "dynamics_model/spacecraft_linear.hpp"
#include "dynamics_model/unicycle.hpp"

#include "cddp_hip.h"   // this repository: include/cddp_hip.h

namespace cddp {
namespace {

void check(int rc) {
  if (rc != 0) throw std::runtime_error(std::string("cddp_hip: ") + cddp_hip_last_error());
}

// Eigen is column-major, the C-ABI is row-major.
std::vector<double> rowMajor(const Eigen::MatrixXd &M) {
  std::vector<double> out((size_t)M.rows() * (size_t)M.cols());
  for (Eigen::Index i = 0; i < M.rows(); ++i)
    for (Eigen::Index j = 0; j < M.cols(); ++j) out[(size_t)i * (size_t)M.cols() + (size_t)j] = M(i, j);
  return out;
}
std::vector<double> toStd(const Eigen::VectorXd &v) { return std::vector<double>(v.data(), v.data() + v.size()); }
void copyRowMajor(const Eigen::MatrixXd &M, int rows, int cols, double *out, const char *what) {
  if (M.rows() != rows || M.cols() != cols) throw std::runtime_error(std::string(what) + ": unexpected shape");
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) out[(size_t)i * cols + j] = M(i, j);
}

// ---- CDDPOptions -> cddp_hip_options: field for field (options.hpp:29-229; include/cddp_hip.h:cddp_hip_options) ------------------------
cddp_hip_options toPOD(const CDDPOptions &c, bool for_msipddp) {
  cddp_hip_options o;
  cddp_hip_default_options(&o);
  o.tolerance = c.tolerance; o.acceptable_tolerance = c.acceptable_tolerance; o.max_iterations = c.max_iterations; o.max_cpu_time = c.max_cpu_time;
  o.use_ilqr = c.use_ilqr ? 1 : 0; o.enable_parallel = c.enable_parallel ? 1 : 0; o.return_iteration_info = c.return_iteration_info ? 1 : 0;
  o.warm_start = c.warm_start ? 1 : 0; o.termination_scaling_max_factor = c.termination_scaling_max_factor;
  o.ls_max_iterations = c.line_search.max_iterations; o.ls_initial_step_size = c.line_search.initial_step_size;
  o.ls_min_step_size = c.line_search.min_step_size; o.ls_step_reduction_factor = c.line_search.step_reduction_factor;
  o.reg_initial_value = c.regularization.initial_value; o.reg_update_factor = c.regularization.update_factor;
  o.reg_max_value = c.regularization.max_value; o.reg_min_value = c.regularization.min_value;
  o.boxqp_max_iterations = c.box_qp.max_iterations; o.boxqp_min_gradient_norm = c.box_qp.min_gradient_norm;
  o.boxqp_min_relative_improvement = c.box_qp.min_relative_improvement; o.boxqp_step_decrease_factor = c.box_qp.step_decrease_factor;
  o.boxqp_min_step_size = c.box_qp.min_step_size; o.boxqp_armijo_constant = c.box_qp.armijo_constant;
  o.filter_merit_acceptance_threshold = c.filter.merit_acceptance_threshold; o.filter_violation_acceptance_threshold = c.filter.violation_acceptance_threshold;
  o.filter_max_violation_threshold = c.filter.max_violation_threshold; o.filter_min_violation_for_armijo_check = c.filter.min_violation_for_armijo_check;
  o.filter_armijo_constant = c.filter.armijo_constant;
  o.ipddp_dual_var_init_scale = c.ipddp.dual_var_init_scale; o.ipddp_slack_var_init_scale = c.ipddp.slack_var_init_scale;
  o.ipddp_barrier_tol_mult = c.ipddp.barrier_tol_mult; o.ipddp_barrier_update_dual_weight = c.ipddp.barrier_update_dual_weight;
  o.ipddp_mu_kappa_epsilon = c.ipddp.mu_kappa_epsilon; o.ipddp_check_state_stationarity = c.ipddp.check_state_stationarity ? 1 : 0;
  o.ipddp_theta_norm_l2 = (c.ipddp.theta_norm == "l2") ? 1 : 0; o.ipddp_max_filter_size = c.ipddp.max_filter_size; o.ipddp_theta_0_floor = c.ipddp.theta_0_floor;
  o.ipddp_warmstart_repair = c.ipddp.warmstart_repair ? 1 : 0; o.ipddp_warmstart_s_min = c.ipddp.warmstart_s_min; o.ipddp_warmstart_y_min = c.ipddp.warmstart_y_min;
  o.ipddp_warmstart_interior_factor = c.ipddp.warmstart_interior_factor;
  o.ipddp_jacobian_regularization_value = c.ipddp.jacobian_regularization_value; o.ipddp_jacobian_regularization_exponent = c.ipddp.jacobian_regularization_exponent;
  const SolverSpecificBarrierOptions &bar = for_msipddp ? c.msipddp.barrier : c.ipddp.barrier;
  o.barrier_mu_initial = bar.mu_initial; o.barrier_mu_min_value = bar.mu_min_value; o.barrier_mu_update_factor = bar.mu_update_factor;
  o.barrier_mu_update_power = bar.mu_update_power; o.barrier_min_fraction_to_boundary = bar.min_fraction_to_boundary;
  o.barrier_strategy = static_cast<int>(bar.strategy);   // ADAPTIVE = 0, MONOTONIC = 1, IPOPT = 2 on both sides (options.hpp:29-34)
  if (for_msipddp) { o.ipddp_dual_var_init_scale = c.msipddp.dual_var_init_scale; o.ipddp_slack_var_init_scale = c.msipddp.slack_var_init_scale; }
  o.logddp_mu_initial = c.log_barrier.barrier.mu_initial; o.logddp_mu_min_value = c.log_barrier.barrier.mu_min_value;
  o.logddp_mu_update_factor = c.log_barrier.barrier.mu_update_factor; o.logddp_relaxed_delta = c.log_barrier.relaxed_log_barrier_delta;
  o.msipddp_costate_var_init_scale = c.msipddp.costate_var_init_scale; o.msipddp_segment_length = c.msipddp.segment_length;
  o.msipddp_rollout_type = c.msipddp.rollout_type == "nonlinear" ? 0 : (c.msipddp.rollout_type == "hybrid" ? 2 : 1);
  o.msipddp_use_controlled_rollout = c.msipddp.use_controlled_rollout ? 1 : 0;
  return o;
}

int integratorId(const std::string &s) {
  if (s == "euler") return CDDP_HIP_EULER;
  if (s == "heun") return CDDP_HIP_HEUN;
  if (s == "rk3") return CDDP_HIP_RK3;
  if (s == "rk4") return CDDP_HIP_RK4;
  return -1;
}

// ---- the plant: dynamic_cast table (src/dynamics_model/*.cpp).  A plant is "resident" when the library has its device form AND the
// reference exposes its parameters.  typeid equality, not dynamic_cast alone: a user class DERIVED from a built-in plant may override
// its dynamics.  Quadrotor / Car / Bicycle / HCW keep their parameters private without accessors in the reference as it stands
// (quadrotor.hpp:122-124, car.hpp:125, bicycle.hpp:102, spacecraft_linear.hpp:113-114): with integration/reference_getters.patch applied
// (five one-line getters) define CDDP_HIP_REFERENCE_HAS_GETTERS and they go resident too; without it they take the plug-in route.
struct ModelDesc {
  int id = -1;
  std::vector<double> params;
  std::vector<double> lti_A, lti_B;   // row-major discrete A, B (LTISystem::getA / getB are the discrete matrices, lti_system.cpp:22-75)
};
bool describeModel(const DynamicalSystem &s, ModelDesc &m) {
  if (integratorId(s.getIntegrationType()) < 0) return false;
  if (typeid(s) == typeid(Pendulum)) {
    const auto &p = static_cast<const Pendulum &>(s);
    m.id = CDDP_HIP_MODEL_PENDULUM; m.params = {p.getLength(), p.getMass(), p.getDamping(), p.getGravity()};
    return true;
  }
  if (typeid(s) == typeid(CartPole)) {
    const auto &p = static_cast<const CartPole &>(s);
    m.id = CDDP_HIP_MODEL_CARTPOLE; m.params = {p.getCartMass(), p.getPoleMass(), p.getPoleLength(), p.getGravity(), p.getDamping()};
    return true;
  }
  if (typeid(s) == typeid(Unicycle)) { m.id = CDDP_HIP_MODEL_UNICYCLE; return true; }
  if (typeid(s) == typeid(Manipulator)) { m.id = CDDP_HIP_MODEL_MANIPULATOR; return true; }
  if (typeid(s) == typeid(LTISystem)) {
    const auto &p = static_cast<const LTISystem &>(s);
    m.id = CDDP_HIP_MODEL_LTI; m.lti_A = rowMajor(p.getA()); m.lti_B = rowMajor(p.getB());
    return true;
  }
#ifdef CDDP_HIP_REFERENCE_HAS_GETTERS
  if (typeid(s) == typeid(Quadrotor)) {
    const auto &p = static_cast<const Quadrotor &>(s);
    const Eigen::Matrix3d I = p.getInertia();
    if (I(0, 1) != 0.0 || I(0, 2) != 0.0 || I(1, 2) != 0.0) return false;   // the device form holds a diagonal inertia
    m.id = CDDP_HIP_MODEL_QUADROTOR; m.params = {p.getMass(), p.getArmLength(), I(0, 0), I(1, 1), I(2, 2), p.getGravity()};
    return true;
  }
  if (typeid(s) == typeid(Bicycle)) { m.id = CDDP_HIP_MODEL_BICYCLE; m.params = {static_cast<const Bicycle &>(s).getWheelbase()}; return true; }
  if (typeid(s) == typeid(Car)) { m.id = CDDP_HIP_MODEL_CAR; m.params = {static_cast<const Car &>(s).getWheelbase()}; return true; }
  if (typeid(s) == typeid(HCW)) { const auto &p = static_cast<const HCW &>(s); m.id = CDDP_HIP_MODEL_HCW; m.params = {p.getMeanMotion(), p.getMass()}; return true; }
#endif
  return false;
}

// ---- the objective: exactly a QuadraticObjective (a subclass may override any virtual) -------------------------------------------------
// getQ() / getR() return the matrices ALREADY scaled by the objective's timestep (objective.cpp:38-39), and that timestep has no accessor;
// the C-ABI takes Q, R unscaled and multiplies by problem.dt as the constructor does.  So the adapter hands over q with
// fl(q * dt) == Q_scaled EXACTLY for every entry: q0 = fl(Q_scaled / dt), then the neighbouring doubles are tried until the product
// reproduces the scaled value bit for bit (x -> fl(x * dt) is monotone, a pre-image lies within a few ulp of q0).
double preimageOfScaled(double scaled, double dt) {
  if (scaled == 0.0 || !std::isfinite(scaled) || !(dt > 0.0)) return scaled == 0.0 ? 0.0 : scaled / dt;
  double q = scaled / dt;
  if (q * dt == scaled) return q;
  double lo = q, hi = q;
  for (int k = 0; k < 8; ++k) {
    lo = std::nextafter(lo, -std::numeric_limits<double>::infinity());
    hi = std::nextafter(hi, std::numeric_limits<double>::infinity());
    if (lo * dt == scaled) return lo;
    if (hi * dt == scaled) return hi;
  }
  throw std::runtime_error("cddp_hip adapter: no double q with q * dt == the objective's scaled weight (was the objective built with another timestep "
                           "than the CDDP context? rebuild it with the context's timestep or use the plug-in route)");
}
struct ObjectiveDesc {
  std::vector<double> Q, R, Qf, x_ref, x_ref_traj;
};
bool describeObjective(const Objective &o, double dt, int nx, int nu, int horizon, ObjectiveDesc &d) {
  if (typeid(o) != typeid(QuadraticObjective)) return false;
  const auto &q = static_cast<const QuadraticObjective &>(o);
  if (q.getQ().rows() != nx || q.getQ().cols() != nx || q.getR().rows() != nu || q.getR().cols() != nu || q.getQf().rows() != nx || q.getQf().cols() != nx) return false;
  d.Q = rowMajor(q.getQ()); d.R = rowMajor(q.getR()); d.Qf = rowMajor(q.getQf());
  for (double &v : d.Q) v = preimageOfScaled(v, dt);
  for (double &v : d.R) v = preimageOfScaled(v, dt);
  const Eigen::VectorXd ref = q.getReferenceState();
  if (ref.size() != nx) return false;
  d.x_ref = toStd(ref);
  const std::vector<Eigen::VectorXd> refs = q.getReferenceStates();   // objective.cpp:83-88: running cost tracks reference_states_[t] when non-empty
  if (!refs.empty()) {
    if ((int)refs.size() != horizon + 1) return false;
    for (const auto &r : refs) { if (r.size() != nx) return false; d.x_ref_traj.insert(d.x_ref_traj.end(), r.data(), r.data() + nx); }
  }
  return true;
}

// ---- constraints: one descriptor per entry of the std::map, in the map's (name) order -------------------------------------------------
// Only what the public interface exposes EXACTLY is described; the scale factors are private, so they are read off a Jacobian / Hessian
// entry that equals them exactly (constraint.hpp:186-216: box Jacobian blocks = +-I * scale; :392-400: ball Hessian = -2 scale I).
struct ConstraintStore {
  std::vector<cddp_hip_constraint> c;
  std::vector<cddp_hip_terminal_constraint> t;
  std::vector<std::unique_ptr<std::vector<double>>> keep;   // backing storage of the pointers above
  const double *hold(std::vector<double> v) { keep.emplace_back(new std::vector<double>(std::move(v))); return keep.back()->data(); }
};
void setName(char (&dst)[CDDP_HIP_NAME_LEN], const std::string &name) {
  if (name.size() >= CDDP_HIP_NAME_LEN) throw std::runtime_error("cddp_hip adapter: constraint name '" + name + "' is longer than CDDP_HIP_NAME_LEN - 1");
  std::memset(dst, 0, sizeof(dst));
  std::memcpy(dst, name.data(), name.size());
}
bool describeConstraint(const std::string &key, const Constraint &k, int nx, int nu, ConstraintStore &st) {
  cddp_hip_constraint d;
  std::memset(&d, 0, sizeof(d));
  setName(d.name, key);   // the MAP KEY (what the user passed to addPathConstraint), not Constraint::getName(): it fixes the dual stacking order
  d.scale = 1.0;
  const Eigen::VectorXd x0 = Eigen::VectorXd::Zero(nx), u0 = Eigen::VectorXd::Zero(nu);
  if (typeid(k) == typeid(ControlConstraint)) {
    const auto &b = static_cast<const ControlConstraint &>(k);
    if (b.rawLowerBound().size() != nu || b.rawUpperBound().size() != nu) return false;
    d.kind = CDDP_HIP_CON_CONTROL_BOX; d.dim = nu; d.lower = st.hold(toStd(b.rawLowerBound())); d.upper = st.hold(toStd(b.rawUpperBound()));
    d.scale = b.getControlJacobian(x0, u0)(nu, 0);          // bottom-right block = +I * scale_factor
    st.c.push_back(d); return true;
  }
  if (typeid(k) == typeid(StateConstraint)) {
    const auto &b = static_cast<const StateConstraint &>(k);
    if (b.rawLowerBound().size() != nx || b.rawUpperBound().size() != nx) return false;
    d.kind = CDDP_HIP_CON_STATE_BOX; d.dim = nx; d.lower = st.hold(toStd(b.rawLowerBound())); d.upper = st.hold(toStd(b.rawUpperBound()));
    d.scale = b.getStateJacobian(x0, u0)(nx, 0);
    st.c.push_back(d); return true;
  }
  if (typeid(k) == typeid(LinearConstraint)) {
    const Eigen::MatrixXd A = k.getStateJacobian(x0, u0);   // = A_ (constraint.hpp:278-284)
    const Eigen::VectorXd b = k.getUpperBound();             // = b_
    if (A.cols() != nx || A.rows() != b.size()) return false;
    d.kind = CDDP_HIP_CON_LINEAR; d.dim = (int)b.size(); d.A = st.hold(rowMajor(A)); d.b = st.hold(toStd(b));
    st.c.push_back(d); return true;
  }
  if (typeid(k) == typeid(BallConstraint)) {
    const auto &b = static_cast<const BallConstraint &>(k);
    const Eigen::VectorXd c = b.getCenter();
    if (c.size() < 1 || c.size() > nx) return false;
    d.kind = CDDP_HIP_CON_BALL; d.dim = (int)c.size(); d.center = st.hold(toStd(c)); d.radius = b.getRadius();
    d.scale = b.getStateHessian(x0, u0)[0](0, 0) / -2.0;     // Hxx = -2 scale I: exact (a power-of-two factor)
    st.c.push_back(d); return true;
  }
#ifdef CDDP_HIP_REFERENCE_HAS_GETTERS
  if (typeid(k) == typeid(SecondOrderConeConstraint)) {
    const auto &s = static_cast<const SecondOrderConeConstraint &>(k);
    d.kind = CDDP_HIP_CON_SOC; d.dim = 3; d.center = st.hold({s.getOrigin()(0), s.getOrigin()(1), s.getOrigin()(2)});
    d.lower = st.hold({s.getAxis()(0), s.getAxis()(1), s.getAxis()(2)}); d.radius = s.getCosFov(); d.scale = s.getEpsilon();
    st.c.push_back(d); return true;
  }
  if (typeid(k) == typeid(ThrustMagnitudeConstraint)) {
    const auto &s = static_cast<const ThrustMagnitudeConstraint &>(k);
    d.kind = CDDP_HIP_CON_THRUST; d.dim = nu; d.lower = st.hold({s.getMinThrustNorm()}); d.radius = s.getMaxThrustNorm(); d.scale = s.getEpsilon();
    st.c.push_back(d); return true;
  }
  if (typeid(k) == typeid(MaxThrustMagnitudeConstraint)) {
    const auto &s = static_cast<const MaxThrustMagnitudeConstraint &>(k);
    d.kind = CDDP_HIP_CON_MAX_THRUST; d.dim = nu; d.radius = s.getMaxThrustNorm(); d.scale = s.getEpsilon();
    st.c.push_back(d); return true;
  }
#endif
  return false;   // PoleConstraint, SOC / thrust rows without the getters patch, user subclasses: plug-in route
}
bool describeTerminal(const std::string &key, const Constraint &k, int nx, ConstraintStore &st) {
  cddp_hip_terminal_constraint d;
  std::memset(&d, 0, sizeof(d));
  setName(d.name, key);
  const Eigen::VectorXd x0 = Eigen::VectorXd::Zero(nx), none;
  if (typeid(k) == typeid(TerminalEqualityConstraint)) {
    const Eigen::VectorXd h0 = k.evaluate(x0, none);        // h(x) = x - target  ->  h(0) = -target exactly
    if (h0.size() != nx) return false;
    std::vector<double> target(nx);
    for (int i = 0; i < nx; ++i) target[i] = -h0(i);
    d.kind = CDDP_HIP_TERM_EQUALITY; d.dim = nx; d.target = st.hold(std::move(target));
    st.t.push_back(d); return true;
  }
  if (typeid(k) == typeid(TerminalInequalityConstraint)) {
    const Eigen::MatrixXd A = k.getStateJacobian(x0, none);  // = A_N
    const Eigen::VectorXd g0 = k.evaluate(x0, none);         // g(x) = A x - b  ->  g(0) = -b exactly
    if (A.cols() != nx || A.rows() != g0.size()) return false;
    std::vector<double> b(g0.size());
    for (Eigen::Index i = 0; i < g0.size(); ++i) b[(size_t)i] = -g0(i);
    d.kind = CDDP_HIP_TERM_INEQUALITY; d.dim = (int)g0.size(); d.A = st.hold(rowMajor(A)); d.b = st.hold(std::move(b));
    st.t.push_back(d); return true;
  }
  return false;
}

int solverKind(const std::string &name) {
  if (name == "IPDDP") return CDDP_HIP_SOLVER_IPDDP;
  if (name == "CLDDP" || name == "CLCDDP") return CDDP_HIP_SOLVER_CLDDP;
  if (name == "LogDDP" || name == "LOGDDP") return CDDP_HIP_SOLVER_LOGDDP;
  if (name == "MSIPDDP") return CDDP_HIP_SOLVER_MSIPDDP;
  return -1;
}
const char *solverName(int kind) {
  return kind == CDDP_HIP_SOLVER_IPDDP ? "IPDDP" : kind == CDDP_HIP_SOLVER_LOGDDP ? "LogDDP" : kind == CDDP_HIP_SOLVER_MSIPDDP ? "MSIPDDP" : "CLDDP";
}

// Everything cddp_hip_create needs, with the storage its pointers refer to.
struct FlatProblem {
  cddp_hip_problem p;
  ModelDesc model;
  ObjectiveDesc obj;
  ConstraintStore cons;
};
// true: *f describes the context completely (resident route).  false: some plug-in has no POD description (plug-in route).
bool flatten(CDDP &ctx, int kind, FlatProblem &f) {
  const DynamicalSystem &sys = ctx.getSystem();
  const int nx = ctx.getStateDim(), nu = ctx.getControlDim(), N = ctx.getHorizon();
  const double dt = ctx.getTimestep();
  if (!describeModel(sys, f.model)) return false;
  if (!describeObjective(ctx.getObjective(), dt, nx, nu, N, f.obj)) return false;
  for (const auto &kv : ctx.getConstraintSet())             // std::map iteration = the reference's dual stacking order
    if (!describeConstraint(kv.first, *kv.second, nx, nu, f.cons)) return false;
  for (const auto &kv : ctx.getTerminalConstraintSet())
    if (!describeTerminal(kv.first, *kv.second, nx, f.cons)) return false;
  // resident LogDDP / MSIPDDP kernels: the shapes the library instantiates (csrc/launch.hpp: nx <= 8, no terminal set; MSIPDDP with path
  // constraints only for nu = 1 or nx = nu, the shapes msipddp_solver.cpp:1398 defines) -- everything else runs on the plug-in route
  // The same routing rule as the other two front ends (pycddp_amd.py `logddp_route` / `msipddp_route`, host/cddp_hip.hpp Route): "auto" keeps
  // nx <= 8 on the resident kernels; CDDP_HIP_F4_ROUTE=resident also sends MSIPDDP up to nx = 13 there (one-lane, scratch-backed sweeps:
  // correct, slow), CDDP_HIP_F4_ROUTE=plugin sends every LogDDP / MSIPDDP problem to the host plug-in route (host libm = the reference's
  // arithmetic; the resident kernels use the library's shared log / sin / cos, so knife-edge iteration counts can differ between the routes).
  const char *f4 = std::getenv("CDDP_HIP_F4_ROUTE");
  const std::string f4route = f4 ? f4 : "auto";
  const int ms_nx_cap = (f4route == "resident") ? 13 : 8;
  if (kind == CDDP_HIP_SOLVER_LOGDDP && (nx > 8 || !f.cons.t.empty())) return false;
  if (kind == CDDP_HIP_SOLVER_MSIPDDP && (nx > ms_nx_cap || !f.cons.t.empty())) return false;
  if (kind == CDDP_HIP_SOLVER_MSIPDDP && !f.cons.c.empty() && !(nu == 1 || nx == nu)) return false;
  if ((kind == CDDP_HIP_SOLVER_LOGDDP || kind == CDDP_HIP_SOLVER_MSIPDDP) && f4route == "plugin") return false;
  cddp_hip_problem &p = f.p;
  std::memset(&p, 0, sizeof(p));
  p.abi_version = CDDP_HIP_ABI_VERSION; p.solver = kind; p.model = f.model.id; p.integrator = integratorId(sys.getIntegrationType());
  p.nx = nx; p.nu = nu; p.horizon = N; p.dt = dt;
  if (f.model.params.size() > CDDP_HIP_MAX_MODEL_PARAMS) return false;
  for (size_t i = 0; i < f.model.params.size(); ++i) p.model_params[i] = f.model.params[i];
  p.lti_A = f.model.lti_A.empty() ? nullptr : f.model.lti_A.data();
  p.lti_B = f.model.lti_B.empty() ? nullptr : f.model.lti_B.data();
  p.Q = f.obj.Q.data(); p.R = f.obj.R.data(); p.Qf = f.obj.Qf.data(); p.x_ref = f.obj.x_ref.data();
  p.x_ref_traj = f.obj.x_ref_traj.empty() ? nullptr : f.obj.x_ref_traj.data();
  p.n_constraints = (int)f.cons.c.size(); p.constraints = f.cons.c.empty() ? nullptr : f.cons.c.data();
  p.n_terminal = (int)f.cons.t.size(); p.terminal = f.cons.t.empty() ? nullptr : f.cons.t.data();
  p.options = toPOD(ctx.getOptions(), kind == CDDP_HIP_SOLVER_MSIPDDP);
  return true;
}

// ---- plug-in trampolines: the context's virtual functions behind the flat callbacks of cddp_hip_plugin (include/cddp_hip.h) ------------
struct PluginCtx {
  const DynamicalSystem *sys = nullptr;
  const Objective *obj = nullptr;
  std::vector<const Constraint *> cons;
  std::vector<const Constraint *> terms;   // terminal set in std::map order (cddp_hip_plugin_solve_terminal)
  int nx = 0, nu = 0, m = 0;
  std::exception_ptr error;   // a C++ exception must not unwind through the C library: parked here, rethrown after the call
  volatile int32_t abort_flag = 0;
  Eigen::VectorXd x, u;
  void load(const double *xp, const double *up) {
    x = Eigen::Map<const Eigen::VectorXd>(xp, nx);
    if (up) u = Eigen::Map<const Eigen::VectorXd>(up, nu);
  }
};
template <class F>
void guarded(PluginCtx *c, F f) {
  if (c->error) return;
  try { f(); } catch (...) { c->error = std::current_exception(); c->abort_flag = 1; }
}
void cbDyn(void *p, const double *x, const double *u, double t, double *xn) {
  auto *c = static_cast<PluginCtx *>(p);
  std::fill(xn, xn + c->nx, std::numeric_limits<double>::quiet_NaN());
  guarded(c, [&] {
    c->load(x, u);
    const Eigen::VectorXd r = c->sys->getDiscreteDynamics(c->x, c->u, t);
    if (r.size() != c->nx) throw std::runtime_error("getDiscreteDynamics: unexpected size");
    std::copy(r.data(), r.data() + c->nx, xn);
  });
}
void cbTerminal(void *p, const double *xN, double *r, double *rx) {   // residual and state-Jacobian rows of the terminal set, stacked in std::map order
  auto *c = static_cast<PluginCtx *>(p);
  guarded(c, [&] {
    c->load(xN, nullptr);
    int row = 0;
    for (const Constraint *k : c->terms) {
      const Eigen::VectorXd v = k->evaluate(c->x, Eigen::VectorXd());
      for (int i = 0; i < v.size(); ++i) r[row + i] = v(i);
      if (rx) {
        const Eigen::MatrixXd J = k->getStateJacobian(c->x, Eigen::VectorXd());
        if (J.rows() != v.size() || J.cols() != c->nx) throw std::runtime_error("terminal getStateJacobian: unexpected shape");
        for (int i = 0; i < J.rows(); ++i) for (int j = 0; j < c->nx; ++j) rx[(size_t)(row + i) * c->nx + j] = J(i, j);
      }
      row += (int)v.size();
    }
  });
}
void cbJac(void *p, const double *x, const double *u, double t, double *fx, double *fu) {   // CONTINUOUS-time f_x, f_u (cddp_solver_base.cpp:340-344 forms A, B)
  auto *c = static_cast<PluginCtx *>(p);
  std::fill(fx, fx + c->nx * c->nx, 0.0); std::fill(fu, fu + c->nx * c->nu, 0.0);
  guarded(c, [&] {
    c->load(x, u);
    copyRowMajor(c->sys->getStateJacobian(c->x, c->u, t), c->nx, c->nx, fx, "getStateJacobian");
    copyRowMajor(c->sys->getControlJacobian(c->x, c->u, t), c->nx, c->nu, fu, "getControlJacobian");
  });
}
void cbHess(void *p, const double *x, const double *u, double t, double *fxx, double *fuu, double *fux) {
  auto *c = static_cast<PluginCtx *>(p);
  const int nx = c->nx, nu = c->nu;
  std::fill(fxx, fxx + nx * nx * nx, 0.0); std::fill(fuu, fuu + nx * nu * nu, 0.0); std::fill(fux, fux + nx * nu * nx, 0.0);
  guarded(c, [&] {
    c->load(x, u);
    const auto a = c->sys->getStateHessian(c->x, c->u, t);
    const auto b = c->sys->getControlHessian(c->x, c->u, t);
    const auto d = c->sys->getCrossHessian(c->x, c->u, t);
    if ((int)a.size() != nx || (int)b.size() != nx || (int)d.size() != nx) throw std::runtime_error("dynamics Hessians: one matrix per state row expected");
    for (int i = 0; i < nx; ++i) {
      copyRowMajor(a[i], nx, nx, fxx + (size_t)i * nx * nx, "getStateHessian");
      copyRowMajor(b[i], nu, nu, fuu + (size_t)i * nu * nu, "getControlHessian");
      copyRowMajor(d[i], nu, nx, fux + (size_t)i * nu * nx, "getCrossHessian");
    }
  });
}
double cbRun(void *p, const double *x, const double *u, int k) {
  auto *c = static_cast<PluginCtx *>(p);
  double v = std::numeric_limits<double>::quiet_NaN();
  guarded(c, [&] { c->load(x, u); v = c->obj->running_cost(c->x, c->u, k); });
  return v;
}
double cbTerm(void *p, const double *x) {
  auto *c = static_cast<PluginCtx *>(p);
  double v = std::numeric_limits<double>::quiet_NaN();
  guarded(c, [&] { c->load(x, nullptr); v = c->obj->terminal_cost(c->x); });
  return v;
}
void cbRunD(void *p, const double *x, const double *u, int k, double *lx, double *lu, double *lxx, double *luu, double *lux) {
  auto *c = static_cast<PluginCtx *>(p);
  const int nx = c->nx, nu = c->nu;
  std::fill(lx, lx + nx, 0.0); std::fill(lu, lu + nu, 0.0); std::fill(lxx, lxx + nx * nx, 0.0); std::fill(luu, luu + nu * nu, 0.0); std::fill(lux, lux + nu * nx, 0.0);
  guarded(c, [&] {
    c->load(x, u);
    const Eigen::VectorXd gx = c->obj->getRunningCostStateGradient(c->x, c->u, k), gu = c->obj->getRunningCostControlGradient(c->x, c->u, k);
    if (gx.size() != nx || gu.size() != nu) throw std::runtime_error("running-cost gradients: unexpected size");
    std::copy(gx.data(), gx.data() + nx, lx); std::copy(gu.data(), gu.data() + nu, lu);
    copyRowMajor(c->obj->getRunningCostStateHessian(c->x, c->u, k), nx, nx, lxx, "getRunningCostStateHessian");
    copyRowMajor(c->obj->getRunningCostControlHessian(c->x, c->u, k), nu, nu, luu, "getRunningCostControlHessian");
    copyRowMajor(c->obj->getRunningCostCrossHessian(c->x, c->u, k), nu, nx, lux, "getRunningCostCrossHessian");
  });
}
void cbTermD(void *p, const double *x, double *lx, double *lxx) {
  auto *c = static_cast<PluginCtx *>(p);
  std::fill(lx, lx + c->nx, 0.0); std::fill(lxx, lxx + c->nx * c->nx, 0.0);
  guarded(c, [&] {
    c->load(x, nullptr);
    const Eigen::VectorXd g = c->obj->getFinalCostGradient(c->x);
    if (g.size() != c->nx) throw std::runtime_error("getFinalCostGradient: unexpected size");
    std::copy(g.data(), g.data() + c->nx, lx);
    copyRowMajor(c->obj->getFinalCostHessian(c->x), c->nx, c->nx, lxx, "getFinalCostHessian");
  });
}
void cbCon(void *p, const double *x, const double *u, int index, double *g, double *gx, double *gu) {
  auto *c = static_cast<PluginCtx *>(p);
  std::fill(g, g + c->m, -1.0);
  if (gx) std::fill(gx, gx + c->m * c->nx, 0.0);
  if (gu) std::fill(gu, gu + c->m * c->nu, 0.0);
  guarded(c, [&] {
    c->load(x, u);
    int off = 0;
    for (const Constraint *k : c->cons) {   // g = evaluate - getUpperBound, constraint by constraint (ipddp_solver.cpp:2145-2250)
      const int d = k->getDualDim();
      const Eigen::VectorXd e = k->evaluate(c->x, c->u, index), ub = k->getUpperBound();
      if (e.size() != d || ub.size() != d) throw std::runtime_error("constraint '" + k->getName() + "': evaluate / getUpperBound size differs from getDualDim");
      for (int i = 0; i < d; ++i) g[off + i] = e(i) - ub(i);
      if (gx) copyRowMajor(k->getStateJacobian(c->x, c->u, index), d, c->nx, gx + (size_t)off * c->nx, "Constraint::getStateJacobian");
      if (gu) copyRowMajor(k->getControlJacobian(c->x, c->u, index), d, c->nu, gu + (size_t)off * c->nu, "Constraint::getControlJacobian");
      off += d;
    }
  });
}
void cbConHess(void *p, const double *x, const double *u, int index, double *gxx, double *guu, double *gux) {
  auto *c = static_cast<PluginCtx *>(p);
  guarded(c, [&] {
    c->load(x, u);
    int off = 0;
    for (const Constraint *k : c->cons) {
      const int d = k->getDualDim();
      try {   // constraints without curvature information throw std::logic_error or return empty lists: their rows stay zero (barrier.hpp:137-213)
        const auto hxx = k->getStateHessian(c->x, c->u, index);
        const auto huu = k->getControlHessian(c->x, c->u, index);
        const auto hux = k->getCrossHessian(c->x, c->u, index);
        for (int r = 0; r < d; ++r) {
          if (r < (int)hxx.size()) copyRowMajor(hxx[r], c->nx, c->nx, gxx + (size_t)(off + r) * c->nx * c->nx, "Constraint::getStateHessian");
          if (r < (int)huu.size()) copyRowMajor(huu[r], c->nu, c->nu, guu + (size_t)(off + r) * c->nu * c->nu, "Constraint::getControlHessian");
          if (r < (int)hux.size()) copyRowMajor(hux[r], c->nu, c->nx, gux + (size_t)(off + r) * c->nu * c->nx, "Constraint::getCrossHessian");
        }
      } catch (const std::logic_error &) {
      }
      off += d;
    }
  });
}

void fillSolution(CDDPSolution &s, const char *name, const cddp_hip_result &r, double solve_ms, int nx, int nu, int N, double dt,
                  const double *X, const double *U, const double *K) {
  s.solver_name = name; s.status_message = cddp_hip_status_string(r.status);
  s.iterations_completed = r.iterations; s.solve_time_ms = solve_ms; s.final_objective = r.final_objective;
  s.final_step_length = r.alpha_pr; s.final_regularization = r.regularization;
  s.final_primal_infeasibility = r.inf_pr; s.final_dual_infeasibility = r.inf_du; s.final_complementary_infeasibility = r.inf_comp; s.final_barrier_mu = r.barrier_mu;
  s.time_points.clear(); s.state_trajectory.clear(); s.control_trajectory.clear(); s.feedback_gains.clear();
  for (int t = 0; t <= N; ++t) {
    s.time_points.push_back(t * dt);
    s.state_trajectory.emplace_back(Eigen::Map<const Eigen::VectorXd>(X + (size_t)t * nx, nx));
  }
  for (int t = 0; t < N; ++t) {
    s.control_trajectory.emplace_back(Eigen::Map<const Eigen::VectorXd>(U + (size_t)t * nu, nu));
    Eigen::MatrixXd Kt(nu, nx);
    for (int i = 0; i < nu; ++i) for (int j = 0; j < nx; ++j) Kt(i, j) = K[((size_t)t * nu + i) * nx + j];
    s.feedback_gains.push_back(Kt);
  }
}

// The strategy object CDDP::createSolver gets from the registry.
class HipBatchSolver : public ISolverAlgorithm {
 public:
  HipBatchSolver(int kind, int device) : kind_(kind), device_(device) {}
  ~HipBatchSolver() override { if (h_) cddp_hip_destroy(h_); }
  std::string getSolverName() const override { return solverName(kind_); }

  // CDDP::solve calls initialize() then solve() on a FRESH solver object (cddp_core.cpp:235-270), so through CDDP::solve a warm start is
  // the reference's "provided trajectory" branch (ipddp_solver.cpp:733-816): options.warm_start travels in the POD, X_ / U_ are uploaded.
  void initialize(CDDP &ctx) override { create(ctx, {ctx.getInitialState()}); }

  CDDPSolution solve(CDDP &ctx) override {
    std::vector<CDDPSolution> s = collect(ctx, 1);
    // leave the context updated as the reference solvers do (cddp_solver_base.cpp:161-171)
    ctx.X_ = s[0].state_trajectory; ctx.U_ = s[0].control_trajectory; ctx.cost_ = s[0].final_objective;
    ctx.alpha_pr_ = s[0].final_step_length; ctx.regularization_ = s[0].final_regularization;
    ctx.inf_pr_ = s[0].final_primal_infeasibility; ctx.inf_du_ = s[0].final_dual_infeasibility; ctx.inf_comp_ = s[0].final_complementary_infeasibility;
    return s[0];
  }

  std::vector<CDDPSolution> solveBatch(CDDP &ctx, const std::vector<Eigen::VectorXd> &x0s) {
    if (x0s.empty()) return {};
    create(ctx, x0s);
    return collect(ctx, (int)x0s.size());
  }
  bool plugin() const { return plugin_; }

  void create(CDDP &ctx, const std::vector<Eigen::VectorXd> &x0s) {
    if (h_) { cddp_hip_destroy(h_); h_ = nullptr; }
    nx_ = ctx.getStateDim(); nu_ = ctx.getControlDim(); N_ = ctx.getHorizon(); dt_ = ctx.getTimestep();
    for (const auto &x : x0s) if (x.size() != nx_) throw std::invalid_argument("cddp_hip adapter: an initial state has the wrong dimension");
    x0s_ = x0s;
    FlatProblem f;
    plugin_ = !flatten(ctx, kind_, f);
    if (plugin_) {
      return;   // nothing to create: cddp_hip_plugin_solve / cddp_hip_plugin_solve_terminal is one call
    }
    const int B = (int)x0s.size();
    check(cddp_hip_create(&f.p, B, device_, &h_));
    ret_hist_ = f.p.options.return_iteration_info != 0; max_it_ = f.p.options.max_iterations;
    std::vector<double> x0((size_t)B * nx_), U0, X0;
    for (int b = 0; b < B; ++b) for (int i = 0; i < nx_; ++i) x0[(size_t)b * nx_ + i] = x0s[b](i);
    flattenGuess(ctx, B, U0, X0);
    check(cddp_hip_set_initial(h_, x0.data(), U0.empty() ? nullptr : U0.data(), X0.empty() ? nullptr : X0.data()));
  }

 private:
  // the context's initial trajectory guess, replicated over the batch (CDDP::setInitialTrajectory, cddp_core.cpp:160-190)
  void flattenGuess(CDDP &ctx, int B, std::vector<double> &U0, std::vector<double> &X0) const {
    if ((int)ctx.U_.size() == N_) {
      U0.resize((size_t)B * N_ * nu_);
      for (int b = 0; b < B; ++b) for (int t = 0; t < N_; ++t) for (int i = 0; i < nu_; ++i) U0[((size_t)b * N_ + t) * nu_ + i] = ctx.U_[t](i);
    }
    if ((int)ctx.X_.size() == N_ + 1) {
      X0.resize((size_t)B * (N_ + 1) * nx_);
      for (int b = 0; b < B; ++b) for (int t = 0; t <= N_; ++t) for (int i = 0; i < nx_; ++i) X0[((size_t)b * (N_ + 1) + t) * nx_ + i] = ctx.X_[t](i);
    }
  }

  std::vector<CDDPSolution> collectPlugin(CDDP &ctx, int B) {
    PluginCtx pc;
    pc.sys = &ctx.getSystem(); pc.obj = &ctx.getObjective(); pc.nx = nx_; pc.nu = nu_;
    cddp_hip_plugin pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.abi_version = CDDP_HIP_ABI_VERSION; pl.options_bytes = (int)sizeof(cddp_hip_options); pl.abort_flag = &pc.abort_flag;
    pl.user = &pc; pl.nx = nx_; pl.nu = nu_;
    std::vector<double> lower, upper;
    if (kind_ == CDDP_HIP_SOLVER_CLDDP) {   // clddp_solver.cpp:85-86: only the constraint literally named "ControlConstraint"
      if (const ControlConstraint *box = ctx.getConstraint<ControlConstraint>("ControlConstraint")) {
        lower = toStd(box->rawLowerBound()); upper = toStd(box->rawUpperBound());
        pl.control_lower = lower.data(); pl.control_upper = upper.data();
      }
    } else {
      for (const auto &kv : ctx.getConstraintSet()) {   // std::map order == dual stacking order
        if ((int)pc.cons.size() == CDDP_HIP_PLUGIN_MAX_CONSTRAINTS) throw std::runtime_error("cddp_hip adapter: more than CDDP_HIP_PLUGIN_MAX_CONSTRAINTS path constraints");
        pl.constraint_dims[pc.cons.size()] = kv.second->getDualDim(); pc.m += kv.second->getDualDim(); pc.cons.push_back(kv.second.get());
      }
      pl.n_constraints = (int)pc.cons.size();
    }
    const CDDPOptions &opt = ctx.getOptions();
    pl.discrete_dynamics = cbDyn; pl.jacobians = cbJac; pl.hessians = opt.use_ilqr ? nullptr : cbHess;
    pl.running_cost = cbRun; pl.terminal_cost = cbTerm; pl.running_cost_derivatives = cbRunD; pl.terminal_cost_derivatives = cbTermD;
    pl.constraints = pc.cons.empty() ? nullptr : cbCon;
    pl.constraint_hessians = ((kind_ == CDDP_HIP_SOLVER_LOGDDP || (kind_ == CDDP_HIP_SOLVER_MSIPDDP && !opt.use_ilqr)) && !pc.cons.empty()) ? cbConHess : nullptr;
    std::vector<double> x0((size_t)B * nx_), U0, X0;
    for (int b = 0; b < B; ++b) for (int i = 0; i < nx_; ++i) x0[(size_t)b * nx_ + i] = x0s_[b](i);
    flattenGuess(ctx, B, U0, X0);
    std::vector<cddp_hip_result> r(B);
    std::vector<double> X((size_t)B * (N_ + 1) * nx_), U((size_t)B * N_ * nu_), K((size_t)B * N_ * nu_ * nx_);
    const cddp_hip_options o = toPOD(opt, kind_ == CDDP_HIP_SOLVER_MSIPDDP);
    // terminal set (only IPDDP reads it, ipddp_solver.cpp:84-215): ANY Constraint subclass -- evaluate(x_N) and getStateJacobian(x_N) through the
    // virtuals, classified as the reference classifies them (dynamic_cast to TerminalEqualityConstraint / TerminalInequalityConstraint, :84-105)
    cddp_hip_plugin_terminal tc; std::memset(&tc, 0, sizeof(tc));
    if (kind_ == CDDP_HIP_SOLVER_IPDDP) {
      for (const auto &kv : ctx.getTerminalConstraintSet()) {
        const bool eq = dynamic_cast<const TerminalEqualityConstraint *>(kv.second.get()) != nullptr;
        if (!eq && dynamic_cast<const TerminalInequalityConstraint *>(kv.second.get()) == nullptr) continue;   // neither layout lists it
        if (tc.n_terminal == CDDP_HIP_PLUGIN_MAX_CONSTRAINTS) throw std::runtime_error("cddp_hip adapter: too many terminal constraints for the plug-in solve");
        tc.dims[tc.n_terminal] = kv.second->getDualDim(); tc.equality[tc.n_terminal] = eq ? 1 : 0; ++tc.n_terminal;
        pc.terms.push_back(kv.second.get());
      }
      tc.evaluate = cbTerminal;
    }
    const int rc = tc.n_terminal > 0
        ? cddp_hip_plugin_solve_terminal(&pl, &tc, kind_, N_, dt_, &o, device_, B, x0.data(), U0.empty() ? nullptr : U0.data(), X0.empty() ? nullptr : X0.data(),
                                         r.data(), X.data(), U.data(), K.data(), nullptr)
        : cddp_hip_plugin_solve(&pl, kind_, N_, dt_, &o, device_, B, x0.data(), U0.empty() ? nullptr : U0.data(), X0.empty() ? nullptr : X0.data(),
                                r.data(), X.data(), U.data(), K.data());
    if (pc.error) std::rethrow_exception(pc.error);   // the plug-in's own exception, as the reference would have propagated it
    check(rc);
    std::vector<CDDPSolution> out(B);
    for (int b = 0; b < B; ++b)
      fillSolution(out[b], solverName(kind_), r[b], 0.0, nx_, nu_, N_, dt_, X.data() + (size_t)b * (N_ + 1) * nx_, U.data() + (size_t)b * N_ * nu_, K.data() + (size_t)b * N_ * nu_ * nx_);
    return out;
  }

  std::vector<CDDPSolution> collect(CDDP &ctx, int B) {
    if (plugin_) return collectPlugin(ctx, B);
    cddp_hip_stats st;
    check(cddp_hip_solve(h_, &st));
    std::vector<cddp_hip_result> r(B);
    check(cddp_hip_get_results(h_, r.data()));
    std::vector<double> X((size_t)B * (N_ + 1) * nx_), U((size_t)B * N_ * nu_), K((size_t)B * N_ * nu_ * nx_);
    check(cddp_hip_get_trajectory(h_, X.data(), U.data()));
    check(cddp_hip_get_gains(h_, K.data(), nullptr));
    const int HB = ret_hist_ ? std::min(B, 64) : 0;   // the library keeps the history of the first min(batch, 64) trajectories
    std::vector<double> hist; std::vector<int32_t> hn;
    if (HB) { hist.resize((size_t)HB * (max_it_ + 1) * 9); hn.resize(HB); check(cddp_hip_get_history(h_, HB, hist.data(), hn.data())); }
    std::vector<CDDPSolution> out(B);
    const bool ip = kind_ != CDDP_HIP_SOLVER_CLDDP;
    for (int b = 0; b < B; ++b) {
      fillSolution(out[b], solverName(kind_), r[b], st.solve_ms, nx_, nu_, N_, dt_, X.data() + (size_t)b * (N_ + 1) * nx_, U.data() + (size_t)b * N_ * nu_, K.data() + (size_t)b * N_ * nu_ * nx_);
      if (b < HB) {
        CDDPSolution::History &h = out[b].history;
        for (int i = 0; i < hn[b]; ++i) {
          const double *row = hist.data() + ((size_t)b * (max_it_ + 1) + i) * 9;
          h.objective.push_back(row[0]); h.merit_function.push_back(row[1]); h.step_length_primal.push_back(row[2]); h.step_length_dual.push_back(row[3]);
          h.dual_infeasibility.push_back(row[4]); h.primal_infeasibility.push_back(row[5]); h.complementary_infeasibility.push_back(row[6]);
          if (ip) h.barrier_mu.push_back(row[7]);
          h.regularization.push_back(row[8]);
        }
      }
    }
    return out;
  }

  int kind_, device_;
  cddp_hip_handle *h_ = nullptr;
  int nx_ = 0, nu_ = 0, N_ = 0, max_it_ = 0;
  double dt_ = 0.0;
  bool ret_hist_ = false, plugin_ = false;
  std::vector<Eigen::VectorXd> x0s_;
};

}  // namespace

void registerHipSolvers(int device) {
  for (const char *name : {"IPDDP", "CLDDP", "CLCDDP", "LogDDP", "LOGDDP", "MSIPDDP"}) {
    const int kind = solverKind(name);
    CDDP::registerSolver(name, [kind, device] { return std::unique_ptr<ISolverAlgorithm>(new HipBatchSolver(kind, device)); });
  }
}

std::vector<CDDPSolution> solveBatchHip(CDDP &context, const std::string &solver_type, const std::vector<Eigen::VectorXd> &x0s, int device) {
  const int kind = solverKind(solver_type);
  if (kind < 0) {   // cddp_core.cpp:243-265: unknown names do not throw
    CDDPSolution s;
    s.solver_name = solver_type; s.status_message = "UnknownSolver - No solver registered for '" + solver_type + "'";
    s.iterations_completed = 0; s.solve_time_ms = 0.0; s.final_objective = 0.0; s.final_step_length = 1.0;
    return std::vector<CDDPSolution>(x0s.size(), s);
  }
  // CDDP::solve sizes X_ / U_ through initializeProblemIfNecessary() (private, cddp_core.cpp:272-306) before the strategy runs; a batch
  // call made before any solve() reaches the adapter with whatever setInitialTrajectory left: an unset guess is uploaded as "none"
  // (the library then uses zeros for U and x0 replicated for X, exactly what initializeProblemIfNecessary builds).
  HipBatchSolver s(kind, device);
  return s.solveBatch(context, x0s);
}

std::string hipRouteOf(CDDP &context, const std::string &solver_type) {
  const int kind = solverKind(solver_type);
  if (kind < 0) return "unknown";
  FlatProblem f;
  return flatten(context, kind, f) ? "resident" : "plugin";
}

}  // namespace cddp
