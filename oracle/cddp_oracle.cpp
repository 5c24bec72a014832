// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// CPU restatement (scalar, one trajectory at a time, fp64) of the hot path of
// astomodynamics/cddp-cpp v0.5.2: CDDP::solve -> {CLDDPSolver, IPDDPSolver} ->
// backwardPass / forwardPass / BoxQP.  Every function cites the reference file:line it
// follows (paths relative to /root/reference).
//
// The reference itself cannot be built here: its arithmetic lives in Eigen 3.4.0 and
// autodiff v1.1.2, both FetchContent dependencies (CMakeLists.txt:65-97,116-125) absent from
// /root/reference and from this image.  Their pieces on the path are restated in
// linalg.hpp / models.hpp.
//
// parity unpinned: the reference's own tests pin NO gain, value-function or iteration-count
// number for this path; what they do pin (status strings, terminal-state
// bounds of the scalar-integrator regressions, model closed forms, filter/barrier
// behaviours) is replayed against this oracle in tests/test_oracle_pins.py.
#include <array>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <atomic>

#include "models.hpp"

namespace oracle {

namespace {
// ipddp_solver.cpp:35-38
constexpr double kSlackInteriorOffset = 1e-4;
constexpr double EPS_SLACK = 1e-10;
constexpr double EPS_DUAL = 1e-10;
constexpr double MAX_BARRIER_RATIO = 1e6;

inline double clampd(double v, double lo, double hi) { return std::min(std::max(v, lo), hi); }  // std::clamp
// ipddp_solver.cpp:222-231
inline double clipPositiveBarrierRatio(double num, double den) { return clampd(num / den, 0.0, MAX_BARRIER_RATIO); }
inline double clipSignedBarrierRatio(double num, double den) { return clampd(num / den, -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO); }
inline Mat cwiseClamp(const Mat &A, double lo, double hi) {  // .cwiseMax(lo).cwiseMin(hi)
  Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = std::min(std::max(A.a[i], lo), hi); return m;
}
}  // namespace

struct ConstraintDesc {
  std::string name;
  int kind = 0, dim = 0, dual_dim = 0, offset = 0;
  Vec lower, upper, center, b, ip_upper;
  Mat A;
  double radius = 0, scale = 1.0;
};
struct TerminalDesc {
  std::string name;
  int kind = 0, dim = 0;
  Vec target, b;
  Mat A;
};
struct FilterPoint {  // cddp_core.hpp:153-175
  double merit_function = 0, constraint_violation = 0;
  bool dominates(const FilterPoint &o) const {
    return merit_function <= o.merit_function && constraint_violation <= o.constraint_violation;
  }
};
struct History {  // cddp_core.hpp:77-102
  std::vector<std::array<double, 9>> rows;  // objective, merit, a_pr, a_du, inf_du, inf_pr, inf_comp, mu, reg
};

// BoxQPSolver (boxqp.cpp:25-250) -------------------------------------------------------------
enum BoxQPStatus { HESSIAN_NOT_PD = -1, NO_DESCENT = 0, MAX_ITER_EXCEEDED = 1, MAX_LS_EXCEEDED = 2,
                   NO_BOUNDS = 3, SUCCESS = 4, ALL_CLAMPED = 5 };
struct BoxQPResult {
  Vec x; int status = MAX_ITER_EXCEEDED; LDLT Hfree; std::vector<int> free_;
  double final_value = 0, final_grad_norm = 0; int iterations = 0, factorizations = 0;
};

static double boxqp_objective(const Vec &x, const Mat &H, const Vec &g) {  // boxqp.cpp:235-239
  return 0.5 * x.dot(H * x) + g.dot(x);
}
static Vec boxqp_project(const Vec &x, const Vec &lo, const Vec &up) {  // boxqp.cpp:241-250
  Vec p = x; for (int i = 0; i < x.size(); ++i) p(i) = std::min(std::max(x(i), lo(i)), up(i)); return p;
}

BoxQPResult boxqp_solve(const cddp_hip_options &o, const Mat &H, const Vec &g, const Vec &lower,
                        const Vec &upper, const Vec &x0) {
  const int n = H.r;
  BoxQPResult result;
  result.status = MAX_ITER_EXCEEDED;
  // initializeX (boxqp.cpp:184-205)
  if (x0.size() == n) result.x = boxqp_project(x0, lower, upper);
  else {
    result.x = Vec(n, 1);
    for (int i = 0; i < n; ++i) {
      if (std::isfinite(lower(i)) && std::isfinite(upper(i))) result.x(i) = 0.5 * (lower(i) + upper(i));
      else if (std::isfinite(lower(i))) result.x(i) = lower(i);
      else if (std::isfinite(upper(i))) result.x(i) = upper(i);
      else result.x(i) = 0.0;
    }
  }
  std::vector<int> clamped(n, 0);
  result.free_.assign(n, 1);
  double value = boxqp_objective(result.x, H, g);
  double old_value = std::numeric_limits<double>::infinity();
  for (int iter = 0; iter < o.boxqp_max_iterations; ++iter) {
    result.iterations = iter + 1;
    if (iter > 0 && std::fabs(old_value - value) < o.boxqp_min_relative_improvement * std::fabs(old_value)) {
      result.status = SUCCESS; break;
    }
    old_value = value;
    Vec grad = g + H * result.x;
    std::vector<int> old_clamped = clamped;
    std::fill(clamped.begin(), clamped.end(), 0);
    int nclamped = 0;
    for (int i = 0; i < n; ++i) {
      if ((result.x(i) == lower(i) && grad(i) > 0) || (result.x(i) == upper(i) && grad(i) < 0)) { clamped[i] = 1; ++nclamped; }
    }
    for (int i = 0; i < n; ++i) result.free_[i] = 1 - clamped[i];
    if (nclamped == n) { result.status = ALL_CLAMPED; break; }
    bool any_different = false;
    for (int i = 0; i < n; ++i) if (old_clamped[i] != clamped[i]) { any_different = true; break; }
    bool factorize = (iter == 0) || any_different;
    std::vector<int> free_idx;
    for (int i = 0; i < n; ++i) if (!clamped[i]) free_idx.push_back(i);
    if (factorize) {
      Mat Hf((int)free_idx.size(), (int)free_idx.size());
      for (size_t i = 0; i < free_idx.size(); ++i)
        for (size_t j = 0; j < free_idx.size(); ++j) Hf((int)i, (int)j) = H(free_idx[i], free_idx[j]);
      result.Hfree.compute(Hf);
      if (!result.Hfree.ok) { result.status = HESSIAN_NOT_PD; break; }
      result.factorizations++;
    }
    double grad_norm = 0;
    for (int i = 0; i < n; ++i) if (!clamped[i]) grad_norm += grad(i) * grad(i);
    grad_norm = std::sqrt(grad_norm);
    result.final_grad_norm = grad_norm;
    if (grad_norm < o.boxqp_min_gradient_norm) { result.status = SUCCESS; break; }
    Vec search = Vec::Zero(n);
    Vec grad_clamped = g;
    for (int i = 0; i < n; ++i) if (clamped[i]) grad_clamped += H.col(i) * result.x(i);
    Vec grad_free((int)free_idx.size(), 1);
    for (size_t i = 0; i < free_idx.size(); ++i) grad_free((int)i) = grad_clamped(free_idx[i]);
    Vec search_free = -result.Hfree.solve(grad_free);
    for (size_t i = 0; i < free_idx.size(); ++i) search(free_idx[i]) = search_free((int)i) - result.x(free_idx[i]);
    double sdotg = search.dot(grad);
    if (sdotg >= 0) { result.status = NO_DESCENT; break; }
    // lineSearch (boxqp.cpp:207-233)
    double step = 1.0;
    bool ls_ok = false;
    Vec x_new = result.x;
    while (step > o.boxqp_min_step_size) {
      Vec cand = boxqp_project(result.x + step * search, lower, upper);
      double value_new = boxqp_objective(cand, H, g);
      if ((value_new - value) <= o.boxqp_armijo_constant * step * sdotg) { ls_ok = true; x_new = cand; break; }
      step *= o.boxqp_step_decrease_factor;
    }
    if (!ls_ok) { result.status = MAX_LS_EXCEEDED; break; }
    result.x = x_new;
    value = boxqp_objective(result.x, H, g);
  }
  result.final_value = value;
  return result;
}

// ForwardPassResult (cddp_core.hpp:105-145) ---------------------------------------------------
struct FPResult {
  std::vector<Vec> X, U;
  double cost = 0, merit = 0, alpha_pr = 1.0, alpha_du = 1.0, alpha = 1.0;
  bool success = false;
  double theta = 0, inf_pr = 0, inf_comp = 0;
  std::vector<Vec> S, Y, G, Lambda;
  std::vector<Vec> F;                    // MSIPDDP: dynamics values f(x_t, u_t) of the trial (dynamics_trajectory)
  std::map<std::string, Vec> S_T, Y_T, G_T;
  Vec Lambda_T_eq;
  bool has_ip = false;
};

struct Solver {
  // ---------------- problem (cddp::CDDP members, cddp_core.hpp:323-423)
  Model model;
  int nx = 0, nu = 0, N = 0, solver_kind = 0;
  double dt = 0;
  Mat Qdt, Rdt, Qf;       // Q*dt, R*dt (objective.cpp:38-39)
  Vec xref;
  std::vector<Vec> xref_traj;
  std::vector<ConstraintDesc> cons;   // std::map order == sorted by name
  std::vector<TerminalDesc> terms;    // sorted by name
  cddp_hip_options opt;
  int m = 0;  // total path dual dim

  // ---------------- context iterate state
  Vec x0;
  std::vector<Vec> X, U;
  double cost = 0, merit = 0, inf_pr = 0, inf_du = 0, inf_comp = 0, step_norm = 0;
  double alpha_pr = 1.0, alpha_du = 0.0, reg = 0;
  std::vector<double> alphas;

  // ---------------- solver state (cddp_solver_base.hpp, ipddp_solver.hpp)
  std::vector<Vec> k_u;
  std::vector<Mat> K_u;
  double dV[2] = {0, 0};
  std::vector<Vec> Vx_t;   // value gradient per step (k_lambda_ for IPDDP)
  std::vector<Mat> Vxx_t;  // value Hessian per step (K_lambda_ for IPDDP)
  std::vector<Mat> F_x, F_u;
  std::vector<std::vector<Mat>> F_xx, F_uu, F_ux;   // dt-scaled Hessian tensors per step (use_ilqr = false)
  double mu = 0.1;
  std::vector<Vec> S, Y, G, dS, dY, k_s, k_y, Lambda, dX, dU;
  std::vector<Mat> K_s, K_y, Gx, Gu;
  std::map<std::string, Vec> S_T, Y_T, G_T, dS_T, dY_T;
  Vec Lambda_T_eq, dLambda_T_eq;
  std::vector<FilterPoint> filter;
  double phi = 0, theta = 0, filter_theta = 0;
  History history;
  int n_backward = 0, n_forward = 0, iterations = 0, status = CDDP_HIP_STATUS_RUNNING;

  // =============================================================== objective (objective.cpp:80-154)
  Vec state_error(const Vec &x, int index) const {
    if (!xref_traj.empty()) return x - xref_traj[index];
    return x - xref;
  }
  double running_cost(const Vec &x, const Vec &u, int index) const {
    Vec e = state_error(x, index);
    return (e.T() * Qdt * e)(0) + (u.T() * Rdt * u)(0);
  }
  double terminal_cost(const Vec &xN) const { Vec e = xN - xref; return (e.T() * Qf * e)(0); }
  double objective_evaluate(const std::vector<Vec> &Xs, const std::vector<Vec> &Us) const {  // :67-77
    double total = 0.0;
    for (int t = 0; t < (int)Xs.size() - 1; ++t) total += running_cost(Xs[t], Us[t], t);
    total += terminal_cost(Xs.back());
    return total;
  }
  Vec l_x(const Vec &x, int index) const { return 2.0 * Qdt * state_error(x, index); }
  Vec l_u(const Vec &u) const { return 2.0 * Rdt * u; }
  Mat l_xx() const { return 2.0 * Qdt; }
  Mat l_uu() const { return 2.0 * Rdt; }
  Mat l_ux() const { return Mat::Zero(nu, nx); }
  Vec final_grad(const Vec &xN) const { return 2.0 * Qf * (xN - xref); }
  Mat final_hess() const { return 2.0 * Qf; }

  // =============================================================== constraints (constraint.hpp)
  // evaluate(x,u) - getUpperBound() for one constraint
  Vec con_g(const ConstraintDesc &c, const Vec &x, const Vec &u) const {
    Vec g(c.dual_dim, 1);
    switch (c.kind) {
      case CDDP_HIP_CON_CONTROL_BOX:
      case CDDP_HIP_CON_STATE_BOX: {  // constraint.hpp:166-176
        const Vec &v = (c.kind == CDDP_HIP_CON_CONTROL_BOX) ? u : x;
        for (int i = 0; i < c.dim; ++i) { g(i) = -v(i); g(c.dim + i) = v(i); }
        g = g * c.scale;
        return g - c.ip_upper;
      }
      case CDDP_HIP_CON_BALL: {  // constraint.hpp:326-343
        double sq = 0; for (int i = 0; i < c.dim; ++i) { double d = x(i) - c.center(i); sq += d * d; }
        g(0) = -(c.scale * sq);
        return g - c.ip_upper;
      }
      case CDDP_HIP_CON_LINEAR: {  // constraint.hpp:263-277
        return c.A * x - c.b;
      }
      case CDDP_HIP_CON_SOC: {     // constraint.hpp:672-692: g = cos(fov) sqrt(|p_s - p_o|^2 + eps) - (p_s - p_o) . axis; upper bound 0
        const double v0 = x(0) - c.center(0), v1 = x(1) - c.center(1), v2 = x(2) - c.center(2);
        const double v_squared = (v0 * v0 + v1 * v1) + v2 * v2;           // Vector3d::squaredNorm (fixed size, unrolled)
        const double reg_norm = std::sqrt(v_squared + c.scale);
        const double dot_prod = (v0 * c.lower(0) + v1 * c.lower(1)) + v2 * c.lower(2);
        g(0) = reg_norm * c.radius - dot_prod;
        return g;
      }
      case CDDP_HIP_CON_THRUST:    // constraint.hpp:840-851: g = [min - |u|, |u| - max]
      case CDDP_HIP_CON_MAX_THRUST: {   // :955-964: g = |u| - max
        double sq = 0; for (int i = 0; i < c.dim; ++i) sq += u(i) * u(i);
        const double u_norm = std::sqrt(sq);
        if (c.kind == CDDP_HIP_CON_THRUST) { g(0) = c.lower(0) - u_norm; g(1) = u_norm - c.radius; }
        else g(0) = u_norm - c.radius;
        return g;
      }
    }
    return g;
  }
  void con_jac(const ConstraintDesc &c, const Vec &x, const Vec &u, Mat &gx, Mat &gu) const {
    gx = Mat(c.dual_dim, nx); gu = Mat(c.dual_dim, nu);
    switch (c.kind) {
      case CDDP_HIP_CON_CONTROL_BOX:  // constraint.hpp:203-219
        for (int i = 0; i < c.dim; ++i) { gu(i, i) = -c.scale; gu(c.dim + i, i) = c.scale; }
        break;
      case CDDP_HIP_CON_STATE_BOX:    // constraint.hpp:183-201
        for (int i = 0; i < c.dim; ++i) { gx(i, i) = -c.scale; gx(c.dim + i, i) = c.scale; }
        break;
      case CDDP_HIP_CON_BALL:         // constraint.hpp:361-374
        for (int i = 0; i < c.dim; ++i) gx(0, i) = -2.0 * c.scale * (x(i) - c.center(i));
        break;
      case CDDP_HIP_CON_LINEAR: gx = c.A; break;
      case CDDP_HIP_CON_SOC: {     // constraint.hpp:710-741
        const double v[3] = {x(0) - c.center(0), x(1) - c.center(1), x(2) - c.center(2)};
        const double reg_norm = std::sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + c.scale);
        for (int i = 0; i < 3; ++i) gx(0, i) = (reg_norm > 1e-9) ? c.radius * (v[i] / reg_norm) - c.lower(i) : -c.lower(i);
        break;
      }
      case CDDP_HIP_CON_THRUST: {  // constraint.hpp:861-880: rows -/+ u^T / sqrt(|u|^2 + eps), zero when that norm is below eps
        double sq = 0; for (int i = 0; i < c.dim; ++i) sq += u(i) * u(i);
        const double u_reg_norm = std::sqrt(sq + c.scale);
        if (!(u_reg_norm < c.scale)) for (int i = 0; i < c.dim; ++i) { gu(0, i) = -(u(i) / u_reg_norm); gu(1, i) = u(i) / u_reg_norm; }
        break;
      }
      case CDDP_HIP_CON_MAX_THRUST: {   // :979-993
        double sq = 0; for (int i = 0; i < c.dim; ++i) sq += u(i) * u(i);
        const double u_reg_norm = std::sqrt(sq + c.scale);
        if (u_reg_norm > std::numeric_limits<double>::min()) for (int i = 0; i < c.dim; ++i) gu(0, i) = u(i) / u_reg_norm;
        break;
      }
    }
  }
  const ConstraintDesc *clddp_control_constraint() const {  // clddp_solver.cpp:85-86
    for (auto &c : cons) if (c.name == "ControlConstraint" && c.kind == CDDP_HIP_CON_CONTROL_BOX) return &c;
    return nullptr;
  }
  bool has_term_ineq() const { for (auto &t : terms) if (t.kind == CDDP_HIP_TERM_INEQUALITY) return true; return false; }
  int term_eq_dim() const { int d = 0; for (auto &t : terms) if (t.kind == CDDP_HIP_TERM_EQUALITY) d += t.dim; return d; }
  Vec term_eq_residual(const Vec &xN) const {  // ipddp_solver.cpp:155-176
    Vec r(term_eq_dim(), 1); int off = 0;
    for (auto &t : terms) if (t.kind == CDDP_HIP_TERM_EQUALITY) { r.setSegment(off, xN - t.target); off += t.dim; }
    return r;
  }
  Mat term_eq_jacobian() const {  // ipddp_solver.cpp:178-201
    Mat J(term_eq_dim(), nx); int off = 0;
    for (auto &t : terms) if (t.kind == CDDP_HIP_TERM_EQUALITY) { for (int i = 0; i < t.dim; ++i) J(off + i, i) = 1.0; off += t.dim; }
    return J;
  }
  Vec term_ineq_eval(const TerminalDesc &t, const Vec &xN) const { return t.A * xN - t.b; }

  // =============================================================== regularisation (cddp_core.cpp:308-346)
  void increaseRegularization() { reg *= opt.reg_update_factor; reg = std::min(reg, opt.reg_max_value); }
  void decreaseRegularization() { reg /= opt.reg_update_factor; reg = std::max(reg, opt.reg_min_value); }
  bool isRegularizationLimitReached() const { return reg >= opt.reg_max_value; }

  // =============================================================== CDDP::initializeProblemIfNecessary
  void set_initial(const double *x0p, const double *U0p, const double *X0p) {
    x0 = Vec::FromPtr(x0p, nx);
    X.assign(N + 1, Vec::Zero(nx));
    U.assign(N, Vec::Zero(nu));
    if (X0p) for (int t = 0; t <= N; ++t) X[t] = Vec::FromPtr(X0p + (size_t)t * nx, nx);
    else for (int t = 0; t <= N; ++t) X[t] = x0;  // cddp::example::makeInitialTrajectory
    if (U0p) for (int t = 0; t < N; ++t) U[t] = Vec::FromPtr(U0p + (size_t)t * nu, nu);
    X[0] = x0;  // cddp_core.cpp:294
    const double inf = std::numeric_limits<double>::infinity();
    cost = inf; merit = inf; inf_pr = inf; inf_du = inf; inf_comp = inf;  // :297-301
    reg = opt.reg_initial_value;
    alpha_pr = opt.ls_initial_step_size; alpha_du = 0.0; step_norm = 0.0;
    n_backward = n_forward = iterations = 0; status = CDDP_HIP_STATUS_RUNNING;
    history.rows.clear();
  }

  void computeCost() {  // cddp_solver_base.cpp:416-424
    cost = 0.0;
    for (int t = 0; t < N; ++t) cost += running_cost(X[t], U[t], t);
    cost += terminal_cost(X.back());
    merit = cost;
  }
  void initializeGains() {  // cddp_solver_base.cpp:396-405
    k_u.assign(N, Vec::Zero(nu)); K_u.assign(N, Mat::Zero(nu, nx)); dV[0] = dV[1] = 0;
  }

  // =============================================================== CLDDP
  bool have_valid_gains() const {  // clddp_solver.cpp:36-49 == ipddp_solver.cpp:655-673
    if ((int)k_u.size() != N || (int)K_u.size() != N || k_u.empty()) return false;
    for (int t = 0; t < N; ++t)
      if (k_u[t].size() != nu || K_u[t].r != nu || K_u[t].c != nx) return false;
    return true;
  }
  void clddp_initialize() {  // clddp_solver.cpp:28-75
    if (opt.warm_start && have_valid_gains()) {   // :51-60: keep the gains (BoxQP warm start x0 = k_u_), cost of the given X, U
      if (Vx_t.size() != (size_t)(N + 1)) { Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx)); }
      computeCost();
      return;
    }
    initializeGains();
    Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx));
    computeCost();
  }

  bool clddp_backward() {  // clddp_solver.cpp:79-204
    ++n_backward;
    const ConstraintDesc *cc = clddp_control_constraint();
    Vec V_x = final_grad(X.back());
    Mat V_xx = final_hess();
    Vx_t[N] = V_x; Vxx_t[N] = V_xx;
    dV[0] = dV[1] = 0;
    double norm_Vx = V_x.lpNorm1();
    double Qu_error = 0.0;
    for (int t = N - 1; t >= 0; --t) {
      const Vec &x = X[t]; const Vec &u = U[t];
      Mat Fx, Fu; model.jacobians(x, u, t * dt, Fx, Fu);
      Mat A = dt * Fx; for (int i = 0; i < nx; ++i) A(i, i) += 1.0;
      Mat B = dt * Fu;
      Vec lx = l_x(x, t), lu = l_u(u);
      Mat lxx = l_xx(), luu = l_uu(), lux = l_ux();
      Vec Q_x = lx + A.T() * V_x;
      Vec Q_u = lu + B.T() * V_x;
      Mat Q_xx = lxx + A.T() * V_xx * A;
      Mat Q_ux = lux + B.T() * V_xx * A;
      Mat Q_uu = luu + B.T() * V_xx * B;
      Mat Q_uu_reg = Q_uu; for (int i = 0; i < nu; ++i) Q_uu_reg(i, i) += reg;
      if (minRealEigenvalue(Q_uu_reg) <= 0) return false;  // :133-140
      Vec k(nu, 1); Mat K(nu, nx);
      if (cc == nullptr) {
        const Mat H = inversePartialPivLU(Q_uu_reg);
        k = (-H) * Q_u;
        K = (-H) * Q_ux;
      } else {
        const Vec lb = cc->lower - u, ub = cc->upper - u;
        const Vec xq0 = k_u[t];
        BoxQPResult qp = boxqp_solve(opt, Q_uu_reg, Q_u, lb, ub, xq0);
        if (qp.status == HESSIAN_NOT_PD || qp.status == NO_DESCENT) return false;
        k = qp.x;
        K = Mat::Zero(nu, nx);
        std::vector<int> free_idx;
        for (int i = 0; i < nu; ++i) if (qp.free_[i]) free_idx.push_back(i);
        if (!free_idx.empty()) {
          Mat Q_ux_free((int)free_idx.size(), nx);
          for (size_t i = 0; i < free_idx.size(); ++i) for (int j = 0; j < nx; ++j) Q_ux_free((int)i, j) = Q_ux(free_idx[i], j);
          Mat K_free = -qp.Hfree.solve(Q_ux_free);
          for (size_t i = 0; i < free_idx.size(); ++i) for (int j = 0; j < nx; ++j) K(free_idx[i], j) = K_free((int)i, j);
        }
      }
      k_u[t] = k; K_u[t] = K;
      dV[0] += Q_u.dot(k);
      dV[1] += 0.5 * k.dot(Q_uu * k);
      V_x = Q_x + K.T() * Q_uu * k + Q_ux.T() * k + K.T() * Q_u;
      V_xx = Q_xx + K.T() * Q_uu * K + Q_ux.T() * K + K.T() * Q_ux;
      V_xx = 0.5 * (V_xx + V_xx.T());
      Vx_t[t] = V_x; Vxx_t[t] = V_xx;
      norm_Vx += V_x.lpNorm1();
      Qu_error = std::max(Qu_error, Q_u.lpNormInf());
    }
    double scaling = opt.termination_scaling_max_factor;
    scaling = std::max(scaling, norm_Vx / (N * nx)) / scaling;
    inf_du = Qu_error / scaling;
    return true;
  }

  FPResult clddp_forward(double a) {  // clddp_solver.cpp:215-262
    FPResult r; r.alpha = a; r.alpha_pr = a; r.success = false;
    r.cost = r.merit = std::numeric_limits<double>::infinity();
    r.X = X; r.U = U; r.X[0] = x0;
    double J_new = 0.0;
    const ConstraintDesc *cc = clddp_control_constraint();
    for (int t = 0; t < N; ++t) {
      const Vec x = r.X[t];
      const Vec delta_x = x - X[t];
      r.U[t] = r.U[t] + a * k_u[t] + K_u[t] * delta_x;
      if (cc) for (int i = 0; i < nu; ++i) r.U[t](i) = std::min(std::max(r.U[t](i), cc->lower(i)), cc->upper(i));
      J_new += running_cost(x, r.U[t], t);
      r.X[t + 1] = model.step(x, r.U[t], t * dt);
    }
    J_new += terminal_cost(r.X.back());
    double dJ = cost - J_new;
    double expected = -a * (dV[0] + 0.5 * a * dV[1]);
    double ratio = expected > 0.0 ? dJ / expected : std::copysign(1.0, dJ);
    r.success = ratio > opt.filter_armijo_constant;
    r.cost = J_new; r.merit = J_new;
    return r;
  }

  // =============================================================== IPDDP
  Vec seg(const Vec &v, const ConstraintDesc &c) const { return v.segment(c.offset, c.dual_dim); }

  void evaluate_G(const std::vector<Vec> &Xs, const std::vector<Vec> &Us, std::vector<Vec> &Gout) const {
    Gout.assign(N, Vec::Zero(m));
    for (int t = 0; t < N; ++t) for (auto &c : cons) Gout[t].setSegment(c.offset, con_g(c, Xs[t], Us[t]));
  }

  double computeTheta(const std::vector<Vec> &Gs, const std::vector<Vec> &Ss, const std::map<std::string, Vec> *GT,
                      const std::map<std::string, Vec> *ST, const Vec *hT) const {  // ipddp_solver.cpp:2778-2848
    const bool use_l2 = opt.ipddp_theta_norm_l2 != 0;
    double total = 0.0, max_entry = 0.0;
    for (auto &c : cons)
      for (int t = 0; t < (int)Gs.size(); ++t) {
        Vec residual = seg(Gs[t], c) + seg(Ss[t], c);
        total += use_l2 ? residual.squaredNorm() : residual.lpNorm1();
        max_entry = std::max(max_entry, residual.lpNormInf());
      }
    if (GT && ST)
      for (auto &kv : *GT) {
        auto it = ST->find(kv.first); if (it == ST->end()) continue;
        Vec residual = kv.second + it->second;
        total += use_l2 ? residual.squaredNorm() : residual.lpNorm1();
        max_entry = std::max(max_entry, residual.lpNormInf());
      }
    if (hT && hT->size() > 0) {
      total += use_l2 ? hT->squaredNorm() : hT->lpNorm1();
      max_entry = std::max(max_entry, hT->lpNormInf());
    }
    const double th = use_l2 ? std::sqrt(total) : total;
    return std::max(th, max_entry);
  }

  double computeBarrierMerit(const std::vector<Vec> &Ss, double c0, const std::map<std::string, Vec> *ST,
                             const Vec *lamT, const Vec *hT) const {  // ipddp_solver.cpp:2850-2880
    double mer = c0;
    for (auto &c : cons)
      for (int t = 0; t < (int)Ss.size(); ++t) {
        double s = 0; for (int i = 0; i < c.dual_dim; ++i) s += olog(std::max(Ss[t](c.offset + i), EPS_SLACK));
        mer -= mu * s;
      }
    if (ST) for (auto &kv : *ST) { double s = 0; for (int i = 0; i < kv.second.size(); ++i) s += olog(std::max(kv.second(i), EPS_SLACK)); mer -= mu * s; }
    if (lamT && hT && lamT->size() == hT->size() && hT->size() > 0) mer += lamT->dot(*hT);
    return mer;
  }

  std::pair<double, double> computePrimalAndComplementarity(const std::vector<Vec> &Gs, const std::vector<Vec> &Ss,
      const std::vector<Vec> &Ys, double mu_, const std::map<std::string, Vec> *GT, const std::map<std::string, Vec> *ST,
      const std::map<std::string, Vec> *YT, const Vec *hT) const {  // ipddp_solver.cpp:2882-2937
    double ipr = 0, icomp = 0;
    for (auto &c : cons)
      for (int t = 0; t < (int)Gs.size(); ++t)
        for (int i = 0; i < c.dual_dim; ++i) {
          int j = c.offset + i;
          ipr = std::max(ipr, std::fabs(Gs[t](j) + Ss[t](j)));
          icomp = std::max(icomp, std::fabs(Ys[t](j) * Ss[t](j) - mu_));
        }
    if (GT && ST && YT)
      for (auto &kv : *GT) {
        auto si = ST->find(kv.first); auto yi = YT->find(kv.first);
        if (si == ST->end() || yi == YT->end()) continue;
        for (int i = 0; i < kv.second.size(); ++i) {
          ipr = std::max(ipr, std::fabs(kv.second(i) + si->second(i)));
          icomp = std::max(icomp, std::fabs(yi->second(i) * si->second(i) - mu_));
        }
      }
    if (hT && hT->size() > 0) ipr = std::max(ipr, hT->lpNormInf());
    return {ipr, icomp};
  }

  bool acceptFilterEntry(double mf, double cv) {  // interior_point_utils.cpp:79-95
    FilterPoint cand{mf, cv};
    for (auto &p : filter) if (p.dominates(cand)) return false;
    std::vector<FilterPoint> keep;
    for (auto &p : filter) if (!cand.dominates(p)) keep.push_back(p);
    filter = keep; filter.push_back(cand);
    return true;
  }
  void pruneFilterToBestPoints() {  // interior_point_utils.cpp:114-139
    if (filter.empty()) return;
    FilterPoint bv = filter[0], bm = filter[0];
    for (auto &p : filter) { if (p.constraint_violation < bv.constraint_violation) bv = p; if (p.merit_function < bm.merit_function) bm = p; }
    filter.clear(); filter.push_back(bv);
    if (std::fabs(bm.constraint_violation - bv.constraint_violation) > 1e-12 ||
        std::fabs(bm.merit_function - bv.merit_function) > 1e-12) filter.push_back(bm);
  }

  void repairWarmstartInterior(Vec &s, Vec &y) const {  // ipddp_solver.cpp:233-262
    if (!opt.ipddp_warmstart_repair) return;
    if (s.size() > 0) {
      for (int i = 0; i < s.size(); ++i) s(i) = std::max(s(i), opt.ipddp_warmstart_s_min);
      if (s.minCoeff() < opt.ipddp_warmstart_s_min * opt.ipddp_warmstart_interior_factor) s = s * opt.ipddp_warmstart_interior_factor;
    }
    if (y.size() > 0) {
      for (int i = 0; i < y.size(); ++i) y(i) = std::max(y(i), opt.ipddp_warmstart_y_min);
      if (y.minCoeff() < opt.ipddp_warmstart_y_min * opt.ipddp_warmstart_interior_factor) y = y * opt.ipddp_warmstart_interior_factor;
    }
  }

  void resetFilter() {  // ipddp_solver.cpp:2484-2519
    const bool hti = has_term_ineq(), hte = term_eq_dim() > 0;
    Vec hT = hte ? term_eq_residual(X.back()) : Vec::Zero(0);
    auto pc = computePrimalAndComplementarity(G, S, Y, mu, hti ? &G_T : nullptr, hti ? &S_T : nullptr, hti ? &Y_T : nullptr, hte ? &hT : nullptr);
    merit = computeBarrierMerit(S, cost, hti ? &S_T : nullptr, hte ? &Lambda_T_eq : nullptr, hte ? &hT : nullptr);
    inf_pr = pc.first; inf_comp = pc.second;
    phi = merit;
    filter_theta = std::max(computeTheta(G, S, hti ? &G_T : nullptr, hti ? &S_T : nullptr, hte ? &hT : nullptr), 1e-8);
    theta = std::max(filter_theta, std::max(opt.ipddp_theta_0_floor, 1e-8));
    filter.clear();
    if (hti || hte) acceptFilterEntry(phi, filter_theta);
  }

  // ---- warm start (ipddp_solver.cpp:264-366, 653-816, 2296-2426)
  bool warmstartNeedsReinit(const Vec &y, const Vec &sv, const Vec &g) const {  // :264-292
    if (y.size() != g.size() || sv.size() != g.size()) return true;
    for (int i = 0; i < g.size(); ++i) if (!std::isfinite(y(i)) || !std::isfinite(sv(i))) return true;
    for (int i = 0; i < g.size(); ++i) {
      if (y(i) <= EPS_DUAL || sv(i) <= EPS_SLACK) return true;
      const double required = std::max(opt.ipddp_slack_var_init_scale, -g(i) + kSlackInteriorOffset);
      if (sv(i) < 0.1 * required) return true;
    }
    return false;
  }
  void evaluateTrajectoryWarmStart() {  // :2296-2343 (X is NOT re-propagated here)
    double c = 0.0;
    G.assign(N, Vec::Zero(m));
    for (int t = 0; t < N; ++t) {
      c += running_cost(X[t], U[t], t);
      for (auto &cd : cons) G[t].setSegment(cd.offset, con_g(cd, X[t], U[t]));
    }
    c += terminal_cost(X.back());
    for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) G_T[td.name] = term_ineq_eval(td, X.back());
    cost = c;
  }
  double computeMaxConstraintViolation() const {  // :2710-2723, interior_point_utils.cpp:141-155
    double mv = 0.0;
    for (auto &cd : cons) for (int t = 0; t < (int)G.size(); ++t) { Vec g = seg(G[t], cd); if (g.size() > 0) mv = std::max(mv, g.maxCoeff()); }
    for (auto &kv : G_T) if (kv.second.size() > 0) mv = std::max(mv, kv.second.maxCoeff());
    return mv;
  }
  void initializeDualSlackVariablesWarmStart(bool has_existing) {  // :2345-2426
    if (!has_existing) { S.assign(N, Vec::Zero(m)); Y.assign(N, Vec::Zero(m)); }
    dS.assign(N, Vec::Zero(m)); dY.assign(N, Vec::Zero(m));
    k_s.assign(N, Vec::Zero(m)); k_y.assign(N, Vec::Zero(m)); K_s.assign(N, Mat::Zero(m, nx)); K_y.assign(N, Mat::Zero(m, nx));
    Gx.assign(N, Mat::Zero(m, nx)); Gu.assign(N, Mat::Zero(m, nu));
    for (auto &cd : cons)
      for (int t = 0; t < N; ++t) {
        Vec g_val = seg(G[t], cd);
        Vec s_cur = Vec::Zero(cd.dual_dim), y_cur = Vec::Zero(cd.dual_dim);
        bool need = !has_existing;
        if (!need) { s_cur = seg(S[t], cd); y_cur = seg(Y[t], cd); need = warmstartNeedsReinit(y_cur, s_cur, g_val); }
        if (need)
          for (int i = 0; i < cd.dual_dim; ++i) {
            s_cur(i) = std::max(opt.ipddp_slack_var_init_scale, -g_val(i) + kSlackInteriorOffset);
            y_cur(i) = (mu * opt.ipddp_dual_var_init_scale) / std::max(s_cur(i), EPS_SLACK);
          }
        repairWarmstartInterior(s_cur, y_cur);
        Y[t].setSegment(cd.offset, y_cur); S[t].setSegment(cd.offset, s_cur);
      }
  }
  void initializeTerminalWarmstartDualSlack() {  // :294-353
    bool has_existing = true;
    for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
      auto yi = Y_T.find(td.name); auto si = S_T.find(td.name);
      if (yi == Y_T.end() || si == S_T.end() || yi->second.size() != td.dim || si->second.size() != td.dim) { has_existing = false; break; }
    }
    for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
      const Vec &g_val = G_T.at(td.name);
      Vec s_cur = Vec::Zero(td.dim), y_cur = Vec::Zero(td.dim);
      bool need = !has_existing;
      if (!need) { s_cur = S_T.at(td.name); y_cur = Y_T.at(td.name); need = warmstartNeedsReinit(y_cur, s_cur, g_val); }
      if (need)
        for (int i = 0; i < td.dim; ++i) {
          s_cur(i) = std::max(opt.ipddp_slack_var_init_scale, -g_val(i) + kSlackInteriorOffset);
          y_cur(i) = (mu * opt.ipddp_dual_var_init_scale) / std::max(s_cur(i), EPS_SLACK);
        }
      repairWarmstartInterior(s_cur, y_cur);
      S_T[td.name] = s_cur; Y_T[td.name] = y_cur; dS_T[td.name] = Vec::Zero(td.dim); dY_T[td.name] = Vec::Zero(td.dim);
    }
  }
  bool path_duals_exist() const {   // has_existing_dual_slack of :2351-2364 in the stacked representation
    if (cons.empty()) return true;
    if ((int)S.size() != N || (int)Y.size() != N) return false;
    for (int t = 0; t < N; ++t) if (S[t].size() != m || Y[t].size() != m) return false;
    return true;
  }
  void ipddp_initialize_warm() {
    const bool existing = have_valid_gains();
    const int pT = term_eq_dim();
    bool lam_ok = Lambda_T_eq.size() == pT;
    for (int i = 0; lam_ok && i < Lambda_T_eq.size(); ++i) lam_ok = std::isfinite(Lambda_T_eq(i));
    if (!lam_ok) Lambda_T_eq = Vec::Zero(pT);        // initializeTerminalEqualityWarmstartMultipliers :355-366
    dLambda_T_eq = Vec::Zero(pT);
    dV[0] = dV[1] = 0;
    const bool has_path = path_duals_exist();
    G_T.clear(); dS_T.clear(); dY_T.clear();        // initializeConstraintStorage (:2662-2708); S_, Y_, S_T_, Y_T_ are restored (:719-727, 756-764)
    if (existing) {   // ---- :675-731 existing solver state
      mu = opt.barrier_mu_initial * 0.1;
      step_norm = 0.0;
      X.assign(N + 1, Vec::Zero(nx)); X[0] = x0;
      for (int t = 0; t < N; ++t) X[t + 1] = model.step(X[t], U[t], t * dt);
      if ((int)Lambda.size() != N + 1) Lambda.assign(N + 1, Vec::Zero(nx));
      if ((int)Vx_t.size() != N + 1) { Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx)); }
      if ((int)dX.size() != N + 1) { dX.assign(N + 1, Vec::Zero(nx)); dU.assign(N, Vec::Zero(nu)); }
      evaluateTrajectoryWarmStart();
      initializeDualSlackVariablesWarmStart(has_path);
      initializeTerminalWarmstartDualSlack();
      resetFilter();
      return;
    }
    // ---- :733-816 provided trajectory, no solver state
    k_u.assign(N, Vec::Zero(nu)); K_u.assign(N, Mat::Zero(nu, nx));
    dX.assign(N + 1, Vec::Zero(nx)); dU.assign(N, Vec::Zero(nu));
    Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx));
    Lambda.assign(N + 1, Vec::Zero(nx));
    if ((int)U.size() != N) U.assign(N, Vec::Zero(nu));
    X.assign(N + 1, Vec::Zero(nx)); X[0] = x0;
    for (int t = 0; t < N; ++t) X[t + 1] = model.step(X[t], U[t], t * dt);
    if (cons.empty() && terms.empty()) {
      mu = std::max(opt.tolerance / 10.0, opt.barrier_mu_min_value);
      G.assign(N, Vec::Zero(m));
    } else {
      evaluateTrajectoryWarmStart();
      const double mv = computeMaxConstraintViolation();
      if (mv <= opt.tolerance) mu = std::max(opt.tolerance, opt.barrier_mu_min_value);
      else if (mv <= 0.1) mu = std::max(opt.tolerance * 10.0, opt.barrier_mu_initial * 0.01);
      else mu = opt.barrier_mu_initial * 0.1;
    }
    reg = opt.reg_initial_value; step_norm = 0.0; alpha_pr = 1.0; alpha_du = 1.0;
    initializeDualSlackVariablesWarmStart(has_path && (int)S.size() == N);
    initializeTerminalWarmstartDualSlack();
    resetFilter();
    inf_du = 0.0;
  }

  void ipddp_initialize() {  // ipddp_solver.cpp:644-914, cold-start path :819-913
    if (opt.warm_start) { ipddp_initialize_warm(); return; }
    k_u.assign(N, Vec::Zero(nu)); K_u.assign(N, Mat::Zero(nu, nx));
    dX.assign(N + 1, Vec::Zero(nx)); dU.assign(N, Vec::Zero(nu));
    Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx));
    Lambda.assign(N + 1, Vec::Zero(nx));
    dV[0] = dV[1] = 0;
    Lambda_T_eq = Vec::Zero(term_eq_dim()); dLambda_T_eq = Vec::Zero(term_eq_dim());
    S_T.clear(); Y_T.clear(); G_T.clear(); dS_T.clear(); dY_T.clear();
    // re-rollout (:868-874)
    X.assign(N + 1, Vec::Zero(nx)); X[0] = x0;
    for (int t = 0; t < N; ++t) X[t + 1] = model.step(X[t], U[t], t * dt);
    mu = (cons.empty() && terms.empty()) ? std::max(opt.tolerance / 10.0, opt.barrier_mu_min_value) : opt.barrier_mu_initial;
    reg = opt.reg_initial_value; step_norm = 0.0; alpha_pr = 1.0; alpha_du = 1.0;
    // evaluateTrajectory (:2252-2296): cost + g, re-propagating X
    {
      double c = 0.0; X[0] = x0;
      G.assign(N, Vec::Zero(m));
      for (int t = 0; t < N; ++t) {
        c += running_cost(X[t], U[t], t);
        for (auto &cd : cons) G[t].setSegment(cd.offset, con_g(cd, X[t], U[t]));
        X[t + 1] = model.step(X[t], U[t], t * dt);
      }
      c += terminal_cost(X.back());
      for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) G_T[td.name] = term_ineq_eval(td, X.back());
      cost = c;
    }
    // initializeDualSlackVariables (:2428-2482)
    S.assign(N, Vec::Zero(m)); Y.assign(N, Vec::Zero(m)); dS.assign(N, Vec::Zero(m)); dY.assign(N, Vec::Zero(m));
    k_s.assign(N, Vec::Zero(m)); k_y.assign(N, Vec::Zero(m)); K_s.assign(N, Mat::Zero(m, nx)); K_y.assign(N, Mat::Zero(m, nx));
    Gx.assign(N, Mat::Zero(m, nx)); Gu.assign(N, Mat::Zero(m, nu));
    for (auto &cd : cons)
      for (int t = 0; t < N; ++t) {
        Vec g_val = con_g(cd, X[t], U[t]);
        G[t].setSegment(cd.offset, g_val);
        Vec s_init(cd.dual_dim, 1), y_init(cd.dual_dim, 1);
        for (int i = 0; i < cd.dual_dim; ++i) {
          s_init(i) = std::max(opt.ipddp_slack_var_init_scale, -g_val(i) + kSlackInteriorOffset);
          y_init(i) = (mu * opt.ipddp_dual_var_init_scale) / std::max(s_init(i), EPS_SLACK);
        }
        repairWarmstartInterior(s_init, y_init);
        Y[t].setSegment(cd.offset, y_init); S[t].setSegment(cd.offset, s_init);
      }
    cost = objective_evaluate(X, U);  // :2481
    for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {  // :889-908
      const Vec &gT = G_T[td.name];
      Vec s_init(td.dim, 1), y_init(td.dim, 1);
      for (int i = 0; i < td.dim; ++i) {
        s_init(i) = std::max(opt.ipddp_slack_var_init_scale, -gT(i) + kSlackInteriorOffset);
        y_init(i) = (mu * opt.ipddp_dual_var_init_scale) / std::max(s_init(i), EPS_SLACK);
      }
      repairWarmstartInterior(s_init, y_init);
      S_T[td.name] = s_init; Y_T[td.name] = y_init; dS_T[td.name] = Vec::Zero(td.dim); dY_T[td.name] = Vec::Zero(td.dim);
    }
    resetFilter();
    inf_du = 0.0;
  }

  double computeScaledDualInfeasibility() const {  // ipddp_solver.cpp:2725-2776
    double v = inf_du;
    if (!opt.ipddp_check_state_stationarity) return v;
    double ss = 0.0;
    for (auto &c : cons)
      for (int t = 0; t < (int)std::min(Gx.size(), Y.size()); ++t) {
        Mat gx = Gx[t].block(c.offset, 0, c.dual_dim, nx);
        Vec st = gx.T() * seg(Y[t], c);
        ss = std::max(ss, st.lpNormInf());
      }
    return std::max(v, ss);
  }

  // rolloutLinearPolicy (ipddp_solver.cpp:368-392), d == 0
  void rolloutLinearPolicy(const std::vector<Mat> &A, const std::vector<Mat> &B, const std::vector<Mat> &K,
                           const std::vector<Vec> &k, const Vec &dx0, std::vector<Vec> &dXo, std::vector<Vec> &dUo) const {
    const int T = (int)K.size();
    dXo.assign(T + 1, Vec::Zero(nx)); dUo.assign(T, Vec::Zero(nu));
    dXo[0] = dx0;
    for (int t = 0; t < T; ++t) {
      dUo[t] = k[t] + K[t] * dXo[t];
      dXo[t + 1] = A[t] * dXo[t] + B[t] * dUo[t] + Vec::Zero(nx);
    }
  }

  // solveSequentialLQR (ipddp_solver.cpp:413-476), d == 0
  bool solveSequentialLQR(const std::vector<Mat> &Q, const std::vector<Vec> &q, const std::vector<Mat> &R,
                          const std::vector<Vec> &r, const std::vector<Mat> &M, const std::vector<Mat> &A,
                          const std::vector<Mat> &B, std::vector<Mat> &K, std::vector<Vec> &k,
                          std::vector<Mat> &P, std::vector<Vec> &p) const {
    const int T = (int)R.size();
    if (T == 0) return true;
    K.assign(T, Mat::Zero(nu, nx)); k.assign(T, Vec::Zero(nu));
    P.assign(T + 1, Mat::Zero(nx, nx)); p.assign(T + 1, Vec::Zero(nx));
    P[T] = 0.5 * (Q[T] + Q[T].T()); p[T] = q[T];
    for (int t = T - 1; t >= 0; --t) {
      const Mat &Pn = P[t + 1]; const Vec &pn = p[t + 1];
      const Mat BtP = B[t].T() * Pn;
      const Mat Q_uu = 0.5 * (R[t] + BtP * B[t] + R[t].T() + B[t].T() * Pn.T() * B[t]);
      const Mat Q_ux = BtP * A[t] + M[t].T();
      const Mat Q_xu = Q_ux.T().plain();
      const Vec drift = pn + Pn * Vec::Zero(nx);
      const Vec Q_x = q[t] + A[t].T() * drift;
      const Vec Q_u = r[t] + B[t].T() * drift;
      LDLT ldlt(Q_uu);
      if (!ldlt.ok) return false;
      K[t] = -ldlt.solve(Q_ux);
      k[t] = -ldlt.solve(Q_u);
      P[t] = Q[t] + A[t].T() * Pn * A[t] + Q_xu * K[t] + K[t].T() * Q_ux + K[t].T() * Q_uu * K[t];
      P[t] = 0.5 * (P[t] + P[t].T());
      p[t] = Q_x + Q_xu * k[t] + K[t].T() * Q_u + K[t].T() * Q_uu * k[t];
      if (!P[t].allFinite() || !p[t].allFinite() || !K[t].allFinite() || !k[t].allFinite()) return false;
    }
    return true;
  }

  // solveTerminalEqualityLQR (ipddp_solver.cpp:478-639)
  bool solveTerminalEqualityLQR(const std::vector<Mat> &Q, const std::vector<Vec> &q, const std::vector<Mat> &R,
                                const std::vector<Vec> &r, const std::vector<Mat> &M, const std::vector<Mat> &A,
                                const std::vector<Mat> &B, const Vec &dx0, const Mat &H_T, const Vec &b_T,
                                const Vec &lambda_prev, std::vector<Mat> &K_out, std::vector<Vec> &k_out,
                                std::vector<Mat> &P_out, std::vector<Vec> &p_out, Vec &lambda_total, Vec &lambda_delta) const {
    const int p_dim = H_T.r;
    if (p_dim == 0) { lambda_total = Vec::Zero(0); lambda_delta = Vec::Zero(0); return solveSequentialLQR(Q, q, R, r, M, A, B, K_out, k_out, P_out, p_out); }
    std::vector<Vec> q_base = q;
    Vec lambda_prev_vec = Vec::Zero(p_dim);
    if (lambda_prev.size() == p_dim) { lambda_prev_vec = lambda_prev; q_base.back() += H_T.T() * lambda_prev_vec; }
    const int T = (int)R.size();
    std::vector<std::vector<Mat>> Kv(p_dim + 1), Pv(p_dim + 1);
    std::vector<std::vector<Vec>> kv(p_dim + 1), pv(p_dim + 1);
    std::vector<Vec> xT(p_dim + 1, Vec::Zero(nx));
    for (int i = 0; i < p_dim + 1; ++i) {
      std::vector<Vec> qv = q_base;
      if (i > 0) qv.back() += H_T.row(i - 1).T();
      if (!solveSequentialLQR(Q, qv, R, r, M, A, B, Kv[i], kv[i], Pv[i], pv[i])) return false;
      std::vector<Vec> dXv, dUv;
      rolloutLinearPolicy(A, B, Kv[i], kv[i], dx0, dXv, dUv);
      xT[i] = dXv.back();
    }
    Mat S_mat(nx, p_dim);
    for (int i = 0; i < p_dim; ++i) { Vec d = xT[i + 1] - xT[0]; for (int r_ = 0; r_ < nx; ++r_) S_mat(r_, i) = d(r_); }
    const Mat A_small = H_T * S_mat;
    const Vec rhs = b_T - H_T * xT[0];
    const Mat AtA = A_small.T() * A_small;
    const Vec Atb = A_small.T() * rhs;
    const double trace_term = (AtA.trace() > 1.0 ? AtA.trace() / std::max(p_dim, 1) : 1.0);
    const double base_floor = std::max(1e-10, opt.ipddp_jacobian_regularization_value *
                                                  opow(std::max(mu, 0.0), opt.ipddp_jacobian_regularization_exponent));
    const double regv = std::max(base_floor, 1e-6 * trace_term);
    std::vector<double> sv = singularValues(A_small);
    double sigma_max = 0.0, sigma_min = 0.0;
    if (!sv.empty()) { sigma_max = *std::max_element(sv.begin(), sv.end()); sigma_min = *std::min_element(sv.begin(), sv.end()); }
    const double svd_reg = std::max(1e-8 * sigma_max - sigma_min, 0.0);
    const double reg_base = std::max(regv, svd_reg);
    const double lambda_norm_cap = 100.0 * (1.0 + rhs.norm());
    const double scales[5] = {1.0, 10.0, 100.0, 1e3, 1e4};
    Vec best_lambda = Vec::Zero(p_dim);
    double best_residual = std::numeric_limits<double>::infinity();
    bool found = false;
    for (double sc : scales) {
      const double reg_i = std::max(reg_base * sc, 1e-12);
      Mat shifted = AtA + reg_i * Mat::Identity(p_dim);
      LDLT ldlt(shifted);
      if (!ldlt.ok) continue;
      Vec lam = ldlt.solve(Atb);
      if (!lam.allFinite()) continue;
      const double ln = lam.norm();
      if (ln > lambda_norm_cap) lam = lam * (lambda_norm_cap / std::max(ln, 1e-12));
      const double residual = (A_small * lam - rhs).norm();
      if (!std::isfinite(residual)) continue;
      if (!found || residual < best_residual) { best_lambda = lam; best_residual = residual; found = true; }
    }
    if (!found) best_lambda = Vec::Zero(p_dim);
    K_out = Kv[0]; k_out = kv[0]; P_out = Pv[0]; p_out = pv[0];
    for (int i = 0; i < p_dim; ++i) {
      const double coeff = best_lambda(i);
      for (int t = 0; t < T; ++t) k_out[t] += coeff * (kv[i + 1][t] - kv[0][t]);
      for (int t = 0; t <= T; ++t) p_out[t] += coeff * (pv[i + 1][t] - pv[0][t]);
    }
    lambda_delta = best_lambda;
    lambda_total = lambda_prev_vec + best_lambda;
    return true;
  }

  void terminal_ineq_directions() {  // ipddp_solver.cpp:1315-1346 == :1534-1561
    for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
      const Vec g_T = term_ineq_eval(td, X.back());
      const Mat &Gtx = td.A;
      const Vec &ST = S_T.at(td.name); const Vec &YT = Y_T.at(td.name);
      const Vec r_p_T = g_T + ST;
      Vec r_d_T(td.dim, 1); for (int i = 0; i < td.dim; ++i) r_d_T(i) = ST(i) * YT(i) - mu;
      dS_T[td.name] = -r_p_T - Gtx * dX.back();
      Vec dYT = Vec::Zero(td.dim);
      for (int i = 0; i < td.dim; ++i) {
        const double s_safe = std::max(ST(i), std::max(mu * 1e-3, EPS_SLACK));
        const double dual_ratio = clampd(YT(i) / s_safe, 0.0, MAX_BARRIER_RATIO);
        const double affine = clampd(-r_d_T(i) / s_safe, -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO);
        dYT(i) = clampd(affine - dual_ratio * dS_T[td.name](i), -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO);
      }
      dY_T[td.name] = dYT;
    }
  }

  bool ipddp_backward() {  // ipddp_solver.cpp:960-1569
    ++n_backward;
    const bool hti = has_term_ineq();
    const bool hte = term_eq_dim() > 0;
    const bool hpc = !cons.empty();
    // precomputeDynamicsDerivatives (cddp_solver_base.cpp:319-394)
    F_x.assign(N, Mat()); F_u.assign(N, Mat());
    for (int t = 0; t < N; ++t) {
      Mat Fx, Fu; model.jacobians(X[t], U[t], t * dt, Fx, Fu);
      F_x[t] = dt * Fx; for (int i = 0; i < nx; ++i) F_x[t](i, i) += 1.0;
      F_u[t] = dt * Fu;
    }
    const bool ddp = !opt.use_ilqr;   // full DDP: second-order dynamics terms (cddp_solver_base.cpp:346-356)
    if (ddp) {
      F_xx.assign(N, {}); F_uu.assign(N, {}); F_ux.assign(N, {});
      for (int t = 0; t < N; ++t) {
        std::vector<Mat> Fxx, Fuu, Fux;
        if (!model.hessians(X[t], U[t], t * dt, Fxx, Fuu, Fux)) { std::fprintf(stderr, "oracle: use_ilqr=false needs a plant with restated Hessians\n"); std::abort(); }
        for (int i = 0; i < nx; ++i) { Fxx[i] = dt * Fxx[i]; Fuu[i] = dt * Fuu[i]; Fux[i] = dt * Fux[i]; }
        F_xx[t] = Fxx; F_uu[t] = Fuu; F_ux[t] = Fux;
      }
    }
    // precomputeConstraintGradients (ipddp_solver.cpp:2145-2250)
    if (hpc) for (int t = 0; t < N; ++t) for (auto &c : cons) { Mat gx, gu; con_jac(c, X[t], U[t], gx, gu); Gx[t].setBlock(c.offset, 0, gx); Gu[t].setBlock(c.offset, 0, gu); }

    Vec V_x = final_grad(X.back());
    Mat V_xx = symmetrize(final_hess());
    dV[0] = dV[1] = 0;
    double l_inf_du = 0.0, l_inf_pr = 0.0, l_inf_comp = 0.0, l_step_norm = 0.0;

    if (hti) {  // :1000-1031
      for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
        G_T[td.name] = term_ineq_eval(td, X.back());
        const Vec &g_T = G_T.at(td.name); const Mat &G_T_x = td.A;
        const Vec &ST = S_T.at(td.name); const Vec &YT = Y_T.at(td.name);
        Vec sigma_T(td.dim, 1), barrier_grad_T(td.dim, 1);
        for (int i = 0; i < td.dim; ++i) {
          const double s_safe = std::max(ST(i), std::max(mu * 1e-3, EPS_SLACK));
          const double y_safe = std::max(YT(i), EPS_DUAL);
          sigma_T(i) = clipPositiveBarrierRatio(y_safe, s_safe);
          const double corr = clipSignedBarrierRatio(y_safe * g_T(i) + mu, s_safe);
          barrier_grad_T(i) = y_safe + corr;
        }
        V_x += G_T_x.T() * barrier_grad_T;
        V_xx += G_T_x.T() * diagTimes(sigma_T, Mat::Identity(td.dim)) * G_T_x;
        V_xx = symmetrize(V_xx);
        l_inf_pr = std::max(l_inf_pr, (g_T + ST).lpNormInf());
        Vec cr(td.dim, 1); for (int i = 0; i < td.dim; ++i) cr(i) = YT(i) * ST(i) - mu;
        l_inf_comp = std::max(l_inf_comp, cr.lpNormInf());
      }
    }
    Vec h_T = Vec::Zero(term_eq_dim());
    Mat H_T = Mat::Zero(term_eq_dim(), nx);
    if (hte) {  // :1036-1042
      h_T = term_eq_residual(X.back()); H_T = term_eq_jacobian();
      l_inf_pr = std::max(l_inf_pr, h_T.lpNormInf());
      dLambda_T_eq = -h_T;
    } else dLambda_T_eq = Vec::Zero(0);

    // ---------------------------------------------------------------- unconstrained branch :1048-1118
    if (!hpc && !hti && !hte) {
      Vx_t[N] = V_x; Vxx_t[N] = V_xx;
      for (int t = N - 1; t >= 0; --t) {
        const Vec &x = X[t]; const Vec &u = U[t];
        const Mat &A = F_x[t]; const Mat &B = F_u[t];
        Vec lx = l_x(x, t), lu = l_u(u);
        Vec Q_x = lx + A.T() * V_x;
        Vec Q_u = lu + B.T() * V_x;
        Mat Q_xx = l_xx() + A.T() * V_xx * A;
        Mat Q_ux = l_ux() + B.T() * V_xx * A;
        Mat Q_uu = l_uu() + B.T() * V_xx * B;
        if (ddp)   // :1070-1082
          for (int i = 0; i < nx; ++i) { Q_xx += V_x(i) * F_xx[t][i]; Q_ux += V_x(i) * F_ux[t][i]; Q_uu += V_x(i) * F_uu[t][i]; }
        Q_uu = symmetrize(Q_uu);
        for (int i = 0; i < nu; ++i) Q_uu(i, i) += reg;
        LDLT ldlt(Q_uu);
        if (!ldlt.ok) return false;
        Vec kk = -ldlt.solve(Q_u);
        Mat KK = -ldlt.solve(Q_ux);
        k_u[t] = kk; K_u[t] = KK;
        V_x = Q_x + KK.T() * Q_u + Q_ux.T() * kk + KK.T() * Q_uu * kk;
        V_xx = Q_xx + KK.T() * Q_ux + Q_ux.T() * KK + KK.T() * Q_uu * KK;
        V_xx = symmetrize(V_xx);
        Vx_t[t] = V_x; Vxx_t[t] = V_xx;
        dV[0] += kk.dot(Q_u);
        dV[1] += 0.5 * kk.dot(Q_uu * kk);
        l_inf_du = std::max(l_inf_du, Q_u.lpNormInf());
        l_step_norm = std::max(l_step_norm, kk.lpNormInf());
      }
      inf_du = l_inf_du; step_norm = l_step_norm; inf_pr = 0.0; inf_comp = 0.0;
      return true;
    }

    // ---------------------------------------------------------------- terminal-equality branch :1120-1353
    if (hte) {
      std::vector<Mat> Q(N + 1, Mat::Zero(nx, nx)), R(N, Mat::Zero(nu, nu)), M(N, Mat::Zero(nx, nu)), A_vec(N), B_vec(N);
      std::vector<Vec> q(N + 1, Vec::Zero(nx)), r(N, Vec::Zero(nu));
      struct PathModel { Vec y, s, primal_residual, rhat; Mat Q_yx, Q_yu, YSinv; };
      std::vector<PathModel> pm(N);
      Q.back() = V_xx; q.back() = V_x;
      for (int t = 0; t < N; ++t) {
        const Vec &x = X[t]; const Vec &u = U[t];
        Q[t] = symmetrize(l_xx()); q[t] = l_x(x, t); R[t] = symmetrize(l_uu()); r[t] = l_u(u);
        M[t] = l_ux().T(); A_vec[t] = F_x[t]; B_vec[t] = F_u[t];
        if (ddp) {   // :1160-1178: the current costate iterate stands in for the value gradient
          const Vec lambda_next = ((int)Lambda.size() == N + 1 && Lambda[t + 1].size() == nx && Lambda[t + 1].allFinite()) ? Lambda[t + 1] : Vec::Zero(nx);
          for (int i = 0; i < nx; ++i) { Q[t] += lambda_next(i) * F_xx[t][i]; M[t] += lambda_next(i) * F_ux[t][i].T(); R[t] += lambda_next(i) * F_uu[t][i]; }
          Q[t] = symmetrize(Q[t]); R[t] = symmetrize(R[t]);
        }
        if (hpc) {
          const Vec &y = Y[t]; const Vec &s = S[t]; const Vec &g = G[t];
          const Mat &Q_yx = Gx[t]; const Mat &Q_yu = Gu[t];
          Mat YSinv = Mat::Zero(m, m);
          for (int i = 0; i < m; ++i) { const double s_safe = std::max(s(i), std::max(mu * 1e-3, EPS_SLACK)); YSinv(i, i) = clipPositiveBarrierRatio(y(i), s_safe); }
          const Vec primal_residual = g + s;
          Vec comp(m, 1); for (int i = 0; i < m; ++i) comp(i) = y(i) * s(i) - mu;
          const Vec rhat = cwiseProduct(y, primal_residual) - comp;
          Vec S_inv_rhat(m, 1);
          for (int i = 0; i < m; ++i) { const double s_safe = std::max(s(i), std::max(mu * 1e-3, EPS_SLACK)); S_inv_rhat(i) = clipSignedBarrierRatio(rhat(i), s_safe); }
          q[t] += Q_yx.T() * (y + S_inv_rhat);
          r[t] += Q_yu.T() * (y + S_inv_rhat);
          Q[t] += Q_yx.T() * YSinv * Q_yx;
          M[t] += (Q_yu.T() * YSinv * Q_yx).T();
          R[t] += Q_yu.T() * YSinv * Q_yu;
          Q[t] = symmetrize(Q[t]); R[t] = symmetrize(R[t]);
          pm[t].y = y; pm[t].s = s; pm[t].Q_yx = Q_yx; pm[t].Q_yu = Q_yu; pm[t].YSinv = YSinv;
          pm[t].primal_residual = primal_residual; pm[t].rhat = rhat;
          l_inf_pr = std::max(l_inf_pr, primal_residual.lpNormInf());
          l_inf_comp = std::max(l_inf_comp, comp.lpNormInf());
        }
        for (int i = 0; i < nu; ++i) R[t](i, i) += reg;
      }
      Vec lambda_total, lambda_delta;
      std::vector<Mat> Kl; std::vector<Vec> kl;
      if (!solveTerminalEqualityLQR(Q, q, R, r, M, A_vec, B_vec, Vec::Zero(nx), H_T, -h_T, Lambda_T_eq, K_u, k_u, Kl, kl, lambda_total, lambda_delta)) return false;
      Vxx_t = Kl; Vx_t = kl;
      dLambda_T_eq = lambda_delta;
      for (int t = 0; t < N; ++t) {
        const Vec Q_u = r[t] + B_vec[t].T() * Vx_t[t + 1];
        l_inf_du = std::max(l_inf_du, Q_u.lpNormInf());
        l_step_norm = std::max(l_step_norm, k_u[t].lpNormInf());
      }
      rolloutLinearPolicy(A_vec, B_vec, K_u, k_u, Vec::Zero(nx), dX, dU);
      if (hpc) {
        for (int t = 0; t < N; ++t) {
          const PathModel &md = pm[t];
          const Vec temp = md.Q_yu * k_u[t];
          Vec ky(m, 1);
          for (int i = 0; i < m; ++i) { const double s_safe = std::max(md.s(i), std::max(mu * 1e-3, EPS_SLACK)); ky(i) = clipSignedBarrierRatio(md.rhat(i) + md.y(i) * temp(i), s_safe); }
          Mat Ky = cwiseClamp(md.YSinv * (md.Q_yx + md.Q_yu * K_u[t]), -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO);
          const Vec ks = -md.primal_residual - temp;
          const Mat Ks = -md.Q_yx - md.Q_yu * K_u[t];
          k_y[t] = ky; K_y[t] = Ky; k_s[t] = ks; K_s[t] = Ks;
          dS[t] = ks + Ks * dX[t];
          dY[t] = cwiseClamp(ky + Ky * dX[t], -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO);
        }
      }
      if (hti) terminal_ineq_directions();
      inf_pr = l_inf_pr; inf_du = l_inf_du; inf_comp = l_inf_comp; step_norm = l_step_norm;
      return true;
    }

    // ---------------------------------------------------------------- path / terminal-ineq branch :1355-1568
    Vx_t[N] = V_x; Vxx_t[N] = V_xx;
    for (int t = N - 1; t >= 0; --t) {
      const Vec &x = X[t]; const Vec &u = U[t];
      const Mat &A = F_x[t]; const Mat &B = F_u[t];
      const Vec y = m ? Y[t] : Vec::Zero(0); const Vec s = m ? S[t] : Vec::Zero(0); const Vec g = m ? G[t] : Vec::Zero(0);
      const Mat Q_yx = m ? Gx[t] : Mat::Zero(0, nx); const Mat Q_yu = m ? Gu[t] : Mat::Zero(0, nu);
      Vec lx = l_x(x, t), lu = l_u(u);
      Vec Q_x = lx + Q_yx.T() * y + A.T() * V_x;
      Vec Q_u = lu + Q_yu.T() * y + B.T() * V_x;
      Mat Q_xx = l_xx() + A.T() * V_xx * A;
      Mat Q_ux = l_ux() + B.T() * V_xx * A;
      Mat Q_uu = l_uu() + B.T() * V_xx * B;
      if (ddp)   // :1396-1408
        for (int i = 0; i < nx; ++i) { Q_xx += V_x(i) * F_xx[t][i]; Q_ux += V_x(i) * F_ux[t][i]; Q_uu += V_x(i) * F_uu[t][i]; }
      Mat YSinv = Mat::Zero(m, m);
      for (int i = 0; i < m; ++i) { const double s_safe = std::max(s(i), std::max(mu * 1e-3, EPS_SLACK)); YSinv(i, i) = clipPositiveBarrierRatio(y(i), s_safe); }
      const Vec primal_residual = g + s;
      Vec comp(m, 1); for (int i = 0; i < m; ++i) comp(i) = y(i) * s(i) - mu;
      const Vec rhat = cwiseProduct(y, primal_residual) - comp;
      Mat Q_uu_reg = symmetrize(Q_uu);
      Q_uu_reg += Q_yu.T() * YSinv * Q_yu;
      for (int i = 0; i < nu; ++i) Q_uu_reg(i, i) += reg;
      LDLT ldlt(Q_uu_reg);
      if (!ldlt.ok) return false;
      Vec S_inv_rhat(m, 1);
      for (int i = 0; i < m; ++i) { const double s_safe = std::max(s(i), std::max(mu * 1e-3, EPS_SLACK)); S_inv_rhat(i) = clipSignedBarrierRatio(rhat(i), s_safe); }
      Mat bigRHS(nu, 1 + nx);
      {
        Vec c0 = Q_u + Q_yu.T() * S_inv_rhat;
        Mat rc = Q_ux + Q_yu.T() * YSinv * Q_yx;
        for (int i = 0; i < nu; ++i) { bigRHS(i, 0) = c0(i); for (int j = 0; j < nx; ++j) bigRHS(i, 1 + j) = rc(i, j); }
      }
      Mat kK = -ldlt.solve(bigRHS);
      Vec kk(nu, 1); Mat KK(nu, nx);
      for (int i = 0; i < nu; ++i) { kk(i) = kK(i, 0); for (int j = 0; j < nx; ++j) KK(i, j) = kK(i, 1 + j); }
      k_u[t] = kk; K_u[t] = KK;
      Vec ky(m, 1);
      const Vec temp = Q_yu * kk;
      for (int i = 0; i < m; ++i) { const double s_safe = std::max(s(i), std::max(mu * 1e-3, EPS_SLACK)); ky(i) = clipSignedBarrierRatio(rhat(i) + y(i) * temp(i), s_safe); }
      Mat Ky = cwiseClamp(YSinv * (Q_yx + Q_yu * KK), -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO);
      const Vec ks = -primal_residual - temp;
      const Mat Ks = -Q_yx - Q_yu * KK;
      if (m) { k_y[t] = ky; K_y[t] = Ky; k_s[t] = ks; K_s[t] = Ks; }
      Q_u += Q_yu.T() * S_inv_rhat;
      Q_x += Q_yx.T() * S_inv_rhat;
      Q_xx += Q_yx.T() * YSinv * Q_yx;
      Q_ux += Q_yu.T() * YSinv * Q_yx;
      Q_uu += Q_yu.T() * YSinv * Q_yu;
      dV[0] += kk.dot(Q_u);
      dV[1] += 0.5 * kk.dot(Q_uu * kk);
      V_x = Q_x + KK.T() * Q_u + Q_ux.T() * kk + KK.T() * Q_uu * kk;
      V_xx = Q_xx + KK.T() * Q_ux + Q_ux.T() * KK + KK.T() * Q_uu * KK;
      V_xx = symmetrize(V_xx);
      Vx_t[t] = V_x; Vxx_t[t] = V_xx;
      l_inf_du = std::max(l_inf_du, Q_u.lpNormInf());
      l_inf_pr = std::max(l_inf_pr, primal_residual.lpNormInf());
      l_inf_comp = std::max(l_inf_comp, comp.lpNormInf());
      l_step_norm = std::max(l_step_norm, kk.lpNormInf());
    }
    rolloutLinearPolicy(F_x, F_u, K_u, k_u, Vec::Zero(nx), dX, dU);  // :1511-1520
    if (m) for (int t = 0; t < N; ++t) {
      dS[t] = k_s[t] + K_s[t] * dX[t];
      dY[t] = cwiseClamp(k_y[t] + K_y[t] * dX[t], -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO);
    }
    if (hti) terminal_ineq_directions();
    inf_pr = l_inf_pr; inf_du = l_inf_du; inf_comp = l_inf_comp; step_norm = l_step_norm;
    return true;
  }

  std::pair<double, double> computeMaxStepSizes() const {  // ipddp_solver.cpp:2939-2988
    const double tau = std::max(opt.barrier_min_fraction_to_boundary, 1.0 - mu);
    double apr = 1.0, adu = 1.0;
    for (auto &c : cons)
      for (int t = 0; t < (int)dS.size(); ++t)
        for (int i = 0; i < c.dual_dim; ++i) {
          int j = c.offset + i;
          if (dS[t](j) < 0.0) apr = std::min(apr, -tau * S[t](j) / dS[t](j));
          if (dY[t](j) < 0.0) adu = std::min(adu, -tau * Y[t](j) / dY[t](j));
        }
    for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
      const Vec &s = S_T.at(td.name), &y = Y_T.at(td.name), &ds = dS_T.at(td.name), &dy = dY_T.at(td.name);
      for (int i = 0; i < td.dim; ++i) {
        if (ds(i) < 0.0) apr = std::min(apr, -tau * s(i) / ds(i));
        if (dy(i) < 0.0) adu = std::min(adu, -tau * y(i) / dy(i));
      }
    }
    return {clampd(apr, 0.0, 1.0), clampd(adu, 0.0, 1.0)};
  }

  FPResult ipddp_forward(double alpha) {  // ipddp_solver.cpp:1571-1876
    const bool hti = has_term_ineq();
    const bool hte = term_eq_dim() > 0;
    auto mx = computeMaxStepSizes();
    FPResult r; r.has_ip = true; r.alpha = alpha;
    r.success = false; r.cost = cost; r.merit = phi; r.theta = theta;
    const double tau = (cons.empty() && !hti) ? 1.0 : std::max(opt.barrier_min_fraction_to_boundary, 1.0 - mu);
    const double a_pr = std::min(alpha, mx.first), a_du = std::min(alpha, mx.second);
    r.alpha_pr = a_pr; r.alpha_du = a_du;
    r.X.assign(N + 1, Vec::Zero(nx)); r.U.assign(N, Vec::Zero(nu)); r.X[0] = x0;
    std::vector<Vec> dx_real(N + 1, Vec::Zero(nx));
    r.Lambda = Lambda; r.S = S; r.Y = Y;
    r.S_T = S_T; r.Y_T = Y_T; r.G_T = G_T; r.Lambda_T_eq = Lambda_T_eq;
    for (int t = 0; t < N; ++t) {
      dx_real[t] = r.X[t] - X[t];
      r.Lambda[t] = Lambda[t] + a_pr * Vx_t[t] + Vxx_t[t] * dx_real[t];
      if (!r.Lambda[t].allFinite()) return r;
      for (auto &c : cons) {
        Vec s_new = seg(S[t], c) + a_pr * seg(k_s[t], c) + K_s[t].block(c.offset, 0, c.dual_dim, nx) * dx_real[t];
        Vec s_min = (1.0 - tau) * seg(S[t], c);
        Vec y_new = seg(Y[t], c) + a_du * seg(k_y[t], c) + K_y[t].block(c.offset, 0, c.dual_dim, nx) * dx_real[t];
        Vec y_min = (1.0 - tau) * seg(Y[t], c);
        for (int i = 0; i < c.dual_dim; ++i) if (s_new(i) < s_min(i) || y_new(i) < y_min(i)) return r;
        if (!s_new.allFinite() || !y_new.allFinite()) return r;
        r.S[t].setSegment(c.offset, s_new); r.Y[t].setSegment(c.offset, y_new);
      }
      r.U[t] = U[t] + a_pr * k_u[t] + K_u[t] * dx_real[t];
      r.X[t + 1] = model.step(r.X[t], r.U[t], t * dt);
      if (!r.X[t + 1].allFinite() || !r.U[t].allFinite()) return r;
    }
    dx_real.back() = r.X.back() - X.back();
    r.Lambda.back() = Lambda.back() + a_pr * Vx_t.back() + Vxx_t.back() * dx_real.back();
    if (!r.Lambda.back().allFinite()) return r;
    if (hti) {  // :1667-1714
      for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
        const Vec g_T0 = term_ineq_eval(td, X.back());
        const Mat &G_T_x0 = td.A;
        const Vec &ST = S_T.at(td.name); const Vec &YT = Y_T.at(td.name);
        const Vec k_s_T = -(g_T0 + ST);
        const Mat K_s_T = -G_T_x0;
        r.S_T[td.name] = ST + a_pr * k_s_T + K_s_T * dx_real.back();
        Vec Y_trial = YT;
        for (int i = 0; i < td.dim; ++i) {
          const double s_safe = std::max(ST(i), std::max(mu * 1e-3, EPS_SLACK));
          const double r_d = YT(i) * ST(i) - mu;
          const double dual_ratio = clipPositiveBarrierRatio(YT(i), s_safe);
          Mat K_y_row = -(dual_ratio * K_s_T.row(i));
          const double kyv = clipSignedBarrierRatio(-r_d - YT(i) * k_s_T(i), s_safe);
          double dotv = 0; for (int j = 0; j < nx; ++j) dotv += K_y_row(0, j) * dx_real.back()(j);
          Y_trial(i) = YT(i) + a_du * kyv + dotv;
        }
        r.Y_T[td.name] = Y_trial;
        const double floor0 = std::max(mu * 1e-3, EPS_SLACK);
        for (int i = 0; i < td.dim; ++i) {
          const double s_floor = std::max((1.0 - tau) * ST(i), floor0);
          if (r.S_T[td.name](i) < s_floor || r.Y_T[td.name](i) < (1.0 - tau) * YT(i)) return r;
        }
        if (!r.S_T[td.name].allFinite() || !r.Y_T[td.name].allFinite()) return r;
      }
    }
    if (hte) { r.Lambda_T_eq = Lambda_T_eq + a_pr * dLambda_T_eq; if (!r.Lambda_T_eq.allFinite()) return r; }
    double cost_new = 0.0;
    r.G.assign(N, Vec::Zero(m));
    for (int t = 0; t < N; ++t) {
      cost_new += running_cost(r.X[t], r.U[t], t);
      for (auto &c : cons) r.G[t].setSegment(c.offset, con_g(c, r.X[t], r.U[t]));
    }
    cost_new += terminal_cost(r.X.back());
    Vec h_T_new = Vec::Zero(term_eq_dim());
    if (hti) for (auto &td : terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) r.G_T[td.name] = term_ineq_eval(td, r.X.back());
    if (hte) h_T_new = term_eq_residual(r.X.back());
    const double phi_new = computeBarrierMerit(r.S, cost_new, hti ? &r.S_T : nullptr, hte ? &r.Lambda_T_eq : nullptr, hte ? &h_T_new : nullptr);
    const double theta_new = computeTheta(r.G, r.S, hti ? &r.G_T : nullptr, hti ? &r.S_T : nullptr, hte ? &h_T_new : nullptr);
    auto pc = computePrimalAndComplementarity(r.G, r.S, r.Y, mu, hti ? &r.G_T : nullptr, hti ? &r.S_T : nullptr, hti ? &r.Y_T : nullptr, hte ? &h_T_new : nullptr);
    if (!std::isfinite(phi_new) || !std::isfinite(theta_new) || !std::isfinite(pc.first) || !std::isfinite(pc.second)) return r;
    bool accept = false;
    if (cons.empty() && !hti && !hte) {  // :1785-1792
      double dJ = cost - cost_new;
      double expected = -a_pr * (dV[0] + 0.5 * a_pr * dV[1]);
      double ratio = expected > 0.0 ? dJ / expected : std::copysign(1.0, dJ);
      accept = ratio > 1e-6;
    } else {  // :1793-1834
      double expected_improvement = a_pr * dV[0];
      double cv_old = filter.empty() ? 0.0 : filter.back().constraint_violation;
      const double high_ref = filter.empty() ? filter_theta : cv_old;
      double merit_old = merit;
      if (theta_new > opt.filter_max_violation_threshold) {
        if (theta_new < (1 - opt.filter_violation_acceptance_threshold) * high_ref) accept = true;
      } else if (std::max(theta_new, cv_old) < opt.filter_min_violation_for_armijo_check && expected_improvement < 0) {
        if (phi_new < merit_old + opt.filter_armijo_constant * expected_improvement) accept = true;
      } else {
        if (phi_new < merit_old - opt.filter_merit_acceptance_threshold * theta_new ||
            theta_new < (1 - opt.filter_violation_acceptance_threshold) * cv_old) accept = true;
      }
    }
    // diagnostics are reported even for rejected trials (the reference discards them)
    r.cost = cost_new; r.merit = phi_new; r.theta = theta_new; r.inf_pr = pc.first; r.inf_comp = pc.second;
    if (!accept) return r;
    r.success = true;
    return r;
  }

  void updateBarrierParameters(bool fp_success) {  // ipddp_solver.cpp:2548-2660
    const bool no_barrier = cons.empty() && !has_term_ineq();
    if (!fp_success) return;
    const double sdu = computeScaledDualInfeasibility();
    const double scomp = inf_comp;
    const double mu_old = mu;
    if (no_barrier) { mu = mu_old; }
    else if (opt.barrier_strategy == CDDP_HIP_BARRIER_ADAPTIVE) {
      const double kkt = std::max(std::max(inf_pr, sdu), scomp);
      const double threshold = std::max(opt.barrier_mu_update_factor * mu, 2.0 * mu);
      if (kkt <= threshold) {
        double factor = opt.barrier_mu_update_factor;
        if (mu > 1e-20) {
          const double ratio = kkt / std::max(mu, 1e-20);
          if (ratio < 0.01) factor = 0.1 * opt.barrier_mu_update_factor;
          else if (ratio < 0.1) factor = 0.3 * opt.barrier_mu_update_factor;
          else if (ratio < 0.5) factor = 0.6 * opt.barrier_mu_update_factor;
        }
        const double linear = factor * mu;
        const double superlinear = opow(mu, opt.barrier_mu_update_power);
        mu = std::max(std::min(linear, superlinear), std::max(opt.barrier_mu_min_value, opt.tolerance / 100.0));
      }
    } else {
      const double wdu = sdu * opt.ipddp_barrier_update_dual_weight;
      const double kkt = std::max(std::max(inf_pr, wdu), scomp);
      if (kkt <= opt.ipddp_mu_kappa_epsilon * mu) {
        const double linear = opt.barrier_mu_update_factor * mu;
        const double superlinear = opow(mu, opt.barrier_mu_update_power);
        mu = std::max(opt.barrier_mu_min_value, std::min(linear, superlinear));
      }
    }
    const bool hti = has_term_ineq(), hte = term_eq_dim() > 0;
    Vec hT = hte ? term_eq_residual(X.back()) : Vec::Zero(0);
    const double ftheta = std::max(computeTheta(G, S, hti ? &G_T : nullptr, hti ? &S_T : nullptr, hte ? &hT : nullptr), 1e-8);
    const bool reset = (mu < mu_old) && (mu > 0.0);
    if (reset) { filter.clear(); if (hte || hti) acceptFilterEntry(phi, ftheta); }
    else { acceptFilterEntry(phi, ftheta); if ((int)filter.size() > opt.ipddp_max_filter_size) pruneFilterToBestPoints(); }
    auto pc = computePrimalAndComplementarity(G, S, Y, mu, hti ? &G_T : nullptr, hti ? &S_T : nullptr, hti ? &Y_T : nullptr, hte ? &hT : nullptr);
    inf_pr = pc.first; inf_comp = pc.second;
    merit = computeBarrierMerit(S, cost, hti ? &S_T : nullptr, hte ? &Lambda_T_eq : nullptr, hte ? &hT : nullptr);
    phi = merit; filter_theta = ftheta;
    theta = std::max(ftheta, std::max(opt.ipddp_theta_0_floor, 1e-8));
  }

  void ipddp_apply(const FPResult &r) {  // ipddp_solver.cpp:1878-1951
    X = r.X; U = r.U; cost = r.cost; merit = r.merit; alpha_pr = r.alpha_pr; alpha_du = r.alpha_du;
    Y = r.Y; S = r.S; G = r.G; Lambda = r.Lambda;
    if (has_term_ineq()) { S_T = r.S_T; Y_T = r.Y_T; G_T = r.G_T; }
    if (term_eq_dim() > 0) Lambda_T_eq = r.Lambda_T_eq;
    inf_pr = r.inf_pr; inf_comp = r.inf_comp; phi = r.merit; filter_theta = r.theta; theta = r.theta;
    updateBarrierParameters(true);
  }

  bool ipddp_checkEarlyConvergence(int &st) {  // ipddp_solver.cpp:925-958
    const bool no_barrier = cons.empty() && !has_term_ineq();
    const double sdu = computeScaledDualInfeasibility();
    if (no_barrier) {
      if (inf_pr < opt.tolerance && sdu < opt.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; return true; }
      return false;
    }
    const double tol = std::max(opt.tolerance, opt.ipddp_barrier_tol_mult * mu);
    const double accepted_step_norm = std::fabs(alpha_pr) * step_norm;
    if (inf_pr < tol && sdu < tol && inf_comp < tol && accepted_step_norm < opt.tolerance * 10.0) { st = CDDP_HIP_STATUS_OPTIMAL; return true; }
    return false;
  }

  bool ipddp_checkConvergence(double dJ, int iter, int &st) {  // ipddp_solver.cpp:1953-2025
    const bool no_barrier = cons.empty() && !has_term_ineq();
    const double sdu = computeScaledDualInfeasibility();
    const double scomp = inf_comp;
    if (no_barrier) {
      if (inf_pr < opt.tolerance && sdu < opt.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; return true; }
      if (opt.acceptable_tolerance > 0.0) {
        const double sq = std::sqrt(opt.acceptable_tolerance);
        bool acc = (inf_pr < sq && sdu < sq && iter > 50);
        if (dJ > 0.0) acc = acc || (dJ < opt.acceptable_tolerance && iter > 50 && inf_pr < sq && sdu < sq);
        if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; return true; }
      }
      return false;
    }
    const double tol = std::max(opt.tolerance, opt.ipddp_barrier_tol_mult * mu);
    if (inf_pr < tol && sdu < tol && scomp < tol && step_norm < opt.tolerance * 10.0) { st = CDDP_HIP_STATUS_OPTIMAL; return true; }
    if (opt.acceptable_tolerance > 0.0) {
      const double at = std::sqrt(opt.acceptable_tolerance);
      const double bat = std::max(opt.barrier_mu_min_value * 100.0, opt.tolerance / 10.0);
      const bool akkt = inf_pr < at && sdu < at && scomp < at;
      const bool bpc = mu <= bat;
      bool acc = akkt && bpc && iter > 10 && std::fabs(dJ) < opt.acceptable_tolerance;
      acc = acc || (akkt && bpc && iter >= 1 && step_norm < opt.tolerance * 10.0 && inf_pr < 1e-4);
      if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; return true; }
    }
    return false;
  }

  bool ipddp_handleForwardPassFailure(int &st) {  // ipddp_solver.cpp:2037-2082
    increaseRegularization();
    const bool no_barrier = cons.empty() && !has_term_ineq();
    if (!no_barrier && term_eq_dim() > 0) increaseRegularization();
    if (isRegularizationLimitReached()) {
      const double sdu = computeScaledDualInfeasibility();
      const double scomp = inf_comp;
      const double base = std::sqrt(std::max(opt.acceptable_tolerance, opt.tolerance));
      const double at = no_barrier ? base : std::max(base, opt.ipddp_barrier_tol_mult * mu);
      const bool acc = opt.acceptable_tolerance > 0.0 && inf_pr < at && sdu < at && (no_barrier || scomp < at);
      st = acc ? CDDP_HIP_STATUS_ACCEPTABLE : CDDP_HIP_STATUS_REG_LIMIT;
      return true;
    }
    return false;
  }


  // =============================================================== LogDDP (logddp_solver.cpp; RelaxedLogBarrier barrier.hpp:37-296)
  double lg_mu = 1e-1, lg_delta = 1e-5, lg_violation = 1e7;   // LogDDPSolver members mu_, relaxation_delta_, constraint_violation_ (:43-44)

  // beta_delta(z) and its first two derivatives (barrier.hpp:274-296)
  static void lg_beta(double z, double delta, double &b0, double &b1, double &b2) {
    if (z > delta) {
      if (z <= 1e-12) { b0 = -std::log(1e-12); b1 = -1.0 / 1e-12; b2 = 1.0 / (1e-12 * 1e-12); }
      else { b0 = -olog(z); b1 = -1.0 / z; b2 = 1.0 / (z * z); }
    } else {
      const double term_div_delta = (z - 2.0 * delta) / delta;
      b0 = 0.5 * (term_div_delta * term_div_delta - 1.0) - olog(delta);
      b1 = term_div_delta / delta;
      b2 = 1.0 / (delta * delta);
    }
  }
  // RelaxedLogBarrier::evaluate (:61-91): every constraint kind here has lower bound -inf, so only the upper side s_U = U - g enters;
  // con_g returns g - U, i.e. s_U = -con_g
  double lg_barrier_value(const ConstraintDesc &c, const Vec &x, const Vec &u) const {
    const Vec g = con_g(c, x, u);
    double total = 0.0;
    for (int i = 0; i < c.dual_dim; ++i) { double b0, b1, b2; lg_beta(-g(i), lg_delta, b0, b1, b2); total += b0; }
    return lg_mu * total;
  }
  // second derivatives of the constraint rows (Constraint::getHessians): false when the constraint throws logic_error (the cone),
  // zero matrices for the kinds that keep the base-class defaults (constraint.hpp:86-120)
  bool con_hess(const ConstraintDesc &c, const Vec &u, std::vector<Mat> &Hxx, std::vector<Mat> &Huu, std::vector<Mat> &Hux) const {
    if (c.kind == CDDP_HIP_CON_SOC) return false;                      // :772-786 throw std::logic_error
    Hxx.assign(c.dual_dim, Mat::Zero(nx, nx)); Huu.assign(c.dual_dim, Mat::Zero(nu, nu)); Hux.assign(c.dual_dim, Mat::Zero(nu, nx));
    if (c.kind == CDDP_HIP_CON_BALL) for (int i = 0; i < c.dim; ++i) Hxx[0](i, i) = -2.0 * c.scale;   // :387-396
    if (c.kind == CDDP_HIP_CON_THRUST || c.kind == CDDP_HIP_CON_MAX_THRUST) {   // :899-920, 1021-1042: (|u|^2 + eps) I - u u^T over (|u|^2 + eps)^1.5
      double sq = 0; for (int i = 0; i < nu; ++i) sq += u(i) * u(i);
      const double term = sq + c.scale, den = opow(term, 1.5);
      Mat H = Mat::Zero(nu, nu);
      if (den > std::numeric_limits<double>::min()) for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) H(i, j) = ((i == j ? term : 0.0) - u(i) * u(j)) / den;
      if (c.kind == CDDP_HIP_CON_THRUST) { Huu[0] = -1.0 * H; Huu[1] = H; } else Huu[0] = H;
    }
    return true;
  }
  // getGradients (:95-135) and getHessians (:137-213) of one constraint, added to the Q blocks by the caller
  void lg_barrier_derivs(const ConstraintDesc &c, const Vec &x, const Vec &u, Vec &gx, Vec &gu, Mat &Hxx, Mat &Huu, Mat &Hux) const {
    const Vec g = con_g(c, x, u);
    Mat Gx, Gu; con_jac(c, x, u, Gx, Gu);
    gx = Vec::Zero(nx); gu = Vec::Zero(nu); Hxx = Mat::Zero(nx, nx); Huu = Mat::Zero(nu, nu); Hux = Mat::Zero(nu, nx);
    std::vector<Mat> Cxx, Cuu, Cux;
    const bool provides = con_hess(c, u, Cxx, Cuu, Cux);
    for (int i = 0; i < c.dual_dim; ++i) {
      double b0, b1, b2; lg_beta(-g(i), lg_delta, b0, b1, b2);
      double dCost = 0.0; dCost -= b1;                 // upper side only
      double t1 = 0.0, t2 = 0.0; t1 += b2; t2 -= b1;
      for (int a = 0; a < nx; ++a) gx(a) += dCost * Gx(i, a);
      for (int a = 0; a < nu; ++a) gu(a) += dCost * Gu(i, a);
      for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) Hxx(a, b) += (t1 * Gx(i, a)) * Gx(i, b);
      for (int a = 0; a < nu; ++a) for (int b = 0; b < nu; ++b) Huu(a, b) += (t1 * Gu(i, a)) * Gu(i, b);
      for (int a = 0; a < nu; ++a) for (int b = 0; b < nx; ++b) Hux(a, b) += (t1 * Gu(i, a)) * Gx(i, b);
      if (provides) { Hxx = Hxx + t2 * Cxx[i]; Huu = Huu + t2 * Cuu[i]; Hux = Hux + t2 * Cux[i]; }
    }
    gx = lg_mu * gx; gu = lg_mu * gu; Hxx = lg_mu * Hxx; Huu = lg_mu * Huu; Hux = lg_mu * Hux;
  }
  void lg_evaluate_trajectory() {   // :316-331
    double c = 0.0;
    for (int t = 0; t < N; ++t) c += running_cost(X[t], U[t], t);
    c += terminal_cost(X.back());
    cost = c;
  }
  void lg_reset_filter() {          // :333-361
    merit = cost; lg_violation = 0.0;
    for (int t = 0; t < N; ++t) for (auto &c : cons) {
      const Vec g = con_g(c, X[t], U[t]);
      merit += lg_barrier_value(c, X[t], U[t]);
      for (int i = 0; i < c.dual_dim; ++i) if (g(i) > 0.0) lg_violation += g(i);
    }
    inf_pr = lg_violation;
  }
  void logddp_initialize() {        // :45-205 (cold start; the warm start branch keeps gains and re-rolls out the same way)
    const bool warm = opt.warm_start && have_valid_gains();
    X[0] = x0;                      // rollOutNominalTrajectory (:30-38): the state guess is only a guess
    for (int t = 0; t < N; ++t) X[t + 1] = model.step(X[t], U[t], t * dt);
    if (!warm) { initializeGains(); cost = objective_evaluate(X, U); }
    alpha_pr = opt.ls_initial_step_size; dV[0] = dV[1] = 0.0;
    reg = opt.reg_initial_value;
    lg_violation = std::numeric_limits<double>::infinity();
    lg_mu = opt.logddp_mu_initial; lg_delta = opt.logddp_relaxed_delta;
    Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx));
    lg_evaluate_trajectory(); lg_reset_filter();
  }
  bool logddp_backward() {          // :365-590
    ++n_backward;
    Vec V_x = final_grad(X.back());
    Mat V_xx = final_hess(); V_xx = 0.5 * (V_xx + V_xx.T());
    Vx_t[N] = V_x; Vxx_t[N] = V_xx;
    dV[0] = dV[1] = 0.0;
    double Qu_err = 0.0;
    for (int t = N - 1; t >= 0; --t) {
      const Vec &x = X[t]; const Vec &u = U[t];
      Mat Fx, Fu; model.jacobians(x, u, t * dt, Fx, Fu);
      Mat A = dt * Fx; for (int i = 0; i < nx; ++i) A(i, i) += 1.0;
      Mat B = dt * Fu;
      Vec Q_x = l_x(x, t) + A.T() * V_x;
      Vec Q_u = l_u(u) + B.T() * V_x;
      Mat Q_xx = l_xx() + A.T() * V_xx * A;
      Mat Q_ux = l_ux() + B.T() * V_xx * A;
      Mat Q_uu = l_uu() + B.T() * V_xx * B;
      if (!opt.use_ilqr) {          // :505-515
        std::vector<Mat> Fxx, Fuu, Fux;
        if (!model.hessians(x, u, t * dt, Fxx, Fuu, Fux)) { std::fprintf(stderr, "oracle: use_ilqr=false needs Hessians\n"); std::abort(); }
        for (int i = 0; i < nx; ++i) { Q_xx = Q_xx + (dt * V_x(i)) * Fxx[i]; Q_ux = Q_ux + (dt * V_x(i)) * Fux[i]; Q_uu = Q_uu + (dt * V_x(i)) * Fuu[i]; }
      }
      for (auto &c : cons) {        // :518-530
        Vec gx, gu; Mat Hxx, Huu, Hux; lg_barrier_derivs(c, x, u, gx, gu, Hxx, Huu, Hux);
        Q_x = Q_x + gx; Q_u = Q_u + gu; Q_xx = Q_xx + Hxx; Q_uu = Q_uu + Huu; Q_ux = Q_ux + Hux;
      }
      Mat Q_uu_reg = Q_uu; for (int i = 0; i < nu; ++i) Q_uu_reg(i, i) += reg;
      Q_uu_reg = 0.5 * (Q_uu_reg + Q_uu_reg.T());
      LDLT ldlt(Q_uu_reg);
      if (!ldlt.ok) return false;
      Mat bigRHS(nu, 1 + nx);
      for (int i = 0; i < nu; ++i) { bigRHS(i, 0) = Q_u(i); for (int c = 0; c < nx; ++c) bigRHS(i, c + 1) = Q_ux(i, c); }
      const Mat kK = -ldlt.solve(bigRHS);
      Vec k(nu, 1); Mat K(nu, nx);
      for (int i = 0; i < nu; ++i) { k(i) = kK(i, 0); for (int c = 0; c < nx; ++c) K(i, c) = kK(i, c + 1); }
      k_u[t] = k; K_u[t] = K;
      dV[0] += Q_u.dot(k);
      dV[1] += 0.5 * k.dot(Q_uu * k);
      V_x = Q_x + K.T() * Q_uu * k + Q_ux.T() * k + K.T() * Q_u;
      V_xx = Q_xx + K.T() * Q_uu * K + Q_ux.T() * K + K.T() * Q_ux;
      V_xx = 0.5 * (V_xx + V_xx.T());
      Vx_t[t] = V_x; Vxx_t[t] = V_xx;
      Qu_err = std::max(Qu_err, Q_u.lpNormInf());
    }
    inf_du = Qu_err;
    return true;
  }
  FPResult logddp_forward(double a) {   // :594-707
    FPResult r; r.alpha = a; r.alpha_pr = a; r.success = false;
    r.cost = r.merit = std::numeric_limits<double>::infinity();
    r.X = X; r.U = U; r.X[0] = x0;
    for (int t = 0; t < N; ++t) {
      const Vec delta_x = r.X[t] - X[t];
      r.U[t] = U[t] + a * k_u[t] + K_u[t] * delta_x;
      r.X[t + 1] = model.step(r.X[t], r.U[t], t * dt);
      if (!r.X[t + 1].allFinite() || !r.U[t].allFinite()) return r;
    }
    double cost_new = 0.0, merit_new = 0.0, rp_err = 0.0;
    for (int t = 0; t < N; ++t) {
      cost_new += running_cost(r.X[t], r.U[t], t);
      for (auto &c : cons) {
        const Vec g = con_g(c, r.X[t], r.U[t]);
        // (lg_barrier_value on the trial point)
        double total = 0.0;
        for (int i = 0; i < c.dual_dim; ++i) { double b0, b1, b2; lg_beta(-g(i), lg_delta, b0, b1, b2); total += b0; }
        merit_new += lg_mu * total;
        for (int i = 0; i < c.dual_dim; ++i) if (g(i) > 0.0) rp_err += g(i);
      }
    }
    cost_new += terminal_cost(r.X.back());
    merit_new += cost_new;
    const double cv_old = lg_violation, cv_new = rp_err, merit_old = merit;
    bool accept = false;
    const double expected = a * dV[0];
    if (cv_new > opt.filter_max_violation_threshold) {
      if (cv_new < (1.0 - opt.filter_violation_acceptance_threshold) * cv_old) accept = true;
    } else if (std::max(cv_new, cv_old) < opt.filter_min_violation_for_armijo_check && expected < 0) {
      if (merit_new < merit_old + opt.filter_armijo_constant * expected) accept = true;
    } else {
      if (merit_new < merit_old - opt.filter_merit_acceptance_threshold * cv_old || cv_new < (1.0 - opt.filter_violation_acceptance_threshold) * cv_old) accept = true;
    }
    if (accept) { r.success = true; r.cost = cost_new; r.merit = merit_new; r.inf_pr = cv_new; }
    return r;
  }
  void logddp_post_iteration(bool fp_success) {   // :263-277
    if (fp_success) lg_mu = std::max(opt.logddp_mu_min_value, lg_mu * opt.logddp_mu_update_factor);
    else lg_mu = std::min(opt.logddp_mu_initial, lg_mu * 5.0);
    lg_reset_filter();
  }


  // =============================================================== MSIPDDP (msipddp_solver.cpp): multiple-shooting interior-point DDP.
  // The iterate carries costates Lambda_t and the dynamics values F_t = f(x_t, u_t); the defect d_t = F_t - x_{t+1} enters the backward
  // pass.  Stacked duals / slacks reuse the IPDDP containers (Y, S, G, k_y, K_y, k_s, K_s: one vector of total dual dimension per step,
  // constraints in std::map order).  Two properties of the reference are restated as they are, not repaired:
  //  * the unconstrained branch caches the LDLT of Q_uu per step and only refactors a step whose cached factor is invalid
  //    (:1169-1185): after the first successful sweep every later sweep solves with the FIRST factor of its step;
  //  * the constrained branch adds the (nx x nu) product Q_yx^T YS^-1 Q_yu to the (nu x nx) block Q_ux (:1398): defined for nu = 1
  //    (same linear layout: the transpose lands) and for nx = nu (elementwise, untransposed); every other shape is refused here.
  std::vector<Vec> ms_F, ms_kl;
  std::vector<Mat> ms_Kl;
  std::vector<LDLT> ms_ldlt;
  std::vector<char> ms_ldlt_valid;
  bool ms_workspace = false;
  int ms_seg = 5;

  void ms_evaluate_trajectory() {            // evaluateTrajectory :425-455
    double c = 0.0;
    X[0] = x0;
    for (int t = 0; t < N; ++t) {
      c += running_cost(X[t], U[t], t);
      for (auto &cd : cons) G[t].setSegment(cd.offset, con_g(cd, X[t], U[t]));
      ms_F[t] = model.step(X[t], U[t], t * dt);
      X[t + 1] = ms_F[t];
    }
    c += terminal_cost(X.back());
    cost = c;
  }
  void ms_evaluate_trajectory_warm() {       // evaluateTrajectoryWarmStart :457-495
    double c = 0.0;
    G.assign(N, Vec::Zero(m));
    for (int t = 0; t < N; ++t) {
      c += running_cost(X[t], U[t], t);
      for (auto &cd : cons) G[t].setSegment(cd.offset, con_g(cd, X[t], U[t]));
      ms_F[t] = model.step(X[t], U[t], t * dt);
      if (opt.msipddp_use_controlled_rollout) X[t + 1] = ms_F[t];
    }
    c += terminal_cost(X.back());
    cost = c;
  }
  void ms_init_pair(const Vec &g, Vec &s_init, Vec &y_init, const ConstraintDesc &cd) const {   // :578-596 == :667-685
    for (int i = 0; i < cd.dual_dim; ++i) {
      const int j = cd.offset + i;
      s_init(j) = std::max(opt.ipddp_slack_var_init_scale, -g(j));
      y_init(j) = (s_init(j) < 1e-12) ? mu / 1e-12 : mu / s_init(j);
      y_init(j) = std::max(opt.ipddp_dual_var_init_scale * 0.01, std::min(y_init(j), opt.ipddp_dual_var_init_scale * 100.0));
    }
  }
  void ms_zero_path_gains() {
    k_y.assign(N, Vec::Zero(m)); k_s.assign(N, Vec::Zero(m)); K_y.assign(N, Mat::Zero(m, nx)); K_s.assign(N, Mat::Zero(m, nx));
  }
  void ms_init_dual_slack_costate() {        // initializeDualSlackCostateVariables :643-709
    G.assign(N, Vec::Zero(m)); Y.assign(N, Vec::Zero(m)); S.assign(N, Vec::Zero(m));
    for (auto &cd : cons)
      for (int t = 0; t < N; ++t) {
        G[t].setSegment(cd.offset, con_g(cd, X[t], U[t]));
        ms_init_pair(G[t], S[t], Y[t], cd);
      }
    ms_zero_path_gains();
    for (int t = 0; t < N; ++t) {
      Lambda[t] = opt.msipddp_costate_var_init_scale * Vec::Ones(nx);
      ms_kl[t] = Vec::Zero(nx); ms_Kl[t] = Mat::Zero(nx, nx);
    }
    cost = objective_evaluate(X, U);
  }
  void ms_init_dual_slack_costate_warm(bool has_existing) {   // initializeDualSlackCostateVariablesWarmStart :497-641
    if (!has_existing) { Y.assign(N, Vec::Zero(m)); S.assign(N, Vec::Zero(m)); }
    for (auto &cd : cons)
      for (int t = 0; t < N; ++t) {
        bool need_reinit = !has_existing;
        if (has_existing)
          for (int i = 0; i < cd.dual_dim; ++i) {
            const int j = cd.offset + i;
            if (Y[t](j) <= 1e-12 || S[t](j) <= 1e-12) { need_reinit = true; break; }
            const double required = std::max(opt.ipddp_slack_var_init_scale, -G[t](j));
            if (S[t](j) < 0.1 * required) { need_reinit = true; break; }
          }
        if (need_reinit) ms_init_pair(G[t], S[t], Y[t], cd);
      }
    ms_zero_path_gains();
    const bool has_costate = (int)Lambda.size() == N;
    if (!has_costate) Lambda.assign(N, opt.msipddp_costate_var_init_scale * Vec::Ones(nx));
    ms_kl.assign(N, Vec::Zero(nx)); ms_Kl.assign(N, Mat::Zero(nx, nx));
  }
  void ms_reset_filter() {                   // resetBarrierFilter :711-763
    double mf = cost, ipr = 0.0, fcv = 0.0, icomp = 0.0, idef = 0.0;
    if (!cons.empty()) {
      for (int t = 0; t < N; ++t) {
        for (auto &cd : cons) {
          double lsum = 0.0, l1 = 0.0;
          for (int i = 0; i < cd.dual_dim; ++i) {
            const int j = cd.offset + i;
            lsum += olog(S[t](j));
            const double pr = G[t](j) + S[t](j);
            ipr = std::max(ipr, std::fabs(pr)); l1 += std::fabs(pr);
            icomp = std::max(icomp, std::fabs(Y[t](j) * S[t](j) - mu));
          }
          mf -= mu * lsum; fcv += l1;
        }
        if (t < (int)ms_F.size() && t + 1 < (int)X.size()) {
          const Vec dres = ms_F[t] - X[t + 1];
          idef = std::max(idef, dres.lpNormInf());
          fcv += dres.lpNorm1();
        }
      }
    }
    inf_pr = std::max(ipr, idef); merit = mf; inf_comp = icomp;
    filter.clear(); filter.push_back(FilterPoint{mf, fcv});
  }
  bool ms_filter_acceptable(double mf, double cv, double expected_improvement) const {   // isFilterAcceptable :771-808
    if (filter.empty()) return true;
    FilterPoint cand{mf, cv};
    for (auto &p : filter) if (p.dominates(cand)) return false;
    double best_v = std::numeric_limits<double>::infinity(), best_m = std::numeric_limits<double>::infinity();
    for (auto &p : filter) if (p.constraint_violation < best_v) { best_v = p.constraint_violation; best_m = p.merit_function; }
    const bool v_imp = cv < best_v * (1.0 - opt.filter_violation_acceptance_threshold);
    const bool m_imp = mf < best_m - opt.filter_merit_acceptance_threshold * cv;
    if (cv < opt.filter_min_violation_for_armijo_check && expected_improvement < 0) return mf < best_m + opt.filter_armijo_constant * expected_improvement;
    if (cv < 1e-6 && mf <= best_m * (1.0 + 1e-8)) return true;
    return v_imp || m_imp;
  }
  double ms_scaled_inf_du() const {          // computeScaledDualInfeasibility :1886-1930
    if (cons.empty()) return inf_du;
    const double smax = 100.0;
    double yn = 0.0, sn = 0.0; int total = 0;
    for (auto &cd : cons)
      for (int t = 0; t < N; ++t) { yn += seg(Y[t], cd).lpNorm1(); sn += seg(S[t], cd).lpNorm1(); total += cd.dual_dim; }
    const int mpn = total + nu * N;
    const double num = mpn > 0 ? (yn + sn) / (double)mpn : 0.0;
    const double sd = std::max(smax, num) / smax;
    return inf_du / sd;
  }
  bool ms_shape_defined() const { return cons.empty() || nu == 1 || nx == nu; }
  void msipddp_initialize() {                // initialize :33-264
    if (!ms_shape_defined()) { std::fprintf(stderr, "oracle: MSIPDDP with path constraints is only defined for nu = 1 or nx = nu (msipddp_solver.cpp:1398)\n"); std::abort(); }
    if (!ms_workspace) { ms_ldlt.assign(N, LDLT()); ms_ldlt_valid.assign(N, 0); ms_workspace = true; }
    ms_seg = opt.msipddp_segment_length;
    Vx_t.assign(N + 1, Vec::Zero(nx)); Vxx_t.assign(N + 1, Mat::Zero(nx, nx));
    if (opt.warm_start) {
      if (have_valid_gains()) {              // :95-106
        mu = opt.barrier_mu_initial * 0.1; step_norm = 0.0;
        if ((int)ms_F.size() != N) ms_F.assign(N, Vec::Zero(nx));
        const bool has_existing = path_duals_exist();
        ms_evaluate_trajectory_warm();
        ms_init_dual_slack_costate_warm(has_existing);
        ms_reset_filter();
        return;
      }
      initializeGains();                     // :108-160
      Lambda.assign(N, opt.msipddp_costate_var_init_scale * Vec::Ones(nx));
      ms_kl.assign(N, Vec::Zero(nx)); ms_Kl.assign(N, Mat::Zero(nx, nx)); ms_F.assign(N, Vec::Zero(nx));
      G.clear(); Y.clear(); S.clear();
      if (cons.empty()) mu = 1e-8;
      else {
        ms_evaluate_trajectory_warm();
        const double mv = computeMaxConstraintViolation();
        if (mv <= opt.tolerance) mu = opt.tolerance * 0.01;
        else if (mv <= 0.1) mu = opt.tolerance;
        else mu = opt.barrier_mu_initial * 0.1;
      }
      reg = opt.reg_initial_value; step_norm = 0.0;
      if (cons.empty()) G.assign(N, Vec::Zero(m));
      ms_init_dual_slack_costate_warm(false);
      ms_reset_filter();
      return;
    }
    initializeGains();                       // cold start :199-263 (the trajectory guess of set_initial counts as provided)
    Lambda.assign(N, Vec::Zero(nx)); ms_kl.assign(N, Vec::Zero(nx)); ms_Kl.assign(N, Mat::Zero(nx, nx)); ms_F.assign(N, Vec::Zero(nx));
    mu = cons.empty() ? 1e-8 : opt.barrier_mu_initial;
    ms_init_dual_slack_costate();
    reg = opt.reg_initial_value; step_norm = 0.0;
    ms_evaluate_trajectory();
    ms_reset_filter();
  }
  bool msipddp_backward() {                  // backwardPass :1112-1430
    ++n_backward;
    Vec V_x = final_grad(X.back());
    Mat V_xx = final_hess(); V_xx = 0.5 * (V_xx + V_xx.T());
    Vx_t[N] = V_x; Vxx_t[N] = V_xx;
    dV[0] = dV[1] = 0.0;
    double idu = 0.0, ipr = 0.0, icomp = 0.0, idef = 0.0, snorm = 0.0;
    if ((int)Gx.size() != N) { Gx.assign(N, Mat::Zero(m, nx)); Gu.assign(N, Mat::Zero(m, nu)); }
    for (int t = N - 1; t >= 0; --t) {
      const Vec &x = X[t]; const Vec &u = U[t]; const Vec &lambda = Lambda[t];
      Vec d = Vec::Zero(nx);
      if (t + 1 < (int)X.size()) d = ms_F[t] - X[t + 1];
      Mat Fx, Fu; model.jacobians(x, u, t * dt, Fx, Fu);
      Mat A = dt * Fx; for (int i = 0; i < nx; ++i) A(i, i) += 1.0;
      Mat B = dt * Fu;
      const Vec lx = l_x(x, t), lu = l_u(u);
      const Vec w = V_x + V_xx * d;
      Vec y = Vec::Zero(m), sv = Vec::Zero(m), g = Vec::Zero(m);
      Mat Q_yu = Mat::Zero(m, nu), Q_yx = Mat::Zero(m, nx);
      if (!cons.empty()) {
        for (auto &cd : cons) {
          Mat gx, gu; con_jac(cd, x, u, gx, gu);
          Q_yx.setBlock(cd.offset, 0, gx); Q_yu.setBlock(cd.offset, 0, gu);
        }
        Gx[t] = Q_yx; Gu[t] = Q_yu;
        y = Y[t]; sv = S[t]; g = G[t];
      }
      Vec Q_x = cons.empty() ? lx + A.T() * w : lx + Q_yx.T() * y + A.T() * w;
      Vec Q_u = cons.empty() ? lu + B.T() * w : lu + Q_yu.T() * y + B.T() * w;
      Mat Q_xx = l_xx() + A.T() * V_xx * A;
      Mat Q_ux = l_ux() + B.T() * V_xx * A;
      Mat Q_uu = l_uu() + B.T() * V_xx * B;
      if (!opt.use_ilqr) {                   // :1151-1163, :1279-1310: the Hessians are weighted with the costates
        std::vector<Mat> Fxx, Fuu, Fux;
        if (!model.hessians(x, u, t * dt, Fxx, Fuu, Fux)) { std::fprintf(stderr, "oracle: use_ilqr=false needs Hessians\n"); std::abort(); }
        for (int i = 0; i < nx; ++i) { Q_xx = Q_xx + (dt * lambda(i)) * Fxx[i]; Q_ux = Q_ux + (dt * lambda(i)) * Fux[i]; Q_uu = Q_uu + (dt * lambda(i)) * Fuu[i]; }
        for (auto &cd : cons) {
          std::vector<Mat> Cxx, Cuu, Cux;
          if (!con_hess(cd, u, Cxx, Cuu, Cux)) { std::fprintf(stderr, "oracle: constraint without Hessians under use_ilqr=false\n"); std::abort(); }
          for (int i = 0; i < cd.dual_dim; ++i) { Q_xx = Q_xx + y(cd.offset + i) * Cxx[i]; Q_ux = Q_ux + y(cd.offset + i) * Cux[i]; Q_uu = Q_uu + y(cd.offset + i) * Cuu[i]; }
        }
      }
      Vec k(nu, 1); Mat K(nu, nx);
      if (cons.empty()) {
        Q_uu = 0.5 * (Q_uu + Q_uu.T());
        for (int i = 0; i < nu; ++i) Q_uu(i, i) += reg;
        const bool need_recompute = !ms_ldlt_valid[t] || (ms_ldlt_valid[t] && ms_ldlt[t].n != nu);   // :1169-1176
        if (need_recompute) { ms_ldlt[t].compute(Q_uu); ms_ldlt_valid[t] = 1; }
        if (!ms_ldlt[t].ok) { ms_ldlt_valid[t] = 0; return false; }
        k = -ms_ldlt[t].solve(Q_u);
        K = -ms_ldlt[t].solve(Q_ux);
        k_u[t] = k; K_u[t] = K;
        ms_kl[t] = -lambda + V_x + V_xx * d;
        ms_Kl[t] = 0.5 * (V_xx + V_xx.T());
        V_x = Q_x + K.T() * Q_u + Q_ux.T() * k + K.T() * Q_uu * k;
        V_xx = Q_xx + K.T() * Q_ux + Q_ux.T() * K + K.T() * Q_uu * K;
        V_xx = 0.5 * (V_xx + V_xx.T());
        dV[0] += k.dot(Q_u);
        dV[1] += 0.5 * k.dot(Q_uu * k);
      } else {
        Mat YSinv = Mat::Zero(m, m);
        for (int i = 0; i < m; ++i) YSinv(i, i) = y(i) / sv(i);
        const Vec pres = g + sv;
        Vec cres(m, 1), rhat(m, 1), Sir(m, 1);
        for (int i = 0; i < m; ++i) { cres(i) = y(i) * sv(i) - mu; rhat(i) = y(i) * pres(i) - cres(i); Sir(i) = rhat(i) / sv(i); }
        Mat Q_uu_reg = 0.5 * (Q_uu + Q_uu.T());
        Q_uu_reg = Q_uu_reg + Q_yu.T() * YSinv * Q_yu;
        for (int i = 0; i < nu; ++i) Q_uu_reg(i, i) += reg;
        LDLT ldlt(Q_uu_reg);
        if (!ldlt.ok) return false;
        Mat bigRHS(nu, 1 + nx);
        const Vec r0 = Q_u + Q_yu.T() * Sir;
        const Mat r1 = Q_ux + Q_yu.T() * YSinv * Q_yx;
        for (int i = 0; i < nu; ++i) { bigRHS(i, 0) = r0(i); for (int c = 0; c < nx; ++c) bigRHS(i, c + 1) = r1(i, c); }
        const Mat kK = -ldlt.solve(bigRHS);
        for (int i = 0; i < nu; ++i) { k(i) = kK(i, 0); for (int c = 0; c < nx; ++c) K(i, c) = kK(i, c + 1); }
        k_u[t] = k; K_u[t] = K;
        const Vec temp = Q_yu * k;
        Vec ky(m, 1);
        for (int i = 0; i < m; ++i) ky(i) = (rhat(i) + y(i) * temp(i)) / sv(i);
        k_y[t] = ky;
        K_y[t] = YSinv * (Q_yx + Q_yu * K);
        k_s[t] = -pres - temp;
        K_s[t] = -Q_yx - Q_yu * K;
        ms_kl[t] = -lambda + V_x + V_xx * d;
        ms_Kl[t] = 0.5 * (V_xx + V_xx.T());
        Q_u = Q_u + Q_yu.T() * Sir;
        Q_x = Q_x + Q_yx.T() * Sir;
        Q_xx = Q_xx + Q_yx.T() * YSinv * Q_yx;
        {   // :1398  Q_ux.noalias() += Q_yx^T * YSinv * Q_yu  -- an (nx x nu) product added to the (nu x nx) block
          const Mat P = Q_yx.T() * YSinv * Q_yu;
          if (nu == 1) for (int c = 0; c < nx; ++c) Q_ux(0, c) += P(c, 0);          // same linear layout
          else for (int i = 0; i < nu; ++i) for (int c = 0; c < nx; ++c) Q_ux(i, c) += P(i, c);   // nx == nu: elementwise
        }
        Q_uu = Q_uu + Q_yu.T() * YSinv * Q_yu;
        dV[0] += k.dot(Q_u);
        dV[1] += 0.5 * k.dot(Q_uu * k);
        V_x = Q_x + K.T() * Q_u + Q_ux.T() * k + K.T() * Q_uu * k;
        V_xx = Q_xx + K.T() * Q_ux + Q_ux.T() * K + K.T() * Q_uu * K;
        V_xx = 0.5 * (V_xx + V_xx.T());
        ipr = std::max(ipr, pres.lpNormInf());
        icomp = std::max(icomp, cres.lpNormInf());
      }
      Vx_t[t] = V_x; Vxx_t[t] = V_xx;
      idu = std::max(idu, Q_u.lpNormInf());
      snorm = std::max(snorm, k.lpNormInf());
      idef = std::max(idef, d.lpNormInf());
    }
    inf_du = idu; step_norm = snorm;
    if (cons.empty()) { inf_pr = idef; inf_comp = 0.0; }
    else { inf_pr = std::max(ipr, idef); inf_comp = icomp; }
    return true;
  }
  Vec ms_next_state(int t, const FPResult &r, const Vec &Fn, const Vec &delta_x, double a) const {   // gap closing :1483-1509 == :1575-1601
    const bool boundary = (ms_seg > 1) && ((t + 1) % ms_seg == 0) && (t + 1 < N);
    if (!boundary) return Fn;
    if (opt.msipddp_rollout_type == 0) return X[t + 1] + (Fn - ms_F[t]) + a * (ms_F[t] - X[t + 1]);
    if (opt.msipddp_rollout_type == 2) {
      Mat Fx, Fu; model.jacobians(X[t], U[t], t * dt, Fx, Fu);
      Mat A = dt * Fx; for (int i = 0; i < nx; ++i) A(i, i) += 1.0;
      const Mat B = dt * Fu;
      return X[t + 1] + (A + B * K_u[t]) * delta_x + a * (B * k_u[t] + ms_F[t] - X[t + 1]);
    }
    (void)r;
    return Fn;
  }
  FPResult msipddp_forward(double a) {       // forwardPass :1432-1724
    FPResult r; r.alpha = a; r.alpha_pr = a; r.success = false;
    r.cost = r.merit = std::numeric_limits<double>::infinity();
    const double tau = std::max(opt.barrier_min_fraction_to_boundary, 1.0 - mu);
    r.X = X; r.U = U; r.X[0] = x0;
    r.F = ms_F; r.Lambda = Lambda; r.Y = Y; r.S = S; r.G = G;
    std::vector<Vec> dxs(N, Vec::Zero(nx));
    double cost_new = 0.0, merit_new = 0.0, cv_new = 0.0;
    if (cons.empty()) {
      for (int t = 0; t < N; ++t) {
        const Vec delta_x = r.X[t] - X[t];
        r.U[t] = U[t] + a * k_u[t] + K_u[t] * delta_x;
        r.Lambda[t] = Lambda[t] + a * ms_kl[t] + ms_Kl[t] * delta_x;
        r.F[t] = model.step(r.X[t], r.U[t], t * dt);
        r.X[t + 1] = ms_next_state(t, r, r.F[t], delta_x, a);
        cost_new += running_cost(r.X[t], r.U[t], t);
      }
      cost_new += terminal_cost(r.X.back());
      const double dJ = cost - cost_new;
      const double expected = -a * (dV[0] + 0.5 * a * dV[1]);
      const double ratio = expected > 0.0 ? dJ / expected : std::copysign(1.0, dJ);
      r.success = ratio > 1e-6;
      r.cost = cost_new; r.merit = cost_new; r.theta = 0.0; r.alpha_du = 1.0; r.has_ip = false;
      return r;
    }
    for (int t = 0; t < N; ++t) {
      const Vec delta_x = r.X[t] - X[t];
      dxs[t] = delta_x;
      const Vec s_new = S[t] + a * k_s[t] + K_s[t] * delta_x;
      for (auto &cd : cons)
        for (int i = 0; i < cd.dual_dim; ++i) {
          const int j = cd.offset + i;
          if (s_new(j) < (1.0 - tau) * S[t](j)) return r;
        }
      r.S[t] = s_new;
      r.U[t] = U[t] + a * k_u[t] + K_u[t] * delta_x;
      r.F[t] = model.step(r.X[t], r.U[t], t * dt);
      r.X[t + 1] = ms_next_state(t, r, r.F[t], delta_x, a);
    }
    bool found = false;
    for (double ay : alphas) {
      bool feasible = true;
      std::vector<Vec> Yt = Y;
      for (int t = 0; t < N && feasible; ++t) {
        const Vec y_new = Y[t] + ay * k_y[t] + K_y[t] * dxs[t];
        for (auto &cd : cons) {
          for (int i = 0; i < cd.dual_dim; ++i) {
            const int j = cd.offset + i;
            if (y_new(j) < (1.0 - tau) * Y[t](j)) { feasible = false; break; }
          }
          if (!feasible) break;
          Yt[t].setSegment(cd.offset, seg(y_new, cd));
        }
        r.Lambda[t] = Lambda[t] + a * ms_kl[t] + ms_Kl[t] * dxs[t];
      }
      if (feasible) { found = true; r.Y = Yt; r.alpha_du = ay; break; }
    }
    if (!found) return r;
    for (int t = 0; t < N; ++t) {
      cost_new += running_cost(r.X[t], r.U[t], t);
      for (auto &cd : cons) {
        r.G[t].setSegment(cd.offset, con_g(cd, r.X[t], r.U[t]));
        double lsum = 0.0, l1 = 0.0;
        for (int i = 0; i < cd.dual_dim; ++i) { const int j = cd.offset + i; lsum += olog(r.S[t](j)); l1 += std::fabs(r.G[t](j) + r.S[t](j)); }
        merit_new -= mu * lsum; cv_new += l1;
      }
      cv_new += (r.F[t] - r.X[t + 1]).lpNorm1();
    }
    cost_new += terminal_cost(r.X.back());
    merit_new += cost_new;
    if (ms_filter_acceptable(merit_new, cv_new, a * dV[0])) {
      r.success = true; r.cost = cost_new; r.merit = merit_new; r.theta = cv_new; r.has_ip = true;
    }
    return r;
  }
  void msipddp_apply(const FPResult &r) {    // applyForwardPassResult :287-304
    X = r.X; U = r.U; cost = r.cost; merit = r.merit; alpha_pr = r.alpha_pr; alpha_du = r.alpha_du;
    if (r.has_ip) { Y = r.Y; S = r.S; G = r.G; }
    ms_F = r.F; Lambda = r.Lambda;
    acceptFilterEntry(r.merit, r.theta);
  }
  bool msipddp_checkConvergence(double dJ, int iter, int &st) {   // :306-364
    const double metric = std::max(std::max(ms_scaled_inf_du(), inf_pr), inf_comp);
    if (metric <= opt.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; return true; }
    if (std::fabs(dJ) < opt.acceptable_tolerance && iter > 10) {
      const double sq = std::sqrt(opt.acceptable_tolerance);
      if (inf_pr < sq && inf_comp < sq) { st = CDDP_HIP_STATUS_ACCEPTABLE; return true; }
    }
    if (iter >= 1 && step_norm < opt.tolerance * 10.0 && inf_pr < 1e-4) { st = CDDP_HIP_STATUS_ACCEPTABLE; return true; }
    return false;
  }
  bool msipddp_handleForwardPassFailure(int &st) {   // :371-398, checkAndPerformFilterRestoration :810-836
    bool needs = filter.size() > 5;
    if (!needs) for (auto &p : filter) if (!std::isfinite(p.merit_function) || !std::isfinite(p.constraint_violation)) { needs = true; break; }
    if (needs && !filter.empty()) { pruneFilterToBestPoints(); return false; }
    increaseRegularization();
    if (isRegularizationLimitReached()) { st = CDDP_HIP_STATUS_REG_LIMIT; return true; }
    return false;
  }
  void msipddp_update_barrier(bool fp_success) {   // updateBarrierParameters :1751-1850
    if (cons.empty()) return;
    if (opt.barrier_strategy == CDDP_HIP_BARRIER_MONOTONIC) {
      mu = std::max(opt.barrier_mu_min_value, opt.barrier_mu_update_factor * mu);
      ms_reset_filter();
    } else if (opt.barrier_strategy == CDDP_HIP_BARRIER_IPOPT) {
      const double err = std::max(std::max(ms_scaled_inf_du(), inf_pr), inf_comp);
      if (err <= 10.0 * mu) {
        const double lin = opt.barrier_mu_update_factor * mu, sup = opow(mu, opt.barrier_mu_update_power);
        mu = std::max(opt.tolerance / 10.0, std::min(lin, sup));
        ms_reset_filter();
      }
    } else {
      const double metric = std::max(std::max(ms_scaled_inf_du(), inf_pr), inf_comp);
      const double threshold = (mu < 1e-5) ? std::max(metric * 10.0, mu * 100.0) : std::max(opt.barrier_mu_update_factor * mu, mu * 2.0);
      const bool slow = fp_success && alpha_pr > 0 && (metric < 1e-3);
      if (metric <= threshold || slow) {
        double factor = opt.barrier_mu_update_factor;
        if (mu > 1e-12) {
          const double ratio = metric / mu;
          if (ratio < 0.01) factor = opt.barrier_mu_update_factor * 0.1;
          else if (ratio < 0.1) factor = opt.barrier_mu_update_factor * 0.3;
          else if (ratio < 0.5) factor = opt.barrier_mu_update_factor * 0.6;
        }
        const double lin = factor * mu, sup = opow(mu, opt.barrier_mu_update_power);
        if (slow && mu > opt.tolerance) mu = std::min(lin, sup);
        else mu = std::max(opt.tolerance / 100.0, std::min(lin, sup));
        ms_reset_filter();
      }
    }
  }

  // =============================================================== dispatch + main loop
  void initialize() {
    if (solver_kind == CDDP_HIP_SOLVER_CLDDP) clddp_initialize(); else if (solver_kind == CDDP_HIP_SOLVER_LOGDDP) logddp_initialize();
    else if (solver_kind == CDDP_HIP_SOLVER_MSIPDDP) msipddp_initialize(); else ipddp_initialize();
  }
  bool backwardPass() {
    return solver_kind == CDDP_HIP_SOLVER_CLDDP ? clddp_backward() : solver_kind == CDDP_HIP_SOLVER_LOGDDP ? logddp_backward()
         : solver_kind == CDDP_HIP_SOLVER_MSIPDDP ? msipddp_backward() : ipddp_backward();
  }
  FPResult forwardPass(double a) {
    ++n_forward;
    return solver_kind == CDDP_HIP_SOLVER_CLDDP ? clddp_forward(a) : solver_kind == CDDP_HIP_SOLVER_LOGDDP ? logddp_forward(a)
         : solver_kind == CDDP_HIP_SOLVER_MSIPDDP ? msipddp_forward(a) : ipddp_forward(a);
  }

  void recordHistory() {  // cddp_solver_base.cpp:220-232, ipddp_solver.cpp:2084-2088
    if (!opt.return_iteration_info) return;
    history.rows.push_back({cost, merit, alpha_pr, alpha_du, inf_du, inf_pr, inf_comp,
                            (solver_kind == CDDP_HIP_SOLVER_IPDDP || solver_kind == CDDP_HIP_SOLVER_MSIPDDP) ? mu : solver_kind == CDDP_HIP_SOLVER_LOGDDP ? lg_mu : 0.0, reg});
  }

  FPResult performForwardPass() {  // cddp_solver_base.cpp:248-317
    FPResult best; best.cost = best.merit = std::numeric_limits<double>::infinity(); best.success = false;
    // failing_alpha_mask() (test hook, 0 by default): the forward pass of alpha index i is evaluated and then DISCARDED, the way the
    // reference's parallel rule discards a forward pass that threw (cddp_solver_base.cpp:280-296; pinned by the reference's
    // ParallelForwardPassKeepsSuccessfulAlphaWhenAnotherThrows, tests/cddp_core/test_cddp_core.cpp:414-435) and the way a trial whose
    // costate is not finite fails (ipddp_solver.cpp:1613-1616).  The HIP library's counterpart is CDDP_HIP_TEST_FAIL_COSTATE.
    const unsigned fail = (unsigned)failing_alpha_mask();
    int idx = 0;
    if (!opt.enable_parallel) {
      for (double a : alphas) { FPResult r = forwardPass(a); if ((fail >> idx++) & 1u) continue; if (r.success) { best = r; break; } }
    } else {
      for (double a : alphas) { FPResult r = forwardPass(a); if ((fail >> idx++) & 1u) continue; if (r.success && r.merit < best.merit) best = r; }
    }
    return best;
  }

  void solve() {  // cddp_solver_base.cpp:29-186
    if (solver_kind == CDDP_HIP_SOLVER_LOGDDP) { lg_evaluate_trajectory(); lg_reset_filter(); }   // preIterationSetup (logddp_solver.cpp:211-214)
    recordHistory();
    int iter = 0; bool converged = false; int reason = CDDP_HIP_STATUS_MAX_ITERATIONS; double dJ = 0.0, dL = 0.0;
    const auto start_time = std::chrono::steady_clock::now();
    while (iter < opt.max_iterations) {
      ++iter;
      if (opt.max_cpu_time > 0) {  // cddp_solver_base.cpp:77-90
        const double el_ms = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start_time).count();
        if (el_ms > opt.max_cpu_time * 1000) { reason = CDDP_HIP_STATUS_MAX_CPU_TIME; break; }   // whole milliseconds, as the reference casts
      }
      bool backward_ok = false;
      while (!backward_ok) {
        backward_ok = backwardPass();
        if (!backward_ok) {
          increaseRegularization();
          if (isRegularizationLimitReached()) {
            if (solver_kind == CDDP_HIP_SOLVER_LOGDDP) { reason = CDDP_HIP_STATUS_REG_LIMIT_CONVERGED; converged = true; }   // logddp_solver.cpp:216-222
            else { reason = CDDP_HIP_STATUS_REG_LIMIT; converged = false; }
            break;
          }
        }
      }
      if (!backward_ok) break;
      bool early = false;
      if (solver_kind == CDDP_HIP_SOLVER_CLDDP) { if (inf_du < opt.tolerance) { reason = CDDP_HIP_STATUS_OPTIMAL; early = true; } }
      else if (solver_kind == CDDP_HIP_SOLVER_LOGDDP || solver_kind == CDDP_HIP_SOLVER_MSIPDDP) early = false;   // base default (cddp_solver_base.hpp)
      else early = ipddp_checkEarlyConvergence(reason);
      if (early) { converged = true; recordHistory(); break; }
      FPResult best = performForwardPass();
      bool fp_success = best.success;
      if (fp_success) {
        dJ = cost - best.cost; dL = merit - best.merit;
        if (solver_kind == CDDP_HIP_SOLVER_CLDDP || solver_kind == CDDP_HIP_SOLVER_LOGDDP) {  // cddp_solver_base.cpp:190-198
          X = best.X; U = best.U; cost = best.cost; merit = best.merit; alpha_pr = best.alpha_pr; alpha_du = best.alpha_du;
          if (solver_kind == CDDP_HIP_SOLVER_LOGDDP) lg_violation = best.inf_pr;   // logddp_solver.cpp:224-231
        } else if (solver_kind == CDDP_HIP_SOLVER_MSIPDDP) msipddp_apply(best);
        else ipddp_apply(best);
        recordHistory();
        decreaseRegularization();
        if (solver_kind == CDDP_HIP_SOLVER_CLDDP) {  // clddp_solver.cpp:264-277
          if (inf_du < opt.tolerance) { reason = CDDP_HIP_STATUS_OPTIMAL; converged = true; }
          else if (dJ > 0.0 && dJ < opt.acceptable_tolerance) { reason = CDDP_HIP_STATUS_ACCEPTABLE; converged = true; }
        } else if (solver_kind == CDDP_HIP_SOLVER_LOGDDP) {  // logddp_solver.cpp:233-261
          if (std::max(inf_du, inf_pr) <= opt.tolerance) { reason = CDDP_HIP_STATUS_OPTIMAL; converged = true; }
          else if (std::fabs(dJ) < opt.acceptable_tolerance && std::fabs(dL) < opt.acceptable_tolerance) { reason = CDDP_HIP_STATUS_ACCEPTABLE; converged = true; }
        } else if (solver_kind == CDDP_HIP_SOLVER_MSIPDDP) converged = msipddp_checkConvergence(dJ, iter, reason);
        else converged = ipddp_checkConvergence(dJ, iter, reason);
      } else {
        bool brk;
        if (solver_kind == CDDP_HIP_SOLVER_CLDDP || solver_kind == CDDP_HIP_SOLVER_LOGDDP) {  // cddp_solver_base.cpp:206-218
          increaseRegularization();
          brk = isRegularizationLimitReached();
          if (brk) reason = CDDP_HIP_STATUS_REG_LIMIT;
        } else if (solver_kind == CDDP_HIP_SOLVER_MSIPDDP) brk = msipddp_handleForwardPassFailure(reason);
        else brk = ipddp_handleForwardPassFailure(reason);
        if (brk) break;
      }
      if (converged) break;
      // postIterationUpdate: IPDDP's only acts on failure, and then returns immediately (:2027-2035, :2556-2559); LogDDP's updates mu
      if (solver_kind == CDDP_HIP_SOLVER_LOGDDP) logddp_post_iteration(fp_success);
      if (solver_kind == CDDP_HIP_SOLVER_MSIPDDP) msipddp_update_barrier(fp_success);   // postIterationUpdate (msipddp_solver.cpp:366-369)
    }
    iterations = iter; status = reason;
  }
};

static Solver *build(const cddp_hip_problem *p) {
  Solver *s = new Solver();
  s->nx = p->nx; s->nu = p->nu; s->N = p->horizon; s->dt = p->dt; s->solver_kind = p->solver; s->opt = p->options;
  s->model.id = p->model; s->model.nx = p->nx; s->model.nu = p->nu; s->model.integrator = p->integrator; s->model.dt = p->dt;
  std::memcpy(s->model.p, p->model_params, sizeof(s->model.p));
  if (p->model == CDDP_HIP_MODEL_LTI) { s->model.A = Mat::FromPtr(p->lti_A, p->nx, p->nx); s->model.B = Mat::FromPtr(p->lti_B, p->nx, p->nu); }
  s->Qdt = Mat::FromPtr(p->Q, p->nx, p->nx) * p->dt;   // objective.cpp:38-39
  s->Rdt = Mat::FromPtr(p->R, p->nu, p->nu) * p->dt;
  s->Qf = Mat::FromPtr(p->Qf, p->nx, p->nx);
  s->xref = Vec::FromPtr(p->x_ref, p->nx);
  if (p->x_ref_traj) for (int t = 0; t <= p->horizon; ++t) s->xref_traj.push_back(Vec::FromPtr(p->x_ref_traj + (size_t)t * p->nx, p->nx));
  for (int i = 0; i < p->n_constraints; ++i) {
    const cddp_hip_constraint &c = p->constraints[i];
    ConstraintDesc d; d.name = c.name; d.kind = c.kind; d.dim = c.dim; d.scale = c.scale; d.radius = c.radius;
    switch (c.kind) {
      case CDDP_HIP_CON_CONTROL_BOX: case CDDP_HIP_CON_STATE_BOX:
        d.dual_dim = 2 * c.dim; d.lower = Vec::FromPtr(c.lower, c.dim); d.upper = Vec::FromPtr(c.upper, c.dim);
        d.ip_upper = Vec(2 * c.dim, 1);
        for (int k = 0; k < c.dim; ++k) { d.ip_upper(k) = -d.lower(k) * d.scale; d.ip_upper(c.dim + k) = d.upper(k) * d.scale; }  // constraint.hpp:155-160
        break;
      case CDDP_HIP_CON_BALL:
        d.dual_dim = 1; d.center = Vec::FromPtr(c.center, c.dim); d.ip_upper = Vec(1, 1); d.ip_upper(0) = -(c.radius * c.radius) * c.scale; break;
      case CDDP_HIP_CON_LINEAR:
        d.dual_dim = c.dim; d.A = Mat::FromPtr(c.A, c.dim, p->nx); d.b = Vec::FromPtr(c.b, c.dim); break;
      case CDDP_HIP_CON_SOC:      // center = cone origin, lower = unit opening direction, radius = cos(fov), scale = eps
        d.dual_dim = 1; d.center = Vec::FromPtr(c.center, 3); d.lower = Vec::FromPtr(c.lower, 3); break;
      case CDDP_HIP_CON_THRUST:   // lower[0] = min, radius = max, scale = eps
        d.dual_dim = 2; d.lower = Vec::FromPtr(c.lower, 1); break;
      case CDDP_HIP_CON_MAX_THRUST:
        d.dual_dim = 1; break;
    }
    s->cons.push_back(d);
  }
  std::stable_sort(s->cons.begin(), s->cons.end(), [](const ConstraintDesc &a, const ConstraintDesc &b) { return a.name < b.name; });
  int off = 0; for (auto &c : s->cons) { c.offset = off; off += c.dual_dim; } s->m = off;
  for (int i = 0; i < p->n_terminal; ++i) {
    const cddp_hip_terminal_constraint &c = p->terminal[i];
    TerminalDesc d; d.name = c.name; d.kind = c.kind; d.dim = c.dim;
    if (c.kind == CDDP_HIP_TERM_EQUALITY) d.target = Vec::FromPtr(c.target, c.dim);
    else { d.A = Mat::FromPtr(c.A, c.dim, p->nx); d.b = Vec::FromPtr(c.b, c.dim); }
    s->terms.push_back(d);
  }
  std::stable_sort(s->terms.begin(), s->terms.end(), [](const TerminalDesc &a, const TerminalDesc &b) { return a.name < b.name; });
  // buildLineSearchAlphas (cddp_context_utils.cpp:37-57)
  {
    double cur = p->options.ls_initial_step_size;
    for (int i = 0; i < p->options.ls_max_iterations; ++i) {
      s->alphas.push_back(cur);
      cur *= p->options.ls_step_reduction_factor;
      if (cur < p->options.ls_min_step_size && i < p->options.ls_max_iterations - 1) { s->alphas.push_back(p->options.ls_min_step_size); break; }
    }
    if (s->alphas.empty()) s->alphas.push_back(p->options.ls_initial_step_size);
  }
  return s;
}

static void fill_result(const Solver *s, cddp_hip_result *r) {
  r->final_objective = s->cost; r->merit_function = s->merit; r->inf_pr = s->inf_pr; r->inf_du = s->inf_du;
  r->inf_comp = s->inf_comp; r->barrier_mu = (s->solver_kind == CDDP_HIP_SOLVER_IPDDP || s->solver_kind == CDDP_HIP_SOLVER_MSIPDDP) ? s->mu : s->solver_kind == CDDP_HIP_SOLVER_LOGDDP ? s->lg_mu : 0.0;
  r->regularization = s->reg; r->alpha_pr = s->alpha_pr; r->alpha_du = s->alpha_du; r->step_norm = s->step_norm;
  r->iterations = s->iterations; r->status = s->status; r->n_backward = s->n_backward; r->n_forward = s->n_forward;
}

}  // namespace oracle

using oracle::Solver;
using oracle::Mat;
using oracle::Vec;

extern "C" {

void *cddp_oracle_create(const cddp_hip_problem *p) { return oracle::build(p); }
void cddp_oracle_destroy(void *o) { delete (Solver *)o; }
int cddp_oracle_dual_dim(void *o) { return ((Solver *)o)->m; }
int cddp_oracle_num_alphas(void *o, double *out, int cap) {
  Solver *s = (Solver *)o; int n = (int)s->alphas.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = s->alphas[i];
  return n;
}
int cddp_oracle_set_initial(void *o, const double *x0, const double *U0, const double *X0) { ((Solver *)o)->set_initial(x0, U0, X0); return 0; }
int cddp_oracle_initialize(void *o) { ((Solver *)o)->initialize(); return 0; }
// ---- warm-start plumbing (tests): options.warm_start, IPDDPSolverTestAccess-style setters, MPC-style x0 / U update
void cddp_oracle_set_warm_start(void *o, int flag) { ((Solver *)o)->opt.warm_start = flag; }
void cddp_oracle_set_path_interior(void *o, double s_val, double y_val) {
  Solver *s = (Solver *)o;
  for (auto &v : s->S) for (int i = 0; i < v.size(); ++i) v(i) = s_val;
  for (auto &v : s->Y) for (int i = 0; i < v.size(); ++i) v(i) = y_val;
}
void cddp_oracle_set_terminal_interior(void *o, double s_val, double y_val) {
  Solver *s = (Solver *)o;
  for (auto &kv : s->S_T) for (int i = 0; i < kv.second.size(); ++i) kv.second(i) = s_val;
  for (auto &kv : s->Y_T) for (int i = 0; i < kv.second.size(); ++i) kv.second(i) = y_val;
}
void cddp_oracle_set_terminal_eq_multiplier(void *o, const double *lam) {
  Solver *s = (Solver *)o;
  for (int i = 0; i < s->Lambda_T_eq.size(); ++i) s->Lambda_T_eq(i) = lam[i];
}
// CDDP::setInitialState (+ setInitialTrajectory when U0 != NULL) on a LIVE context: nothing else is reset
void cddp_oracle_update_initial(void *o, const double *x0, const double *U0) {
  Solver *s = (Solver *)o;
  s->x0 = Vec::FromPtr(x0, s->nx);
  if (U0) for (int t = 0; t < s->N; ++t) s->U[t] = Vec::FromPtr(U0 + (size_t)t * s->nu, s->nu);
  if (!s->X.empty()) s->X[0] = s->x0;
}
// one backwardPass; retry != 0 adds the regularisation-retry loop of cddp_solver_base.cpp:93-111
int cddp_oracle_backward(void *o, int retry) {
  Solver *s = (Solver *)o;
  bool ok = false;
  while (!ok) {
    ok = s->backwardPass();
    if (ok || !retry) break;
    s->increaseRegularization();
    if (s->isRegularizationLimitReached()) break;
  }
  return ok ? 1 : 0;
}
int cddp_oracle_forward(void *o, double alpha, cddp_hip_trial *out) {
  Solver *s = (Solver *)o;
  oracle::FPResult r = s->forwardPass(alpha);
  out->alpha = alpha; out->alpha_pr = r.alpha_pr; out->alpha_du = r.alpha_du; out->cost = r.cost; out->merit_function = r.merit;
  out->theta = r.theta; out->inf_pr = r.inf_pr; out->inf_comp = r.inf_comp; out->success = r.success ? 1 : 0; out->_pad = 0;
  return 0;
}
int cddp_oracle_solve(void *o, cddp_hip_result *res) {
  Solver *s = (Solver *)o;
  s->n_backward = s->n_forward = 0;   // per-solve work counters
  s->history.rows.clear();            // cddp_solver_base.cpp:47-50
  s->initialize();
  s->solve();
  if (res) oracle::fill_result(s, res);
  return 0;
}
int cddp_oracle_get_result(void *o, cddp_hip_result *res) { oracle::fill_result((Solver *)o, res); return 0; }
int cddp_oracle_get_trajectory(void *o, double *X, double *U) {
  Solver *s = (Solver *)o;
  if (X) for (int t = 0; t <= s->N; ++t) for (int i = 0; i < s->nx; ++i) X[(size_t)t * s->nx + i] = s->X[t](i);
  if (U) for (int t = 0; t < s->N; ++t) for (int i = 0; i < s->nu; ++i) U[(size_t)t * s->nu + i] = s->U[t](i);
  return 0;
}
int cddp_oracle_get_gains(void *o, double *K, double *k) {
  Solver *s = (Solver *)o;
  for (int t = 0; t < s->N; ++t) {
    if (K) for (int i = 0; i < s->nu * s->nx; ++i) K[(size_t)t * s->nu * s->nx + i] = s->K_u[t].a[i];
    if (k) for (int i = 0; i < s->nu; ++i) k[(size_t)t * s->nu + i] = s->k_u[t](i);
  }
  return 0;
}
int cddp_oracle_get_value(void *o, double *Vx, double *Vxx) {
  Solver *s = (Solver *)o;
  for (int t = 0; t <= s->N; ++t) {
    if (Vx) for (int i = 0; i < s->nx; ++i) Vx[(size_t)t * s->nx + i] = s->Vx_t[t](i);
    if (Vxx) for (int i = 0; i < s->nx * s->nx; ++i) Vxx[(size_t)t * s->nx * s->nx + i] = s->Vxx_t[t].a[i];
  }
  return 0;
}
int cddp_oracle_get_duals(void *o, double *S, double *Y, double *G) {
  Solver *s = (Solver *)o;
  for (int t = 0; t < s->N; ++t) for (int i = 0; i < s->m; ++i) {
    if (S) S[(size_t)t * s->m + i] = s->S[t](i);
    if (Y) Y[(size_t)t * s->m + i] = s->Y[t](i);
    if (G) G[(size_t)t * s->m + i] = s->G[t](i);
  }
  return 0;
}
// costate trajectory of the current iterate (IPDDP: N + 1 rows, MSIPDDP: N rows); returns the row count, Lam may be null
int cddp_oracle_get_costates(void *o, double *Lam) {
  Solver *s = (Solver *)o;
  const int rows = (int)s->Lambda.size();
  if (Lam)
    for (int t = 0; t < rows; ++t)
      for (int i = 0; i < s->nx; ++i) Lam[(size_t)t * s->nx + i] = (s->Lambda[t].size() == s->nx) ? s->Lambda[t](i) : 0.0;
  return rows;
}
// stacked terminal state in std::map order: inequality (S_T, Y_T, G_T) and equality multipliers
int cddp_oracle_get_terminal(void *o, double *ST, double *YT, double *GT, double *LamT, int *dims) {
  Solver *s = (Solver *)o; int mT = 0;
  for (auto &td : s->terms) if (td.kind == CDDP_HIP_TERM_INEQUALITY) {
    for (int i = 0; i < td.dim; ++i) {
      if (ST) ST[mT + i] = s->S_T.at(td.name)(i);
      if (YT) YT[mT + i] = s->Y_T.at(td.name)(i);
      if (GT) GT[mT + i] = s->G_T.at(td.name)(i);
    }
    mT += td.dim;
  }
  int pT = s->term_eq_dim();
  if (LamT) for (int i = 0; i < pT; ++i) LamT[i] = s->Lambda_T_eq(i);
  if (dims) { dims[0] = mT; dims[1] = pT; }
  return 0;
}
int cddp_oracle_get_backward_scalars(void *o, double *dV, double *reg) {
  Solver *s = (Solver *)o; if (dV) { dV[0] = s->dV[0]; dV[1] = s->dV[1]; } if (reg) *reg = s->reg; return 0;
}
int cddp_oracle_get_history(void *o, double *hist, int cap_rows) {
  Solver *s = (Solver *)o; int n = (int)s->history.rows.size();
  for (int i = 0; i < n && i < cap_rows; ++i) for (int j = 0; j < 9; ++j) hist[i * 9 + j] = s->history.rows[i][j];
  return n;
}
// filter / barrier inspection (replays IPDDPSolverTestAccess of tests/cddp_core/test_ipddp_solver.cpp:30-135)
int cddp_oracle_filter_size(void *o) { return (int)((Solver *)o)->filter.size(); }
double cddp_oracle_filter_theta(void *o) { return ((Solver *)o)->filter_theta; }
double cddp_oracle_filter_back_violation(void *o) { Solver *s = (Solver *)o; return s->filter.empty() ? -1.0 : s->filter.back().constraint_violation; }
void cddp_oracle_update_barrier(void *o, int fp_success) { ((Solver *)o)->updateBarrierParameters(fp_success != 0); }
double cddp_oracle_scaled_inf_du(void *o) { return ((Solver *)o)->computeScaledDualInfeasibility(); }
double cddp_oracle_get_mu(void *o) { return ((Solver *)o)->mu; }
// libm-noise knob of models.hpp (process-wide): 0 = off (default), 1 = sin / cos results moved by -1 / 0 / +1 ulp
void cddp_oracle_set_trig_noise(int v) { oracle::trig_noise() = v; }
void cddp_oracle_set_failing_alphas(int mask) { oracle::failing_alpha_mask() = mask; }   // test hook, see Solver::performForwardPass
void cddp_oracle_set_trig_mode(int v) { oracle::trig_mode() = v; }   // 0 = glibc, 1 = the HIP parity build's routine (models.hpp)
// summation-order noise knob of linalg.hpp (process-wide): 0 = off (default), 1 = matrix-product entries moved by <= 1 ulp
void cddp_oracle_set_matmul_noise(int v) { oracle::matmul_noise() = v; }
// summation-order model of the reference's Eigen build (linalg.hpp::assoc_mode): 0 = serial (default), 1 = Eigen 3.4 SSE2 packet order
void cddp_oracle_set_assoc_mode(int v) { oracle::assoc_mode() = v; }
// the three summation orders of linalg.hpp on a caller's terms (unit tests): kind 0 serial, 1 redux (dot / norm / sum), 2 row-major gemv
double cddp_oracle_sum_order(int kind, const double *p, int n) { return kind == 1 ? oracle::redux_order(p, n) : (kind == 2 ? oracle::gemv_row_order(p, n) : oracle::serial_order(p, n)); }
void cddp_oracle_set_inf_du(void *o, double v) { ((Solver *)o)->inf_du = v; }
void cddp_oracle_set_check_state_stationarity(void *o, int v) { ((Solver *)o)->opt.ipddp_check_state_stationarity = v; }

// plugin-surface probes ----------------------------------------------------------------------
int cddp_oracle_dynamics(void *o, const double *x, const double *u, double time, double *xdot, double *xnext, double *Fx, double *Fu) {
  Solver *s = (Solver *)o;
  Vec xv = Vec::FromPtr(x, s->nx), uv = Vec::FromPtr(u, s->nu);
  if (xdot) s->model.f(x, u, time, xdot);
  if (xnext) { Vec xn = s->model.step(xv, uv, time); for (int i = 0; i < s->nx; ++i) xnext[i] = xn(i); }
  if (Fx || Fu) {
    Mat A, B; s->model.jacobians(xv, uv, time, A, B);
    if (Fx) for (int i = 0; i < s->nx * s->nx; ++i) Fx[i] = A.a[i];
    if (Fu) for (int i = 0; i < s->nx * s->nu; ++i) Fu[i] = B.a[i];
  }
  return 0;
}
// continuous-time Hessian tensors f_xx (nx*nx*nx), f_uu (nx*nu*nu), f_ux (nx*nu*nx); returns 0 for plants without them
int cddp_oracle_hessians(void *o, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) {
  Solver *s = (Solver *)o;
  std::vector<Mat> a, b, c;
  if (!s->model.hessians(Vec::FromPtr(x, s->nx), Vec::FromPtr(u, s->nu), 0.0, a, b, c)) return 0;
  const int nx = s->nx, nu = s->nu;
  for (int i = 0; i < nx; ++i) {
    for (int e = 0; e < nx * nx; ++e) Fxx[i * nx * nx + e] = a[i].a[e];
    for (int e = 0; e < nu * nu; ++e) Fuu[i * nu * nu + e] = b[i].a[e];
    for (int e = 0; e < nu * nx; ++e) Fux[i * nu * nx + e] = c[i].a[e];
  }
  return 1;
}
int cddp_oracle_constraint_eval(void *o, const double *x, const double *u, double *g, double *gx, double *gu) {
  Solver *s = (Solver *)o;
  Vec xv = Vec::FromPtr(x, s->nx), uv = Vec::FromPtr(u, s->nu);
  for (auto &c : s->cons) {
    Vec gv = s->con_g(c, xv, uv); Mat jx, ju; s->con_jac(c, xv, uv, jx, ju);
    for (int i = 0; i < c.dual_dim; ++i) {
      if (g) g[c.offset + i] = gv(i);
      if (gx) for (int j = 0; j < s->nx; ++j) gx[(c.offset + i) * s->nx + j] = jx(i, j);
      if (gu) for (int j = 0; j < s->nu; ++j) gu[(c.offset + i) * s->nu + j] = ju(i, j);
    }
  }
  return 0;
}
double cddp_oracle_cost(void *o, const double *X, const double *U) {
  Solver *s = (Solver *)o;
  std::vector<Vec> Xs, Us;
  for (int t = 0; t <= s->N; ++t) Xs.push_back(Vec::FromPtr(X + (size_t)t * s->nx, s->nx));
  for (int t = 0; t < s->N; ++t) Us.push_back(Vec::FromPtr(U + (size_t)t * s->nu, s->nu));
  return s->objective_evaluate(Xs, Us);
}
// BoxQP probe: returns status; x (n), free (n), iterations
int cddp_oracle_boxqp(const cddp_hip_options *opt, int n, const double *H, const double *g, const double *lo, const double *up,
                      const double *x0, double *x, int *free_, int *iterations, int *factorizations) {
  oracle::BoxQPResult r = oracle::boxqp_solve(*opt, Mat::FromPtr(H, n, n), Vec::FromPtr(g, n), Vec::FromPtr(lo, n), Vec::FromPtr(up, n),
                                             x0 ? Vec::FromPtr(x0, n) : Vec::Zero(0));
  for (int i = 0; i < n; ++i) { x[i] = r.x(i); if (free_) free_[i] = r.free_[i]; }
  if (iterations) *iterations = r.iterations;
  if (factorizations) *factorizations = r.factorizations;
  return r.status;
}
// LDLT probe: solve A X = B, returns info()==Success
int cddp_oracle_ldlt_solve(int n, int nrhs, const double *A, const double *B, double *X) {
  oracle::LDLT f(Mat::FromPtr(A, n, n));
  Mat Xm = f.solve(Mat::FromPtr(B, n, nrhs));
  for (int i = 0; i < n * nrhs; ++i) X[i] = Xm.a[i];
  return f.ok ? 1 : 0;
}

// Whole-batch solve on host threads: the CPU baseline ("port") and the parity reference for
// cddp_hip_solve.  x0: B*nx, U0: B*N*nu or NULL, X0: B*(N+1)*nx or NULL.
// Outputs (any may be NULL): results[B], X[B][(N+1)][nx], U[B][N][nu], K[B][N][nu][nx].
// Returns wall-clock milliseconds through *elapsed_ms.
int cddp_oracle_solve_batch(const cddp_hip_problem *p, int batch, const double *x0, const double *U0, const double *X0,
                            int n_threads, cddp_hip_result *results, double *X, double *U, double *K, double *elapsed_ms) {
  if (n_threads < 1) n_threads = 1;
  const int nx = p->nx, nu = p->nu, N = p->horizon;
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    Solver *s = oracle::build(p);
    for (;;) {
      int b = next.fetch_add(1);
      if (b >= batch) break;
      // MSIPDDP keeps state that belongs to the solver OBJECT across initialize() calls (the per-step factor cache of
      // msipddp_solver.cpp:1169-1185, and gains that send a warm start down the "existing state" branch): independent trajectories
      // of a batch are independent solver objects, as each trajectory of a device handle owns its cache
      if (p->solver == CDDP_HIP_SOLVER_MSIPDDP) { delete s; s = oracle::build(p); }
      s->set_initial(x0 + (size_t)b * nx, U0 ? U0 + (size_t)b * N * nu : nullptr, X0 ? X0 + (size_t)b * (N + 1) * nx : nullptr);
      s->initialize();
      s->solve();
      if (results) oracle::fill_result(s, &results[b]);
      if (X || U) cddp_oracle_get_trajectory(s, X ? X + (size_t)b * (N + 1) * nx : nullptr, U ? U + (size_t)b * N * nu : nullptr);
      if (K) cddp_oracle_get_gains(s, K + (size_t)b * N * nu * nx, nullptr);
    }
    delete s;
  };
  std::vector<std::thread> th;
  for (int i = 0; i < n_threads - 1; ++i) th.emplace_back(worker);
  worker();
  for (auto &t : th) t.join();
  auto t1 = std::chrono::steady_clock::now();
  if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  return 0;
}

}  // extern "C"
