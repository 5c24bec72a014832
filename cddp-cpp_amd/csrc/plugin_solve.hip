// cddp_hip_plugin_solve: CDDP::solve() for HOST plug-ins -- arbitrary DynamicalSystem / Objective / Constraint subclasses
// that cannot run on the GPU (the north_star's "keeps the DynamicsModel / Constraint / Objective plugin surface").
//
// Division of labour (SURVEY.md 7 hard parts (ii) / (iii)): the caller's plug-ins are evaluated on
// the host through the callbacks of cddp_hip_plugin; the (N x batch) derivative stacks of every iterate go to the GPU, which
// runs the backward pass of the whole batch in one launch (stack-fed sweeps, stacks.hip); the line-searched forward pass needs
// the plug-in's f(x, u) at every step of every trial and therefore runs on the host, as does the outer loop.  The host side
// below restates, per trajectory, exactly what the device-resident state machine of the built-in plants does:
//
//   outer loop, regularisation schedule, line search        cddp_solver_base.cpp:29-186, 248-317; cddp_core.cpp:308-346
//   IPDDP  initialize / forwardPass / applyForwardPassResult / updateBarrierParameters / filter / convergence
//                                                           ipddp_solver.cpp:819-913, 925-958, 1571-1876, 1878-2082, 2428-2519,
//                                                           2548-2660, 2778-2937; interior_point_utils.cpp:79-139
//   CLDDP  initialize / forwardPass / convergence           clddp_solver.cpp:28-75, 206-277
//
// (kernels.hpp holds the same logic as device code: k_init, k_forward_ipddp, k_forward_clddp, k_update.)  Round 6: terminal constraints
// (cddp_hip_plugin_solve_terminal, ipddp_terminal_solve below) and the "provided trajectory" warm start.
// This file uses only the public C-ABI of include/cddp_hip.h for the GPU part.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/cddp_hip.h"

extern "C" int cddp_hip_internal_set_error(int code, const char *msg);   // capi.hip (thread-local last-error string)

namespace {

int pfail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  return cddp_hip_internal_set_error(code, buf);
}

constexpr double kEpsSlack = 1e-10, kSlackOffset = 1e-4, kMaxRatio = 1e6;   // ipddp_solver.cpp:35-38
const double kInf = std::numeric_limits<double>::infinity();
inline bool fin(double v) { return std::fabs(v) <= std::numeric_limits<double>::max(); }
inline double clampd(double v, double lo, double hi) { return std::min(std::max(v, lo), hi); }

// Host threads of the per-trajectory work (derivative / stack fill, forward passes, updates): the trajectories of a batch are independent
// problems, and the reference itself fans its line search out with std::async (cddp_solver_base.cpp:264-314).  1 (the default) keeps every
// callback on the calling thread, as documented since round 3; cddp_hip_plugin_set_host_threads(n) / CDDP_HIP_PLUGIN_THREADS=n (0 = one per
// hardware thread) let n threads call the plug-in's callbacks CONCURRENTLY -- for thread-safe (C / C++) plug-ins only; the Python facade
// keeps 1 (its callbacks serialise on the interpreter lock anyway).  Results do not depend on the count: every trajectory's arithmetic is
// its own (tests/test_host_plugins.py::test_plugin_host_threads_do_not_change_results).
int g_host_threads = -1;   // -1: unset (environment, then 1)
// Time split of the LAST IPDDP / CLDDP plug-in solve of this process (cddp_hip_plugin_last_stats; bench.py's plug-in line): wall time of the
// whole call, of the GPU sections (upload of the stacks, the sweep launch, download of the gains -- cddp_hip_set_stacks .. cddp_hip_stacks_get_*),
// the sweeps' kernel time (hipEvents, cddp_hip_stacks_last_kernel_ms) and the batch sweeps run
struct PluginStats { double total_ms = 0, gpu_section_ms = 0, kernel_ms = 0; int sweeps = 0, threads = 1; };
PluginStats g_last_stats;
struct StatClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
int host_threads(int batch) {
  int n = g_host_threads;
  if (n < 0) { const char *e = std::getenv("CDDP_HIP_PLUGIN_THREADS"); n = e ? std::atoi(e) : 1; }
  if (n == 0) n = (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(n, batch));
}
template <class F> void par_for(size_t n, int threads, F &&f) {
  if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
  std::atomic<size_t> next{0};
  auto work = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) return; f(i); } };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; ++t) th.emplace_back(work);
  work();
  for (auto &x : th) x.join();
}

struct Trial {   // ForwardPassResult (cddp_core.hpp:105-145)
  bool success = false;
  double alpha_pr = 1.0, alpha_du = 1.0, cost = 0, merit = 0, theta = 0, inf_pr = 0, inf_comp = 0;
  std::vector<double> X, U, S, Y, G, Lam;
};

struct Traj {
  std::vector<double> X, U, S, Y, G, Lam;
  double cost = 0, merit = 0, inf_pr = 0, inf_du = 0, inf_comp = 0, step_norm = 0, alpha_pr = 1.0, alpha_du = 1.0, reg = 0, mu = 0;
  double dV0 = 0, dV1 = 0, phi = 0, theta = 0, filter_theta = 0, apr_max = 1.0, adu_max = 1.0;
  std::vector<std::pair<double, double>> filter;   // (merit, violation)
  int iter = 0, status = CDDP_HIP_STATUS_RUNNING, n_bwd = 0, n_fwd = 0;
  bool done = false;
};

struct Ctx {
  const cddp_hip_plugin *pl;
  const cddp_hip_options *o;
  int solver, nx, nu, m, N;
  double dt;
  std::vector<double> alphas;
  bool ipddp() const { return solver == CDDP_HIP_SOLVER_IPDDP; }
};

// regularisation schedule (cddp_core.cpp:308-346); from exactly 0 the step restarts at reg_min_value (kernels.hpp::reg_increase)
double reg_increase(const cddp_hip_options &o, double r) {
  r *= o.reg_update_factor;
  if (!(r > 0.0)) r = (o.reg_min_value > 0.0) ? o.reg_min_value : o.reg_max_value;
  return std::min(r, o.reg_max_value);
}
double reg_decrease(const cddp_hip_options &o, double r) { r /= o.reg_update_factor; return std::max(r, o.reg_min_value); }

// computeTheta / computeBarrierMerit / computePrimalAndComplementarity, constraint-major then t (ipddp_solver.cpp:2778-2937)
void ip_reductions(const Ctx &c, const double *S, const double *Y, const double *G, double mu, double cost0,
                   double &phi, double &theta, double &inf_pr, double &inf_comp) {
  const int m = c.m, N = c.N;
  const bool l2 = c.o->ipddp_theta_norm_l2 != 0;
  double total = 0.0, max_entry = 0.0, ipr = 0.0, icomp = 0.0, mer = cost0;
  int off = 0;
  for (int s = 0; s < c.pl->n_constraints; ++s) {
    const int dim = c.pl->constraint_dims[s];
    for (int t = 0; t < N; ++t) {
      double n1 = 0.0, ninf = 0.0;
      for (int i = 0; i < dim; ++i) {
        const size_t j = (size_t)t * m + off + i;
        const double r = G[j] + S[j];
        n1 += l2 ? r * r : std::fabs(r);
        ninf = std::max(ninf, std::fabs(r));
        icomp = std::max(icomp, std::fabs(Y[j] * S[j] - mu));
      }
      total += n1; max_entry = std::max(max_entry, ninf); ipr = std::max(ipr, ninf);
    }
    off += dim;
  }
  off = 0;
  for (int s = 0; s < c.pl->n_constraints; ++s) {
    const int dim = c.pl->constraint_dims[s];
    for (int t = 0; t < N; ++t) {
      double ls = 0.0;
      for (int i = 0; i < dim; ++i) ls += std::log(std::max(S[(size_t)t * m + off + i], kEpsSlack));
      mer -= mu * ls;
    }
    off += dim;
  }
  const double th = l2 ? std::sqrt(total) : total;
  theta = std::max(th, max_entry);
  phi = mer; inf_pr = ipr; inf_comp = icomp;
}

// filter helpers (interior_point_utils.cpp:79-139)
void filter_accept(std::vector<std::pair<double, double>> &f, double mf, double cv) {
  for (auto &p : f) if (p.first <= mf && p.second <= cv) return;   // dominated by an existing point
  std::vector<std::pair<double, double>> keep;
  for (auto &p : f) if (!(mf <= p.first && cv <= p.second)) keep.push_back(p);
  keep.emplace_back(mf, cv);
  f.swap(keep);
}
void filter_prune(std::vector<std::pair<double, double>> &f) {
  if (f.empty()) return;
  auto bv = f[0], bm = f[0];
  for (size_t i = 1; i < f.size(); ++i) { if (f[i].second < bv.second) bv = f[i]; if (f[i].first < bm.first) bm = f[i]; }
  std::vector<std::pair<double, double>> out{bv};
  if (std::fabs(bm.second - bv.second) > 1e-12 || std::fabs(bm.first - bv.first) > 1e-12) out.push_back(bm);
  f.swap(out);
}

double total_cost(const Ctx &c, const double *X, const double *U) {   // CDDPSolverBase::computeCost (:416-424)
  double J = 0.0;
  for (int t = 0; t < c.N; ++t) J += c.pl->running_cost(c.pl->user, X + (size_t)t * c.nx, U + (size_t)t * c.nu, t);
  J += c.pl->terminal_cost(c.pl->user, X + (size_t)c.N * c.nx);
  return J;
}

void repair_interior(const cddp_hip_options &o, double *s, double *y, int dim);   // (defined with the terminal-constraint code below)
// ---- ISolverAlgorithm::initialize ------------------------------------------------------------------------------------
void initialize(const Ctx &c, Traj &T, const double *x0, const double *U0, const double *X0) {
  const int nx = c.nx, nu = c.nu, N = c.N, m = c.m;
  const cddp_hip_options &o = *c.o;
  T.X.assign((size_t)(N + 1) * nx, 0.0); T.U.assign((size_t)N * nu, 0.0);
  if (U0) std::copy(U0, U0 + (size_t)N * nu, T.U.begin());
  if (X0) std::copy(X0, X0 + (size_t)(N + 1) * nx, T.X.begin());
  else for (int t = 0; t <= N; ++t) std::copy(x0, x0 + nx, T.X.begin() + (size_t)t * nx);
  std::copy(x0, x0 + nx, T.X.begin());   // X_[0] = initial_state (cddp_core.cpp:294)
  T.reg = o.reg_initial_value; T.iter = 0; T.status = CDDP_HIP_STATUS_RUNNING; T.done = false; T.n_bwd = T.n_fwd = 0;
  T.dV0 = T.dV1 = 0.0; T.step_norm = 0.0; T.filter.clear();
  if (!c.ipddp()) {   // clddp_solver.cpp:62-74: cost of the GIVEN (X, U)
    T.cost = total_cost(c, T.X.data(), T.U.data()); T.merit = T.cost;
    T.inf_pr = T.inf_du = T.inf_comp = kInf; T.alpha_pr = o.ls_initial_step_size; T.alpha_du = 0.0; T.mu = 0.0;
    return;
  }
  // IPDDP cold start (ipddp_solver.cpp:819-913): re-rollout X from U, mu, g, slack / dual initialisation, cost, filter reset; with
  // options.warm_start (round 6) the "warm start with provided trajectory" branch (:733-816): a stateless call has no existing solver
  // state, so the barrier parameter follows the seed's largest constraint value and the duals are initialised from it
  T.mu = (m == 0) ? std::max(o.tolerance / 10.0, o.barrier_mu_min_value) : o.barrier_mu_initial;
  T.alpha_pr = T.alpha_du = 1.0;
  T.S.assign((size_t)N * m, 0.0); T.Y = T.S; T.G = T.S; T.Lam.assign((size_t)(N + 1) * nx, 0.0);
  double cost = 0.0;
  std::vector<double> xn(nx);
  for (int t = 0; t < N; ++t) {
    double *x = T.X.data() + (size_t)t * nx, *u = T.U.data() + (size_t)t * nu;
    cost += c.pl->running_cost(c.pl->user, x, u, t);
    if (m > 0) c.pl->constraints(c.pl->user, x, u, t, T.G.data() + (size_t)t * m, nullptr, nullptr);
    c.pl->discrete_dynamics(c.pl->user, x, u, t * c.dt, xn.data());
    std::copy(xn.begin(), xn.end(), T.X.begin() + (size_t)(t + 1) * nx);
  }
  cost += c.pl->terminal_cost(c.pl->user, T.X.data() + (size_t)N * nx);
  T.cost = cost;
  if (o.warm_start && m > 0) {   // :779-799
    double mv = 0.0;
    for (size_t j = 0; j < T.G.size(); ++j) mv = std::max(mv, T.G[j]);
    if (mv <= o.tolerance) T.mu = std::max(o.tolerance, o.barrier_mu_min_value);
    else if (mv <= 0.1) T.mu = std::max(o.tolerance * 10.0, o.barrier_mu_initial * 0.01);
    else T.mu = o.barrier_mu_initial * 0.1;
  }
  {   // initializeDualSlackVariables (:2456-2468) / initializeDualSlackVariablesWarmStart without existing duals (:2345-2426), object by object
    int off = 0;
    for (int sidx = 0; sidx < (m > 0 ? c.pl->n_constraints : 0); ++sidx) {
      const int dim = c.pl->constraint_dims[sidx];
      for (int t = 0; t < N; ++t) {
        double *g = T.G.data() + (size_t)t * m + off, *s = T.S.data() + (size_t)t * m + off, *y = T.Y.data() + (size_t)t * m + off;
        for (int i = 0; i < dim; ++i) {
          s[i] = std::max(o.ipddp_slack_var_init_scale, -g[i] + kSlackOffset);
          y[i] = (T.mu * o.ipddp_dual_var_init_scale) / std::max(s[i], kEpsSlack);
        }
        repair_interior(o, s, y, dim);
      }
      off += dim;
    }
  }
  double phi = cost, theta = 0.0, ipr = 0.0, icomp = 0.0;   // resetFilter (:2484-2519)
  if (m > 0) ip_reductions(c, T.S.data(), T.Y.data(), T.G.data(), T.mu, cost, phi, theta, ipr, icomp);
  T.merit = T.phi = phi; T.inf_pr = ipr; T.inf_comp = icomp; T.inf_du = 0.0;
  T.filter_theta = std::max(theta, 1e-8);
  T.theta = std::max(T.filter_theta, std::max(o.ipddp_theta_0_floor, 1e-8));
}

// ---- forwardPass ------------------------------------------------------------------------------------------------------
struct Gains {   // results of the GPU sweep for one trajectory (batch-major slices)
  const double *K, *k, *Vx, *Vxx, *ky, *Ky, *ks, *Ks;
};

Trial forward_clddp(const Ctx &c, const Traj &T, const Gains &g, double alpha) {   // clddp_solver.cpp:215-262
  const int nx = c.nx, nu = c.nu, N = c.N;
  Trial r; r.alpha_pr = alpha; r.alpha_du = 1.0;
  r.X.assign(T.X.size(), 0.0); r.U.assign(T.U.size(), 0.0);
  std::copy(T.X.begin(), T.X.begin() + nx, r.X.begin());
  std::vector<double> dx(nx);
  double J = 0.0;
  for (int t = 0; t < N; ++t) {
    const double *x = r.X.data() + (size_t)t * nx, *xo = T.X.data() + (size_t)t * nx, *uo = T.U.data() + (size_t)t * nu;
    double *u = r.U.data() + (size_t)t * nu;
    for (int i = 0; i < nx; ++i) dx[i] = x[i] - xo[i];
    for (int i = 0; i < nu; ++i) {
      double s = 0.0;
      for (int j = 0; j < nx; ++j) s += g.K[((size_t)t * nu + i) * nx + j] * dx[j];
      u[i] = (uo[i] + alpha * g.k[(size_t)t * nu + i]) + s;
      if (c.pl->control_lower && c.pl->control_upper) u[i] = std::min(std::max(u[i], c.pl->control_lower[i]), c.pl->control_upper[i]);
    }
    J += c.pl->running_cost(c.pl->user, x, u, t);
    c.pl->discrete_dynamics(c.pl->user, x, u, t * c.dt, r.X.data() + (size_t)(t + 1) * nx);
  }
  J += c.pl->terminal_cost(c.pl->user, r.X.data() + (size_t)N * nx);
  const double dJ = T.cost - J;
  const double expected = -alpha * (T.dV0 + 0.5 * alpha * T.dV1);
  const double ratio = expected > 0.0 ? dJ / expected : std::copysign(1.0, dJ);
  r.cost = r.merit = J;
  r.success = ratio > c.o->filter_armijo_constant;
  return r;
}

Trial forward_ipddp(const Ctx &c, const Traj &T, const Gains &g, double alpha) {   // ipddp_solver.cpp:1571-1876
  const int nx = c.nx, nu = c.nu, N = c.N, m = c.m;
  const cddp_hip_options &o = *c.o;
  Trial r;
  const double mu = T.mu;
  const double tau = (m == 0) ? 1.0 : std::max(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  const double a_pr = std::min(alpha, T.apr_max), a_du = std::min(alpha, T.adu_max);
  r.alpha_pr = a_pr; r.alpha_du = a_du;
  r.cost = T.cost; r.merit = T.phi; r.theta = T.theta;   // diagnostics of a failed trial (ForwardPassResult defaults)
  r.X.assign(T.X.size(), 0.0); r.U.assign(T.U.size(), 0.0); r.Lam.assign(T.Lam.size(), 0.0);
  r.S.assign(T.S.size(), 0.0); r.Y = r.S; r.G = r.S;
  std::copy(T.X.begin(), T.X.begin() + nx, r.X.begin());
  std::vector<double> dx(nx);
  double cost_new = 0.0;
  for (int t = 0; t <= N; ++t) {
    const double *x = r.X.data() + (size_t)t * nx, *xo = T.X.data() + (size_t)t * nx;
    for (int i = 0; i < nx; ++i) dx[i] = x[i] - xo[i];
    // costate trial: Lambda' = Lambda + alpha_pr V_x + V_xx dx (:1613-1616, 1660-1663); a non-finite entry fails the trial
    for (int i = 0; i < nx; ++i) {
      double s = 0.0;
      for (int j = 0; j < nx; ++j) s += g.Vxx[((size_t)t * nx + i) * nx + j] * dx[j];
      const double lam = (T.Lam[(size_t)t * nx + i] + a_pr * g.Vx[(size_t)t * nx + i]) + s;
      if (!fin(lam)) return r;
      r.Lam[(size_t)t * nx + i] = lam;
    }
    if (t == N) break;
    if (m > 0) {   // slack / dual trial + fraction-to-boundary test (:1629-1658)
      for (int q = 0; q < m; ++q) {
        const size_t j = (size_t)t * m + q;
        double ps = 0.0, py = 0.0;
        for (int i = 0; i < nx; ++i) { ps = ps + g.Ks[j * nx + i] * dx[i]; py = py + g.Ky[j * nx + i] * dx[i]; }
        const double sn = (T.S[j] + a_pr * g.ks[j]) + ps;
        const double yn = (T.Y[j] + a_du * g.ky[j]) + py;
        if (sn < (1.0 - tau) * T.S[j] || yn < (1.0 - tau) * T.Y[j]) return r;
        if (!fin(sn) || !fin(yn)) return r;
        r.S[j] = sn; r.Y[j] = yn;
      }
    }
    const double *uo = T.U.data() + (size_t)t * nu;
    double *u = r.U.data() + (size_t)t * nu;
    for (int i = 0; i < nu; ++i) {   // no clamping (:1618-1622)
      double s = 0.0;
      for (int j = 0; j < nx; ++j) s += g.K[((size_t)t * nu + i) * nx + j] * dx[j];
      u[i] = (uo[i] + a_pr * g.k[(size_t)t * nu + i]) + s;
      if (!fin(u[i])) return r;
    }
    double *xn = r.X.data() + (size_t)(t + 1) * nx;
    c.pl->discrete_dynamics(c.pl->user, x, u, t * c.dt, xn);
    for (int i = 0; i < nx; ++i) if (!fin(xn[i])) return r;
    cost_new += c.pl->running_cost(c.pl->user, x, u, t);
    if (m > 0) c.pl->constraints(c.pl->user, x, u, t, r.G.data() + (size_t)t * m, nullptr, nullptr);
  }
  cost_new += c.pl->terminal_cost(c.pl->user, r.X.data() + (size_t)N * nx);
  double phi_new = cost_new, theta_new = 0.0, ipr = 0.0, icomp = 0.0;
  if (m > 0) ip_reductions(c, r.S.data(), r.Y.data(), r.G.data(), mu, cost_new, phi_new, theta_new, ipr, icomp);
  if (!fin(phi_new) || !fin(theta_new) || !fin(ipr) || !fin(icomp)) return r;
  bool accept = false;
  if (m == 0) {   // :1785-1792
    const double dJ = T.cost - cost_new;
    const double expected = -a_pr * (T.dV0 + 0.5 * a_pr * T.dV1);
    const double ratio = expected > 0.0 ? dJ / expected : std::copysign(1.0, dJ);
    accept = ratio > 1e-6;
  } else {        // :1793-1834
    const double expected_improvement = a_pr * T.dV0;
    const bool fe = T.filter.empty();
    const double cv_old = fe ? 0.0 : T.filter.back().second;
    const double high_ref = fe ? T.filter_theta : cv_old;
    const double merit_old = T.merit;
    if (theta_new > o.filter_max_violation_threshold) {
      if (theta_new < (1 - o.filter_violation_acceptance_threshold) * high_ref) accept = true;
    } else if (std::max(theta_new, cv_old) < o.filter_min_violation_for_armijo_check && expected_improvement < 0) {
      if (phi_new < merit_old + o.filter_armijo_constant * expected_improvement) accept = true;
    } else {
      if (phi_new < merit_old - o.filter_merit_acceptance_threshold * theta_new ||
          theta_new < (1 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
    }
  }
  r.cost = cost_new; r.merit = phi_new; r.theta = theta_new; r.inf_pr = ipr; r.inf_comp = icomp;
  r.success = accept;
  return r;
}

// computeScaledDualInfeasibility (ipddp_solver.cpp:2725-2776): G_x of the last backward pass (kept per trajectory), Y current
double scaled_inf_du(const Ctx &c, const Traj &T, const std::vector<double> &Gx) {
  double v = T.inf_du;
  if (!c.o->ipddp_check_state_stationarity || c.m == 0) return v;
  double ss = 0.0;
  int off = 0;
  for (int s = 0; s < c.pl->n_constraints; ++s) {
    const int dim = c.pl->constraint_dims[s];
    for (int t = 0; t < c.N; ++t)
      for (int j = 0; j < c.nx; ++j) {
        double a = 0.0;
        for (int i = 0; i < dim; ++i) a += Gx[((size_t)t * c.m + off + i) * c.nx + j] * T.Y[(size_t)t * c.m + off + i];
        ss = std::max(ss, std::fabs(a));
      }
    off += dim;
  }
  return std::max(v, ss);
}


// ======================================================================================================================
// LogDDP (logddp_solver.cpp:43-707 on the CDDPSolverBase loop): single shooting; every path constraint enters through the relaxed
// log barrier (barrier.hpp:37-296), whose value joins the merit function and whose gradients / Hessians are FOLDED into the cost
// derivative stacks here on the host; the GPU runs the unconstrained Riccati sweep of the batch (CDDP_HIP_STACKS_LOGDDP: Q_uu + reg I
// symmetrised, LDLT, dV, inf_du = max |Q_u|).
// ======================================================================================================================
struct LTraj {
  std::vector<double> X, U;
  double cost = 0, merit = 0, violation = 0, inf_pr = 0, inf_du = kInf, alpha_pr = 1.0, reg = 0, mu = 0, dV0 = 0;
  int iter = 0, status = CDDP_HIP_STATUS_RUNNING, n_bwd = 0, n_fwd = 0;
  bool done = false;
};

void lg_beta(double z, double delta, double &b0, double &b1, double &b2) {   // calculate_beta_derivatives (barrier.hpp:274-296)
  if (z > delta) {
    if (z <= 1e-12) { b0 = -std::log(1e-12); b1 = -1.0 / 1e-12; b2 = 1.0 / (1e-12 * 1e-12); }
    else { b0 = -std::log(z); b1 = -1.0 / z; b2 = 1.0 / (z * z); }
  } else {
    const double td = (z - 2.0 * delta) / delta;
    b0 = 0.5 * (td * td - 1.0) - std::log(delta); b1 = td / delta; b2 = 1.0 / (delta * delta);
  }
}

// barrier value of the whole trajectory and its constraint violation (resetFilter :333-361 / forwardPass :637-660): constraint by
// constraint inside a step, as the reference's loops run; every constraint kind has lower bound -inf, so s_U = U - g = -(g - U)
void lg_merit_terms(const Ctx &c, const double *X, const double *U, double mu, double delta, double &barrier, double &violation, std::vector<double> &g) {
  const int m = c.m, N = c.N;
  barrier = 0.0; violation = 0.0;
  if (m == 0) return;
  g.resize(m);
  for (int t = 0; t < N; ++t) {
    c.pl->constraints(c.pl->user, X + (size_t)t * c.nx, U + (size_t)t * c.nu, t, g.data(), nullptr, nullptr);
    int off = 0;
    for (int s = 0; s < c.pl->n_constraints; ++s) {
      const int dim = c.pl->constraint_dims[s];
      double tot = 0.0;
      for (int i = 0; i < dim; ++i) { double b0, b1, b2; lg_beta(-g[off + i], delta, b0, b1, b2); tot += b0; }
      barrier += mu * tot;
      for (int i = 0; i < dim; ++i) if (g[off + i] > 0.0) violation += g[off + i];
      off += dim;
    }
  }
}

inline bool aborted(const cddp_hip_plugin *pl) { return pl->abort_flag && *pl->abort_flag != 0; }

int logddp_solve(const Ctx &c, int device, int batch, const double *x0, const double *U0, cddp_hip_result *results, double *Xout, double *Uout, double *Kout) {
  const cddp_hip_plugin *pl = c.pl; const cddp_hip_options &o = *c.o;
  const int nx = c.nx, nu = c.nu, m = c.m, N = c.N; const double dt = c.dt;
  const size_t B = (size_t)batch;
  cddp_hip_stack_handle *sh = nullptr;
  { int rc = cddp_hip_stacks_create(device, batch, nx, nu, 0, N, &sh); if (rc) return rc; }
  struct Guard { cddp_hip_stack_handle *h; ~Guard() { if (h) cddp_hip_stacks_destroy(h); } } guard{sh};
  std::vector<double> alphas;   // logddp_solver.cpp:171-177: the plain geometric ladder
  { double a = o.ls_initial_step_size; for (int i = 0; i < o.ls_max_iterations; ++i) { alphas.push_back(a); a *= o.ls_step_reduction_factor; } }
  const double delta = o.logddp_relaxed_delta;
  if (!(delta > 0.0)) return pfail(-2, "Relaxation delta must be positive.");
  std::vector<double> gtmp;

  std::vector<LTraj> T(B);
  for (size_t b = 0; b < B; ++b) {   // initialize (:45-205, cold start): roll the control guess out, cost, barrier merit
    LTraj &t = T[b];
    t.X.assign((size_t)(N + 1) * nx, 0.0); t.U.assign((size_t)N * nu, 0.0);
    if (U0) std::copy(U0 + b * N * nu, U0 + (b + 1) * N * nu, t.U.begin());
    std::copy(x0 + b * nx, x0 + (b + 1) * nx, t.X.begin());
    for (int s = 0; s < N; ++s) pl->discrete_dynamics(pl->user, t.X.data() + (size_t)s * nx, t.U.data() + (size_t)s * nu, s * dt, t.X.data() + (size_t)(s + 1) * nx);
    t.reg = o.reg_initial_value; t.mu = o.logddp_mu_initial; t.alpha_pr = o.ls_initial_step_size;
    t.cost = total_cost(c, t.X.data(), t.U.data());
    double bar, viol; lg_merit_terms(c, t.X.data(), t.U.data(), t.mu, delta, bar, viol, gtmp);
    t.merit = t.cost + bar; t.violation = viol; t.inf_pr = viol;
  }

  std::vector<double> fx(B * N * nx * nx), fu(B * N * nx * nu), lx(B * N * nx), lu(B * N * nu), lxx(B * N * nx * nx), luu(B * N * nu * nu),
      lux(B * N * nu * nx), VxN(B * nx), VxxN(B * nx * nx), Fxx, Fuu, Fux;
  if (!o.use_ilqr) { Fxx.resize(B * N * nx * nx * nx); Fuu.resize(B * N * nx * nu * nu); Fux.resize(B * N * nx * nu * nx); }
  std::vector<double> Kb(B * N * nu * nx), kb(B * N * nu), Vxb(B * (N + 1) * nx), Vxxb(B * (N + 1) * nx * nx), dVb(B * 2);
  // K_u_ of each trajectory's LAST backward pass (the reference's solver object stops sweeping a problem when it ends; the batch keeps
  // sweeping the others): the slice is kept at every sweep the trajectory still takes part in
  std::vector<double> Kfin(Kout ? B * N * nu * nx : 0, 0.0);
  std::vector<double> regv(B), s_reg(B), s_du(B), s_pr(B), s_comp(B), s_sn(B), s_apr(B), s_adu(B);
  std::vector<int32_t> okv(B);
  std::vector<double> tfx(nx * nx), tfu(nx * nu), g(std::max(m, 1)), Gx((size_t)std::max(m, 1) * nx), Gu((size_t)std::max(m, 1) * nu);
  std::vector<double> Cxx, Cuu, Cux;
  if (m > 0 && pl->constraint_hessians) { Cxx.resize((size_t)m * nx * nx); Cuu.resize((size_t)m * nu * nu); Cux.resize((size_t)m * nu * nx); }
  std::vector<double> Xn((size_t)(N + 1) * nx), Un((size_t)N * nu), Xb, Ub;
  const bool first_rule = !o.enable_parallel;
  const auto wall0 = std::chrono::steady_clock::now();

  for (int it = 1; it <= o.max_iterations; ++it) {
    bool any = false;
    for (auto &t : T) any = any || !t.done;
    if (!any) break;
    if (aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
    if (o.max_cpu_time > 0.0) {
      const double el_ms = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count();
      if (el_ms > o.max_cpu_time * 1000.0) { for (auto &t : T) if (!t.done) { t.iter += 1; t.status = CDDP_HIP_STATUS_MAX_CPU_TIME; t.done = true; } break; }
    }
    for (size_t b = 0; b < B; ++b) {
      LTraj &t = T[b];
      regv[b] = t.done ? std::max(t.reg, o.reg_min_value) : t.reg;
      if (t.done) continue;
      t.iter += 1;
      for (int s = 0; s < N; ++s) {
        const double *x = t.X.data() + (size_t)s * nx, *u = t.U.data() + (size_t)s * nu;
        const size_t bs = b * N + s;
        pl->jacobians(pl->user, x, u, s * dt, tfx.data(), tfu.data());
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double a = dt * tfx[i * nx + j]; if (i == j) a += 1.0; fx[(bs * nx + i) * nx + j] = a; }
        for (int i = 0; i < nx * nu; ++i) fu[bs * nx * nu + i] = dt * tfu[i];
        double *plx = lx.data() + bs * nx, *plu = lu.data() + bs * nu, *plxx = lxx.data() + bs * nx * nx, *pluu = luu.data() + bs * nu * nu, *plux = lux.data() + bs * nu * nx;
        pl->running_cost_derivatives(pl->user, x, u, s, plx, plu, plxx, pluu, plux);
        if (m > 0) {   // getGradients / getHessians of every constraint (barrier.hpp:95-213), scaled by the barrier coefficient, folded in
          pl->constraints(pl->user, x, u, s, g.data(), Gx.data(), Gu.data());
          if (!Cxx.empty()) {
            std::fill(Cxx.begin(), Cxx.end(), 0.0); std::fill(Cuu.begin(), Cuu.end(), 0.0); std::fill(Cux.begin(), Cux.end(), 0.0);
            pl->constraint_hessians(pl->user, x, u, s, Cxx.data(), Cuu.data(), Cux.data());
          }
          for (int r = 0; r < m; ++r) {
            double b0, b1, b2; lg_beta(-g[r], delta, b0, b1, b2);
            const double dC = t.mu * (-b1), t1 = t.mu * b2, t2 = t.mu * (-b1);
            const double *gx = Gx.data() + (size_t)r * nx, *gu = Gu.data() + (size_t)r * nu;
            for (int a = 0; a < nx; ++a) plx[a] += dC * gx[a];
            for (int a = 0; a < nu; ++a) plu[a] += dC * gu[a];
            for (int a = 0; a < nx; ++a) for (int e = 0; e < nx; ++e) plxx[a * nx + e] += (t1 * gx[a]) * gx[e];
            for (int a = 0; a < nu; ++a) for (int e = 0; e < nu; ++e) pluu[a * nu + e] += (t1 * gu[a]) * gu[e];
            for (int a = 0; a < nu; ++a) for (int e = 0; e < nx; ++e) plux[a * nx + e] += (t1 * gu[a]) * gx[e];
            if (!Cxx.empty()) {
              for (int e = 0; e < nx * nx; ++e) plxx[e] += t2 * Cxx[(size_t)r * nx * nx + e];
              for (int e = 0; e < nu * nu; ++e) pluu[e] += t2 * Cuu[(size_t)r * nu * nu + e];
              for (int e = 0; e < nu * nx; ++e) plux[e] += t2 * Cux[(size_t)r * nu * nx + e];
            }
          }
        }
        if (!o.use_ilqr) {
          double *pxx = Fxx.data() + bs * nx * nx * nx, *puu = Fuu.data() + bs * nx * nu * nu, *pux = Fux.data() + bs * nx * nu * nx;
          pl->hessians(pl->user, x, u, s * dt, pxx, puu, pux);
          for (int e = 0; e < nx * nx * nx; ++e) pxx[e] = dt * pxx[e];
          for (int e = 0; e < nx * nu * nu; ++e) puu[e] = dt * puu[e];
          for (int e = 0; e < nx * nu * nx; ++e) pux[e] = dt * pux[e];
        }
      }
      pl->terminal_cost_derivatives(pl->user, t.X.data() + (size_t)N * nx, VxN.data() + b * nx, VxxN.data() + b * nx * nx);
    }
    { int rc = cddp_hip_set_stacks(sh, fx.data(), fu.data(), lx.data(), lu.data(), lxx.data(), luu.data(), lux.data(), VxN.data(), VxxN.data()); if (rc) return rc; }
    if (!o.use_ilqr) { int rc = cddp_hip_set_hessian_stacks(sh, Fxx.data(), Fuu.data(), Fux.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_backward(sh, CDDP_HIP_STACKS_LOGDDP, c.o, regv.data(), nullptr, 1, okv.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_get_gains(sh, Kb.data(), kb.data(), Vxb.data(), Vxxb.data(), dVb.data()); if (rc) return rc; }
    if (Kout) for (size_t b = 0; b < B; ++b) if (!T[b].done) std::copy(Kb.begin() + b * N * nu * nx, Kb.begin() + (b + 1) * N * nu * nx, Kfin.begin() + b * N * nu * nx);
    { int rc = cddp_hip_stacks_get_scalars(sh, s_reg.data(), s_du.data(), s_pr.data(), s_comp.data(), s_sn.data(), s_apr.data(), s_adu.data()); if (rc) return rc; }

    for (size_t b = 0; b < B; ++b) {
      LTraj &t = T[b];
      if (t.done) continue;
      if (aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
      { int nb = 1; double r = t.reg; while (r < s_reg[b] && nb < 64) { r = reg_increase(o, r); ++nb; }
        if (!okv[b] && nb > 1) --nb;
        t.n_bwd += nb; }
      t.reg = s_reg[b];
      if (!okv[b]) { t.status = CDDP_HIP_STATUS_REG_LIMIT_CONVERGED; t.done = true; continue; }   // handleBackwardPassRegularizationLimit (:216-222)
      t.dV0 = dVb[b * 2]; t.inf_du = s_du[b];
      const double *K = Kb.data() + b * N * nu * nx, *k = kb.data() + b * N * nu;
      // ---- performForwardPass over forwardPass(alpha) (:594-707)
      bool have = false; double best_cost = 0, best_merit = kInf, best_viol = 0, best_alpha = 0; int walked = 0;
      for (double a : alphas) {
        ++walked;
        std::copy(t.X.begin(), t.X.begin() + nx, Xn.begin());
        bool finite = true;
        for (int s = 0; s < N && finite; ++s) {
          const double *xs = Xn.data() + (size_t)s * nx, *xo = t.X.data() + (size_t)s * nx;
          double *us = Un.data() + (size_t)s * nu;
          for (int i = 0; i < nu; ++i) {
            double acc = 0.0;
            for (int j = 0; j < nx; ++j) acc += K[((size_t)s * nu + i) * nx + j] * (xs[j] - xo[j]);
            us[i] = (t.U[(size_t)s * nu + i] + a * k[(size_t)s * nu + i]) + acc;
          }
          pl->discrete_dynamics(pl->user, xs, us, s * dt, Xn.data() + (size_t)(s + 1) * nx);
          for (int i = 0; i < nx; ++i) finite = finite && fin(Xn[(size_t)(s + 1) * nx + i]);
          for (int i = 0; i < nu; ++i) finite = finite && fin(us[i]);
        }
        if (!finite) continue;
        const double cost_new = total_cost(c, Xn.data(), Un.data());
        double bar, cv_new; lg_merit_terms(c, Xn.data(), Un.data(), t.mu, delta, bar, cv_new, gtmp);
        const double merit_new = bar + cost_new;
        const double cv_old = t.violation, expected = a * t.dV0;
        bool accept = false;
        if (cv_new > o.filter_max_violation_threshold) accept = cv_new < (1.0 - o.filter_violation_acceptance_threshold) * cv_old;
        else if (std::max(cv_new, cv_old) < o.filter_min_violation_for_armijo_check && expected < 0) accept = merit_new < t.merit + o.filter_armijo_constant * expected;
        else accept = merit_new < t.merit - o.filter_merit_acceptance_threshold * cv_old || cv_new < (1.0 - o.filter_violation_acceptance_threshold) * cv_old;
        if (!accept) continue;
        if (first_rule || !have || merit_new < best_merit) { Xb = Xn; Ub = Un; best_cost = cost_new; best_merit = merit_new; best_viol = cv_new; best_alpha = a; have = true; }
        if (first_rule) break;
      }
      t.n_fwd += first_rule ? walked : (int)alphas.size();
      bool converged = false;
      if (have) {
        const double dJ = t.cost - best_cost, dL = t.merit - best_merit;
        t.X.swap(Xb); t.U.swap(Ub); t.cost = best_cost; t.merit = best_merit; t.alpha_pr = best_alpha; t.violation = best_viol;
        t.reg = reg_decrease(o, t.reg);
        if (std::max(t.inf_du, t.inf_pr) <= o.tolerance) { t.status = CDDP_HIP_STATUS_OPTIMAL; converged = true; }          // checkConvergence (:233-261);
        else if (std::fabs(dJ) < o.acceptable_tolerance && std::fabs(dL) < o.acceptable_tolerance) { t.status = CDDP_HIP_STATUS_ACCEPTABLE; converged = true; }   // inf_pr is still the last resetFilter's
      } else {
        t.reg = reg_increase(o, t.reg);
        if (t.reg >= o.reg_max_value) { t.status = CDDP_HIP_STATUS_REG_LIMIT; t.done = true; continue; }
      }
      if (converged) { t.done = true; continue; }
      // ---- postIterationUpdate (:263-277): barrier coefficient, then resetFilter with it
      t.mu = have ? std::max(o.logddp_mu_min_value, t.mu * o.logddp_mu_update_factor) : std::min(o.logddp_mu_initial, t.mu * 5.0);
      double bar, viol; lg_merit_terms(c, t.X.data(), t.U.data(), t.mu, delta, bar, viol, gtmp);
      t.merit = t.cost + bar; t.violation = viol; t.inf_pr = viol;
      if (it == o.max_iterations) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
    }
  }
  for (auto &t : T) if (!t.done) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
  if (Kout) std::copy(Kfin.begin(), Kfin.end(), Kout);
  for (size_t b = 0; b < B; ++b) {   // CDDPSolution (+ populateSolverSpecificSolution :286-291)
    const LTraj &t = T[b];
    cddp_hip_result &r = results[b];
    std::memset(&r, 0, sizeof(r));
    r.final_objective = t.cost; r.merit_function = t.merit; r.inf_pr = t.violation; r.inf_du = t.inf_du; r.inf_comp = kInf;
    r.barrier_mu = t.mu; r.regularization = t.reg; r.alpha_pr = t.alpha_pr; r.alpha_du = 0.0;
    r.iterations = t.iter; r.status = t.status; r.n_backward = t.n_bwd; r.n_forward = t.n_fwd;
    if (Xout) std::copy(t.X.begin(), t.X.end(), Xout + b * (N + 1) * nx);
    if (Uout) std::copy(t.U.begin(), t.U.end(), Uout + b * N * nu);
  }
  return 0;
}


// ---- MSIPDDP (msipddp_solver.cpp:33-1930): multiple-shooting interior-point DDP on the host loop ---------------------------------
// The iterate carries costates and the dynamics values F_t = f(x_t, u_t); the defects d_t = F_t - x_{t+1} go to the GPU as a stack.
// Backward pass: CDDP_HIP_STACKS_MSIPDDP with the per-step factor cache (no path constraints: the reference solves every sweep after
// the first with the factor of the step's FIRST sweep, :1169-1185) or CDDP_HIP_STACKS_MSIPDDP_PATH (plain y / s condensation,
// :1222-1420).  Forward pass with gap closing at the segment boundaries ("nonlinear" / "hybrid" / plain), filter, barrier update and
// convergence tests on the host.  use_ilqr = false: the costate-weighted dynamics Hessians and the dual-weighted constraint
// Hessians (:1151-1163, 1279-1310) are folded into the cost-Hessian stacks (last-bit association difference against the reference's
// "Q += ..." after the A^T V A products; the tests hold that case to 1e-9, not to equality).
struct MTraj {
  std::vector<double> X, U, F, Lam, S, Y, G;
  std::vector<std::pair<double, double>> filter;
  double cost = kInf, merit = kInf, inf_pr = kInf, inf_du = kInf, inf_comp = kInf, step_norm = 0, alpha_pr = 1.0, alpha_du = 0.0, reg = 0, mu = 0;
  double dV0 = 0, dV1 = 0;
  int iter = 0, status = CDDP_HIP_STATUS_RUNNING, n_bwd = 0, n_fwd = 0;
  bool done = false;
};

void ms_reset_filter(const Ctx &c, MTraj &t) {   // resetBarrierFilter :711-763
  const int nx = c.nx, m = c.m, N = c.N;
  double mf = t.cost, ipr = 0.0, fcv = 0.0, icomp = 0.0, idef = 0.0;
  if (m > 0) {
    for (int s = 0; s < N; ++s) {
      int off = 0;
      for (int q = 0; q < c.pl->n_constraints; ++q) {
        const int dim = c.pl->constraint_dims[q];
        double lsum = 0.0, l1 = 0.0;
        for (int i = 0; i < dim; ++i) {
          const size_t j = (size_t)s * m + off + i;
          lsum += std::log(t.S[j]);
          const double pr = t.G[j] + t.S[j];
          ipr = std::max(ipr, std::fabs(pr)); l1 += std::fabs(pr);
          icomp = std::max(icomp, std::fabs(t.Y[j] * t.S[j] - t.mu));
        }
        mf -= t.mu * lsum; fcv += l1; off += dim;
      }
      double dn = 0.0, d1 = 0.0;
      for (int i = 0; i < nx; ++i) { const double d = t.F[(size_t)s * nx + i] - t.X[(size_t)(s + 1) * nx + i]; dn = std::max(dn, std::fabs(d)); d1 += std::fabs(d); }
      idef = std::max(idef, dn); fcv += d1;
    }
  }
  t.inf_pr = std::max(ipr, idef); t.merit = mf; t.inf_comp = icomp;
  t.filter.clear(); t.filter.emplace_back(mf, fcv);
}

void ms_init_pair(const cddp_hip_options &o, double mu, const double *g, double *s, double *y, int m) {   // :578-596 == :667-685
  for (int i = 0; i < m; ++i) {
    s[i] = std::max(o.ipddp_slack_var_init_scale, -g[i]);
    y[i] = (s[i] < 1e-12) ? mu / 1e-12 : mu / s[i];
    y[i] = std::max(o.ipddp_dual_var_init_scale * 0.01, std::min(y[i], o.ipddp_dual_var_init_scale * 100.0));
  }
}

double ms_scaled_inf_du(const Ctx &c, const MTraj &t) {   // computeScaledDualInfeasibility :1886-1930
  if (c.m == 0) return t.inf_du;
  double yn = 0.0, sn = 0.0;
  int off = 0;
  for (int q = 0; q < c.pl->n_constraints; ++q) {   // constraint-major, then t, as the reference's maps are walked
    const int dim = c.pl->constraint_dims[q];
    for (int s = 0; s < c.N; ++s) {
      double a = 0.0, b = 0.0;
      for (int i = 0; i < dim; ++i) { a += std::fabs(t.Y[(size_t)s * c.m + off + i]); b += std::fabs(t.S[(size_t)s * c.m + off + i]); }
      yn += a; sn += b;
    }
    off += dim;
  }
  const int mpn = c.m * c.N + c.nu * c.N;
  const double num = mpn > 0 ? (yn + sn) / (double)mpn : 0.0;
  return t.inf_du / (std::max(100.0, num) / 100.0);
}

bool ms_filter_acceptable(const cddp_hip_options &o, const std::vector<std::pair<double, double>> &f, double mf, double cv, double expected) {   // :771-808
  if (f.empty()) return true;
  for (auto &p : f) if (p.first <= mf && p.second <= cv) return false;
  double best_v = kInf, best_m = kInf;
  for (auto &p : f) if (p.second < best_v) { best_v = p.second; best_m = p.first; }
  const bool v_imp = cv < best_v * (1.0 - o.filter_violation_acceptance_threshold);
  const bool m_imp = mf < best_m - o.filter_merit_acceptance_threshold * cv;
  if (cv < o.filter_min_violation_for_armijo_check && expected < 0) return mf < best_m + o.filter_armijo_constant * expected;
  if (cv < 1e-6 && mf <= best_m * (1.0 + 1e-8)) return true;
  return v_imp || m_imp;
}

int msipddp_solve(const Ctx &c, int device, int batch, const double *x0, const double *U0, const double *X0, cddp_hip_result *results,
                  double *Xout, double *Uout, double *Kout) {
  const cddp_hip_plugin *pl = c.pl; const cddp_hip_options &o = *c.o;
  const int nx = c.nx, nu = c.nu, m = c.m, N = c.N; const double dt = c.dt;
  const size_t B = (size_t)batch;
  if (m > 0 && !(nu == 1 || nx == nu))
    return pfail(-3, "MSIPDDP with path constraints is only defined for nu = 1 or nx = nu: the reference adds an (nx x nu) product to its (nu x nx) block Q_ux (msipddp_solver.cpp:1398); got nx = %d, nu = %d", nx, nu);
  if (o.msipddp_segment_length < 0) return pfail(-2, "MSIPDDP: ms_segment_length must be non-negative");
  if (!o.use_ilqr && m > 0 && !pl->constraint_hessians) return pfail(-3, "MSIPDDP with use_ilqr=false needs the constraint Hessian callback");
  cddp_hip_stack_handle *sh = nullptr;
  { int rc = cddp_hip_stacks_create(device, batch, nx, nu, m, N, &sh); if (rc) return rc; }
  struct Guard { cddp_hip_stack_handle *h; ~Guard() { if (h) cddp_hip_stacks_destroy(h); } } guard{sh};
  if (m == 0) { int rc = cddp_hip_stacks_factor_cache(sh, 1); if (rc) return rc; }
  const int seg = o.msipddp_segment_length, rtype = o.msipddp_rollout_type;
  std::vector<double> alphas;
  { double a = o.ls_initial_step_size; for (int i = 0; i < o.ls_max_iterations; ++i) { alphas.push_back(a); a *= o.ls_step_reduction_factor; } }

  std::vector<MTraj> T(B);
  std::vector<double> xn(nx);
  for (size_t b = 0; b < B; ++b) {   // initialize :33-264
    MTraj &t = T[b];
    t.X.assign((size_t)(N + 1) * nx, 0.0); t.U.assign((size_t)N * nu, 0.0);
    if (U0) std::copy(U0 + b * N * nu, U0 + (b + 1) * N * nu, t.U.begin());
    if (X0) std::copy(X0 + b * (N + 1) * nx, X0 + (b + 1) * (N + 1) * nx, t.X.begin());
    else for (int s = 0; s <= N; ++s) std::copy(x0 + b * nx, x0 + (b + 1) * nx, t.X.begin() + (size_t)s * nx);
    std::copy(x0 + b * nx, x0 + (b + 1) * nx, t.X.begin());
    t.F.assign((size_t)N * nx, 0.0); t.Lam.assign((size_t)N * nx, o.msipddp_costate_var_init_scale);
    t.S.assign((size_t)N * m, 0.0); t.Y = t.S; t.G = t.S;
    t.alpha_pr = o.ls_initial_step_size; t.alpha_du = 0.0; t.reg = o.reg_initial_value; t.step_norm = 0.0;
    if (o.warm_start) {   // no gains of an earlier solve in this call: the branch :108-160 -- the state guess is kept as it is
      if (m == 0) t.mu = 1e-8;
      else {
        double cost = 0.0, mv = 0.0;   // evaluateTrajectoryWarmStart :457-495
        for (int s = 0; s < N; ++s) {
          double *x = t.X.data() + (size_t)s * nx, *u = t.U.data() + (size_t)s * nu;
          cost += pl->running_cost(pl->user, x, u, s);
          pl->constraints(pl->user, x, u, s, t.G.data() + (size_t)s * m, nullptr, nullptr);
          pl->discrete_dynamics(pl->user, x, u, s * dt, t.F.data() + (size_t)s * nx);
          if (o.msipddp_use_controlled_rollout) std::copy(t.F.begin() + (size_t)s * nx, t.F.begin() + (size_t)(s + 1) * nx, t.X.begin() + (size_t)(s + 1) * nx);
        }
        cost += pl->terminal_cost(pl->user, t.X.data() + (size_t)N * nx);
        t.cost = cost;
        for (double g : t.G) mv = std::max(mv, g);   // detail::computeMaxConstraintViolation (interior_point_utils.cpp:141-155)
        t.mu = (mv <= o.tolerance) ? o.tolerance * 0.01 : (mv <= 0.1) ? o.tolerance : o.barrier_mu_initial * 0.1;
        for (int s = 0; s < N; ++s) ms_init_pair(o, t.mu, t.G.data() + (size_t)s * m, t.S.data() + (size_t)s * m, t.Y.data() + (size_t)s * m, m);
      }
      ms_reset_filter(c, t);
      continue;
    }
    t.mu = (m == 0) ? 1e-8 : o.barrier_mu_initial;   // cold start :199-263
    for (int s = 0; s < N && m > 0; ++s) {           // initializeDualSlackCostateVariables, on the GUESS trajectory
      pl->constraints(pl->user, t.X.data() + (size_t)s * nx, t.U.data() + (size_t)s * nu, s, t.G.data() + (size_t)s * m, nullptr, nullptr);
      ms_init_pair(o, t.mu, t.G.data() + (size_t)s * m, t.S.data() + (size_t)s * m, t.Y.data() + (size_t)s * m, m);
    }
    double cost = 0.0;                               // evaluateTrajectory :425-455
    for (int s = 0; s < N; ++s) {
      double *x = t.X.data() + (size_t)s * nx, *u = t.U.data() + (size_t)s * nu;
      cost += pl->running_cost(pl->user, x, u, s);
      if (m > 0) pl->constraints(pl->user, x, u, s, t.G.data() + (size_t)s * m, nullptr, nullptr);
      pl->discrete_dynamics(pl->user, x, u, s * dt, t.F.data() + (size_t)s * nx);
      std::copy(t.F.begin() + (size_t)s * nx, t.F.begin() + (size_t)(s + 1) * nx, t.X.begin() + (size_t)(s + 1) * nx);
    }
    cost += pl->terminal_cost(pl->user, t.X.data() + (size_t)N * nx);
    t.cost = cost;
    ms_reset_filter(c, t);
  }

  std::vector<double> fx(B * N * nx * nx), fu(B * N * nx * nu), lx(B * N * nx), lu(B * N * nu), lxx(B * N * nx * nx), luu(B * N * nu * nu),
      lux(B * N * nu * nx), VxN(B * nx), VxxN(B * nx * nx), dfc(B * N * nx);
  std::vector<double> ys(B * N * m), ss(B * N * m), gs(B * N * m), Gxs(B * N * m * nx), Gus(B * N * m * nu);
  std::vector<double> Kb(B * N * nu * nx), kb(B * N * nu), Vxb(B * (N + 1) * nx), Vxxb(B * (N + 1) * nx * nx), dVb(B * 2);
  // K_u_ of each trajectory's LAST backward pass (the reference's solver object stops sweeping a problem when it ends; the batch keeps
  // sweeping the others): the slice is kept at every sweep the trajectory still takes part in
  std::vector<double> Kfin(Kout ? B * N * nu * nx : 0, 0.0);
  std::vector<double> kyb(B * N * m), Kyb(B * N * m * nx), ksb(B * N * m), Ksb(B * N * m * nx), dXb(m > 0 ? B * (N + 1) * nx : 0);
  std::vector<double> regv(B), muv(B), s_reg(B), s_du(B), s_pr(B), s_comp(B), s_sn(B), s_apr(B), s_adu(B);
  std::vector<int32_t> okv(B);
  std::vector<double> tfx(nx * nx), tfu(nx * nu), Hxx, Huu, Hux, Cxx, Cuu, Cux, gtmp(std::max(m, 1));
  if (!o.use_ilqr) {
    Hxx.resize((size_t)nx * nx * nx); Huu.resize((size_t)nx * nu * nu); Hux.resize((size_t)nx * nu * nx);
    if (m > 0) { Cxx.resize((size_t)m * nx * nx); Cuu.resize((size_t)m * nu * nu); Cux.resize((size_t)m * nu * nx); }
  }
  std::vector<double> kl((size_t)N * nx), Kl((size_t)N * nx * nx), dxs((size_t)N * nx), Xn, Un, Fn, Ln, Sn, Yn, Gn, Yt, ynew(std::max(m, 1)), snew(std::max(m, 1));
  std::vector<double> Abuf((size_t)nx * nx), Bbuf((size_t)nx * nu), tmpx(nx);
  const bool first_rule = !o.enable_parallel;
  const auto wall0 = std::chrono::steady_clock::now();

  for (int it = 1; it <= o.max_iterations; ++it) {
    bool any = false;
    for (auto &t : T) any = any || !t.done;
    if (!any) break;
    if (aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
    if (o.max_cpu_time > 0.0) {
      const double el_ms = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count();
      if (el_ms > o.max_cpu_time * 1000.0) { for (auto &t : T) if (!t.done) { t.iter += 1; t.status = CDDP_HIP_STATUS_MAX_CPU_TIME; t.done = true; } break; }
    }
    // ---- derivative stacks of every running iterate (precomputeDynamicsDerivatives / precomputeConstraintGradients :846-1110)
    for (size_t b = 0; b < B; ++b) {
      MTraj &t = T[b];
      regv[b] = t.done ? std::max(t.reg, o.reg_min_value) : t.reg;
      muv[b] = (t.mu > 0.0) ? t.mu : 1e-8;
      if (t.done) continue;
      t.iter += 1;
      for (int s = 0; s < N; ++s) {
        const double *x = t.X.data() + (size_t)s * nx, *u = t.U.data() + (size_t)s * nu;
        const size_t bs = b * N + s;
        for (int i = 0; i < nx; ++i) dfc[bs * nx + i] = t.F[(size_t)s * nx + i] - t.X[(size_t)(s + 1) * nx + i];
        pl->jacobians(pl->user, x, u, s * dt, tfx.data(), tfu.data());
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double a = dt * tfx[i * nx + j]; if (i == j) a += 1.0; fx[(bs * nx + i) * nx + j] = a; }
        for (int i = 0; i < nx * nu; ++i) fu[bs * nx * nu + i] = dt * tfu[i];
        double *plxx = lxx.data() + bs * nx * nx, *pluu = luu.data() + bs * nu * nu, *plux = lux.data() + bs * nu * nx;
        pl->running_cost_derivatives(pl->user, x, u, s, lx.data() + bs * nx, lu.data() + bs * nu, plxx, pluu, plux);
        if (m > 0) {
          std::copy(t.Y.begin() + (size_t)s * m, t.Y.begin() + (size_t)(s + 1) * m, ys.begin() + bs * m);
          std::copy(t.S.begin() + (size_t)s * m, t.S.begin() + (size_t)(s + 1) * m, ss.begin() + bs * m);
          std::copy(t.G.begin() + (size_t)s * m, t.G.begin() + (size_t)(s + 1) * m, gs.begin() + bs * m);
          pl->constraints(pl->user, x, u, s, gtmp.data(), Gxs.data() + bs * m * nx, Gus.data() + bs * m * nu);
        }
        if (!o.use_ilqr) {   // :1151-1163, 1279-1310, folded into the cost Hessians
          pl->hessians(pl->user, x, u, s * dt, Hxx.data(), Huu.data(), Hux.data());
          for (int i = 0; i < nx; ++i) {
            const double w = dt * t.Lam[(size_t)s * nx + i];
            for (int e = 0; e < nx * nx; ++e) plxx[e] += w * Hxx[(size_t)i * nx * nx + e];
            for (int e = 0; e < nu * nx; ++e) plux[e] += w * Hux[(size_t)i * nu * nx + e];
            for (int e = 0; e < nu * nu; ++e) pluu[e] += w * Huu[(size_t)i * nu * nu + e];
          }
          if (m > 0) {
            std::fill(Cxx.begin(), Cxx.end(), 0.0); std::fill(Cuu.begin(), Cuu.end(), 0.0); std::fill(Cux.begin(), Cux.end(), 0.0);
            pl->constraint_hessians(pl->user, x, u, s, Cxx.data(), Cuu.data(), Cux.data());
            for (int r = 0; r < m; ++r) {
              const double w = t.Y[(size_t)s * m + r];
              for (int e = 0; e < nx * nx; ++e) plxx[e] += w * Cxx[(size_t)r * nx * nx + e];
              for (int e = 0; e < nu * nx; ++e) plux[e] += w * Cux[(size_t)r * nu * nx + e];
              for (int e = 0; e < nu * nu; ++e) pluu[e] += w * Cuu[(size_t)r * nu * nu + e];
            }
          }
        }
      }
      pl->terminal_cost_derivatives(pl->user, t.X.data() + (size_t)N * nx, VxN.data() + b * nx, VxxN.data() + b * nx * nx);
    }
    { int rc = cddp_hip_set_stacks(sh, fx.data(), fu.data(), lx.data(), lu.data(), lxx.data(), luu.data(), lux.data(), VxN.data(), VxxN.data()); if (rc) return rc; }
    { int rc = cddp_hip_set_defect_stack(sh, dfc.data()); if (rc) return rc; }
    if (m > 0) { int rc = cddp_hip_set_constraint_stacks(sh, ys.data(), ss.data(), gs.data(), Gxs.data(), Gus.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_backward(sh, m > 0 ? CDDP_HIP_STACKS_MSIPDDP_PATH : CDDP_HIP_STACKS_MSIPDDP, c.o, regv.data(), m > 0 ? muv.data() : nullptr, 1, okv.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_get_gains(sh, Kb.data(), kb.data(), Vxb.data(), Vxxb.data(), dVb.data()); if (rc) return rc; }
    if (Kout) for (size_t b = 0; b < B; ++b) if (!T[b].done) std::copy(Kb.begin() + b * N * nu * nx, Kb.begin() + (b + 1) * N * nu * nx, Kfin.begin() + b * N * nu * nx);
    if (m > 0) { int rc = cddp_hip_stacks_get_constraint_gains(sh, kyb.data(), Kyb.data(), ksb.data(), Ksb.data(), dXb.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_get_scalars(sh, s_reg.data(), s_du.data(), s_pr.data(), s_comp.data(), s_sn.data(), s_apr.data(), s_adu.data()); if (rc) return rc; }

    for (size_t b = 0; b < B; ++b) {
      MTraj &t = T[b];
      if (t.done) continue;
      if (aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
      { int nb = 1; double r = t.reg; while (r < s_reg[b] && nb < 64) { r = reg_increase(o, r); ++nb; }
        if (!okv[b] && nb > 1) --nb;
        t.n_bwd += nb; }
      t.reg = s_reg[b];
      if (!okv[b]) { t.status = CDDP_HIP_STATUS_REG_LIMIT; t.done = true; continue; }   // handleBackwardPassRegularizationLimit (base)
      t.dV0 = dVb[b * 2]; t.dV1 = dVb[b * 2 + 1]; t.inf_du = s_du[b]; t.step_norm = s_sn[b];
      double idef = 0.0;
      for (size_t e = 0; e < (size_t)N * nx; ++e) idef = std::max(idef, std::fabs(dfc[b * N * nx + e]));
      if (m > 0) { t.inf_pr = std::max(s_pr[b], idef); t.inf_comp = s_comp[b]; } else { t.inf_pr = idef; t.inf_comp = 0.0; }
      const double *K = Kb.data() + b * N * nu * nx, *k = kb.data() + b * N * nu, *Vx = Vxb.data() + b * (N + 1) * nx, *Vxx = Vxxb.data() + b * (N + 1) * nx * nx;
      const double *ky = kyb.data() + b * N * m, *Ky = Kyb.data() + b * N * m * nx, *ks = ksb.data() + b * N * m, *Ks = Ksb.data() + b * N * m * nx;
      for (int s = 0; s < N; ++s) {   // k_lambda = -lambda + V_x + V_xx d, K_lambda = sym(V_xx) of step s + 1 (:1196-1198)
        const double *vx = Vx + (size_t)(s + 1) * nx, *vxx = Vxx + (size_t)(s + 1) * nx * nx, *d = dfc.data() + (b * N + s) * nx;
        for (int i = 0; i < nx; ++i) {
          double acc = 0.0;
          for (int j = 0; j < nx; ++j) acc += vxx[i * nx + j] * d[j];
          kl[(size_t)s * nx + i] = (-t.Lam[(size_t)s * nx + i] + vx[i]) + acc;
          for (int j = 0; j < nx; ++j) Kl[((size_t)s * nx + i) * nx + j] = 0.5 * (vxx[i * nx + j] + vxx[j * nx + i]);
        }
      }
      const double tau = std::max(o.barrier_min_fraction_to_boundary, 1.0 - t.mu);
      // ---- performForwardPass over forwardPass(alpha) :1432-1724
      bool have = false; int walked = 0;
      double b_cost = 0, b_merit = kInf, b_cv = 0, b_alpha = 0, b_adu = 1.0;
      std::vector<double> bX, bU, bF, bL, bS, bY, bG;
      for (double a : alphas) {
        ++walked;
        Xn = t.X; Un = t.U; Fn = t.F; Ln = t.Lam; Sn = t.S;
        std::copy(x0 + b * nx, x0 + (b + 1) * nx, Xn.begin());
        bool s_ok = true;
        double cost_new = 0.0;
        for (int s = 0; s < N && s_ok; ++s) {
          const double *xs = Xn.data() + (size_t)s * nx, *xo = t.X.data() + (size_t)s * nx;
          double *dx = dxs.data() + (size_t)s * nx;
          for (int i = 0; i < nx; ++i) dx[i] = xs[i] - xo[i];
          if (m > 0) {
            for (int r = 0; r < m; ++r) {
              double acc = 0.0;
              for (int j = 0; j < nx; ++j) acc += Ks[((size_t)s * m + r) * nx + j] * dx[j];
              snew[r] = (t.S[(size_t)s * m + r] + a * ks[(size_t)s * m + r]) + acc;
              if (snew[r] < (1.0 - tau) * t.S[(size_t)s * m + r]) { s_ok = false; break; }
            }
            if (!s_ok) break;
            std::copy(snew.begin(), snew.begin() + m, Sn.begin() + (size_t)s * m);
          }
          double *us = Un.data() + (size_t)s * nu;
          for (int i = 0; i < nu; ++i) {
            double acc = 0.0;
            for (int j = 0; j < nx; ++j) acc += K[((size_t)s * nu + i) * nx + j] * dx[j];
            us[i] = (t.U[(size_t)s * nu + i] + a * k[(size_t)s * nu + i]) + acc;
          }
          if (m == 0) {
            for (int i = 0; i < nx; ++i) {
              double acc = 0.0;
              for (int j = 0; j < nx; ++j) acc += Kl[((size_t)s * nx + i) * nx + j] * dx[j];
              Ln[(size_t)s * nx + i] = (t.Lam[(size_t)s * nx + i] + a * kl[(size_t)s * nx + i]) + acc;
            }
          }
          double *fn = Fn.data() + (size_t)s * nx, *xnext = Xn.data() + (size_t)(s + 1) * nx;
          pl->discrete_dynamics(pl->user, xs, us, s * dt, fn);
          const bool boundary = (seg > 1) && ((s + 1) % seg == 0) && (s + 1 < N);
          const double *fo = t.F.data() + (size_t)s * nx, *xon = t.X.data() + (size_t)(s + 1) * nx;
          if (boundary && rtype == 0) {
            for (int i = 0; i < nx; ++i) xnext[i] = (xon[i] + (fn[i] - fo[i])) + a * (fo[i] - xon[i]);
          } else if (boundary && rtype == 2) {
            pl->jacobians(pl->user, xo, t.U.data() + (size_t)s * nu, s * dt, tfx.data(), tfu.data());
            for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double v = dt * tfx[i * nx + j]; if (i == j) v = 1.0 + v; Abuf[i * nx + j] = v; }
            for (int i = 0; i < nx * nu; ++i) Bbuf[i] = dt * tfu[i];
            for (int i = 0; i < nx; ++i) {
              double p1 = 0.0;   // ((A + B K) dx)_i
              for (int j = 0; j < nx; ++j) {
                double bk = 0.0;
                for (int q = 0; q < nu; ++q) bk += Bbuf[i * nu + q] * K[((size_t)s * nu + q) * nx + j];
                p1 += (Abuf[i * nx + j] + bk) * dx[j];
              }
              double p2 = 0.0;   // (B k)_i
              for (int q = 0; q < nu; ++q) p2 += Bbuf[i * nu + q] * k[(size_t)s * nu + q];
              xnext[i] = (xon[i] + p1) + a * ((p2 + fo[i]) - xon[i]);
            }
          } else {
            for (int i = 0; i < nx; ++i) xnext[i] = fn[i];
          }
          if (m == 0) cost_new += pl->running_cost(pl->user, xs, us, s);
        }
        if (!s_ok) continue;
        if (m == 0) {
          cost_new += pl->terminal_cost(pl->user, Xn.data() + (size_t)N * nx);
          const double dJ = t.cost - cost_new;
          const double expected = -a * (t.dV0 + 0.5 * a * t.dV1);
          const double ratio = expected > 0.0 ? dJ / expected : std::copysign(1.0, dJ);
          if (!(ratio > 1e-6)) continue;
          if (first_rule || !have || cost_new < b_merit) { bX = Xn; bU = Un; bF = Fn; bL = Ln; b_cost = cost_new; b_merit = cost_new; b_cv = 0.0; b_alpha = a; b_adu = 1.0; have = true; }
          if (first_rule) break;
          continue;
        }
        bool found = false; double adu = 1.0;
        for (double ay : alphas) {   // dual step: the first ladder entry that keeps every y above (1 - tau) y_old
          bool feas = true;
          Yt = t.Y;
          for (int s = 0; s < N && feas; ++s) {
            const double *dx = dxs.data() + (size_t)s * nx;
            for (int r = 0; r < m; ++r) {
              double acc = 0.0;
              for (int j = 0; j < nx; ++j) acc += Ky[((size_t)s * m + r) * nx + j] * dx[j];
              ynew[r] = (t.Y[(size_t)s * m + r] + ay * ky[(size_t)s * m + r]) + acc;
            }
            int off = 0;
            for (int q = 0; q < pl->n_constraints && feas; ++q) {   // a constraint's block is stored only when ALL its rows pass
              const int dim = pl->constraint_dims[q];
              for (int i = 0; i < dim; ++i) if (ynew[off + i] < (1.0 - tau) * t.Y[(size_t)s * m + off + i]) { feas = false; break; }
              if (feas) std::copy(ynew.begin() + off, ynew.begin() + off + dim, Yt.begin() + (size_t)s * m + off);
              off += dim;
            }
          }
          if (feas) { found = true; adu = ay; break; }
        }
        if (!found) continue;
        for (int s = 0; s < N; ++s) {
          const double *dx = dxs.data() + (size_t)s * nx;
          for (int i = 0; i < nx; ++i) {
            double acc = 0.0;
            for (int j = 0; j < nx; ++j) acc += Kl[((size_t)s * nx + i) * nx + j] * dx[j];
            Ln[(size_t)s * nx + i] = (t.Lam[(size_t)s * nx + i] + a * kl[(size_t)s * nx + i]) + acc;
          }
        }
        Gn = t.G;
        double merit_new = 0.0, cv_new = 0.0;
        for (int s = 0; s < N; ++s) {
          const double *xs = Xn.data() + (size_t)s * nx, *us = Un.data() + (size_t)s * nu;
          cost_new += pl->running_cost(pl->user, xs, us, s);
          pl->constraints(pl->user, xs, us, s, Gn.data() + (size_t)s * m, nullptr, nullptr);
          int off = 0;
          for (int q = 0; q < pl->n_constraints; ++q) {
            const int dim = pl->constraint_dims[q];
            double lsum = 0.0, l1 = 0.0;
            for (int i = 0; i < dim; ++i) { const size_t j = (size_t)s * m + off + i; lsum += std::log(Sn[j]); l1 += std::fabs(Gn[j] + Sn[j]); }
            merit_new -= t.mu * lsum; cv_new += l1; off += dim;
          }
          double d1 = 0.0;
          for (int i = 0; i < nx; ++i) d1 += std::fabs(Fn[(size_t)s * nx + i] - Xn[(size_t)(s + 1) * nx + i]);
          cv_new += d1;
        }
        cost_new += pl->terminal_cost(pl->user, Xn.data() + (size_t)N * nx);
        merit_new += cost_new;
        if (!ms_filter_acceptable(o, t.filter, merit_new, cv_new, a * t.dV0)) continue;
        if (first_rule || !have || merit_new < b_merit) { bX = Xn; bU = Un; bF = Fn; bL = Ln; bS = Sn; bY = Yt; bG = Gn; b_cost = cost_new; b_merit = merit_new; b_cv = cv_new; b_alpha = a; b_adu = adu; have = true; }
        if (first_rule) break;
      }
      t.n_fwd += first_rule ? walked : (int)alphas.size();
      bool converged = false;
      if (have) {   // applyForwardPassResult :287-304, decreaseRegularization, checkConvergence :306-364
        const double dJ = t.cost - b_cost;
        t.X.swap(bX); t.U.swap(bU); t.F.swap(bF); t.Lam.swap(bL);
        if (m > 0) { t.S.swap(bS); t.Y.swap(bY); t.G.swap(bG); }
        t.cost = b_cost; t.merit = b_merit; t.alpha_pr = b_alpha; t.alpha_du = b_adu;
        filter_accept(t.filter, b_merit, b_cv);
        t.reg = reg_decrease(o, t.reg);
        const double metric = std::max(std::max(ms_scaled_inf_du(c, t), t.inf_pr), t.inf_comp);
        if (metric <= o.tolerance) { t.status = CDDP_HIP_STATUS_OPTIMAL; converged = true; }
        else if (std::fabs(dJ) < o.acceptable_tolerance && it > 10 && t.inf_pr < std::sqrt(o.acceptable_tolerance) && t.inf_comp < std::sqrt(o.acceptable_tolerance)) { t.status = CDDP_HIP_STATUS_ACCEPTABLE; converged = true; }
        else if (it >= 1 && t.step_norm < o.tolerance * 10.0 && t.inf_pr < 1e-4) { t.status = CDDP_HIP_STATUS_ACCEPTABLE; converged = true; }
      } else {      // handleForwardPassFailure :371-398
        bool needs = t.filter.size() > 5;
        if (!needs) for (auto &p : t.filter) if (!fin(p.first) || !fin(p.second)) { needs = true; break; }
        if (needs && !t.filter.empty()) filter_prune(t.filter);
        else {
          t.reg = reg_increase(o, t.reg);
          if (t.reg >= o.reg_max_value) { t.status = CDDP_HIP_STATUS_REG_LIMIT; t.done = true; continue; }
        }
      }
      if (converged) { t.done = true; continue; }
      if (m > 0) {   // postIterationUpdate -> updateBarrierParameters :1751-1850
        if (o.barrier_strategy == CDDP_HIP_BARRIER_MONOTONIC) {
          t.mu = std::max(o.barrier_mu_min_value, o.barrier_mu_update_factor * t.mu); ms_reset_filter(c, t);
        } else if (o.barrier_strategy == CDDP_HIP_BARRIER_IPOPT) {
          const double err = std::max(std::max(ms_scaled_inf_du(c, t), t.inf_pr), t.inf_comp);
          if (err <= 10.0 * t.mu) {
            t.mu = std::max(o.tolerance / 10.0, std::min(o.barrier_mu_update_factor * t.mu, std::pow(t.mu, o.barrier_mu_update_power)));
            ms_reset_filter(c, t);
          }
        } else {
          const double metric = std::max(std::max(ms_scaled_inf_du(c, t), t.inf_pr), t.inf_comp);
          const double thr = (t.mu < 1e-5) ? std::max(metric * 10.0, t.mu * 100.0) : std::max(o.barrier_mu_update_factor * t.mu, t.mu * 2.0);
          const bool slow = have && t.alpha_pr > 0 && (metric < 1e-3);
          if (metric <= thr || slow) {
            double fac = o.barrier_mu_update_factor;
            if (t.mu > 1e-12) {
              const double ratio = metric / t.mu;
              if (ratio < 0.01) fac = o.barrier_mu_update_factor * 0.1;
              else if (ratio < 0.1) fac = o.barrier_mu_update_factor * 0.3;
              else if (ratio < 0.5) fac = o.barrier_mu_update_factor * 0.6;
            }
            const double lin = fac * t.mu, sup = std::pow(t.mu, o.barrier_mu_update_power);
            if (slow && t.mu > o.tolerance) t.mu = std::min(lin, sup);
            else t.mu = std::max(o.tolerance / 100.0, std::min(lin, sup));
            ms_reset_filter(c, t);
          }
        }
      }
      if (it == o.max_iterations) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
    }
  }
  for (auto &t : T) if (!t.done) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
  if (Kout) std::copy(Kfin.begin(), Kfin.end(), Kout);
  for (size_t b = 0; b < B; ++b) {   // CDDPSolution + populateSolverSpecificSolution :406-413
    const MTraj &t = T[b];
    cddp_hip_result &r = results[b];
    std::memset(&r, 0, sizeof(r));
    r.final_objective = t.cost; r.merit_function = t.merit; r.inf_pr = t.inf_pr; r.inf_du = t.inf_du; r.inf_comp = t.inf_comp;
    r.barrier_mu = t.mu; r.regularization = t.reg; r.alpha_pr = t.alpha_pr; r.alpha_du = t.alpha_du;
    r.iterations = t.iter; r.status = t.status; r.n_backward = t.n_bwd; r.n_forward = t.n_fwd;
    if (Xout) std::copy(t.X.begin(), t.X.end(), Xout + b * (N + 1) * nx);
    if (Uout) std::copy(t.U.begin(), t.U.end(), Uout + b * N * nu);
  }
  return 0;
}

// ======================================================================================================================
// IPDDP with TERMINAL constraints for host plug-ins (round 6).  The reference's own user-plant regression pairs a
// DynamicalSystem subclass with a TerminalEqualityConstraint and use_ilqr = false (tests/cddp_core/test_ipddp_solver.cpp:292-346,
// 1512-1578).  Division of labour as above: callbacks, LQ-model assembly (the path constraints condensed into Q, q, R, r, M exactly as
// ipddp_solver.cpp:1143-1245 does; terminal-inequality barrier terms added to V_x, V_xx, :1000-1031) and the filter line search on the
// host; the Riccati work on the GPU -- the reduced LQR with terminal-equality rows (CDDP_HIP_STACKS_IPDDP_TERM_EQ, stacks_te.hpp) or,
// with terminal inequalities only, the ordinary stack-fed sweeps on the modified terminal value.  Also here: the "warm start with
// provided trajectory" initialisation (:733-816) when options.warm_start is set (a stateless call has no existing solver state).
// ======================================================================================================================
constexpr double kEpsDual = 1e-10;   // EPS_DUAL ipddp_solver.cpp:37

struct TermInfo {
  const cddp_hip_plugin_terminal *tc = nullptr;
  int nobj = 0, rows = 0, mT = 0, pT = 0;
  int dim[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS], eq[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS], src[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS], dst[CDDP_HIP_PLUGIN_MAX_CONSTRAINTS];
  // evaluate -> (g_T rows of the inequality objects in object order | h_T rows of the equality objects in object order) (+ Jacobian rows)
  void eval(void *user, int nx, const double *xN, double *gT, double *GTx, double *hT, double *HT, std::vector<double> &r, std::vector<double> &rx) const {
    if (!tc || rows == 0) return;
    r.assign((size_t)rows, 0.0); rx.assign((size_t)rows * nx, 0.0);
    tc->evaluate(user, xN, r.data(), (GTx || HT) ? rx.data() : nullptr);
    for (int s = 0; s < nobj; ++s)
      for (int i = 0; i < dim[s]; ++i) {
        if (eq[s]) { if (hT) hT[dst[s] + i] = r[src[s] + i]; if (HT) std::copy(rx.begin() + (size_t)(src[s] + i) * nx, rx.begin() + (size_t)(src[s] + i + 1) * nx, HT + (size_t)(dst[s] + i) * nx); }
        else { if (gT) gT[dst[s] + i] = r[src[s] + i]; if (GTx) std::copy(rx.begin() + (size_t)(src[s] + i) * nx, rx.begin() + (size_t)(src[s] + i + 1) * nx, GTx + (size_t)(dst[s] + i) * nx); }
      }
  }
};

struct TTraj : Traj {
  std::vector<double> ST, YT, GT, dST, dYT, LamT, dLamT;              // terminal slack / dual / residual (mT), multipliers (pT)
  std::vector<double> ky, Ky, ks, Ks;                                  // path gains of the last sweep (host-formed in the terminal-equality branch)
  std::vector<double> gGx;                                             // G_x of the last backward pass (computeScaledDualInfeasibility)
  double l_pr = 0.0, l_comp = 0.0;                                     // terminal (and, in the reduced-LQR branch, path) residual maxima of the last sweep
};
struct TTrial : Trial { std::vector<double> ST, YT, GT, LamT; };

void repair_interior(const cddp_hip_options &o, double *s, double *y, int dim) {   // repairWarmstartInterior (:233-262), one constraint object
  if (!o.ipddp_warmstart_repair || dim <= 0) return;
  double mn = kInf, mny = kInf;
  for (int i = 0; i < dim; ++i) { s[i] = std::max(s[i], o.ipddp_warmstart_s_min); mn = std::min(mn, s[i]); }
  if (mn < o.ipddp_warmstart_s_min * o.ipddp_warmstart_interior_factor) for (int i = 0; i < dim; ++i) s[i] = s[i] * o.ipddp_warmstart_interior_factor;
  for (int i = 0; i < dim; ++i) { y[i] = std::max(y[i], o.ipddp_warmstart_y_min); mny = std::min(mny, y[i]); }
  if (mny < o.ipddp_warmstart_y_min * o.ipddp_warmstart_interior_factor) for (int i = 0; i < dim; ++i) y[i] = y[i] * o.ipddp_warmstart_interior_factor;
}

// computeTheta / computeBarrierMerit / computePrimalAndComplementarity with the terminal terms (ipddp_solver.cpp:2778-2937): path objects
// (constraint-major, then t), then the terminal-inequality objects, then the stacked terminal-equality residual
void ip_reductions_t(const Ctx &c, const TermInfo &ti, const double *S, const double *Y, const double *G, const double *ST, const double *YT, const double *GT,
                     const double *LamT, const double *hT, double mu, double cost0, double &phi, double &theta, double &inf_pr, double &inf_comp) {
  const int m = c.m, N = c.N;
  const bool l2 = c.o->ipddp_theta_norm_l2 != 0;
  double total = 0.0, max_entry = 0.0, ipr = 0.0, icomp = 0.0, mer = cost0;
  int off = 0;
  for (int s = 0; s < (m > 0 ? c.pl->n_constraints : 0); ++s) {
    const int dim = c.pl->constraint_dims[s];
    for (int t = 0; t < N; ++t) {
      double n1 = 0.0, ninf = 0.0;
      for (int i = 0; i < dim; ++i) {
        const size_t j = (size_t)t * m + off + i;
        const double r = G[j] + S[j];
        n1 += l2 ? r * r : std::fabs(r);
        ninf = std::max(ninf, std::fabs(r));
      }
      total += n1; max_entry = std::max(max_entry, ninf);
    }
    off += dim;
  }
  for (int s = 0; s < ti.nobj; ++s) if (!ti.eq[s]) {
    double n1 = 0.0, ninf = 0.0;
    for (int i = 0; i < ti.dim[s]; ++i) { const double r = GT[ti.dst[s] + i] + ST[ti.dst[s] + i]; n1 += l2 ? r * r : std::fabs(r); ninf = std::max(ninf, std::fabs(r)); }
    total += n1; max_entry = std::max(max_entry, ninf);
  }
  if (ti.pT > 0) {
    double n1 = 0.0, ninf = 0.0;
    for (int i = 0; i < ti.pT; ++i) { n1 += l2 ? hT[i] * hT[i] : std::fabs(hT[i]); ninf = std::max(ninf, std::fabs(hT[i])); }
    total += n1; max_entry = std::max(max_entry, ninf);
  }
  off = 0;
  for (int s = 0; s < (m > 0 ? c.pl->n_constraints : 0); ++s) {
    const int dim = c.pl->constraint_dims[s];
    for (int t = 0; t < N; ++t) {
      double ls = 0.0;
      for (int i = 0; i < dim; ++i) ls += std::log(std::max(S[(size_t)t * m + off + i], kEpsSlack));
      mer -= mu * ls;
    }
    off += dim;
  }
  for (int s = 0; s < ti.nobj; ++s) if (!ti.eq[s]) {
    double ls = 0.0;
    for (int i = 0; i < ti.dim[s]; ++i) ls += std::log(std::max(ST[ti.dst[s] + i], kEpsSlack));
    mer -= mu * ls;
  }
  if (ti.pT > 0) { double dp = 0.0; for (int i = 0; i < ti.pT; ++i) dp += LamT[i] * hT[i]; mer += dp; }
  off = 0;
  for (int s = 0; s < (m > 0 ? c.pl->n_constraints : 0); ++s) {
    const int dim = c.pl->constraint_dims[s];
    for (int t = 0; t < N; ++t)
      for (int i = 0; i < dim; ++i) {
        const size_t j = (size_t)t * m + off + i;
        ipr = std::max(ipr, std::fabs(G[j] + S[j])); icomp = std::max(icomp, std::fabs(Y[j] * S[j] - mu));
      }
    off += dim;
  }
  for (int i = 0; i < ti.mT; ++i) { ipr = std::max(ipr, std::fabs(GT[i] + ST[i])); icomp = std::max(icomp, std::fabs(YT[i] * ST[i] - mu)); }
  for (int i = 0; i < ti.pT; ++i) ipr = std::max(ipr, std::fabs(hT[i]));
  const double th = l2 ? std::sqrt(total) : total;
  theta = std::max(th, max_entry);
  phi = mer; inf_pr = ipr; inf_comp = icomp;
}

// resetFilter (:2484-2519)
void reset_filter_t(const Ctx &c, const TermInfo &ti, TTraj &T, std::vector<double> &r, std::vector<double> &rx) {
  std::vector<double> hT((size_t)std::max(ti.pT, 1), 0.0);
  if (ti.pT > 0) ti.eval(c.pl->user, c.nx, T.X.data() + (size_t)c.N * c.nx, nullptr, nullptr, hT.data(), nullptr, r, rx);
  double phi, theta, ipr, icomp;
  ip_reductions_t(c, ti, T.S.data(), T.Y.data(), T.G.data(), T.ST.data(), T.YT.data(), T.GT.data(), T.LamT.data(), hT.data(), T.mu, T.cost, phi, theta, ipr, icomp);
  T.merit = T.phi = phi; T.inf_pr = ipr; T.inf_comp = icomp;
  T.filter_theta = std::max(theta, 1e-8);
  T.theta = std::max(T.filter_theta, std::max(c.o->ipddp_theta_0_floor, 1e-8));
  T.filter.clear();
  if (ti.mT > 0 || ti.pT > 0) filter_accept(T.filter, T.phi, T.filter_theta);
}

// IPDDPSolver::initialize: cold start (:819-913) or, with options.warm_start, "warm start with provided trajectory" (:733-816)
void initialize_t(const Ctx &c, const TermInfo &ti, TTraj &T, const double *x0, const double *U0, const double *X0) {
  const int nx = c.nx, nu = c.nu, N = c.N, m = c.m;
  const cddp_hip_options &o = *c.o;
  const bool warm = o.warm_start != 0;
  T.X.assign((size_t)(N + 1) * nx, 0.0); T.U.assign((size_t)N * nu, 0.0);
  if (U0) std::copy(U0, U0 + (size_t)N * nu, T.U.begin());
  (void)X0;   // both branches re-roll X out from the controls (:868-874, :771-777)
  std::copy(x0, x0 + nx, T.X.begin());
  T.reg = o.reg_initial_value; T.iter = 0; T.status = CDDP_HIP_STATUS_RUNNING; T.done = false; T.n_bwd = T.n_fwd = 0;
  T.dV0 = T.dV1 = 0.0; T.step_norm = 0.0; T.filter.clear(); T.alpha_pr = T.alpha_du = 1.0;
  T.S.assign((size_t)N * m, 0.0); T.Y = T.S; T.G = T.S; T.Lam.assign((size_t)(N + 1) * nx, 0.0);
  T.ST.assign((size_t)std::max(ti.mT, 1), 0.0); T.YT = T.ST; T.GT = T.ST; T.dST = T.ST; T.dYT = T.ST;
  T.LamT.assign((size_t)std::max(ti.pT, 1), 0.0); T.dLamT = T.LamT;
  std::vector<double> r, rx, xn(nx);
  double cost = 0.0;
  for (int t = 0; t < N; ++t) {   // evaluateTrajectory (:2252-2296) / re-rollout + evaluateTrajectoryWarmStart (:2296-2343): same numbers
    double *x = T.X.data() + (size_t)t * nx, *u = T.U.data() + (size_t)t * nu;
    cost += c.pl->running_cost(c.pl->user, x, u, t);
    if (m > 0) c.pl->constraints(c.pl->user, x, u, t, T.G.data() + (size_t)t * m, nullptr, nullptr);
    c.pl->discrete_dynamics(c.pl->user, x, u, t * c.dt, xn.data());
    std::copy(xn.begin(), xn.end(), T.X.begin() + (size_t)(t + 1) * nx);
  }
  cost += c.pl->terminal_cost(c.pl->user, T.X.data() + (size_t)N * nx);
  T.cost = cost;
  if (ti.mT > 0) ti.eval(c.pl->user, nx, T.X.data() + (size_t)N * nx, T.GT.data(), nullptr, nullptr, nullptr, r, rx);
  if (!warm) T.mu = o.barrier_mu_initial;   // (a terminal set exists: never the unconstrained value)
  else {   // :779-799: barrier parameter from the seed's largest constraint value
    double mv = 0.0;
    for (size_t j = 0; j < T.G.size(); ++j) mv = std::max(mv, T.G[j]);
    for (int i = 0; i < ti.mT; ++i) mv = std::max(mv, T.GT[i]);
    if (mv <= o.tolerance) T.mu = std::max(o.tolerance, o.barrier_mu_min_value);
    else if (mv <= 0.1) T.mu = std::max(o.tolerance * 10.0, o.barrier_mu_initial * 0.01);
    else T.mu = o.barrier_mu_initial * 0.1;
  }
  {   // initializeDualSlackVariables (:2428-2482) / ...WarmStart without existing duals (:2345-2426)
    int off = 0;
    for (int s = 0; s < (m > 0 ? c.pl->n_constraints : 0); ++s) {
      const int dim = c.pl->constraint_dims[s];
      for (int t = 0; t < N; ++t) {
        double *g = T.G.data() + (size_t)t * m + off, *sv = T.S.data() + (size_t)t * m + off, *y = T.Y.data() + (size_t)t * m + off;
        for (int i = 0; i < dim; ++i) { sv[i] = std::max(o.ipddp_slack_var_init_scale, -g[i] + kSlackOffset); y[i] = (T.mu * o.ipddp_dual_var_init_scale) / std::max(sv[i], kEpsSlack); }
        repair_interior(o, sv, y, dim);
      }
      off += dim;
    }
    for (int s = 0; s < ti.nobj; ++s) if (!ti.eq[s]) {   // :889-908 / initializeTerminalWarmstartDualSlack (:294-353)
      double *g = T.GT.data() + ti.dst[s], *sv = T.ST.data() + ti.dst[s], *y = T.YT.data() + ti.dst[s];
      for (int i = 0; i < ti.dim[s]; ++i) { sv[i] = std::max(o.ipddp_slack_var_init_scale, -g[i] + kSlackOffset); y[i] = (T.mu * o.ipddp_dual_var_init_scale) / std::max(sv[i], kEpsSlack); }
      repair_interior(o, sv, y, ti.dim[s]);
    }
  }
  reset_filter_t(c, ti, T, r, rx);
  T.inf_du = 0.0;
}

// forwardPass with terminal sets (ipddp_solver.cpp:1571-1876)
TTrial forward_ipddp_t(const Ctx &c, const TermInfo &ti, const TTraj &T, const Gains &g, double alpha, const double *GTx0) {
  const int nx = c.nx, nu = c.nu, N = c.N, m = c.m;
  const cddp_hip_options &o = *c.o;
  const bool hti = ti.mT > 0, hte = ti.pT > 0;
  TTrial r;
  const double mu = T.mu;
  const double tau = (m == 0 && !hti) ? 1.0 : std::max(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  const double a_pr = std::min(alpha, T.apr_max), a_du = std::min(alpha, T.adu_max);
  r.alpha_pr = a_pr; r.alpha_du = a_du;
  r.cost = T.cost; r.merit = T.phi; r.theta = T.theta;
  r.X.assign(T.X.size(), 0.0); r.U.assign(T.U.size(), 0.0); r.Lam.assign(T.Lam.size(), 0.0);
  r.S = T.S; r.Y = T.Y; r.G.assign(T.S.size(), 0.0);
  r.ST = T.ST; r.YT = T.YT; r.GT = T.GT; r.LamT = T.LamT;
  std::copy(T.X.begin(), T.X.begin() + nx, r.X.begin());
  std::vector<double> dx(nx), rr, rrx;
  for (int t = 0; t <= N; ++t) {
    const double *x = r.X.data() + (size_t)t * nx, *xo = T.X.data() + (size_t)t * nx;
    for (int i = 0; i < nx; ++i) dx[i] = x[i] - xo[i];
    for (int i = 0; i < nx; ++i) {
      double s = 0.0;
      for (int j = 0; j < nx; ++j) s += g.Vxx[((size_t)t * nx + i) * nx + j] * dx[j];
      const double lam = (T.Lam[(size_t)t * nx + i] + a_pr * g.Vx[(size_t)t * nx + i]) + s;
      if (!fin(lam)) return r;
      r.Lam[(size_t)t * nx + i] = lam;
    }
    if (t == N) break;
    for (int q = 0; q < m; ++q) {
      const size_t j = (size_t)t * m + q;
      double ps = 0.0, py = 0.0;
      for (int i = 0; i < nx; ++i) { ps = ps + g.Ks[j * nx + i] * dx[i]; py = py + g.Ky[j * nx + i] * dx[i]; }
      const double sn = (T.S[j] + a_pr * g.ks[j]) + ps;
      const double yn = (T.Y[j] + a_du * g.ky[j]) + py;
      if (sn < (1.0 - tau) * T.S[j] || yn < (1.0 - tau) * T.Y[j]) return r;
      if (!fin(sn) || !fin(yn)) return r;
      r.S[j] = sn; r.Y[j] = yn;
    }
    const double *uo = T.U.data() + (size_t)t * nu;
    double *u = r.U.data() + (size_t)t * nu;
    for (int i = 0; i < nu; ++i) {
      double s = 0.0;
      for (int j = 0; j < nx; ++j) s += g.K[((size_t)t * nu + i) * nx + j] * dx[j];
      u[i] = (uo[i] + a_pr * g.k[(size_t)t * nu + i]) + s;
    }
    double *xn = r.X.data() + (size_t)(t + 1) * nx;
    c.pl->discrete_dynamics(c.pl->user, x, u, t * c.dt, xn);
    for (int i = 0; i < nx; ++i) if (!fin(xn[i])) return r;
    for (int i = 0; i < nu; ++i) if (!fin(u[i])) return r;
  }
  // dx holds x_N' - x_N here
  if (hti) {   // terminal slack / dual trial (:1667-1714); Jacobian and residual of the CURRENT iterate's x_N
    const double floor0 = std::max(mu * 1e-3, kEpsSlack);
    for (int s = 0; s < ti.nobj; ++s) if (!ti.eq[s]) {
      for (int i = 0; i < ti.dim[s]; ++i) {
        const int j = ti.dst[s] + i;
        const double ksT = -(T.GT[j] + T.ST[j]);
        double kd = 0.0;
        for (int k = 0; k < nx; ++k) kd += (-GTx0[(size_t)j * nx + k]) * dx[k];                 // K_s_T * dx
        r.ST[j] = (T.ST[j] + a_pr * ksT) + kd;
        const double s_safe = std::max(T.ST[j], floor0);
        const double r_d = T.YT[j] * T.ST[j] - mu;
        const double dual_ratio = clampd(T.YT[j] / s_safe, 0.0, kMaxRatio);
        const double kyv = clampd((-r_d - T.YT[j] * ksT) / s_safe, -kMaxRatio, kMaxRatio);
        double dotv = 0.0;
        for (int k = 0; k < nx; ++k) dotv += (-(dual_ratio * (-GTx0[(size_t)j * nx + k]))) * dx[k];
        r.YT[j] = (T.YT[j] + a_du * kyv) + dotv;
      }
      for (int i = 0; i < ti.dim[s]; ++i) {
        const int j = ti.dst[s] + i;
        const double s_floor = std::max((1.0 - tau) * T.ST[j], floor0);
        if (r.ST[j] < s_floor || r.YT[j] < (1.0 - tau) * T.YT[j]) return r;
      }
      for (int i = 0; i < ti.dim[s]; ++i) if (!fin(r.ST[ti.dst[s] + i]) || !fin(r.YT[ti.dst[s] + i])) return r;
    }
  }
  if (hte) for (int i = 0; i < ti.pT; ++i) { r.LamT[i] = T.LamT[i] + a_pr * T.dLamT[i]; if (!fin(r.LamT[i])) return r; }
  double cost_new = 0.0;
  for (int t = 0; t < N; ++t) {
    const double *x = r.X.data() + (size_t)t * nx, *u = r.U.data() + (size_t)t * nu;
    cost_new += c.pl->running_cost(c.pl->user, x, u, t);
    if (m > 0) c.pl->constraints(c.pl->user, x, u, t, r.G.data() + (size_t)t * m, nullptr, nullptr);
  }
  cost_new += c.pl->terminal_cost(c.pl->user, r.X.data() + (size_t)N * nx);
  std::vector<double> hTn((size_t)std::max(ti.pT, 1), 0.0);
  ti.eval(c.pl->user, nx, r.X.data() + (size_t)N * nx, hti ? r.GT.data() : nullptr, nullptr, hte ? hTn.data() : nullptr, nullptr, rr, rrx);
  double phi_new, theta_new, ipr, icomp;
  ip_reductions_t(c, ti, r.S.data(), r.Y.data(), r.G.data(), r.ST.data(), r.YT.data(), r.GT.data(), r.LamT.data(), hTn.data(), mu, cost_new, phi_new, theta_new, ipr, icomp);
  if (!fin(phi_new) || !fin(theta_new) || !fin(ipr) || !fin(icomp)) return r;
  bool accept = false;
  {   // filter acceptance (:1793-1834): a terminal set exists
    const double expected_improvement = a_pr * T.dV0;
    const bool fe = T.filter.empty();
    const double cv_old = fe ? 0.0 : T.filter.back().second;
    const double high_ref = fe ? T.filter_theta : cv_old;
    const double merit_old = T.merit;
    if (theta_new > o.filter_max_violation_threshold) {
      if (theta_new < (1 - o.filter_violation_acceptance_threshold) * high_ref) accept = true;
    } else if (std::max(theta_new, cv_old) < o.filter_min_violation_for_armijo_check && expected_improvement < 0) {
      if (phi_new < merit_old + o.filter_armijo_constant * expected_improvement) accept = true;
    } else {
      if (phi_new < merit_old - o.filter_merit_acceptance_threshold * theta_new ||
          theta_new < (1 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
    }
  }
  r.cost = cost_new; r.merit = phi_new; r.theta = theta_new; r.inf_pr = ipr; r.inf_comp = icomp;
  r.success = accept;
  return r;
}

int ipddp_terminal_solve(const Ctx &c, const TermInfo &ti, int device, int batch, const double *x0, const double *U0, const double *X0,
                         cddp_hip_result *results, double *Xout, double *Uout, double *Kout, double *Tout) {
  const cddp_hip_plugin *pl = c.pl; const cddp_hip_options &o = *c.o; const cddp_hip_options *opt = c.o;
  const int nx = c.nx, nu = c.nu, m = c.m, N = c.N, mT = ti.mT, pT = ti.pT; const double dt = c.dt;
  const bool hti = mT > 0, hte = pT > 0, no_barrier = (m == 0 && !hti);
  const size_t B = (size_t)batch;
  // terminal-equality rows: the reduced LQR takes the path constraints condensed (handle with m = 0); otherwise the ordinary sweeps
  cddp_hip_stack_handle *sh = nullptr;
  { int rc = cddp_hip_stacks_create(device, batch, nx, nu, hte ? 0 : m, N, &sh); if (rc) return rc; }
  struct Guard { cddp_hip_stack_handle *h; ~Guard() { if (h) cddp_hip_stacks_destroy(h); } } guard{sh};
  std::vector<TTraj> T(B);
  const int n_threads = host_threads((int)B);
  par_for(B, n_threads, [&](size_t b) { initialize_t(c, ti, T[b], x0 + b * nx, U0 ? U0 + b * N * nu : nullptr, X0 ? X0 + b * (N + 1) * nx : nullptr); });

  std::vector<double> fx(B * N * nx * nx), fu(B * N * nx * nu), lx(B * N * nx), lu(B * N * nu), lxx(B * N * nx * nx), luu(B * N * nu * nu),
      lux(B * N * nu * nx), VxN(B * nx), VxxN(B * nx * nx);
  std::vector<double> gy, gs, gg, gGx, gGu, Fxx, Fuu, Fux;
  if (m > 0) { gy.resize(B * N * m); gs = gy; gg = gy; gGx.resize(B * N * m * nx); gGu.resize(B * N * m * nu); }
  if (!o.use_ilqr) { Fxx.resize(B * N * nx * nx * nx); Fuu.resize(B * N * nx * nu * nu); Fux.resize(B * N * nx * nu * nx); }
  std::vector<double> Kb(B * N * nu * nx), kb(B * N * nu), Vxb(B * (N + 1) * nx), Vxxb(B * (N + 1) * nx * nx), dVb(B * 2);
  std::vector<double> Kfin(Kout ? B * N * nu * nx : 0, 0.0);
  std::vector<double> kyb, Kyb, ksb, Ksb, dXb(B * (N + 1) * nx, 0.0);
  if (m > 0) { kyb.resize(B * N * m); ksb = kyb; Kyb.resize(B * N * m * nx); Ksb = Kyb; }
  std::vector<double> GTxb(B * (size_t)std::max(mT, 1) * nx, 0.0), HTb(B * (size_t)std::max(pT, 1) * nx, 0.0), bTb(B * (size_t)std::max(pT, 1), 0.0),
      lamb(B * (size_t)std::max(pT, 1), 0.0), floorb(B, 0.0), dlamb(B * (size_t)std::max(pT, 1), 0.0);
  std::vector<double> regv(B), muv(B), s_reg(B), s_du(B), s_pr(B), s_comp(B), s_sn(B), s_apr(B), s_adu(B);
  std::vector<int32_t> okv(B);
  const bool first_rule = !o.enable_parallel;
  std::atomic<bool> abort_seen{false};
  const auto wall0 = std::chrono::steady_clock::now();

  for (int it = 1; it <= o.max_iterations; ++it) {
    bool any = false;
    for (auto &t : T) any = any || !t.done;
    if (!any) break;
    if (aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
    if (o.max_cpu_time > 0.0) {
      const double el_ms = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count();
      if (el_ms > o.max_cpu_time * 1000.0) { for (auto &t : T) if (!t.done) { t.iter += 1; t.status = CDDP_HIP_STATUS_MAX_CPU_TIME; t.done = true; } break; }
    }
    // ---- host: derivatives, terminal value, LQ model
    auto fill = [&](size_t b) {
      TTraj &t = T[b];
      regv[b] = t.done ? std::max(t.reg, o.reg_min_value) : t.reg;
      muv[b] = (t.mu > 0.0) ? t.mu : 1.0;
      floorb[b] = std::max(1e-10, o.ipddp_jacobian_regularization_value * std::pow(std::max(t.mu, 0.0), o.ipddp_jacobian_regularization_exponent));
      if (t.done) return;
      t.iter += 1;
      const double mu = t.mu, fl0 = std::max(mu * 1e-3, kEpsSlack);
      std::vector<double> tfx((size_t)nx * nx), tfu((size_t)nx * nu), r, rx, hT((size_t)std::max(pT, 1), 0.0);
      double *Vx = VxN.data() + b * nx, *Vxx = VxxN.data() + b * nx * nx;
      const double *xN = t.X.data() + (size_t)N * nx;
      pl->terminal_cost_derivatives(pl->user, xN, Vx, Vxx);
      { std::vector<double> S2((size_t)nx * nx); for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) S2[i * nx + j] = 0.5 * (Vxx[i * nx + j] + Vxx[j * nx + i]); std::copy(S2.begin(), S2.end(), Vxx); }
      t.l_pr = 0.0; t.l_comp = 0.0;
      double *GTx = GTxb.data() + b * (size_t)std::max(mT, 1) * nx, *HT = HTb.data() + b * (size_t)std::max(pT, 1) * nx;
      ti.eval(pl->user, nx, xN, hti ? t.GT.data() : nullptr, hti ? GTx : nullptr, hte ? hT.data() : nullptr, hte ? HT : nullptr, r, rx);
      for (int s = 0; s < ti.nobj; ++s) if (!ti.eq[s]) {   // terminal-inequality barrier terms (:1000-1031), object by object
        const int d0 = ti.dst[s], dim = ti.dim[s];
        std::vector<double> sig(dim), bg(dim);
        for (int i = 0; i < dim; ++i) {
          const double s_safe = std::max(t.ST[d0 + i], fl0), y_safe = std::max(t.YT[d0 + i], kEpsDual);
          sig[i] = clampd(y_safe / s_safe, 0.0, kMaxRatio);
          bg[i] = y_safe + clampd((y_safe * t.GT[d0 + i] + mu) / s_safe, -kMaxRatio, kMaxRatio);
        }
        for (int j = 0; j < nx; ++j) { double a = 0.0; for (int i = 0; i < dim; ++i) a += GTx[(size_t)(d0 + i) * nx + j] * bg[i]; Vx[j] += a; }
        for (int j = 0; j < nx; ++j) for (int k = 0; k < nx; ++k) { double a = 0.0; for (int i = 0; i < dim; ++i) a += (GTx[(size_t)(d0 + i) * nx + j] * sig[i]) * GTx[(size_t)(d0 + i) * nx + k]; Vxx[j * nx + k] += a; }
        { std::vector<double> S2((size_t)nx * nx); for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) S2[i * nx + j] = 0.5 * (Vxx[i * nx + j] + Vxx[j * nx + i]); std::copy(S2.begin(), S2.end(), Vxx); }
        for (int i = 0; i < dim; ++i) { t.l_pr = std::max(t.l_pr, std::fabs(t.GT[d0 + i] + t.ST[d0 + i])); t.l_comp = std::max(t.l_comp, std::fabs(t.YT[d0 + i] * t.ST[d0 + i] - mu)); }
      }
      if (hte) {
        for (int i = 0; i < pT; ++i) { t.l_pr = std::max(t.l_pr, std::fabs(hT[i])); bTb[b * pT + i] = -hT[i]; lamb[b * pT + i] = t.LamT[i]; t.dLamT[i] = -hT[i]; }
      }
      if (m > 0) t.gGx.assign((size_t)N * m * nx, 0.0);
      for (int s = 0; s < N; ++s) {
        const double *x = t.X.data() + (size_t)s * nx, *u = t.U.data() + (size_t)s * nu;
        const size_t bs = b * N + s;
        pl->jacobians(pl->user, x, u, s * dt, tfx.data(), tfu.data());
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double a = dt * tfx[i * nx + j]; if (i == j) a += 1.0; fx[(bs * nx + i) * nx + j] = a; }
        for (int i = 0; i < nx * nu; ++i) fu[bs * nx * nu + i] = dt * tfu[i];
        double *q = lx.data() + bs * nx, *rr = lu.data() + bs * nu, *Q = lxx.data() + bs * nx * nx, *R = luu.data() + bs * nu * nu, *Mx = lux.data() + bs * nu * nx;
        pl->running_cost_derivatives(pl->user, x, u, s, q, rr, Q, R, Mx);
        if (!o.use_ilqr) {
          double *pxx = Fxx.data() + bs * nx * nx * nx, *puu = Fuu.data() + bs * nx * nu * nu, *pux = Fux.data() + bs * nx * nu * nx;
          pl->hessians(pl->user, x, u, s * dt, pxx, puu, pux);
          for (int e = 0; e < nx * nx * nx; ++e) pxx[e] = dt * pxx[e];
          for (int e = 0; e < nx * nu * nu; ++e) puu[e] = dt * puu[e];
          for (int e = 0; e < nx * nu * nx; ++e) pux[e] = dt * pux[e];
        }
        if (m > 0) {
          pl->constraints(pl->user, x, u, s, gg.data() + bs * m, gGx.data() + bs * m * nx, gGu.data() + bs * m * nu);
          std::copy(t.G.begin() + (size_t)s * m, t.G.begin() + (size_t)(s + 1) * m, gg.begin() + bs * m);
          std::copy(t.S.begin() + (size_t)s * m, t.S.begin() + (size_t)(s + 1) * m, gs.begin() + bs * m);
          std::copy(t.Y.begin() + (size_t)s * m, t.Y.begin() + (size_t)(s + 1) * m, gy.begin() + bs * m);
          std::copy(gGx.begin() + bs * m * nx, gGx.begin() + (bs + 1) * m * nx, t.gGx.begin() + (size_t)s * m * nx);
        }
        if (!hte) continue;
        // ---- LQ model of the reduced LQR (:1143-1247): Q = sym(l_xx), R = sym(l_uu), M = l_ux^T, (+ second-order terms weighed with the
        //      costate iterate, :1160-1178), path constraints condensed (:1180-1245); the regularisation is added by the sweep
        std::vector<double> Qs((size_t)nx * nx), Rs((size_t)nu * nu), Mm((size_t)nx * nu);
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) Qs[i * nx + j] = 0.5 * (Q[i * nx + j] + Q[j * nx + i]);
        for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) Rs[i * nu + j] = 0.5 * (R[i * nu + j] + R[j * nu + i]);
        for (int a = 0; a < nu; ++a) for (int cix = 0; cix < nx; ++cix) Mm[cix * nu + a] = Mx[a * nx + cix];
        if (!o.use_ilqr) {
          const double *lam = t.Lam.data() + (size_t)(s + 1) * nx;
          bool lf = true; for (int i = 0; i < nx; ++i) lf = lf && fin(lam[i]);
          const double *pxx = Fxx.data() + bs * nx * nx * nx, *puu = Fuu.data() + bs * nx * nu * nu, *pux = Fux.data() + bs * nx * nu * nx;
          for (int i = 0; i < nx; ++i) {
            const double li = lf ? lam[i] : 0.0;
            for (int e = 0; e < nx * nx; ++e) Qs[e] = Qs[e] + li * pxx[(size_t)i * nx * nx + e];
            for (int a = 0; a < nu; ++a) for (int cix = 0; cix < nx; ++cix) Mm[cix * nu + a] = Mm[cix * nu + a] + li * pux[(size_t)i * nu * nx + a * nx + cix];
            for (int e = 0; e < nu * nu; ++e) Rs[e] = Rs[e] + li * puu[(size_t)i * nu * nu + e];
          }
          std::vector<double> Q2(Qs), R2(Rs);
          for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) Qs[i * nx + j] = 0.5 * (Q2[i * nx + j] + Q2[j * nx + i]);
          for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) Rs[i * nu + j] = 0.5 * (R2[i * nu + j] + R2[j * nu + i]);
        }
        if (m > 0) {
          const double *y = gy.data() + bs * m, *sv = gs.data() + bs * m, *g = gg.data() + bs * m, *Qyx = gGx.data() + bs * m * nx, *Qyu = gGu.data() + bs * m * nu;
          std::vector<double> YS(m), ypS(m);
          for (int i = 0; i < m; ++i) {
            const double ss = std::max(sv[i], fl0);
            YS[i] = clampd(y[i] / ss, 0.0, kMaxRatio);
            const double rp = g[i] + sv[i], rc = y[i] * sv[i] - mu, rhat = y[i] * rp - rc;
            ypS[i] = y[i] + clampd(rhat / ss, -kMaxRatio, kMaxRatio);
            t.l_pr = std::max(t.l_pr, std::fabs(rp)); t.l_comp = std::max(t.l_comp, std::fabs(rc));
          }
          for (int i = 0; i < nx; ++i) { double a = 0.0; for (int r2 = 0; r2 < m; ++r2) a += Qyx[r2 * nx + i] * ypS[r2]; q[i] += a; }
          for (int i = 0; i < nu; ++i) { double a = 0.0; for (int r2 = 0; r2 < m; ++r2) a += Qyu[r2 * nu + i] * ypS[r2]; rr[i] += a; }
          std::vector<double> Qn(Qs), Rn(Rs);
          for (int i = 0; i < nx; ++i) for (int cix = 0; cix < nx; ++cix) { double a = 0.0; for (int r2 = 0; r2 < m; ++r2) a += (Qyx[r2 * nx + i] * YS[r2]) * Qyx[r2 * nx + cix]; Qn[i * nx + cix] = Qs[i * nx + cix] + a; }
          for (int i = 0; i < nu; ++i) for (int cix = 0; cix < nx; ++cix) { double a = 0.0; for (int r2 = 0; r2 < m; ++r2) a += (Qyu[r2 * nu + i] * YS[r2]) * Qyx[r2 * nx + cix]; Mm[cix * nu + i] += a; }
          for (int i = 0; i < nu; ++i) for (int cix = 0; cix < nu; ++cix) { double a = 0.0; for (int r2 = 0; r2 < m; ++r2) a += (Qyu[r2 * nu + i] * YS[r2]) * Qyu[r2 * nu + cix]; Rn[i * nu + cix] = Rs[i * nu + cix] + a; }
          for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) Qs[i * nx + j] = 0.5 * (Qn[i * nx + j] + Qn[j * nx + i]);
          for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) Rs[i * nu + j] = 0.5 * (Rn[i * nu + j] + Rn[j * nu + i]);
        }
        std::copy(Qs.begin(), Qs.end(), Q); std::copy(Rs.begin(), Rs.end(), R); std::copy(Mm.begin(), Mm.end(), Mx);   // (the lux slot carries M as nx x nu)
      }
    };
    par_for(B, n_threads, fill);
    { int rc = cddp_hip_set_stacks(sh, fx.data(), fu.data(), lx.data(), lu.data(), lxx.data(), luu.data(), lux.data(), VxN.data(), VxxN.data()); if (rc) return rc; }
    if (hte) {
      { int rc = cddp_hip_set_terminal_equality(sh, pT, HTb.data(), bTb.data(), lamb.data(), floorb.data()); if (rc) return rc; }
      { int rc = cddp_hip_stacks_backward(sh, CDDP_HIP_STACKS_IPDDP_TERM_EQ, opt, regv.data(), nullptr, 1, okv.data()); if (rc) return rc; }
      { int rc = cddp_hip_stacks_get_terminal(sh, dlamb.data(), dXb.data()); if (rc) return rc; }
    } else {
      if (m > 0) { int rc = cddp_hip_set_constraint_stacks(sh, gy.data(), gs.data(), gg.data(), gGx.data(), gGu.data()); if (rc) return rc; }
      if (!o.use_ilqr) { int rc = cddp_hip_set_hessian_stacks(sh, Fxx.data(), Fuu.data(), Fux.data()); if (rc) return rc; }
      { int rc = cddp_hip_stacks_backward(sh, m > 0 ? CDDP_HIP_STACKS_IPDDP_PATH : CDDP_HIP_STACKS_IPDDP, opt, regv.data(), m > 0 ? muv.data() : nullptr, 1, okv.data()); if (rc) return rc; }
      if (m > 0) { int rc = cddp_hip_stacks_get_constraint_gains(sh, kyb.data(), Kyb.data(), ksb.data(), Ksb.data(), dXb.data()); if (rc) return rc; }
    }
    { int rc = cddp_hip_stacks_get_gains(sh, Kb.data(), kb.data(), Vxb.data(), Vxxb.data(), dVb.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_get_scalars(sh, s_reg.data(), s_du.data(), s_pr.data(), s_comp.data(), s_sn.data(), s_apr.data(), s_adu.data()); if (rc) return rc; }
    if (Kout) for (size_t b = 0; b < B; ++b) if (!T[b].done) std::copy(Kb.begin() + b * N * nu * nx, Kb.begin() + (b + 1) * N * nu * nx, Kfin.begin() + b * N * nu * nx);

    auto advance = [&](size_t b) {
      TTraj &t = T[b];
      if (t.done) return;
      if (aborted(pl)) { abort_seen.store(true); return; }
      { int nb = 1; double r = t.reg; while (r < s_reg[b] && nb < 64) { r = reg_increase(o, r); ++nb; }
        if (!okv[b] && nb > 1) --nb;
        t.n_bwd += nb; }
      t.reg = s_reg[b];
      if (!okv[b]) { t.status = CDDP_HIP_STATUS_REG_LIMIT; t.done = true; return; }
      const double mu = t.mu, fl0 = std::max(mu * 1e-3, kEpsSlack);
      const double tau = std::max(o.barrier_min_fraction_to_boundary, 1.0 - mu);
      const double *Kt = Kb.data() + b * N * nu * nx, *kt = kb.data() + b * N * nu;
      double *dX = dXb.data() + b * (N + 1) * nx;
      double apr = 1.0, adu = 1.0;
      t.inf_du = s_du[b]; t.step_norm = s_sn[b];
      if (hte) {
        t.dV0 = 0.0; t.dV1 = 0.0;
        t.inf_pr = t.l_pr; t.inf_comp = t.l_comp;
        for (int i = 0; i < pT; ++i) t.dLamT[i] = dlamb[b * pT + i];
        if (m > 0) {   // slack / dual gains and directions from the final K, k and dX (:1270-1312)
          t.ky.assign((size_t)N * m, 0.0); t.ks = t.ky; t.Ky.assign((size_t)N * m * nx, 0.0); t.Ks = t.Ky;
          for (int s = 0; s < N; ++s) {
            const size_t bs = b * N + s;
            const double *y = gy.data() + bs * m, *sv = gs.data() + bs * m, *g = gg.data() + bs * m, *Qyx = gGx.data() + bs * m * nx, *Qyu = gGu.data() + bs * m * nu;
            for (int r2 = 0; r2 < m; ++r2) {
              const double ss = std::max(sv[r2], fl0), YSr = clampd(y[r2] / ss, 0.0, kMaxRatio);
              const double rp = g[r2] + sv[r2], rc = y[r2] * sv[r2] - mu, rhat = y[r2] * rp - rc;
              double temp = 0.0;
              for (int i = 0; i < nu; ++i) temp += Qyu[r2 * nu + i] * kt[(size_t)s * nu + i];
              const double kyv = clampd((rhat + y[r2] * temp) / ss, -kMaxRatio, kMaxRatio), ksv = (-rp) - temp;
              double a = 0.0, cc = 0.0;
              for (int j = 0; j < nx; ++j) {
                double s2 = 0.0;
                for (int i = 0; i < nu; ++i) s2 += Qyu[r2 * nu + i] * Kt[((size_t)s * nu + i) * nx + j];
                const double inner = Qyx[r2 * nx + j] + s2;
                const double Kyv = std::min(std::max(YSr * inner, -kMaxRatio), kMaxRatio), Ksv = (-Qyx[r2 * nx + j]) - s2;
                t.Ky[((size_t)s * m + r2) * nx + j] = Kyv; t.Ks[((size_t)s * m + r2) * nx + j] = Ksv;
                a += Ksv * dX[(size_t)s * nx + j]; cc += Kyv * dX[(size_t)s * nx + j];
              }
              t.ky[(size_t)s * m + r2] = kyv; t.ks[(size_t)s * m + r2] = ksv;
              const double ds = ksv + a, dy = std::min(std::max(kyv + cc, -kMaxRatio), kMaxRatio);
              if (ds < 0.0) apr = std::min(apr, -tau * sv[r2] / ds);
              if (dy < 0.0) adu = std::min(adu, -tau * y[r2] / dy);
            }
          }
        }
      } else {
        t.dV0 = dVb[b * 2]; t.dV1 = dVb[b * 2 + 1];
        t.inf_pr = std::max((m > 0) ? s_pr[b] : 0.0, t.l_pr); t.inf_comp = std::max((m > 0) ? s_comp[b] : 0.0, t.l_comp);
        if (m > 0) {
          apr = s_apr[b]; adu = s_adu[b];
          t.ky.assign(kyb.begin() + b * N * m, kyb.begin() + (b + 1) * N * m); t.ks.assign(ksb.begin() + b * N * m, ksb.begin() + (b + 1) * N * m);
          t.Ky.assign(Kyb.begin() + b * N * m * nx, Kyb.begin() + (b + 1) * N * m * nx); t.Ks.assign(Ksb.begin() + b * N * m * nx, Ksb.begin() + (b + 1) * N * m * nx);
        } else {   // rolloutLinearPolicy on the host (:1511-1520): the unconstrained stack sweep does not form dX
          std::vector<double> dx(nx, 0.0), dxn(nx), du(nu);
          for (int s = 0; s < N; ++s) {
            std::copy(dx.begin(), dx.end(), dX + (size_t)s * nx);
            const size_t bs = b * N + s;
            for (int i = 0; i < nu; ++i) { double a = 0.0; for (int j = 0; j < nx; ++j) a += Kt[((size_t)s * nu + i) * nx + j] * dx[j]; du[i] = kt[(size_t)s * nu + i] + a; }
            for (int i = 0; i < nx; ++i) { double a = 0.0, cc = 0.0; for (int j = 0; j < nx; ++j) a += fx[(bs * nx + i) * nx + j] * dx[j]; for (int j = 0; j < nu; ++j) cc += fu[(bs * nx + i) * nu + j] * du[j]; dxn[i] = (a + cc) + 0.0; }
            dx = dxn;
          }
          std::copy(dx.begin(), dx.end(), dX + (size_t)N * nx);
        }
      }
      const double *GTx = GTxb.data() + b * (size_t)std::max(mT, 1) * nx;
      for (int i = 0; i < mT; ++i) {   // terminal-inequality directions (:1315-1346 == :1534-1561) and their step caps (:2939-2988)
        const double r_p = t.GT[i] + t.ST[i], r_d = t.ST[i] * t.YT[i] - mu;
        double gd = 0.0;
        for (int j = 0; j < nx; ++j) gd += GTx[(size_t)i * nx + j] * dX[(size_t)N * nx + j];
        const double dsT = (-r_p) - gd;
        const double s_safe = std::max(t.ST[i], fl0);
        const double dual_ratio = clampd(t.YT[i] / s_safe, 0.0, kMaxRatio), affine = clampd(-r_d / s_safe, -kMaxRatio, kMaxRatio);
        const double dyT = clampd(affine - dual_ratio * dsT, -kMaxRatio, kMaxRatio);
        t.dST[i] = dsT; t.dYT[i] = dyT;
        if (dsT < 0.0) apr = std::min(apr, -tau * t.ST[i] / dsT);
        if (dyT < 0.0) adu = std::min(adu, -tau * t.YT[i] / dyT);
      }
      t.apr_max = clampd(apr, 0.0, 1.0); t.adu_max = clampd(adu, 0.0, 1.0);
      Gains g;
      g.K = Kt; g.k = kt; g.Vx = Vxb.data() + b * (N + 1) * nx; g.Vxx = Vxxb.data() + b * (N + 1) * nx * nx;
      g.ky = t.ky.data(); g.Ky = t.Ky.data(); g.ks = t.ks.data(); g.Ks = t.Ks.data();
      // ---- checkEarlyConvergence (:925-958)
      bool conv = false;
      {
        const double sdu = scaled_inf_du(c, t, t.gGx);
        if (no_barrier) conv = (t.inf_pr < o.tolerance && sdu < o.tolerance);
        else {
          const double tol = std::max(o.tolerance, o.ipddp_barrier_tol_mult * t.mu);
          conv = (t.inf_pr < tol && sdu < tol && t.inf_comp < tol && std::fabs(t.alpha_pr) * t.step_norm < o.tolerance * 10.0);
        }
      }
      if (conv) { t.status = CDDP_HIP_STATUS_OPTIMAL; t.done = true; return; }
      TTrial best; bool have = false;
      int walked = 0;
      for (double a : c.alphas) {
        TTrial r = forward_ipddp_t(c, ti, t, g, a, GTx);
        ++walked;
        if (!r.success) continue;
        if (first_rule) { best = std::move(r); have = true; break; }
        if (!have || r.merit < best.merit) { best = std::move(r); have = true; }
      }
      t.n_fwd += first_rule ? walked : (int)c.alphas.size();
      if (have) {
        const double dJ = t.cost - best.cost;
        t.X.swap(best.X); t.U.swap(best.U); t.Lam.swap(best.Lam);
        t.cost = best.cost; t.merit = best.merit; t.alpha_pr = best.alpha_pr; t.alpha_du = best.alpha_du;
        if (m > 0) { t.S.swap(best.S); t.Y.swap(best.Y); t.G.swap(best.G); }
        if (hti) { t.ST = best.ST; t.YT = best.YT; t.GT = best.GT; }
        if (hte) t.LamT = best.LamT;
        t.inf_pr = best.inf_pr; t.inf_comp = best.inf_comp; t.phi = best.merit; t.filter_theta = best.theta; t.theta = best.theta;
        // ---- updateBarrierParameters(true) (:2548-2660)
        const double sdu = scaled_inf_du(c, t, t.gGx);
        double mu2 = t.mu; const double mu_old = mu2;
        if (!no_barrier) {
          if (o.barrier_strategy == CDDP_HIP_BARRIER_ADAPTIVE) {
            const double kkt = std::max(std::max(t.inf_pr, sdu), t.inf_comp);
            const double threshold = std::max(o.barrier_mu_update_factor * mu2, 2.0 * mu2);
            if (kkt <= threshold) {
              double factor = o.barrier_mu_update_factor;
              if (mu2 > 1e-20) {
                const double ratio = kkt / std::max(mu2, 1e-20);
                if (ratio < 0.01) factor = 0.1 * o.barrier_mu_update_factor;
                else if (ratio < 0.1) factor = 0.3 * o.barrier_mu_update_factor;
                else if (ratio < 0.5) factor = 0.6 * o.barrier_mu_update_factor;
              }
              const double linear = factor * mu2, superlinear = std::pow(mu2, o.barrier_mu_update_power);
              mu2 = std::max(std::min(linear, superlinear), std::max(o.barrier_mu_min_value, o.tolerance / 100.0));
            }
          } else {
            const double kkt = std::max(std::max(t.inf_pr, sdu * o.ipddp_barrier_update_dual_weight), t.inf_comp);
            if (kkt <= o.ipddp_mu_kappa_epsilon * mu2) {
              const double linear = o.barrier_mu_update_factor * mu2, superlinear = std::pow(mu2, o.barrier_mu_update_power);
              mu2 = std::max(o.barrier_mu_min_value, std::min(linear, superlinear));
            }
          }
        }
        t.mu = mu2;
        std::vector<double> r, rx, hT((size_t)std::max(pT, 1), 0.0);
        if (hte) ti.eval(pl->user, nx, t.X.data() + (size_t)N * nx, nullptr, nullptr, hT.data(), nullptr, r, rx);
        double phi_n, theta_n, ipr, icomp;
        ip_reductions_t(c, ti, t.S.data(), t.Y.data(), t.G.data(), t.ST.data(), t.YT.data(), t.GT.data(), t.LamT.data(), hT.data(), mu2, t.cost, phi_n, theta_n, ipr, icomp);
        const double ftheta = std::max(theta_n, 1e-8);
        const bool reset = (mu2 < mu_old) && (mu2 > 0.0);
        if (reset) { t.filter.clear(); filter_accept(t.filter, t.phi, ftheta); }   // (:2629-2637: re-seeded because a terminal set exists)
        else { filter_accept(t.filter, t.phi, ftheta); if ((int)t.filter.size() > o.ipddp_max_filter_size) filter_prune(t.filter); }
        t.inf_pr = ipr; t.inf_comp = icomp; t.merit = t.phi = phi_n; t.filter_theta = ftheta;
        t.theta = std::max(ftheta, std::max(o.ipddp_theta_0_floor, 1e-8));
        t.reg = reg_decrease(o, t.reg);
        // ---- checkConvergence (:1953-2025)
        const double sdu2 = scaled_inf_du(c, t, t.gGx);
        const double pr = t.inf_pr, scomp = t.inf_comp, sn = t.step_norm;
        int st = CDDP_HIP_STATUS_RUNNING; bool done = false;
        if (no_barrier) {
          if (pr < o.tolerance && sdu2 < o.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; done = true; }
          else if (o.acceptable_tolerance > 0.0) {
            const double sq = std::sqrt(o.acceptable_tolerance);
            bool acc = (pr < sq && sdu2 < sq && t.iter > 50);
            if (dJ > 0.0) acc = acc || (dJ < o.acceptable_tolerance && t.iter > 50 && pr < sq && sdu2 < sq);
            if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; done = true; }
          }
        } else {
          const double tol = std::max(o.tolerance, o.ipddp_barrier_tol_mult * mu2);
          if (pr < tol && sdu2 < tol && scomp < tol && sn < o.tolerance * 10.0) { st = CDDP_HIP_STATUS_OPTIMAL; done = true; }
          else if (o.acceptable_tolerance > 0.0) {
            const double at = std::sqrt(o.acceptable_tolerance);
            const double bat = std::max(o.barrier_mu_min_value * 100.0, o.tolerance / 10.0);
            const bool akkt = pr < at && sdu2 < at && scomp < at, bpc = mu2 <= bat;
            bool acc = akkt && bpc && t.iter > 10 && std::fabs(dJ) < o.acceptable_tolerance;
            acc = acc || (akkt && bpc && t.iter >= 1 && sn < o.tolerance * 10.0 && pr < 1e-4);
            if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; done = true; }
          }
        }
        if (done) { t.status = st; t.done = true; }
      } else {
        // ---- handleForwardPassFailure (:2037-2082): a second increase when a barrier and terminal-equality rows coexist
        t.reg = reg_increase(o, t.reg);
        if (!no_barrier && hte) t.reg = reg_increase(o, t.reg);
        if (t.reg >= o.reg_max_value) {
          const double sdu = scaled_inf_du(c, t, t.gGx);
          const double base = std::sqrt(std::max(o.acceptable_tolerance, o.tolerance));
          const double at = no_barrier ? base : std::max(base, o.ipddp_barrier_tol_mult * t.mu);
          const bool acc = o.acceptable_tolerance > 0.0 && t.inf_pr < at && sdu < at && (no_barrier || t.inf_comp < at);
          t.status = acc ? CDDP_HIP_STATUS_ACCEPTABLE : CDDP_HIP_STATUS_REG_LIMIT; t.done = true;
        }
      }
      if (!t.done && it == o.max_iterations) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
    };
    par_for(B, n_threads, advance);
    if (abort_seen.load() || aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
  }
  for (auto &t : T) if (!t.done) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
  if (Kout) std::copy(Kfin.begin(), Kfin.end(), Kout);
  for (size_t b = 0; b < B; ++b) {
    const TTraj &t = T[b];
    cddp_hip_result &r = results[b];
    std::memset(&r, 0, sizeof(r));
    r.final_objective = t.cost; r.merit_function = t.merit; r.inf_pr = t.inf_pr; r.inf_du = t.inf_du; r.inf_comp = t.inf_comp;
    r.barrier_mu = t.mu; r.regularization = t.reg; r.alpha_pr = t.alpha_pr; r.alpha_du = t.alpha_du; r.step_norm = t.step_norm;
    r.iterations = t.iter; r.status = t.status; r.n_backward = t.n_bwd; r.n_forward = t.n_fwd;
    if (Xout) std::copy(t.X.begin(), t.X.end(), Xout + b * (N + 1) * nx);
    if (Uout) std::copy(t.U.begin(), t.U.end(), Uout + b * N * nu);
    if (Tout) {   // [S_T (mT) | Y_T (mT) | Lambda_T (pT)] per trajectory
      double *o2 = Tout + b * (size_t)(2 * mT + pT);
      for (int i = 0; i < mT; ++i) { o2[i] = t.ST[i]; o2[mT + i] = t.YT[i]; }
      for (int i = 0; i < pT; ++i) o2[2 * mT + i] = t.LamT[i];
    }
  }
  return 0;
}

}  // namespace

extern "C" int cddp_hip_plugin_last_stats(double *total_ms, double *gpu_section_ms, double *kernel_ms, int *sweeps, int *threads) {
  if (total_ms) *total_ms = g_last_stats.total_ms;
  if (gpu_section_ms) *gpu_section_ms = g_last_stats.gpu_section_ms;
  if (kernel_ms) *kernel_ms = g_last_stats.kernel_ms;
  if (sweeps) *sweeps = g_last_stats.sweeps;
  if (threads) *threads = g_last_stats.threads;
  return 0;
}

extern "C" int cddp_hip_plugin_set_host_threads(int n) {
  if (n < 0) return pfail(-2, "host thread count must be >= 0 (0 = one per hardware thread)");
  g_host_threads = n;
  return 0;
}

static int plugin_solve_impl(const cddp_hip_plugin *pl, const cddp_hip_plugin_terminal *tc, int solver, int horizon, double dt, const cddp_hip_options *opt,
                             int device, int batch, const double *x0, const double *U0, const double *X0,
                             cddp_hip_result *results, double *Xout, double *Uout, double *Kout, double *Tout);

extern "C" int cddp_hip_plugin_solve(const cddp_hip_plugin *pl, int solver, int horizon, double dt, const cddp_hip_options *opt,
                                     int device, int batch, const double *x0, const double *U0, const double *X0,
                                     cddp_hip_result *results, double *Xout, double *Uout, double *Kout) {
  return plugin_solve_impl(pl, nullptr, solver, horizon, dt, opt, device, batch, x0, U0, X0, results, Xout, Uout, Kout, nullptr);
}

extern "C" int cddp_hip_plugin_solve_terminal(const cddp_hip_plugin *pl, const cddp_hip_plugin_terminal *tc, int solver, int horizon, double dt,
                                              const cddp_hip_options *opt, int device, int batch, const double *x0, const double *U0, const double *X0,
                                              cddp_hip_result *results, double *Xout, double *Uout, double *Kout, double *terminal_out) {
  return plugin_solve_impl(pl, tc, solver, horizon, dt, opt, device, batch, x0, U0, X0, results, Xout, Uout, Kout, terminal_out);
}

static int plugin_solve_impl(const cddp_hip_plugin *pl, const cddp_hip_plugin_terminal *tc, int solver, int horizon, double dt, const cddp_hip_options *opt,
                             int device, int batch, const double *x0, const double *U0, const double *X0,
                             cddp_hip_result *results, double *Xout, double *Uout, double *Kout, double *Tout) {
  if (!pl || !opt || !x0 || !results) return pfail(-1, "null argument");
  if (pl->abi_version != CDDP_HIP_ABI_VERSION || pl->options_bytes != (int)sizeof(cddp_hip_options))
    return pfail(-2, "ABI mismatch: caller built against version %d with a %d-byte cddp_hip_options, library has version %d and %d bytes",
                 pl->abi_version, pl->options_bytes, CDDP_HIP_ABI_VERSION, (int)sizeof(cddp_hip_options));
  if (solver != CDDP_HIP_SOLVER_CLDDP && solver != CDDP_HIP_SOLVER_IPDDP && solver != CDDP_HIP_SOLVER_LOGDDP && solver != CDDP_HIP_SOLVER_MSIPDDP)
    return pfail(-2, "UnknownSolver - No solver registered for id %d", solver);
  if (!pl->discrete_dynamics || !pl->jacobians) return pfail(-2, "Dynamical system must be set before solving.");
  if (!pl->running_cost || !pl->terminal_cost || !pl->running_cost_derivatives || !pl->terminal_cost_derivatives)
    return pfail(-2, "Objective function must be set before solving.");
  const int nx = pl->nx, nu = pl->nu, N = horizon;
  int m = 0;
  if (pl->n_constraints < 0 || pl->n_constraints > CDDP_HIP_PLUGIN_MAX_CONSTRAINTS) return pfail(-3, "too many path constraints (%d)", pl->n_constraints);
  for (int s = 0; s < pl->n_constraints; ++s) m += pl->constraint_dims[s];
  if (solver == CDDP_HIP_SOLVER_CLDDP) m = 0;   // CLDDP only honours the box named "ControlConstraint" (control_lower / control_upper)
  if (m > 0 && !pl->constraints) return pfail(-2, "Cannot add null constraint.");
  if (nx <= 0 || nu <= 0 || N <= 0 || batch <= 0 || !(dt > 0)) return pfail(-2, "bad dimensions nx=%d nu=%d N=%d batch=%d dt=%g", nx, nu, N, batch, dt);
  if (!(opt->reg_update_factor > 1.0) || !(opt->reg_max_value > 0.0)) return pfail(-2, "regularization.update_factor must be > 1 and max_value > 0");
  if (!opt->use_ilqr && !pl->hessians) return pfail(-3, "use_ilqr=false needs the plug-in's Hessian callback");

  Ctx c; c.pl = pl; c.o = opt; c.solver = solver; c.nx = nx; c.nu = nu; c.m = m; c.N = N; c.dt = dt;
  { double al[CDDP_HIP_MAX_ALPHAS]; const int na = cddp_hip_build_alphas(opt, al, CDDP_HIP_MAX_ALPHAS); c.alphas.assign(al, al + na); }
  const cddp_hip_options &o = *opt;
  // Terminal constraints: only IPDDP reads the terminal set (clddp / logddp / msipddp_solver.cpp never touch getTerminalConstraintSet)
  if (tc && tc->n_terminal > 0 && solver == CDDP_HIP_SOLVER_IPDDP) {
    if (tc->n_terminal > CDDP_HIP_PLUGIN_MAX_CONSTRAINTS || !tc->evaluate) return pfail(-2, "bad terminal-constraint description (%d objects)", tc->n_terminal);
    TermInfo ti; ti.tc = tc; ti.nobj = tc->n_terminal;
    for (int s2 = 0; s2 < ti.nobj; ++s2) {
      ti.dim[s2] = tc->dims[s2]; ti.eq[s2] = tc->equality[s2] ? 1 : 0; ti.src[s2] = ti.rows; ti.rows += ti.dim[s2];
      if (ti.dim[s2] <= 0) return pfail(-2, "terminal constraint %d has %d rows", s2, ti.dim[s2]);
      if (ti.eq[s2]) { ti.dst[s2] = ti.pT; ti.pT += ti.dim[s2]; } else { ti.dst[s2] = ti.mT; ti.mT += ti.dim[s2]; }
    }
    if (ti.pT > 8) return pfail(-3, "more than 8 terminal-equality rows (%d) on the plug-in route", ti.pT);
    return ipddp_terminal_solve(c, ti, device, batch, x0, U0, X0, results, Xout, Uout, Kout, Tout);
  }
  if (solver == CDDP_HIP_SOLVER_LOGDDP) return logddp_solve(c, device, batch, x0, U0, results, Xout, Uout, Kout);
  if (solver == CDDP_HIP_SOLVER_MSIPDDP) return msipddp_solve(c, device, batch, x0, U0, X0, results, Xout, Uout, Kout);

  cddp_hip_stack_handle *sh = nullptr;
  { int rc = cddp_hip_stacks_create(device, batch, nx, nu, m, N, &sh); if (rc) return rc; }
  struct Guard { cddp_hip_stack_handle *h; ~Guard() { if (h) cddp_hip_stacks_destroy(h); } } guard{sh};

  const size_t B = (size_t)batch;
  std::vector<Traj> T(B);
  for (size_t b = 0; b < B; ++b)
    initialize(c, T[b], x0 + b * nx, U0 ? U0 + b * N * nu : nullptr, X0 ? X0 + b * (N + 1) * nx : nullptr);

  // batch-major host stacks of the current iterates
  std::vector<double> fx(B * N * nx * nx), fu(B * N * nx * nu), lx(B * N * nx), lu(B * N * nu), lxx(B * N * nx * nx), luu(B * N * nu * nu),
      lux(B * N * nu * nx), VxN(B * nx), VxxN(B * nx * nx), Ubuf(B * N * nu);
  std::vector<double> gy, gs, gg, gGx, gGu, Fxx, Fuu, Fux;
  if (m > 0) { gy.resize(B * N * m); gs = gy; gg = gy; gGx.resize(B * N * m * nx); gGu.resize(B * N * m * nu); }
  if (!o.use_ilqr) { Fxx.resize(B * N * nx * nx * nx); Fuu.resize(B * N * nx * nu * nu); Fux.resize(B * N * nx * nu * nx); }
  std::vector<double> Kb(B * N * nu * nx), kb(B * N * nu), Vxb(B * (N + 1) * nx), Vxxb(B * (N + 1) * nx * nx), dVb(B * 2);
  // K_u_ of each trajectory's LAST backward pass (the reference's solver object stops sweeping a problem when it ends; the batch keeps
  // sweeping the others): the slice is kept at every sweep the trajectory still takes part in
  std::vector<double> Kfin(Kout ? B * N * nu * nx : 0, 0.0);
  std::vector<double> kyb, Kyb, ksb, Ksb, dXb;
  if (m > 0) { kyb.resize(B * N * m); ksb = kyb; Kyb.resize(B * N * m * nx); Ksb = Kyb; dXb.resize(B * (N + 1) * nx); }
  std::vector<double> regv(B), muv(B), s_reg(B), s_du(B), s_pr(B), s_comp(B), s_sn(B), s_apr(B), s_adu(B);
  std::vector<int32_t> okv(B);
  const bool first_rule = !o.enable_parallel;
  std::atomic<bool> abort_seen{false};
  const int n_threads = host_threads((int)B);
  const StatClock total_clock;
  g_last_stats = PluginStats(); g_last_stats.threads = n_threads;
  const auto wall0 = std::chrono::steady_clock::now();

  for (int it = 1; it <= o.max_iterations; ++it) {
    bool any = false;
    for (auto &t : T) any = any || !t.done;
    if (!any) break;
    if (aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
    if (o.max_cpu_time > 0.0) {   // cddp_solver_base.cpp:77-90 (whole elapsed milliseconds)
      const double el_ms = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count();
      if (el_ms > o.max_cpu_time * 1000.0) {
        for (auto &t : T) if (!t.done) { t.iter += 1; t.status = CDDP_HIP_STATUS_MAX_CPU_TIME; t.done = true; }
        break;
      }
    }
    // ---- precomputeDynamicsDerivatives / precomputeConstraintGradients on the host (cddp_solver_base.cpp:319-394,
    //      ipddp_solver.cpp:2145-2250), cost derivatives (objective.hpp): the stacks of every running trajectory
    auto fill_stacks = [&](size_t b) {   // (one host thread per trajectory when the caller allows it: cddp_hip_plugin_set_host_threads)
      Traj &t = T[b];
      std::vector<double> tfx((size_t)nx * nx), tfu((size_t)nx * nu);
      regv[b] = t.done ? std::max(t.reg, o.reg_min_value) : t.reg;
      muv[b] = (t.mu > 0.0) ? t.mu : 1.0;
      if (t.done) return;
      t.iter += 1;
      for (int s = 0; s < N; ++s) {
        const double *x = t.X.data() + (size_t)s * nx, *u = t.U.data() + (size_t)s * nu;
        const size_t bs = b * N + s;
        pl->jacobians(pl->user, x, u, s * dt, tfx.data(), tfu.data());
        for (int i = 0; i < nx; ++i)
          for (int j = 0; j < nx; ++j) { double a = dt * tfx[i * nx + j]; if (i == j) a += 1.0; fx[(bs * nx + i) * nx + j] = a; }   // A = I + dt f_x
        for (int i = 0; i < nx * nu; ++i) fu[bs * nx * nu + i] = dt * tfu[i];                                                      // B = dt f_u
        pl->running_cost_derivatives(pl->user, x, u, s, lx.data() + bs * nx, lu.data() + bs * nu, lxx.data() + bs * nx * nx,
                                     luu.data() + bs * nu * nu, lux.data() + bs * nu * nx);
        if (m > 0) {
          pl->constraints(pl->user, x, u, s, gg.data() + bs * m, gGx.data() + bs * m * nx, gGu.data() + bs * m * nu);
          // (g itself is the iterate's residual evaluated by the last accepted rollout; Jacobians are what is needed here)
          std::copy(t.G.begin() + (size_t)s * m, t.G.begin() + (size_t)(s + 1) * m, gg.begin() + bs * m);
          std::copy(t.S.begin() + (size_t)s * m, t.S.begin() + (size_t)(s + 1) * m, gs.begin() + bs * m);
          std::copy(t.Y.begin() + (size_t)s * m, t.Y.begin() + (size_t)(s + 1) * m, gy.begin() + bs * m);
        }
        if (!o.use_ilqr) {
          double *pxx = Fxx.data() + bs * nx * nx * nx, *puu = Fuu.data() + bs * nx * nu * nu, *pux = Fux.data() + bs * nx * nu * nx;
          pl->hessians(pl->user, x, u, s * dt, pxx, puu, pux);
          for (int e = 0; e < nx * nx * nx; ++e) pxx[e] = dt * pxx[e];   // F_xx_[t][i] = dt f_xx[i] (cddp_solver_base.cpp:346-356)
          for (int e = 0; e < nx * nu * nu; ++e) puu[e] = dt * puu[e];
          for (int e = 0; e < nx * nu * nx; ++e) pux[e] = dt * pux[e];
        }
      }
      pl->terminal_cost_derivatives(pl->user, t.X.data() + (size_t)N * nx, VxN.data() + b * nx, VxxN.data() + b * nx * nx);
      std::copy(t.U.begin(), t.U.end(), Ubuf.begin() + b * N * nu);
    };
    par_for(B, n_threads, fill_stacks);
    const StatClock gpu_clock;
    { int rc = cddp_hip_set_stacks(sh, fx.data(), fu.data(), lx.data(), lu.data(), lxx.data(), luu.data(), lux.data(), VxN.data(), VxxN.data()); if (rc) return rc; }
    if (m > 0) { int rc = cddp_hip_set_constraint_stacks(sh, gy.data(), gs.data(), gg.data(), gGx.data(), gGu.data()); if (rc) return rc; }
    if (!o.use_ilqr && c.ipddp()) { int rc = cddp_hip_set_hessian_stacks(sh, Fxx.data(), Fuu.data(), Fux.data()); if (rc) return rc; }
    if (!c.ipddp() && pl->control_lower && pl->control_upper) {
      int rc = cddp_hip_set_control_box(sh, pl->control_lower, pl->control_upper, Ubuf.data()); if (rc) return rc;
    }
    // ---- backwardPass of the whole batch on the GPU, incl. the "increase regularisation and retry" loop (:93-111)
    const int branch = !c.ipddp() ? CDDP_HIP_STACKS_CLDDP : (m > 0 ? CDDP_HIP_STACKS_IPDDP_PATH : CDDP_HIP_STACKS_IPDDP);
    { int rc = cddp_hip_stacks_backward(sh, branch, opt, regv.data(), m > 0 ? muv.data() : nullptr, 1, okv.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_get_gains(sh, Kb.data(), kb.data(), Vxb.data(), Vxxb.data(), dVb.data()); if (rc) return rc; }
    if (Kout) for (size_t b = 0; b < B; ++b) if (!T[b].done) std::copy(Kb.begin() + b * N * nu * nx, Kb.begin() + (b + 1) * N * nu * nx, Kfin.begin() + b * N * nu * nx);
    if (m > 0) { int rc = cddp_hip_stacks_get_constraint_gains(sh, kyb.data(), Kyb.data(), ksb.data(), Ksb.data(), dXb.data()); if (rc) return rc; }
    { int rc = cddp_hip_stacks_get_scalars(sh, s_reg.data(), s_du.data(), s_pr.data(), s_comp.data(), s_sn.data(), s_apr.data(), s_adu.data()); if (rc) return rc; }
    g_last_stats.gpu_section_ms += gpu_clock.ms(); g_last_stats.kernel_ms += cddp_hip_stacks_last_kernel_ms(sh); g_last_stats.sweeps += 1;

    auto advance = [&](size_t b) {
      Traj &t = T[b];
      if (t.done) return;
      if (aborted(pl)) { abort_seen.store(true); return; }
      // sweeps the retry loop ran: replay the schedule from the regularisation it started with
      { int nb = 1; double r = t.reg; while (r < s_reg[b] && nb < 64) { r = reg_increase(o, r); ++nb; }
        if (!okv[b] && nb > 1) --nb;   // the loop stops when the schedule reaches reg_max: no sweep is run there
        t.n_bwd += nb; }
      t.reg = s_reg[b];
      if (!okv[b]) { t.status = CDDP_HIP_STATUS_REG_LIMIT; t.done = true; return; }   // handleBackwardPassRegularizationLimit
      t.dV0 = dVb[b * 2]; t.dV1 = dVb[b * 2 + 1]; t.inf_du = s_du[b];
      Gains g;
      g.K = Kb.data() + b * N * nu * nx; g.k = kb.data() + b * N * nu; g.Vx = Vxb.data() + b * (N + 1) * nx; g.Vxx = Vxxb.data() + b * (N + 1) * nx * nx;
      g.ky = g.Ky = g.ks = g.Ks = nullptr;
      std::vector<double> Gx_t;
      if (c.ipddp()) {
        t.step_norm = s_sn[b];
        t.inf_pr = (m > 0) ? s_pr[b] : 0.0; t.inf_comp = (m > 0) ? s_comp[b] : 0.0;
        t.apr_max = (m > 0) ? s_apr[b] : 1.0; t.adu_max = (m > 0) ? s_adu[b] : 1.0;
        if (m > 0) {
          g.ky = kyb.data() + b * N * m; g.ks = ksb.data() + b * N * m; g.Ky = Kyb.data() + b * N * m * nx; g.Ks = Ksb.data() + b * N * m * nx;
          Gx_t.assign(gGx.begin() + b * N * m * nx, gGx.begin() + (b + 1) * N * m * nx);
        }
      }
      // ---- checkEarlyConvergence (clddp_solver.cpp:206-213 / ipddp_solver.cpp:925-958)
      bool conv = false;
      if (!c.ipddp()) conv = t.inf_du < o.tolerance;
      else {
        const double sdu = scaled_inf_du(c, t, Gx_t);
        if (m == 0) conv = (t.inf_pr < o.tolerance && sdu < o.tolerance);
        else {
          const double tol = std::max(o.tolerance, o.ipddp_barrier_tol_mult * t.mu);
          conv = (t.inf_pr < tol && sdu < tol && t.inf_comp < tol && std::fabs(t.alpha_pr) * t.step_norm < o.tolerance * 10.0);
        }
      }
      if (conv) { t.status = CDDP_HIP_STATUS_OPTIMAL; t.done = true; return; }
      // ---- performForwardPass (cddp_solver_base.cpp:248-317): first success, or lowest merit among the successes
      Trial best; bool have = false;
      int walked = 0;
      for (double a : c.alphas) {
        Trial r = c.ipddp() ? forward_ipddp(c, t, g, a) : forward_clddp(c, t, g, a);
        ++walked;
        if (!r.success) continue;
        if (first_rule) { best = std::move(r); have = true; break; }
        if (!have || r.merit < best.merit) { best = std::move(r); have = true; }
      }
      t.n_fwd += first_rule ? walked : (int)c.alphas.size();
      if (have) {
        // ---- applyForwardPassResult (cddp_solver_base.cpp:190-198, ipddp_solver.cpp:1878-1951)
        const double dJ = t.cost - best.cost;
        t.X.swap(best.X); t.U.swap(best.U);
        t.cost = best.cost; t.merit = best.merit; t.alpha_pr = best.alpha_pr; t.alpha_du = c.ipddp() ? best.alpha_du : 1.0;
        int st = CDDP_HIP_STATUS_RUNNING; bool done = false;
        if (c.ipddp()) {
          t.Lam.swap(best.Lam);
          if (m > 0) { t.S.swap(best.S); t.Y.swap(best.Y); t.G.swap(best.G); }
          t.inf_pr = best.inf_pr; t.inf_comp = best.inf_comp; t.phi = best.merit; t.filter_theta = best.theta; t.theta = best.theta;
          // ---- updateBarrierParameters(true) (ipddp_solver.cpp:2548-2660)
          const double sdu = scaled_inf_du(c, t, Gx_t);
          double mu = t.mu; const double mu_old = mu;
          if (m > 0) {
            if (o.barrier_strategy == CDDP_HIP_BARRIER_ADAPTIVE) {
              const double kkt = std::max(std::max(t.inf_pr, sdu), t.inf_comp);
              const double threshold = std::max(o.barrier_mu_update_factor * mu, 2.0 * mu);
              if (kkt <= threshold) {
                double factor = o.barrier_mu_update_factor;
                if (mu > 1e-20) {
                  const double ratio = kkt / std::max(mu, 1e-20);
                  if (ratio < 0.01) factor = 0.1 * o.barrier_mu_update_factor;
                  else if (ratio < 0.1) factor = 0.3 * o.barrier_mu_update_factor;
                  else if (ratio < 0.5) factor = 0.6 * o.barrier_mu_update_factor;
                }
                const double linear = factor * mu, superlinear = std::pow(mu, o.barrier_mu_update_power);
                mu = std::max(std::min(linear, superlinear), std::max(o.barrier_mu_min_value, o.tolerance / 100.0));
              }
            } else {
              const double kkt = std::max(std::max(t.inf_pr, sdu * o.ipddp_barrier_update_dual_weight), t.inf_comp);
              if (kkt <= o.ipddp_mu_kappa_epsilon * mu) {
                const double linear = o.barrier_mu_update_factor * mu, superlinear = std::pow(mu, o.barrier_mu_update_power);
                mu = std::max(o.barrier_mu_min_value, std::min(linear, superlinear));
              }
            }
          }
          t.mu = mu;
          double phi_n = t.cost, theta_n = 0.0, ipr = 0.0, icomp = 0.0;
          if (m > 0) ip_reductions(c, t.S.data(), t.Y.data(), t.G.data(), mu, t.cost, phi_n, theta_n, ipr, icomp);
          const double ftheta = std::max(theta_n, 1e-8);
          const bool reset = (mu < mu_old) && (mu > 0.0);
          if (reset) t.filter.clear();   // re-seeded only when terminal constraints exist (:2629-2637): none here
          else { filter_accept(t.filter, t.phi, ftheta); if ((int)t.filter.size() > o.ipddp_max_filter_size) filter_prune(t.filter); }
          t.inf_pr = ipr; t.inf_comp = icomp; t.merit = t.phi = phi_n; t.filter_theta = ftheta;
          t.theta = std::max(ftheta, std::max(o.ipddp_theta_0_floor, 1e-8));
          t.reg = reg_decrease(o, t.reg);
          // ---- checkConvergence (ipddp_solver.cpp:1953-2025)
          const double sdu2 = scaled_inf_du(c, t, Gx_t);
          const double pr = t.inf_pr, scomp = t.inf_comp, sn = t.step_norm;
          if (m == 0) {
            if (pr < o.tolerance && sdu2 < o.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; done = true; }
            else if (o.acceptable_tolerance > 0.0) {
              const double sq = std::sqrt(o.acceptable_tolerance);
              bool acc = (pr < sq && sdu2 < sq && t.iter > 50);
              if (dJ > 0.0) acc = acc || (dJ < o.acceptable_tolerance && t.iter > 50 && pr < sq && sdu2 < sq);
              if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; done = true; }
            }
          } else {
            const double tol = std::max(o.tolerance, o.ipddp_barrier_tol_mult * mu);
            if (pr < tol && sdu2 < tol && scomp < tol && sn < o.tolerance * 10.0) { st = CDDP_HIP_STATUS_OPTIMAL; done = true; }
            else if (o.acceptable_tolerance > 0.0) {
              const double at = std::sqrt(o.acceptable_tolerance);
              const double bat = std::max(o.barrier_mu_min_value * 100.0, o.tolerance / 10.0);
              const bool akkt = pr < at && sdu2 < at && scomp < at, bpc = mu <= bat;
              bool acc = akkt && bpc && t.iter > 10 && std::fabs(dJ) < o.acceptable_tolerance;
              acc = acc || (akkt && bpc && t.iter >= 1 && sn < o.tolerance * 10.0 && pr < 1e-4);
              if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; done = true; }
            }
          }
        } else {
          t.reg = reg_decrease(o, t.reg);
          if (t.inf_du < o.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; done = true; }                      // clddp_solver.cpp:264-277
          else if (dJ > 0.0 && dJ < o.acceptable_tolerance) { st = CDDP_HIP_STATUS_ACCEPTABLE; done = true; }
        }
        if (done) { t.status = st; t.done = true; }
      } else {
        // ---- handleForwardPassFailure (cddp_solver_base.cpp:206-218, ipddp_solver.cpp:2037-2082)
        t.reg = reg_increase(o, t.reg);
        if (t.reg >= o.reg_max_value) {
          int st = CDDP_HIP_STATUS_REG_LIMIT;
          if (c.ipddp()) {
            const double sdu = scaled_inf_du(c, t, Gx_t);
            const double base = std::sqrt(std::max(o.acceptable_tolerance, o.tolerance));
            const double at = (m == 0) ? base : std::max(base, o.ipddp_barrier_tol_mult * t.mu);
            if (o.acceptable_tolerance > 0.0 && t.inf_pr < at && sdu < at && (m == 0 || t.inf_comp < at)) st = CDDP_HIP_STATUS_ACCEPTABLE;
          }
          t.status = st; t.done = true;
        }
      }
      if (!t.done && it == o.max_iterations) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }
    };
    par_for(B, n_threads, advance);
    if (abort_seen.load() || aborted(pl)) return pfail(-50, "aborted by the caller (cddp_hip_plugin::abort_flag)");
  }
  for (auto &t : T) if (!t.done) { t.status = CDDP_HIP_STATUS_MAX_ITERATIONS; t.done = true; }   // max_iterations <= 0
  g_last_stats.total_ms = total_clock.ms();

  // ---- CDDPSolution fields (cddp_solver_base.cpp:161-171, ipddp_solver.cpp:2090-2097); feedback gains = K_u_ of the last sweep
  if (Kout) std::copy(Kfin.begin(), Kfin.end(), Kout);
  for (size_t b = 0; b < B; ++b) {
    const Traj &t = T[b];
    cddp_hip_result &r = results[b];
    std::memset(&r, 0, sizeof(r));
    r.final_objective = t.cost; r.merit_function = t.merit; r.inf_pr = t.inf_pr; r.inf_du = t.inf_du; r.inf_comp = t.inf_comp;
    r.barrier_mu = t.mu; r.regularization = t.reg; r.alpha_pr = t.alpha_pr; r.alpha_du = t.alpha_du; r.step_norm = t.step_norm;
    r.iterations = t.iter; r.status = t.status; r.n_backward = t.n_bwd; r.n_forward = t.n_fwd;
    if (Xout) std::copy(t.X.begin(), t.X.end(), Xout + b * (N + 1) * nx);
    if (Uout) std::copy(t.U.begin(), t.U.end(), Uout + b * N * nu);
  }
  return 0;
}
