// Registry and drop-in checks of the reference-side adapter, against the REAL cddp-cpp library (built by oracle/ref_pin/build_ref.sh when
// Eigen 3.4.0 / autodiff 1.1.2 are given; plain main(), no gtest).  What it restates from the reference's own tests:
//   tests/cddp_core/test_cddp_core.cpp:316-411   registerSolver / isSolverRegistered / getRegisteredSolvers, UnknownSolver result
//   tests/cddp_core/test_cddp_core.cpp:463-483   SolverPrecedence: a registered name wins over the built-in solver of the same name
//   tests/cddp_core/test_ipddp_solver.cpp:349-472, test_clddp_solver.cpp:28-151   pendulum solves: status in {Optimal, Acceptable},
//                                                iterations > 0, final_objective < initial cost
// and what only a machine with BOTH the reference and a GPU can check: the registered (GPU) "IPDDP" / "CLDDP" and the reference's own
// IPDDPSolver / CLDDPSolver give the same iteration count and status on the pendulum and cart-pole examples (north_star: "identical
// iteration counts on pendulum / cartpole").  Exit code 0 = all checks passed; run on a box with an MI355X.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>

#include "cddp_core/cddp_core.hpp"
#include "cddp_core/constraint.hpp"
#include "cddp_core/objective.hpp"
#include "dynamics_model/cartpole.hpp"
#include "dynamics_model/pendulum.hpp"
#include "hip_batch_solver.hpp"

static int g_failures = 0;
#define CHECK(cond) do { if (!(cond)) { ++g_failures; std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

// (std::unique_ptr: cddp::CDDP holds unique_ptr members and declares no move constructor of its own)
static std::unique_ptr<cddp::CDDP> makePendulum(const cddp::CDDPOptions &opt, bool box) {   // examples/cddp_pendulum.cpp:24-65
  const int N = 100; const double dt = 0.02;
  Eigen::VectorXd x0(2), goal(2); x0 << M_PI, 0.0; goal << 0.0, 0.0;
  Eigen::MatrixXd Q = Eigen::MatrixXd::Zero(2, 2), R = 0.1 * Eigen::MatrixXd::Identity(1, 1), Qf = 100.0 * Eigen::MatrixXd::Identity(2, 2);
  auto s = std::make_unique<cddp::CDDP>(x0, goal, N, dt, std::make_unique<cddp::Pendulum>(dt, 0.5, 1.0, 0.01, "euler"),
                                        std::make_unique<cddp::QuadraticObjective>(Q, R, Qf, goal, std::vector<Eigen::VectorXd>(), dt), opt);
  if (box) { Eigen::VectorXd ub(1); ub << 20.0; s->addPathConstraint("ControlConstraint", std::make_unique<cddp::ControlConstraint>(-ub, ub)); }
  return s;
}
static std::unique_ptr<cddp::CDDP> makeCartPole(const cddp::CDDPOptions &opt) {              // examples/cddp_cartpole.cpp:24-66
  const int N = 100; const double dt = 0.05;
  Eigen::VectorXd x0 = Eigen::VectorXd::Zero(4), goal(4); goal << 0.0, M_PI, 0.0, 0.0;
  Eigen::MatrixXd Q = Eigen::MatrixXd::Zero(4, 4), R = 0.1 * Eigen::MatrixXd::Identity(1, 1), Qf = 100.0 * Eigen::MatrixXd::Identity(4, 4);
  auto s = std::make_unique<cddp::CDDP>(x0, goal, N, dt, std::make_unique<cddp::CartPole>(dt, "rk4", 1.0, 0.2, 0.5, 9.81, 0.0),
                                        std::make_unique<cddp::QuadraticObjective>(Q, R, Qf, goal, std::vector<Eigen::VectorXd>(), dt), opt);
  Eigen::VectorXd ub(1); ub << 5.0;
  s->addPathConstraint("ControlConstraint", std::make_unique<cddp::ControlConstraint>(-ub, ub));
  return s;
}

int main() {
  cddp::CDDPOptions opt;
  opt.max_iterations = 30; opt.tolerance = 1e-4; opt.acceptable_tolerance = 1e-5; opt.regularization.initial_value = 1e-6;
  opt.verbose = false; opt.print_solver_header = false; opt.return_iteration_info = true;

  // 1. the reference's own solvers FIRST (nothing registered yet): the numbers the GPU path must reproduce
  cddp::CDDPSolution ref_ip = makePendulum(opt, true)->solve("IPDDP");
  cddp::CDDPSolution ref_cl = makePendulum(opt, true)->solve("CLDDP");
  cddp::CDDPOptions opt_cp = opt; opt_cp.max_iterations = 80; opt_cp.tolerance = 1e-6; opt_cp.regularization.initial_value = 1e-5;
  cddp::CDDPSolution ref_cp = makeCartPole(opt_cp)->solve("IPDDP");
  cddp::CDDPSolution ref_cpc = makeCartPole(opt_cp)->solve("CLDDP");

  // 2. registry (test_cddp_core.cpp:316-411, 463-483)
  CHECK(!cddp::CDDP::isSolverRegistered("IPDDP"));
  cddp::registerHipSolvers();
  for (const char *n : {"IPDDP", "CLDDP", "CLCDDP", "LogDDP", "LOGDDP", "MSIPDDP"}) CHECK(cddp::CDDP::isSolverRegistered(n));
  CHECK(cddp::CDDP::getRegisteredSolvers().size() >= 6);
  {
    std::unique_ptr<cddp::CDDP> s = makePendulum(opt, true);
    cddp::CDDPSolution u = s->solve("NonExistentSolver");      // unknown names do not throw (cddp_core.cpp:243-265)
    CHECK(u.solver_name == "NonExistentSolver" && u.status_message.find("UnknownSolver") != std::string::npos && u.iterations_completed == 0);
    CHECK(cddp::hipRouteOf(*s, "IPDDP") == "resident");
  }

  // 3. the registered solvers win over the built-ins (SolverPrecedence) and reproduce them
  auto same = [](const cddp::CDDPSolution &a, const cddp::CDDPSolution &b, const char *what) {
    std::printf("%-28s reference: %-28s %3d iterations J = %.12g | hip: %-28s %3d iterations J = %.12g\n", what, a.status_message.c_str(), a.iterations_completed,
                a.final_objective, b.status_message.c_str(), b.iterations_completed, b.final_objective);
    CHECK(a.status_message == b.status_message);
    CHECK(a.iterations_completed == b.iterations_completed);
    CHECK(std::fabs(a.final_objective - b.final_objective) <= 1e-8 * std::max(1.0, std::fabs(a.final_objective)));
    CHECK(a.state_trajectory.size() == b.state_trajectory.size() && a.feedback_gains.size() == b.feedback_gains.size());
    double ek = 0.0;
    for (size_t t = 0; t < a.feedback_gains.size() && t < b.feedback_gains.size(); ++t)
      ek = std::max(ek, (a.feedback_gains[t] - b.feedback_gains[t]).cwiseAbs().maxCoeff() / std::max(1.0, a.feedback_gains[t].cwiseAbs().maxCoeff()));
    std::printf("%-28s max relative gain difference %.3e\n", what, ek);
    CHECK(ek <= 1e-8);   // BASELINE.json: "gains within 1e-8 of Eigen reference"
  };
  same(ref_ip, makePendulum(opt, true)->solve("IPDDP"), "pendulum IPDDP");
  same(ref_cl, makePendulum(opt, true)->solve(cddp::SolverType::CLDDP), "pendulum CLDDP (enum)");
  same(ref_cp, makeCartPole(opt_cp)->solve("IPDDP"), "cart-pole IPDDP");
  same(ref_cpc, makeCartPole(opt_cp)->solve("CLDDP"), "cart-pole CLDDP");
  for (const cddp::CDDPSolution *s : {&ref_ip, &ref_cl}) {
    CHECK(s->status_message == "OptimalSolutionFound" || s->status_message == "AcceptableSolutionFound");
    CHECK(s->iterations_completed > 0 && !s->history.objective.empty() && s->final_objective < s->history.objective.front());
  }

  // 4. the batched entry point: 64 perturbed cart-pole starts, trajectory 0 = the single solve
  {
    std::unique_ptr<cddp::CDDP> s = makeCartPole(opt_cp);
    std::vector<Eigen::VectorXd> x0s;
    for (int b = 0; b < 64; ++b) { Eigen::VectorXd x = Eigen::VectorXd::Zero(4); x(1) = 0.004 * b; x0s.push_back(x); }
    std::vector<cddp::CDDPSolution> sols = cddp::solveBatchHip(*s, "IPDDP", x0s);
    CHECK(sols.size() == 64);
    CHECK(sols[0].iterations_completed == ref_cp.iterations_completed && sols[0].status_message == ref_cp.status_message);
    for (const auto &q : sols) for (const auto &u : q.control_trajectory) CHECK(std::fabs(u(0)) <= 5.0 + 1e-9);
  }
  std::printf("%s (%d failed checks)\n", g_failures ? "FAILED" : "all adapter checks passed", g_failures);
  return g_failures ? 1 : 0;
}
