// C++ tests of the host-side mirror (cddp-cpp_amd/host/cddp_hip.hpp), written after the reference's own
// gtests: tests/cddp_core/test_cddp_core.cpp (registry / dispatch / error conventions) and
// tests/cddp_core/test_{clddp,ipddp}_solver.cpp (pendulum solves).  usage: test_host_api cpu|gpu
#include <cmath>
#include <cstdio>
#include <iostream>
#include "../../cddp-cpp_amd/host/cddp_hip.hpp"

static int g_fail = 0;
#define EXPECT_TRUE(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))

namespace {
class MockExternalSolver : public cddp::ISolverAlgorithm {   // test_cddp_core.cpp:35-74
 public:
  void initialize(cddp::CDDP &) override { initialized = true; }
  cddp::CDDPSolution solve(cddp::CDDP &) override {
    cddp::CDDPSolution s; s.solver_name = "MockExternalSolver"; s.status_message = "MockSolved"; s.iterations_completed = 7; return s;
  }
  std::string getSolverName() const override { return "MockExternalSolver"; }
  bool initialized = false;
};

cddp::CDDP makePendulum(const cddp::CDDPOptions &options, int horizon = 100) {
  const double dt = 0.02;
  cddp::Vector x0 = {3.14159265358979323846, 0.0}, goal = {0.0, 0.0};
  cddp::Matrix Q = cddp::Matrix::Zero(2, 2), R = 0.1 * cddp::Matrix::Identity(1), Qf = 100.0 * cddp::Matrix::Identity(2);
  cddp::CDDP solver(x0, goal, horizon, dt, std::make_unique<cddp::Pendulum>(dt, 0.5, 1.0, 0.01, "euler"),
                    std::make_unique<cddp::QuadraticObjective>(Q, R, Qf, goal, std::vector<cddp::Vector>{}, dt), options);
  solver.addPathConstraint("ControlConstraint", std::make_unique<cddp::ControlConstraint>(cddp::Vector{-20.0}, cddp::Vector{20.0}));
  std::vector<cddp::Vector> X(horizon + 1, x0), U(horizon, cddp::Vector{0.0});
  solver.setInitialTrajectory(X, U);
  return solver;
}
}  // namespace

static void cpu_tests() {
  // registry: register / query / list (test_cddp_core.cpp:316-370)
  EXPECT_TRUE(!cddp::CDDP::isSolverRegistered("MockSolver"));
  cddp::CDDP::registerSolver("MockSolver", [] { return std::make_unique<MockExternalSolver>(); });
  EXPECT_TRUE(cddp::CDDP::isSolverRegistered("MockSolver"));
  bool listed = false;
  for (auto &n : cddp::CDDP::getRegisteredSolvers()) listed = listed || n == "MockSolver";
  EXPECT_TRUE(listed);
  cddp::CDDPOptions opt; opt.max_iterations = 30; opt.verbose = false;
  {
    cddp::CDDP solver = makePendulum(opt);
    cddp::CDDPSolution s = solver.solve("MockSolver");          // external solver is dispatched
    EXPECT_EQ(s.status_message, std::string("MockSolved"));
    EXPECT_EQ(s.iterations_completed, 7);
    // unknown solver -> status string, no throw (cddp_core.cpp:243-265)
    cddp::CDDPSolution u = solver.solve("NoSuchSolver");
    EXPECT_EQ(u.status_message, std::string("UnknownSolver - No solver registered for 'NoSuchSolver'"));
    EXPECT_EQ(u.iterations_completed, 0);
    EXPECT_TRUE(u.final_step_length == 1.0);
    // dual-dim bookkeeping (test_cddp_core.cpp:637-677)
    EXPECT_EQ(solver.getTotalDualDim(), 2);
    solver.addPathConstraint("Obstacle", std::make_unique<cddp::BallConstraint>(0.4, cddp::Vector{1.0, 1.0}));
    EXPECT_EQ(solver.getTotalDualDim(), 3);
    solver.addPathConstraint("ControlConstraint", std::make_unique<cddp::ControlConstraint>(cddp::Vector{-1.0}, cddp::Vector{1.0}));
    EXPECT_EQ(solver.getTotalDualDim(), 3);
    EXPECT_TRUE(solver.removePathConstraint("Obstacle"));
    EXPECT_EQ(solver.getTotalDualDim(), 2);
  }
  {   // missing system / objective -> runtime_error with the reference's messages (cddp_core.cpp:277-282)
    cddp::CDDP bare(cddp::Vector{0.0}, cddp::Vector{0.0}, 4, 1.0);
    bool threw = false;
    try { bare.solve("MockSolver"); } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "Dynamical system must be set before solving."; }
    EXPECT_TRUE(threw);
    bare.setDynamicalSystem(std::make_unique<cddp::LTISystem>(cddp::Matrix::Identity(1), cddp::Matrix::Identity(1), 1.0));
    threw = false;
    try { bare.solve("MockSolver"); } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "Objective function must be set before solving."; }
    EXPECT_TRUE(threw);
    threw = false;
    try { bare.addPathConstraint("x", nullptr); } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "Cannot add null constraint."; }
    EXPECT_TRUE(threw);
  }
  {   // options flatten to the POD field-for-field
    cddp::CDDPOptions o; o.tolerance = 1e-4; o.ipddp.barrier.mu_initial = 0.1; o.line_search.max_iterations = 15; o.enable_parallel = true;
    cddp_hip_options p = o.toPOD();
    EXPECT_TRUE(p.tolerance == 1e-4 && p.barrier_mu_initial == 0.1 && p.ls_max_iterations == 15 && p.enable_parallel == 1);
    EXPECT_TRUE(p.reg_initial_value == 1e-6 && p.boxqp_max_iterations == 100 && p.ipddp_max_filter_size == 5);
  }
  {   // registering the GPU core under the reference's names overrides nothing else (drop-in)
    cddp::registerHipSolvers();
    EXPECT_TRUE(cddp::CDDP::isSolverRegistered("IPDDP") && cddp::CDDP::isSolverRegistered("CLDDP"));
    if (cddp_hip_device_count() == 0) {   // no GPU: the product refuses, it never falls back to a CPU path
      cddp::CDDP solver = makePendulum(opt);
      bool threw = false;
      try { solver.solve(cddp::SolverType::IPDDP); } catch (const std::runtime_error &e) { threw = std::string(e.what()).find("no CPU fallback") != std::string::npos; }
      EXPECT_TRUE(threw);
    }
  }
}

static void gpu_tests() {
  cddp::registerHipSolvers();
  cddp::CDDPOptions opt; opt.max_iterations = 30; opt.tolerance = 1e-4; opt.acceptable_tolerance = 1e-5;
  opt.regularization.initial_value = 1e-6; opt.verbose = false; opt.return_iteration_info = true;
  for (const char *name : {"IPDDP", "CLDDP"}) {   // examples/cddp_pendulum.cpp:24-65
    cddp::CDDP solver = makePendulum(opt);
    cddp::CDDPSolution s = solver.solve(name);
    std::cout << name << ": " << s.status_message << " iterations " << s.iterations_completed << " cost " << s.final_objective << "\n";
    EXPECT_TRUE(s.status_message == "OptimalSolutionFound" || s.status_message == "AcceptableSolutionFound");
    EXPECT_TRUE(s.iterations_completed > 0);
    EXPECT_EQ((int)s.state_trajectory.size(), 101);
    EXPECT_EQ((int)s.control_trajectory.size(), 100);
    EXPECT_EQ((int)s.feedback_gains.size(), 100);
    EXPECT_EQ(s.feedback_gains[0].rows, 1); EXPECT_EQ(s.feedback_gains[0].cols, 2);
    EXPECT_EQ((int)s.time_points.size(), 101);
    EXPECT_TRUE(std::fabs(s.state_trajectory.back()[0]) < 0.01);           // upright
    EXPECT_TRUE(s.final_objective < s.history.objective.front());          // cost decreased
    EXPECT_TRUE(!s.history.objective.empty() && (int)s.history.objective.size() == s.iterations_completed + 1);
    for (auto &u : s.control_trajectory) EXPECT_TRUE(u[0] <= 20.0 + 1e-9 && u[0] >= -20.0 - 1e-9);
    EXPECT_TRUE(std::fabs(solver.cost_ - s.final_objective) == 0.0);      // context left updated
  }
  {   // batched API: trajectory 0 of the batch equals the single solve
    cddp::CDDP solver = makePendulum(opt);
    cddp::CDDPSolution single = solver.solve("IPDDP");
    cddp::CDDP solver2 = makePendulum(opt);
    std::vector<cddp::Vector> x0s;
    for (int b = 0; b < 128; ++b) x0s.push_back({3.14159265358979323846 + 0.001 * b, 0.0});
    std::vector<cddp::CDDPSolution> sols = solver2.solveBatch("IPDDP", x0s);
    EXPECT_EQ((int)sols.size(), 128);
    EXPECT_EQ(sols[0].iterations_completed, single.iterations_completed);
    EXPECT_TRUE(sols[0].final_objective == single.final_objective);
    int conv = 0; for (auto &s : sols) conv += (s.status_message == "OptimalSolutionFound" || s.status_message == "AcceptableSolutionFound");
    EXPECT_TRUE(conv >= 120);
  }
  {   // terminal equality constraint (tests/cddp_core/test_ipddp_solver.cpp:1580-1637 style): x_N pinned to the target
    cddp::CDDPOptions o2 = opt; o2.max_iterations = 100;
    cddp::CDDP solver = makePendulum(o2, 60);
    solver.addTerminalConstraint("TerminalTarget", std::make_unique<cddp::TerminalEqualityConstraint>(cddp::Vector{0.0, 0.0}));
    cddp::CDDPSolution s = solver.solve("IPDDP");
    std::cout << "IPDDP + terminal equality: " << s.status_message << " iterations " << s.iterations_completed
              << " |x_N| " << std::fabs(s.state_trajectory.back()[0]) << "\n";
    EXPECT_TRUE(s.iterations_completed > 0);
    EXPECT_TRUE(std::fabs(s.state_trajectory.back()[0]) < 1e-2 && std::fabs(s.state_trajectory.back()[1]) < 1e-2);
  }
  {   // warm start (tests/cddp_core/test_ipddp_solver.cpp:474-549): previous solution as the initial trajectory of a
      // NEW solver -> converges within cold + 5 iterations; then solver-object reuse from a perturbed state
    cddp::CDDP cold = makePendulum(opt);
    cddp::CDDPSolution c = cold.solve("IPDDP");
    cddp::CDDPOptions wopt = opt; wopt.warm_start = true;
    cddp::CDDP warm = makePendulum(wopt);
    warm.setInitialTrajectory(c.state_trajectory, c.control_trajectory);
    cddp::CDDPSolution w = warm.solve("IPDDP");
    std::cout << "warm start: " << w.status_message << " iterations " << w.iterations_completed << " (cold " << c.iterations_completed << ")\n";
    EXPECT_TRUE(w.status_message == "OptimalSolutionFound" || w.status_message == "AcceptableSolutionFound");
    EXPECT_TRUE(w.iterations_completed <= c.iterations_completed + 5);
    cddp::HipBatchSolver obj(CDDP_HIP_SOLVER_IPDDP);   // one solver object, two solves (MPC restart)
    cddp::CDDP ctx = makePendulum(opt);
    obj.initialize(ctx); cddp::CDDPSolution s1 = obj.solve(ctx);
    ctx.setOptions(wopt);
    ctx.setInitialState({3.14159265358979323846 - 0.05, 0.0});
    obj.initialize(ctx); cddp::CDDPSolution s2 = obj.solve(ctx);
    std::cout << "solver reuse: " << s2.status_message << " iterations " << s2.iterations_completed << " (first " << s1.iterations_completed << ")\n";
    EXPECT_TRUE(s2.iterations_completed > 0);
    EXPECT_TRUE(std::fabs(s2.state_trajectory.front()[0] - (3.14159265358979323846 - 0.05)) < 1e-15);
  }
  {   // a layout that is not instantiated on the device: loud error, never a silent fallback
    cddp::CDDP solver = makePendulum(opt);
    solver.addPathConstraint("Extra", std::make_unique<cddp::BallConstraint>(0.5, cddp::Vector{1.0, 1.0}));
    bool threw = false;
    try { solver.solve("IPDDP"); } catch (const std::runtime_error &e) { threw = std::string(e.what()).find("no kernel instantiation") != std::string::npos; }
    EXPECT_TRUE(threw);
  }
}

int main(int argc, char **argv) {
  std::string mode = argc > 1 ? argv[1] : "cpu";
  if (mode == "cpu") cpu_tests(); else gpu_tests();
  if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
  std::printf("host API %s tests passed\n", mode.c_str());
  return 0;
}
