// Reference-side adapter: the MI355X batched solver core registered under cddp-cpp's own solver names.
//
// This header and hip_batch_solver.cpp are written against the REAL cddp-cpp headers (include/cddp-cpp/cddp_core/cddp_core.hpp and
// friends, Eigen 3.4) -- not against the Eigen-free mirror in cddp-cpp_amd/host/cddp_hip.hpp -- and against this repository's C-ABI
// (include/cddp_hip.h).  They are what a cddp-cpp maintainer adds to the tree:
//
//     #include "hip_batch_solver.hpp"
//     cddp::registerHipSolvers();                       // once: "IPDDP", "CLDDP" / "CLCDDP", "LogDDP" / "LOGDDP", "MSIPDDP"
//     cddp::CDDPSolution s = solver.solve("IPDDP");      // CDDP::createSolver consults the external registry first (cddp_core.cpp:213-219)
//     auto sols = cddp::solveBatchHip(solver, "IPDDP", x0s);   // NEW: one device-resident batch of the same problem
//
// Neither Eigen nor autodiff exists in the image this repository is built in, so this translation unit cannot be compiled there;
// oracle/ref_pin/build_ref.sh compiles it (and integration/test_hip_registry.cpp) as soon as EIGEN3_INCLUDE_DIR / AUTODIFF_INCLUDE_DIR
// point at the pinned checkouts -- same fail-loudly rule as the reference pin itself.
#pragma once
#include <string>
#include <vector>

#include "cddp_core/cddp_core.hpp"   // cddp::CDDP, ISolverAlgorithm, CDDPSolution (cddp_core.hpp:54-103, 186-210, 214-423)

namespace cddp {

// CDDP::registerSolver for every solver name the library serves (cddp_core.hpp:305-319, cddp_core.cpp:578-595).  Idempotent.
void registerHipSolvers(int device = 0);

// One device-resident batch: solution i starts from x0s[i]; problem, options, constraints and the initial trajectory guess (ctx.X_, ctx.U_)
// are the context's.  No reference counterpart (the reference solves one problem per CDDP object).
std::vector<CDDPSolution> solveBatchHip(CDDP &context, const std::string &solver_type, const std::vector<Eigen::VectorXd> &x0s, int device = 0);

// Which path a problem takes (for logs / tests): "resident" = device-resident batch kernels (built-in plant with accessible parameters,
// exactly a QuadraticObjective, constraint kinds with accessible parameters), "plugin" = the context's virtual functions behind the flat
// callbacks of cddp_hip_plugin (host forward passes, batched GPU backward passes).
std::string hipRouteOf(CDDP &context, const std::string &solver_type);

}  // namespace cddp
