// Terminal constraints of the IPDDP core (reference include/cddp-cpp/cddp_core/terminal_constraint.hpp):
//   TerminalInequalityConstraint  g_T = A_N x_N - b_N <= 0   (slack s_T, dual y_T)
//   TerminalEqualityConstraint    h_T = x_N - target  = 0     (multiplier lambda_T)
// Only instantiated for kernel sets with TERM = true; dimensions are run-time (<= kMTMax / kPTMax).
#pragma once
#include "dev_constraints.hpp"

namespace cddp_dev {

constexpr int kMTMax = 8;    // stacked terminal-inequality rows
constexpr int kPTMax = 16;   // stacked terminal-equality rows (>= max nx)

// per-lane copy of the terminal variables of one iterate
struct TermState {
  double g[kMTMax], s[kMTMax], y[kMTMax];
  double lam[kPTMax], h[kPTMax];
};

// g_T = A x_N - b for every terminal-inequality object, stacked in std::map (name) order
template <int NX>
DEV void term_ineq_eval(const ProblemDev *P, const double *xN, double *g) {
  for (int c = 0; c < P->n_term; ++c) {
    const TermDev &td = P->terms[c];
    if (td.kind != CDDP_HIP_TERM_INEQUALITY) continue;
    for (int r = 0; r < td.dim; ++r) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += P->pool[td.off_A + r * NX + j] * xN[j];
      g[td.offset + r] = s - P->pool[td.off_b + r];
    }
  }
}
// h_T = x_N - target, stacked (ipddp_solver.cpp:155-176)
template <int NX>
DEV void term_eq_residual(const ProblemDev *P, const double *xN, double *h) {
  for (int c = 0; c < P->n_term; ++c) {
    const TermDev &td = P->terms[c];
    if (td.kind != CDDP_HIP_TERM_EQUALITY) continue;
    for (int r = 0; r < td.dim; ++r) h[td.offset + r] = xN[r] - P->pool[td.off_target + r];
  }
}
// row r of the stacked terminal-inequality Jacobian
DEV const double *term_ineq_row(const ProblemDev *P, int row) {
  for (int c = 0; c < P->n_term; ++c) {
    const TermDev &td = P->terms[c];
    if (td.kind != CDDP_HIP_TERM_INEQUALITY) continue;
    if (row >= td.offset && row < td.offset + td.dim) return P->pool + td.off_A + (row - td.offset) * P->nx;
  }
  return P->pool;
}
// state index that stacked terminal-equality row `row` selects (H_T rows are identity rows)
DEV int term_eq_col(const ProblemDev *P, int row) {
  for (int c = 0; c < P->n_term; ++c) {
    const TermDev &td = P->terms[c];
    if (td.kind != CDDP_HIP_TERM_EQUALITY) continue;
    if (row >= td.offset && row < td.offset + td.dim) return row - td.offset;
  }
  return 0;
}

DEV void term_load(const DevBuf &d, int b, int mT, int pT, TermState &ts) {
  for (int i = 0; i < mT; ++i) { ts.g[i] = d.GT[(size_t)i * d.Bp + b]; ts.s[i] = d.ST[(size_t)i * d.Bp + b]; ts.y[i] = d.YT[(size_t)i * d.Bp + b]; }
  for (int i = 0; i < pT; ++i) ts.lam[i] = d.LamT[(size_t)i * d.Bp + b];
}

// Terminal parts of computeTheta / computeBarrierMerit / computePrimalAndComplementarity
// (ipddp_solver.cpp:2812-2845, 2866-2878, 2912-2935), appended in the reference's order.
DEV void term_reductions(const ProblemDev *P, const TermState &ts, int mT, int pT, double mu, bool l2,
                         double &total, double &max_entry, double &mer, double &ipr, double &icomp) {
  for (int c = 0; c < P->n_term; ++c) {
    const TermDev &td = P->terms[c];
    if (td.kind != CDDP_HIP_TERM_INEQUALITY) continue;
    double n1 = 0.0, ninf = 0.0;
    for (int r = 0; r < td.dim; ++r) {
      const int j = td.offset + r;
      const double res = ts.g[j] + ts.s[j];
      n1 += l2 ? res * res : fabs(res);
      ninf = dmax(ninf, fabs(res));
      icomp = dmax(icomp, fabs(ts.y[j] * ts.s[j] - mu));
    }
    total += n1; max_entry = dmax(max_entry, ninf); ipr = dmax(ipr, ninf);
  }
  if (pT > 0) {
    double n1 = 0.0, ninf = 0.0;
    for (int r = 0; r < pT; ++r) { n1 += l2 ? ts.h[r] * ts.h[r] : fabs(ts.h[r]); ninf = dmax(ninf, fabs(ts.h[r])); }
    total += n1; max_entry = dmax(max_entry, ninf); ipr = dmax(ipr, ninf);
  }
  for (int c = 0; c < P->n_term; ++c) {
    const TermDev &td = P->terms[c];
    if (td.kind != CDDP_HIP_TERM_INEQUALITY) continue;
    double ls = 0.0;
    for (int r = 0; r < td.dim; ++r) ls += solver_log(dmax(ts.s[td.offset + r], 1e-10));
    mer -= mu * ls;
  }
  if (pT > 0) {
    double dp = 0.0;
    for (int r = 0; r < pT; ++r) dp += ts.lam[r] * ts.h[r];
    mer += dp;
  }
  (void)mT;
}

}  // namespace cddp_dev
