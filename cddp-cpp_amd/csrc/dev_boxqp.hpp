// BoxQP: projected-Newton solver of  min 0.5 x^T H x + g^T x  s.t. lower <= x <= upper  (src/cddp_core/boxqp.cpp:25-250),
// shared by the CLDDP sweeps of the solver core (kernels.hpp, kernels_coop.hpp) and of the stack-fed mode (stacks.hip).
#pragma once
#include "dev_linalg.hpp"
#include "../../include/cddp_hip.h"

namespace cddp_dev {

enum { BQ_HESSIAN_NOT_PD = -1, BQ_NO_DESCENT = 0, BQ_MAX_ITER = 1, BQ_MAX_LS = 2, BQ_SUCCESS = 4, BQ_ALL_CLAMPED = 5 };

template <int N>
DEV double boxqp_objective(const double *x, const double *H, const double *g) {
  double q = 0.0, l = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double hx = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) hx += H[i * N + j] * x[j];
    q += x[i] * hx;
    l += g[i] * x[i];
  }
  return 0.5 * q + l;
}

// Returns status; x (in: warm start, out: solution), free mask, Hfree factor of the final free block.
template <int N>
DEV int boxqp_solve(const cddp_hip_options &o, const double *H, const double *g, const double *lower,
                    const double *upper, double *x, int *free_, LDLTd<N> &Hfree) {
  int status = BQ_MAX_ITER;
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = dmin(dmax(x[i], lower[i]), upper[i]);
  int clamped[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { clamped[i] = 0; free_[i] = 1; }
  double value = boxqp_objective<N>(x, H, g);
  double old_value = INFINITY;
  for (int iter = 0; iter < o.boxqp_max_iterations; ++iter) {
    if (iter > 0 && fabs(old_value - value) < o.boxqp_min_relative_improvement * fabs(old_value)) { status = BQ_SUCCESS; break; }
    old_value = value;
    double grad[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double hx = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) hx += H[i * N + j] * x[j];
      grad[i] = g[i] + hx;
    }
    int old_clamped[N];
    int nclamped = 0;
    bool any_different = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      old_clamped[i] = clamped[i];
      clamped[i] = ((x[i] == lower[i] && grad[i] > 0) || (x[i] == upper[i] && grad[i] < 0)) ? 1 : 0;
      nclamped += clamped[i];
      free_[i] = 1 - clamped[i];
      if (old_clamped[i] != clamped[i]) any_different = true;
    }
    if (nclamped == N) { status = BQ_ALL_CLAMPED; break; }
    const bool factorize = (iter == 0) || any_different;
    int free_idx[N];
    int nf = 0;
    for (int i = 0; i < N; ++i) if (!clamped[i]) free_idx[nf++] = i;
    if (factorize) {
      double Hf[N * N];
      for (int i = 0; i < nf; ++i) for (int j = 0; j < nf; ++j) Hf[i * N + j] = H[free_idx[i] * N + free_idx[j]];
      Hfree.compute(Hf, nf);
      if (!Hfree.ok) { status = BQ_HESSIAN_NOT_PD; break; }
    }
    double grad_norm = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) if (!clamped[i]) grad_norm += grad[i] * grad[i];
    grad_norm = sqrt(grad_norm);
    if (grad_norm < o.boxqp_min_gradient_norm) { status = BQ_SUCCESS; break; }
    double search[N], grad_clamped[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { search[i] = 0.0; grad_clamped[i] = g[i]; }
    for (int i = 0; i < N; ++i)
      if (clamped[i]) {
#pragma unroll
        for (int r = 0; r < N; ++r) grad_clamped[r] += H[r * N + i] * x[i];
      }
    double sf[N];
    for (int i = 0; i < nf; ++i) sf[i] = grad_clamped[free_idx[i]];
    Hfree.solve(sf);
    for (int i = 0; i < nf; ++i) search[free_idx[i]] = (-sf[i]) - x[free_idx[i]];
    double sdotg = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) sdotg += search[i] * grad[i];
    if (sdotg >= 0) { status = BQ_NO_DESCENT; break; }
    double step = 1.0;
    bool ls_ok = false;
    double xn[N];
    while (step > o.boxqp_min_step_size) {
#pragma unroll
      for (int i = 0; i < N; ++i) xn[i] = dmin(dmax(x[i] + step * search[i], lower[i]), upper[i]);
      double value_new = boxqp_objective<N>(xn, H, g);
      if ((value_new - value) <= o.boxqp_armijo_constant * step * sdotg) { ls_ok = true; break; }
      step *= o.boxqp_step_decrease_factor;
    }
    if (!ls_ok) { status = BQ_MAX_LS; break; }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = xn[i];
    value = boxqp_objective<N>(x, H, g);
  }
  return status;
}

// The BoxQP options as loop-invariant values.  A sweep kernel loads them ONCE in front of its step loop: read through the options
// struct inside the step, every field is a scalar load plus an s_waitcnt inside conditionally executed code (which the compiler does
// not hoist) -- eight dependent ~150-cycle round trips per step on the chain of the cart-pole CLDDP sweep (round 4, ISA of
// k_backward_coop_plain<CartPoleModel, true>).
struct BoxQPConst {
  int max_it;
  double min_grad, min_rel, step_dec, min_step, armijo;
  DEV void load(const cddp_hip_options &o) {
    max_it = o.boxqp_max_iterations; min_grad = o.boxqp_min_gradient_norm; min_rel = o.boxqp_min_relative_improvement;
    step_dec = o.boxqp_step_decrease_factor; min_step = o.boxqp_min_step_size; armijo = o.boxqp_armijo_constant;
  }
};

// N = 1: the same projected-Newton iteration written on scalars (no free-index lists, no factor object: the 1x1 "factor" of the
// free block is H itself and LDLT's solve is D^+ with tolerance DBL_MIN, dev_linalg.hpp::ldlt1_solve).  Statement for statement
// the N = 1 trace of boxqp_solve<N> above -- same values, same order of decisions, same exits -- so a control-limited single-input
// plant (pendulum, cart-pole: BASELINE config[1] read literally) gets the reference's BoxQP answer bit for bit.  free_ = 0 iff
// the solution is clamped.  Three evaluations of the generic trace are REUSED instead of repeated; each yields the same bits:
//   * the Newton target -H^+ g is the same quotient in every iteration (H, g do not change): computed once;
//   * ||grad|| = sqrt(grad^2) equals |grad| exactly in binary floating point when grad^2 neither overflows nor underflows
//     (S. Boldo, "Stupid is as stupid does: taking the square root of the square of a floating-point number", 2015); outside
//     that range the square root is taken as written;
//   * the objective at the accepted point was already evaluated by the line search (same function, same argument).
// A dependent f64 division or square root costs ~150 cycles on gfx950 and the loop is replicated by every lane of a trajectory's
// group, on the critical path of every sweep step.
DEV int boxqp_solve1(const BoxQPConst &o, const double H, const double g, const double lower, const double upper, double &x, int &free_) {
  int status = BQ_MAX_ITER;
  x = dmin(dmax(x, lower), upper);
  int clamped = 0;
  free_ = 1;
  auto objective = [&](double xv) { const double hx = 0.0 + H * xv; const double q = 0.0 + xv * hx; const double l = 0.0 + g * xv; return 0.5 * q + l; };
  double value = objective(x);
  double old_value = INFINITY;
  const double newton = -ldlt1_solve(H, g);      // -(H_free^+ grad_clamped): grad_clamped = g while the variable is free
  for (int iter = 0; iter < o.max_it; ++iter) {
    if (iter > 0 && fabs(old_value - value) < o.min_rel * fabs(old_value)) { status = BQ_SUCCESS; break; }
    old_value = value;
    const double grad = g + (0.0 + H * x);
    clamped = ((x == lower && grad > 0) || (x == upper && grad < 0)) ? 1 : 0;
    free_ = 1 - clamped;
    if (clamped) { status = BQ_ALL_CLAMPED; break; }
    const double ag = fabs(grad);
    double grad_norm = ag;
    if (!(ag > 0x1p-500 && ag < 0x1p500)) grad_norm = sqrt(0.0 + grad * grad);   // also NaN
    if (grad_norm < o.min_grad) { status = BQ_SUCCESS; break; }
    const double search = newton - x;
    const double sdotg = 0.0 + search * grad;
    if (sdotg >= 0) { status = BQ_NO_DESCENT; break; }
    double step = 1.0;
    bool ls_ok = false;
    double xn = x, value_new = value;
    while (step > o.min_step) {
      xn = dmin(dmax(x + step * search, lower), upper);
      value_new = objective(xn);
      if ((value_new - value) <= o.armijo * step * sdotg) { ls_ok = true; break; }
      step *= o.step_dec;
    }
    if (!ls_ok) { status = BQ_MAX_LS; break; }
    x = xn;
    value = value_new;
  }
  return status;
}

DEV int boxqp_solve1(const cddp_hip_options &o, const double H, const double g, const double lower, const double upper, double &x, int &free_) {
  BoxQPConst c; c.load(o);
  return boxqp_solve1(c, H, g, lower, upper, x, free_);
}

// boxqp_solve1 with its COMMON traces in straight-line code (round 4).  The iteration loop above costs ~80 instructions per pass plus
// a dozen branches on a kernel whose every sweep step is one dependent chain replicated by the lanes of a trajectory (profiles/
// r04_kernel_stats_clddp_base.md: the BoxQP is ~200 of the ~650 instructions of a cart-pole sweep step).  Nearly every call ends in
// one of five ways, all inside the first two passes:
//   A  pass 0: the clamped warm start is held against its bound by the gradient            -> ALL_CLAMPED, x = x0, not free
//   B  pass 0: |grad| < min_gradient_norm                                                  -> SUCCESS, x = x0
//   C  pass 0 takes the full Newton step (Armijo holds at step 1); pass 1: relative improvement below the threshold -> SUCCESS, x = x1
//   D  ... pass 1: x1 sits on a bound with the gradient pushing outward                     -> ALL_CLAMPED, x = x1, not free
//   E  ... pass 1: |grad| < min_gradient_norm                                              -> SUCCESS, x = x1
// The routine evaluates exactly the statements of those traces (same operands, same order, selects instead of branches) and
// returns their result; when ANY lane of the wavefront leaves them (a shortened step, a third pass, no descent, an out-of-range
// gradient square, fewer than two passes allowed) every lane redoes the call with the loop -- a prefix of the same trace, so the
// answer is the loop's in both cases.  tests/test_boxqp.py replays the reference-held inputs through both.
DEV int boxqp_solve1_fast(const BoxQPConst &o, const double H, const double g, const double lower, const double upper, double &x, int &free_) {
  const double xw = x;
  const double x0 = dmin(dmax(xw, lower), upper);
  auto objective = [&](double xv) { const double hx = 0.0 + H * xv; const double q = 0.0 + xv * hx; const double l = 0.0 + g * xv; return 0.5 * q + l; };
  auto norm_ok = [](double ag) { return ag == 0.0 || (ag > 0x1p-500 && ag < 0x1p500); };   // sqrt(grad^2) == |grad| exactly
  const double value0 = objective(x0);
  const double newton = -ldlt1_solve(H, g);
  // pass 0
  const double grad0 = g + (0.0 + H * x0);
  const bool clamped0 = (x0 == lower && grad0 > 0) || (x0 == upper && grad0 < 0);
  const double ag0 = fabs(grad0);
  const bool small0 = ag0 < o.min_grad;
  const double search0 = newton - x0;
  const double sdotg0 = 0.0 + search0 * grad0;
  const double x1 = dmin(dmax(x0 + 1.0 * search0, lower), upper);
  const double value1 = objective(x1);
  const bool step1 = (sdotg0 < 0) && (1.0 > o.min_step) && ((value1 - value0) <= o.armijo * 1.0 * sdotg0);
  // pass 1
  const bool rel1 = fabs(value0 - value1) < o.min_rel * fabs(value0);
  const double grad1 = g + (0.0 + H * x1);
  const bool clamped1 = (x1 == lower && grad1 > 0) || (x1 == upper && grad1 < 0);
  const double ag1 = fabs(grad1);
  const bool small1 = ag1 < o.min_grad;
  const bool exitA = clamped0;
  const bool exitB = !clamped0 && norm_ok(ag0) && small0;
  const bool pass1 = !clamped0 && norm_ok(ag0) && !small0 && step1;
  const bool exitC = pass1 && rel1;
  const bool exitD = pass1 && !rel1 && clamped1;
  const bool exitE = pass1 && !rel1 && !clamped1 && norm_ok(ag1) && small1;
  const bool fast = (o.max_it >= 2) && (exitA || exitB || exitC || exitD || exitE);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast) != 0ull, 0)) return boxqp_solve1(o, H, g, lower, upper, x, free_);
  x = (exitA || exitB) ? x0 : x1;
  free_ = (exitA || exitD) ? 0 : 1;
  return (exitA || exitD) ? BQ_ALL_CLAMPED : BQ_SUCCESS;
}

}  // namespace cddp_dev
