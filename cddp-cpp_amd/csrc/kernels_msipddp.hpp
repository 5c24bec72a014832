// MSIPDDP on the device: the reference's multiple-shooting interior-point DDP (msipddp_solver.cpp:33-1930) for
// the built-in plants, as K0 / K2 / K4 / K5 variants of the batched core -- the same handle, trial slots, phase machine, ladder shapes
// and host loop as CLDDP / IPDDP / LogDDP (capi.hip::SolveRun), K1 (k_derivs: A_t = I + dt f_x, B_t = dt f_u) unchanged.  Rounds 3-4
// served MSIPDDP through the plug-in boundary only (cddp_hip_plugin_solve: GPU sweeps on caller-built stacks, host rollouts, one
// trajectory at a time); here the whole solve is resident:
//
//   K0  k_init_msipddp      grid (batch)        initialize: cold start, multiple-shooting start (warm_start with a state guess that is
//                                               not rolled out), warm re-solve; duals / slacks / costates, evaluateTrajectory,
//                                               resetBarrierFilter                              (msipddp_solver.cpp:33-264, 425-763)
//   K2  k_backward_msipddp  grid (batch)        Riccati sweep with the defects d_t = F_t - x_{t+1}: unconstrained branch with the
//                                               reference's per-step factor cache, path-constrained KKT condensation; costate gains
//                                               k_lambda, K_lambda; retry loop                  (msipddp_solver.cpp:1112-1430)
//   K4  k_forward_msipddp   grid (batch x n_a)  multiple-shooting rollout with the three gap-closing rules at the segment boundaries,
//                                               slack / dual / costate trials, dual step-size search, cost, barrier merit, violation,
//                                               multi-point filter test                         (msipddp_solver.cpp:1432-1724, 771-808)
//   K5  k_update_msipddp    grid (batch)        selection rule, applyForwardPassResult, checkConvergence, forward-pass failure handling
//                                               (filter restoration / regularisation), barrier update + resetBarrierFilter
//                                               (cddp_solver_base.cpp:29-186, msipddp_solver.cpp:287-398, 1751-1930)
//
// One trajectory per lane; every sum in the reference's order (plain triple loops, k ascending: what the CPU checker runs), FMA contraction off,
// the logarithm / power the shared straight-line routines (dev_trig.hpp) -- the checker's MSIPDDP in its trig_mode 1 runs the same
// operations, so tests/test_msipddp_device.py compares decisions exactly.
//
// Two properties of the reference are restated AS THEY ARE (the checker and its numpy twin do the same):
//   * msipddp_solver.cpp:1169-1185 -- the unconstrained sweep keeps one LDLT of Q_uu per step and refactors a step only while its cached
//     factor is invalid: after the first sweep that reaches a step, every later sweep solves that step with the FIRST factor.  The cache
//     ([N][nu nu + nu + 1] per trajectory, d.fac) lives as long as the handle (the reference's workspace outlives initialize());
//   * msipddp_solver.cpp:1398 -- the constrained sweep adds the (nx x nu) product Q_yx^T Y S^-1 Q_yu to the (nu x nx) block Q_ux: defined
//     for nu = 1 (same linear layout) and nx = nu (elementwise); the kernels are instantiated for exactly those shapes.
#pragma once
#include "kernels_logddp.hpp"
#include "kernels_lean.hpp"   // first_surviving_trial (K4b)

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

// ---- multi-point filter on the per-trajectory columns d.filt [2 * d.filt_cap][Bp] (merit rows, then violation rows).  MSIPDDP never
// bounds the filter between barrier updates (it prunes only inside a failed iteration, :371-398), so the capacity is max_iterations + 2.
DEV void ms_filter_add(const DevBuf &d, int b, double mf, double cv) {   // acceptFilterEntry (interior_point_utils.cpp:79-95)
  const int cap = d.filt_cap;
  const int n = d.filt_n[b];
  for (int i = 0; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(cap + i) * d.Bp + b];
    if (fm <= mf && fv <= cv) return;
  }
  int w = 0;
  for (int i = 0; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(cap + i) * d.Bp + b];
    if (!(mf <= fm && cv <= fv)) { d.filt[(size_t)w * d.Bp + b] = fm; d.filt[(size_t)(cap + w) * d.Bp + b] = fv; ++w; }
  }
  if (w < cap) { d.filt[(size_t)w * d.Bp + b] = mf; d.filt[(size_t)(cap + w) * d.Bp + b] = cv; ++w; }
  d.filt_n[b] = w;
}
DEV void ms_filter_prune(const DevBuf &d, int b) {   // pruneFilterToBestPoints (interior_point_utils.cpp:114-139)
  const int cap = d.filt_cap;
  const int n = d.filt_n[b];
  if (n == 0) return;
  double bvm = d.filt[b], bvv = d.filt[(size_t)cap * d.Bp + b];
  double bmm = bvm, bmv = bvv;
  for (int i = 1; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(cap + i) * d.Bp + b];
    if (fv < bvv) { bvm = fm; bvv = fv; }
    if (fm < bmm) { bmm = fm; bmv = fv; }
  }
  d.filt[b] = bvm; d.filt[(size_t)cap * d.Bp + b] = bvv;
  int w = 1;
  if (fabs(bmv - bvv) > 1e-12 || fabs(bmm - bvm) > 1e-12) { d.filt[(size_t)1 * d.Bp + b] = bmm; d.filt[(size_t)(cap + 1) * d.Bp + b] = bmv; w = 2; }
  d.filt_n[b] = w;
}
DEV bool ms_filter_acceptable(const DevBuf &d, int b, const cddp_hip_options &o, double mf, double cv, double expected) {   // isFilterAcceptable :771-808
  const int cap = d.filt_cap;
  const int n = d.filt_n[b];
  if (n == 0) return true;
  double best_v = INFINITY, best_m = INFINITY;
  bool dominated = false;
  for (int i = 0; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(cap + i) * d.Bp + b];
    if (fm <= mf && fv <= cv) dominated = true;
    if (fv < best_v) { best_v = fv; best_m = fm; }
  }
  if (dominated) return false;
  const bool v_imp = cv < best_v * (1.0 - o.filter_violation_acceptance_threshold);
  const bool m_imp = mf < best_m - o.filter_merit_acceptance_threshold * cv;
  if (cv < o.filter_min_violation_for_armijo_check && expected < 0) return mf < best_m + o.filter_armijo_constant * expected;
  if (cv < 1e-6 && mf <= best_m * (1.0 + 1e-8)) return true;
  return v_imp || m_imp;
}

// initial slack / dual pair of one row (msipddp_solver.cpp:578-596 == :667-685)
DEV void ms_init_pair(const cddp_hip_options &o, double mu, double g, double &s, double &y) {
  s = dmax(o.ipddp_slack_var_init_scale, -g);
  y = (s < 1e-12) ? mu / 1e-12 : mu / s;
  y = dmax(o.ipddp_dual_var_init_scale * 0.01, dmin(y, o.ipddp_dual_var_init_scale * 100.0));
}

// resetBarrierFilter (:711-763) on the iterate of `slot`: merit, violation, residuals under `mu`; the filter restarts from that point.
// Round 5: the per-step sums of log s (one per constraint object) are PARKED per slot (d.ev, [slot][N][NSEG]; written here and by the
// rollout for its trial) and the mu-independent pieces of the iterate (max |g + s|, max |F_t - x_{t+1}|, the violation sum) are kept in
// d.ms_res, so that K5 replays the chain  merit -= mu_new * lsum_tc  instead of walking the horizon through two logarithms per row again
// (ms_replay_filter below: same terms, same order).
template <class Cons> DEV size_t ms_ev_plane(const DevBuf &d) { return (size_t)d.N * (Cons::NSEG > 0 ? Cons::NSEG : 1) * d.Bp; }
template <class Model, class Cons>
DEV void ms_reset_filter(const DevBuf &d, int b, int slot, double mu, double cost) {
  constexpr int NX = Model::NX, M = Cons::M, MM = M > 0 ? M : 1, NSEG = Cons::NSEG, NS = NSEG > 0 ? NSEG : 1;
  double mf = cost, ipr = 0.0, fcv = 0.0, icomp = 0.0, idef = 0.0;
  if constexpr (M > 0) {
    double *evs = d.ev + (size_t)slot * ms_ev_plane<Cons>(d);
    const double *Sc = d.S + (size_t)slot * d.planeM, *Yc = d.Y + (size_t)slot * d.planeM, *Gc = d.G + (size_t)slot * d.planeM;
    const double *Fc = d.F + (size_t)slot * d.planeX, *Xc = d.X + (size_t)slot * d.planeX;
    // rows of step t + 1 in flight while step t is reduced (one lane per trajectory: a load issued at the top of its own step is a
    // memory round trip on the chain of K5)
    struct Row { double s[MM], y[MM], g[MM], f[NX], x1[NX]; };
    auto fetch = [&](int tt, Row &r) {
      ld<M>(Sc + GI(tt, M, 0), kLS, r.s); ld<M>(Yc + GI(tt, M, 0), kLS, r.y); ld<M>(Gc + GI(tt, M, 0), kLS, r.g);
      ld<NX>(Fc + GI(tt, NX, 0), kLS, r.f); ld<NX>(Xc + GI(tt + 1, NX, 0), kLS, r.x1);
    };
    auto reduce = [&](const int t, const Row &c, Row &n) {
      fetch(t + 1 < d.N ? t + 1 : d.N - 1, n);
      PIPELINE_FENCE();
#pragma unroll
      for (int cs = 0; cs < NSEG; ++cs) {
        const int off = Cons::seg_off(cs), dim = Cons::seg_dim(cs);
        double lsum = 0.0, l1 = 0.0;
#pragma unroll
        for (int i = 0; i < MM; ++i) {
          if (i < dim) {
            const int j = off + i;
            lsum += solver_log(c.s[j]);
            const double pr = c.g[j] + c.s[j];
            ipr = dmax(ipr, fabs(pr)); l1 += fabs(pr);
            icomp = dmax(icomp, fabs(c.y[j] * c.s[j] - mu));
          }
        }
        mf -= mu * lsum; fcv += l1;
        evs[GI(t, NS, cs)] = lsum;
      }
      double n1 = 0.0, ni = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) { const double r = c.f[i] - c.x1[i]; ni = dmax(ni, fabs(r)); n1 += fabs(r); }
      idef = dmax(idef, ni);
      fcv += n1;
    };
    Row ra, rb;
    fetch(0, ra);
    int t = 0;
    for (; t + 1 < d.N; t += 2) { reduce(t, ra, rb); reduce(t + 1, rb, ra); }
    if (t < d.N) reduce(t, ra, rb);
  }
  d.inf_pr[b] = dmax(ipr, idef); d.merit[b] = mf; d.phi[b] = mf; d.inf_comp[b] = icomp;
  d.filter_theta[b] = fcv; d.theta[b] = fcv;
  d.filt[b] = mf; d.filt[(size_t)d.filt_cap * d.Bp + b] = fcv; d.filt_n[b] = 1;
  if constexpr (M > 0) { d.ms_res[b] = ipr; d.ms_res[(size_t)d.Bp + b] = idef; d.ms_res[(size_t)2 * d.Bp + b] = fcv; }
}

// resetBarrierFilter (:711-763) of K5 on the iterate of `slot` under a NEW mu, from what the rollout (or the reset above) left behind:
//   merit      = cost - sum_t sum_c mu * lsum_tc       the parked sums replayed in the order of the loop above
//   violation  = the iterate's own violation sum (d.ms_res[2]: the accepted trial's theta -- the same terms in the same order)
//   inf_pr     = max(max |g + s|, max |F - x_next|)    (d.ms_res[0], [1]; maxima are order-free)
//   inf_comp   = max |y s - mu| = the larger of |min(y s) - mu|, |max(y s) - mu|: x -> fl(x - mu) is monotone, so the extreme residuals
//                sit at the extreme products (NaN products are skipped by either form); ys_lo / ys_hi come from K5's pass over the Y, S rows
template <class Model, class Cons>
DEV void ms_replay_filter(const DevBuf &d, int b, int slot, double mu, double cost, double ys_lo, double ys_hi) {
  constexpr int M = Cons::M, NSEG = Cons::NSEG, NS = NSEG > 0 ? NSEG : 1;
  static_assert(M > 0, "the barrier filter is reset for path-constrained problems only");
  const double *evs = d.ev + (size_t)slot * ms_ev_plane<Cons>(d);
  double mf = cost;
  constexpr int kTB = 16;   // rows per round trip
  int t = 0;
  for (; t + kTB - 1 < d.N; t += kTB) {
    double v[kTB][NS];
#pragma unroll
    for (int k = 0; k < kTB; ++k)
#pragma unroll
      for (int c = 0; c < NSEG; ++c) v[k][c] = evs[GI(t + k, NS, c)];
#pragma unroll
    for (int k = 0; k < kTB; ++k)
#pragma unroll
      for (int c = 0; c < NSEG; ++c) mf -= mu * v[k][c];
  }
  for (; t < d.N; ++t)
    for (int c = 0; c < NSEG; ++c) mf -= mu * evs[GI(t, NS, c)];
  double icomp = 0.0;
  if (ys_lo <= ys_hi) { icomp = dmax(icomp, fabs(ys_lo - mu)); icomp = dmax(icomp, fabs(ys_hi - mu)); }
  const double ipr = d.ms_res[b], idef = d.ms_res[(size_t)d.Bp + b], fcv = d.ms_res[(size_t)2 * d.Bp + b];
  d.inf_pr[b] = dmax(ipr, idef); d.merit[b] = mf; d.phi[b] = mf; d.inf_comp[b] = icomp;
  d.filter_theta[b] = fcv; d.theta[b] = fcv;
  d.filt[b] = mf; d.filt[(size_t)d.filt_cap * d.Bp + b] = fcv; d.filt_n[b] = 1;
}

// computeScaledDualInfeasibility (:1886-1930): inf_du / max(1, mean(|y|_1 + |s|_1) / 100), sums constraint-major as the reference walks them
template <class Model, class Cons>
DEV double ms_scaled_inf_du(const DevBuf &d, int b, int slot, double inf_du, double &ys_lo, double &ys_hi) {
  constexpr int NU = Model::NU, M = Cons::M, NSEG = Cons::NSEG;
  ys_lo = INFINITY; ys_hi = -INFINITY;   // extreme products y s of the iterate (the same pass; ms_replay_filter)
  if constexpr (M == 0) return inf_du;
  else {
    const double *Sc = d.S + (size_t)slot * d.planeM, *Yc = d.Y + (size_t)slot * d.planeM;
    double yn = 0.0, sn = 0.0;
#pragma unroll
    for (int c = 0; c < NSEG; ++c) {
      const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
      constexpr int kTB = 8;   // steps per round trip: the loads of kTB steps in flight, then the ordered sums
      int t = 0;
      for (; t + kTB - 1 < d.N; t += kTB) {
        double yv[kTB][M], sv[kTB][M];
#pragma unroll
        for (int k = 0; k < kTB; ++k)
#pragma unroll
          for (int i = 0; i < M; ++i) if (i < dim) { yv[k][i] = Yc[GI(t + k, M, off + i)]; sv[k][i] = Sc[GI(t + k, M, off + i)]; }
#pragma unroll
        for (int k = 0; k < kTB; ++k) {
          double ya = 0.0, sa = 0.0;
#pragma unroll
          for (int i = 0; i < M; ++i) if (i < dim) { ya += fabs(yv[k][i]); sa += fabs(sv[k][i]); const double ys = yv[k][i] * sv[k][i]; ys_lo = dmin(ys_lo, ys); ys_hi = dmax(ys_hi, ys); }
          yn += ya; sn += sa;
        }
      }
      for (; t < d.N; ++t) {
        double ya = 0.0, sa = 0.0;
        for (int i = 0; i < dim; ++i) { const double yv1 = Yc[GI(t, M, off + i)], sv1 = Sc[GI(t, M, off + i)]; ya += fabs(yv1); sa += fabs(sv1); const double ys = yv1 * sv1; ys_lo = dmin(ys_lo, ys); ys_hi = dmax(ys_hi, ys); }
        yn += ya; sn += sa;
      }
    }
    const int total = d.N * M;
    const int mpn = total + NU * d.N;
    const double num = mpn > 0 ? (yn + sn) / (double)mpn : 0.0;
    const double sd = dmax(100.0, num) / 100.0;
    return inf_du / sd;
  }
}

// ================================================================================ K0
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_init_msipddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int mode) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = M > 0 ? M : 1;
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.n_active = d.B;
  if (b >= d.B) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const bool existing = (mode == kInitWarmExisting);
  d.cur[b] = 0;
  d.iter[b] = 0; d.status[b] = CDDP_HIP_STATUS_RUNNING; d.phase[b] = PH_ACTIVE;
  d.n_bwd[b] = 0; d.n_fwd[b] = 0; d.n_fwd_steps[b] = 0; d.bwd_ok[b] = 0; d.filt_n[b] = 0;
  if (b < d.hist_batch) d.hist_n[b] = 0;
  if (!existing) d.reg[b] = o.reg_initial_value;
  d.dV0[b] = 0.0; d.dV1[b] = 0.0; d.step_norm[b] = 0.0; d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;
  double *X0 = d.X, *U0 = d.U, *F0 = d.F, *L0 = d.Lam;
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  if constexpr (M == 0) {   // the factor cache belongs to the solver object (ms_workspace): cleared on the handle's first initialize only
    if (d.ms_fresh) for (int t = 0; t < N; ++t) d.fac[GI(t, NU * NU + NU + 1, NU * NU + NU)] = 0.0;
  }
  double mu, cost = 0.0;
  if (mode == kInitCold) {   // :199-263
    mu = (M == 0) ? 1e-8 : o.barrier_mu_initial;
    // initializeDualSlackCostateVariables (:643-709): duals / slacks from the constraint values on the GUESS (X is rolled out below)
    for (int t = 0; t < N; ++t) {
      if constexpr (M > 0) {
        double x[NX], u[NU], g[MM], s[MM], y[MM];
        ld<NX>(X0 + GI(t, NX, 0), kLS, x); ld<NU>(U0 + GI(t, NU, 0), kLS, u);
        Cons::template eval<NX, NU>(cc, x, u, g);
#pragma unroll
        for (int j = 0; j < M; ++j) ms_init_pair(o, mu, g[j], s[j], y[j]);
        st<M>(d.S + GI(t, M, 0), kLS, s); st<M>(d.Y + GI(t, M, 0), kLS, y);
      }
      double lam[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) lam[i] = o.msipddp_costate_var_init_scale * 1.0;
      st<NX>(L0 + GI(t, NX, 0), kLS, lam);
    }
    // evaluateTrajectory (:425-455): rollout from the initial state, constraint values and dynamics values of the rolled-out iterate
    double x[NX];
    ld<NX>(X0 + GI(0, NX, 0), kLS, x);
    for (int t = 0; t < N; ++t) {
      double u[NU], xn[NX];
      ld<NU>(U0 + GI(t, NU, 0), kLS, u);
      cost += Obj::running_cost(P, xrt, t, x, u);
      if constexpr (M > 0) { double g[MM]; Cons::template eval<NX, NU>(cc, x, u, g); st<M>(d.G + GI(t, M, 0), kLS, g); }
      Stepper<Model>::step(P->integrator, P->dt, P->mp, x, u, xn);
      st<NX>(F0 + GI(t, NX, 0), kLS, xn);
      st<NX>(X0 + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = xn[i];
    }
    cost += Obj::terminal_cost(P, x);
  } else {
    // warm start (:95-197).  existing: gains / duals / costates of the last solve (staged into slot 0 by k_stage); provided: the state
    // guess is kept as it is -- the multiple-shooting start -- and the first iterate carries its defects
    if (!existing) {
      double lam[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) lam[i] = o.msipddp_costate_var_init_scale * 1.0;
      for (int t = 0; t < N; ++t) st<NX>(L0 + GI(t, NX, 0), kLS, lam);
    }
    cost = INFINITY;   // CDDP::initializeProblemIfNecessary (cddp_core.cpp:297): kept when no trajectory evaluation follows
    if (existing || M > 0) {   // evaluateTrajectoryWarmStart (:457-495)
      cost = 0.0;
      double x[NX];
      ld<NX>(X0 + GI(0, NX, 0), kLS, x);
      for (int t = 0; t < N; ++t) {
        double u[NU], xn[NX];
        ld<NU>(U0 + GI(t, NU, 0), kLS, u);
        cost += Obj::running_cost(P, xrt, t, x, u);
        if constexpr (M > 0) { double g[MM]; Cons::template eval<NX, NU>(cc, x, u, g); st<M>(d.G + GI(t, M, 0), kLS, g); }
        Stepper<Model>::step(P->integrator, P->dt, P->mp, x, u, xn);
        st<NX>(F0 + GI(t, NX, 0), kLS, xn);
        if (o.msipddp_use_controlled_rollout) {
          st<NX>(X0 + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
          for (int i = 0; i < NX; ++i) x[i] = xn[i];
        } else ld<NX>(X0 + GI(t + 1, NX, 0), kLS, x);
      }
      cost += Obj::terminal_cost(P, x);
    } else {
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      for (int t = 0; t < N; ++t) st<NX>(F0 + GI(t, NX, 0), kLS, z);   // dynamics_trajectory_ stays zero (:120)
    }
    if (existing) mu = o.barrier_mu_initial * 0.1;
    else if (M == 0) mu = 1e-8;
    else {
      double mv = 0.0;   // computeMaxConstraintViolation (interior_point_utils.cpp:141-155), constraint-major; a maximum is order-free
      if constexpr (M > 0) {
        for (int t = 0; t < N; ++t) { double g[MM]; ld<M>(d.G + GI(t, M, 0), kLS, g);
#pragma unroll
          for (int j = 0; j < M; ++j) mv = dmax(mv, g[j]); }
      }
      if (mv <= o.tolerance) mu = o.tolerance * 0.01;
      else if (mv <= 0.1) mu = o.tolerance;
      else mu = o.barrier_mu_initial * 0.1;
    }
    if constexpr (M > 0) {   // initializeDualSlackCostateVariablesWarmStart (:497-641)
      constexpr int NSEG = Cons::NSEG;
      for (int t = 0; t < N; ++t) {
        double g[MM], s[MM], y[MM];
        ld<M>(d.G + GI(t, M, 0), kLS, g);
        if (existing) { ld<M>(d.S + GI(t, M, 0), kLS, s); ld<M>(d.Y + GI(t, M, 0), kLS, y); }
#pragma unroll
        for (int c = 0; c < NSEG; ++c) {
          const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
          bool need = !existing;
          if (existing) {
            bool stop = false;
#pragma unroll
            for (int i = 0; i < MM; ++i)
              if (i < dim && !stop) {
                const int j = off + i;
                if (y[j] <= 1e-12 || s[j] <= 1e-12) { need = true; stop = true; }
                else {
                  const double required = dmax(o.ipddp_slack_var_init_scale, -g[j]);
                  if (s[j] < 0.1 * required) { need = true; stop = true; }
                }
              }
          }
          if (need) {
#pragma unroll
            for (int i = 0; i < MM; ++i) if (i < dim) ms_init_pair(o, mu, g[off + i], s[off + i], y[off + i]);
          }
        }
        st<M>(d.S + GI(t, M, 0), kLS, s); st<M>(d.Y + GI(t, M, 0), kLS, y);
      }
    }
  }
  d.cost[b] = cost; d.mu[b] = mu;
  d.alpha_pr[b] = o.ls_initial_step_size; d.alpha_du[b] = 0.0;
  d.inf_du[b] = INFINITY;
  ms_reset_filter<Model, Cons>(d, b, 0, mu, cost);
  hist_push(d, b, mu);
}

// ================================================================================ K2
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_msipddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = M > 0 ? M : 1;
  constexpr int FAC = NU * NU + NU + 1;
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  const double *Fc = d.F + (size_t)cur * d.planeX;
  const double *Lc = d.Lam + (size_t)cur * d.planeX;
  if (count_iter) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, idu = 0, ipr = 0, icomp = 0, idef = 0, snorm = 0;
  for (;;) {
    ++nb;
    double xN[NX], Vx[NX], Vxx[NX * NX];
    ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
    Obj::final_grad(P, xN, Vx);
    const double *Qf = P->pool + P->off_Qf;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * ((2.0 * Qf[i * NX + c]) + (2.0 * Qf[c * NX + i]));   // :1122
    st<NX>(d.Vx + GI(N, NX, 0), kLS, Vx);
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Vxx);
    dV0 = 0; dV1 = 0; idu = 0; ipr = 0; icomp = 0; idef = 0; snorm = 0;
    bool fail = false;
    // One step's inputs, fetched one step AHEAD (round 5): the sweep is one dependent instruction stream per wavefront, and the loads
    // of a step issued at its own top were a memory round trip on that chain every step (same arithmetic, same order).
    struct Rec { double A[NX * NX], Bm[NX * NU], x[NX], u[NU], lam[NX], f[NX], x1[NX], y[MM], sv[MM], g[MM]; };
    auto fetch = [&](int tt, Rec &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.x);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.u);
      ld<NX>(Lc + GI(tt, NX, 0), kLS, r.lam);
      ld<NX>(Fc + GI(tt, NX, 0), kLS, r.f); ld<NX>(Xc + GI(tt + 1, NX, 0), kLS, r.x1);
      if constexpr (M > 0) {
        ld<M>(d.Y + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.y);
        ld<M>(d.S + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.sv);
        ld<M>(d.G + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.g);
      }
    };
#ifndef CDDP_MS_SWEEP_PF
#define CDDP_MS_SWEEP_PF 1
#endif
    constexpr bool kPF = CDDP_MS_SWEEP_PF && sizeof(Rec) <= 56 * sizeof(double);
    auto step = [&](const int t, const Rec &c, Rec &n) -> bool {
      if constexpr (kPF) { fetch(t > 0 ? t - 1 : 0, n); PIPELINE_FENCE(); }
      const double (&A)[NX * NX] = c.A; const double (&Bm)[NX * NU] = c.Bm; const double (&x)[NX] = c.x; const double (&u)[NU] = c.u;
      const double (&lam)[NX] = c.lam;
      double dd[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dd[i] = c.f[i] - c.x1[i];   // defect (:1129-1131)
      double Vd[NX], w[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Vxx[i * NX + j] * dd[j];
        Vd[i] = s; w[i] = Vx[i] + s; }
      double Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU];
      Obj::lx(P, xrt, t, x, Qx);
      Obj::lu(P, u, Qu);
      [[maybe_unused]] double y[MM], sv[MM], g[MM], Gx[MM * NX], Gu[MM * NU];
      if constexpr (M > 0) {
#pragma unroll
        for (int i = 0; i < MM * NX; ++i) Gx[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MM * NU; ++i) Gu[i] = 0.0;
        Cons::template jac<NX, NU>(cc, x, u, Gx, Gu);
#pragma unroll
        for (int i = 0; i < M; ++i) { y[i] = c.y[i]; sv[i] = c.sv[i]; g[i] = c.g[i]; }
#pragma unroll
        for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s += Gx[r * NX + i] * y[r];
          Qx[i] = Qx[i] + s; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s += Gu[r * NU + i] * y[r];
          Qu[i] = Qu[i] + s; }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += A[k * NX + i] * w[k];
        Qx[i] = Qx[i] + s; }
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bm[k * NU + i] * w[k];
        Qu[i] = Qu[i] + s; }
      q_blocks<NX, NU>(P, A, Bm, Vx, Vxx, Qxx, Qux, Quu);
      if constexpr (Model::kHasHess) {   // full DDP: the tensors weighted with the costates (:1151-1163, :1279-1310); the rows of the
        if (!o.use_ilqr) lg_tensor_terms<Model>(P, x, u, lam, Qxx, Qux, Quu);   // layouts served here have no second derivatives
      }
      double kk[NU], KK[NU * NX];
      // k_lambda, K_lambda (:1195-1196 == :1383-1384): K_lambda = sym(V_xx(t+1)) is V_xx(t+1) itself (stored symmetric, d.Vxx[t+1])
      {
        double kl[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) kl[i] = ((0.0 - lam[i]) + Vx[i]) + Vd[i];
        st<NX>(d.kl + GI(t, NX, 0), kLS, kl);
      }
      if constexpr (M == 0) {
        double Qs[NU * NU];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]);
#pragma unroll
        for (int i = 0; i < NU; ++i) Qs[i * NU + i] += reg;
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Quu[i] = Qs[i];   // the reference regularises Q_uu itself in this branch (:1166-1167)
        LDLTs<NU> f;
        double *fc = d.fac + GI(t, FAC, 0);
        const bool valid = fc[(size_t)(NU * NU + NU) * kLS] != 0.0;
        if (!valid) {
          f.compute(Qs, NU);
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) fc[(size_t)i * kLS] = f.m[i];
#pragma unroll
          for (int i = 0; i < NU; ++i) fc[(size_t)(NU * NU + i) * kLS] = (double)f.tr[i];
          fc[(size_t)(NU * NU + NU) * kLS] = f.ok ? 1.0 : 0.0;   // a failed factor is dropped (:1178-1182)
        } else {
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) f.m[i] = fc[(size_t)i * kLS];
#pragma unroll
          for (int i = 0; i < NU; ++i) f.tr[i] = (int)fc[(size_t)(NU * NU + i) * kLS];
          f.ok = true;
        }
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
        }
      } else {
        double YS[MM], pres[MM], cres[MM], rhat[MM], Sir[MM];
#pragma unroll
        for (int i = 0; i < M; ++i) {
          YS[i] = y[i] / sv[i];
          pres[i] = g[i] + sv[i];
          cres[i] = y[i] * sv[i] - mu; rhat[i] = y[i] * pres[i] - cres[i]; Sir[i] = rhat[i] / sv[i];
        }
        // T_u = Q_yu^T Y S^-1 (nu x m), T_x = Q_yx^T Y S^-1 (nx x m): one non-zero product per entry of the diagonal factor
        double GuYG[NU * NU], GuYGx[NU * NX], GxYGx[NX * NX], GxYGu[NX * NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
          for (int c = 0; c < NU; ++c) { double s = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) s += (Gu[r * NU + i] * YS[r]) * Gu[r * NU + c];
            GuYG[i * NU + c] = s; }
#pragma unroll
          for (int c = 0; c < NX; ++c) { double s = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) s += (Gu[r * NU + i] * YS[r]) * Gx[r * NX + c];
            GuYGx[i * NX + c] = s; }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
          for (int c = 0; c < NX; ++c) { double s = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) s += (Gx[r * NX + i] * YS[r]) * Gx[r * NX + c];
            GxYGx[i * NX + c] = s; }
#pragma unroll
          for (int c = 0; c < NU; ++c) { double s = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) s += (Gx[r * NX + i] * YS[r]) * Gu[r * NU + c];
            GxYGu[i * NU + c] = s; }
        }
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int c = 0; c < NU; ++c) Qr[i * NU + c] = (0.5 * (Quu[i * NU + c] + Quu[c * NU + i])) + GuYG[i * NU + c];
#pragma unroll
        for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
        LDLTs<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) return false;
        double GuS[NU], GxS[NX];
#pragma unroll
        for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s += Gu[r * NU + i] * Sir[r];
          GuS[i] = s; }
#pragma unroll
        for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s += Gx[r * NX + i] * Sir[r];
          GxS[i] = s; }
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i] + GuS[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c] + GuYGx[i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
        }
        double temp[MM], ky[MM], ksv[MM], Ky[MM * NX], Ks[MM * NX];
#pragma unroll
        for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
          for (int i = 0; i < NU; ++i) s += Gu[r * NU + i] * kk[i];
          temp[r] = s;
          ky[r] = (rhat[r] + y[r] * s) / sv[r];
          ksv[r] = (0.0 - pres[r]) - s; }
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
          for (int c = 0; c < NX; ++c) { double s = 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) s += Gu[r * NU + i] * KK[i * NX + c];
            Ky[r * NX + c] = YS[r] * (Gx[r * NX + c] + s);
            Ks[r * NX + c] = (0.0 - Gx[r * NX + c]) - s; }
        st<M>(d.ky + GI(t, M, 0), kLS, ky); st<M>(d.ks + GI(t, M, 0), kLS, ksv);
        st<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky); st<M * NX>(d.Ks + GI(t, M * NX, 0), kLS, Ks);
        // the condensed blocks (:1391-1400), Q_ux with the reference's layout (:1398)
#pragma unroll
        for (int i = 0; i < NU; ++i) Qu[i] = Qu[i] + GuS[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) Qx[i] = Qx[i] + GxS[i];
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) Qxx[i] = Qxx[i] + GxYGx[i];
        if constexpr (NU == 1) {
#pragma unroll
          for (int c = 0; c < NX; ++c) Qux[c] = Qux[c] + GxYGu[c];
        } else {
          static_assert(NX == NU, "msipddp_solver.cpp:1398 defines the constrained recursion for nu = 1 or nx = nu only");
#pragma unroll
          for (int i = 0; i < NU * NX; ++i) Qux[i] = Qux[i] + GxYGu[i];
        }
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Quu[i] = Quu[i] + GuYG[i];
#pragma unroll
        for (int r = 0; r < M; ++r) { ipr = dmax(ipr, fabs(pres[r])); icomp = dmax(icomp, fabs(cres[r])); }
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      // dV, V_x, V_xx (:1198-1206 == :1402-1410):  V_x = Q_x + K^T Q_u + Q_ux^T k + (K^T Q_uu) k,  V_xx likewise, then symmetrised
      { double s0 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) s0 += kk[i] * Qu[i];
        dV0 += s0;
        double s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) { double q = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) q += Quu[i * NU + j] * kk[j];
          s1 += kk[i] * q; }
        dV1 += 0.5 * s1; }
      double KtQ[NX * NU];
      mm_tn<NX, NU, NU>(KK, Quu, KtQ);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KK[j * NX + i] * Qu[j]; bb += Qux[j * NX + i] * kk[j]; c += KtQ[i * NU + j] * kk[j]; }
        Vx[i] = ((Qx[i] + a) + bb) + c;
      }
      double Vn[NX * NX];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { a += KK[j * NX + i] * Qux[j * NX + c]; bb += Qux[j * NX + i] * KK[j * NX + c]; e += KtQ[i * NU + j] * KK[j * NX + c]; }
          Vn[i * NX + c] = ((Qxx[i * NX + c] + a) + bb) + e;
        }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (Vn[i * NX + c] + Vn[c * NX + i]);
      st<NX>(d.Vx + GI(t, NX, 0), kLS, Vx);
      st<NX * NX>(d.Vxx + GI(t, NX * NX, 0), kLS, Vxx);
#pragma unroll
      for (int i = 0; i < NU; ++i) { idu = dmax(idu, fabs(Qu[i])); snorm = dmax(snorm, fabs(kk[i])); }
#pragma unroll
      for (int i = 0; i < NX; ++i) idef = dmax(idef, fabs(dd[i]));
      return true;
    };
    if constexpr (kPF) {
      Rec ra, rb;
      fetch(N - 1, ra);
      int t = N - 1;
      for (; t >= 1; t -= 2) {
        if (!step(t, ra, rb)) { fail = true; break; }
        if (!step(t - 1, rb, ra)) { fail = true; break; }
      }
      if (!fail && t == 0) fail = !step(0, ra, rb);
    } else {
      for (int t = N - 1; t >= 0; --t) { Rec r; fetch(t, r); if (!step(t, r, r)) { fail = true; break; } }
    }
    if (!fail) { ok = true; break; }
    if (force == 2) break;   // single un-retried pass (step-level API)
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = idu; d.step_norm[b] = snorm;
    if constexpr (M == 0) { d.inf_pr[b] = idef; d.inf_comp[b] = 0.0; }
    else { d.inf_pr[b] = dmax(ipr, idef); d.inf_comp[b] = icomp; }
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }   // cddp_solver_base.cpp:98-109
  d.phase[b] = PH_FWD1;   // no early convergence test (base-class default)
}

// ================================================================================ K2, path-constrained, split (round 5)
// The path-constrained sweep above spends most of its one-wave chain on work that does not depend on the value function: the constraint
// Jacobians, Y S^-1, the residuals (three divisions per row) and the condensation products in front of the factorisation, the slack / dual
// gains (another division per row) behind it.  Split as the IPDDP sweep is (kernels_lean.hpp):
//   k_ms_condense   grid (batch x N)   l_x + G_x^T y, l_u + G_u^T y, the defect, Y S^-1, r_hat, the primal residual, the condensation
//                                      products G^T (Y S^-1) G, G^T S^-1 r_hat and the step's residual maxima      -> d.cst [N][MsCst::SIZE]
//   k_backward_msipddp_lean  (batch)   the value recursion only: w = V_x + V_xx d, Q blocks, regularised LDLT, k, K, k_lambda, V update
//   k_ms_post       grid (batch x N)   k_y, k_s, K_y, K_s from k, K                                                  (:1367-1384)
// Every entry is formed by the expression of the fused kernel in the same order (tests/test_msipddp_device.py::
// test_split_and_fused_sweeps_agree_bitwise); full DDP (costate-weighted tensor terms) keeps the fused kernel.
template <class Model, class Cons>
struct MsCst {
  static constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M > 0 ? Cons::M : 1;
  static constexpr int oQx = 0, oQu = oQx + NX, oDd = oQu + NU, oGuYG = oDd + NX, oGuYGx = oGuYG + NU * NU, oGxYGx = oGuYGx + NU * NX,
                       oGxYGu = oGxYGx + NX * NX, oGuS = oGxYGu + NX * NU, oGxS = oGuS + NU, oRes = oGxS + NX /* ipr, icomp of the step */,
                       oSweep = oRes + 2 /* what the sweep reads ends here */, oYS = oSweep, oRhat = oYS + M, oPres = oRhat + M, SIZE = oPres + M;
};

// WITH_DERIVS (register-resident plants): the same (batch x N) pass also writes A_t = I + dt f_x, B_t = dt f_u (K1: x, u are read once and the
// iteration has one launch less, as k_condense<.., true> does for IPDDP); it is then the first kernel of the iteration and resets the counter.
template <class Model, class Cons, bool WITH_DERIVS = false>
__global__ __launch_bounds__(64) void k_ms_condense(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = M > 0 ? M : 1;
  typedef Objective<NX, NU> Obj;
  typedef MsCst<Model, Cons> L;
  if constexpr (M > 0) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int t = blockIdx.y;
    if constexpr (WITH_DERIVS) { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && !force) *d.n_active = 0; }
    if (b >= d.B) return;
    if (!force && d.phase[b] != PH_ACTIVE) return;
    const ProblemDev *__restrict__ P = Pk;
    const int cur = d.cur[b];
    const double mu = d.mu[b];
    typename Cons::Ctx cc;
    Cons::load(P, cc);
    double x[NX], u[NU], f[NX], x1[NX], y[MM], sv[MM], g[MM];
    ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, x);
    ld<NU>(d.U + (size_t)cur * d.planeU + GI(t, NU, 0), kLS, u);
    if constexpr (WITH_DERIVS) {   // identical to k_derivs (plain layout: MSIPDDP handles never use the sub-tile-minor stacks)
      double Fx[NX * NX], Fu[NX * NU];
      Model::jac(P->mp, x, u, Fx, Fu);
      const double dt = P->dt;
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          double a = dt * Fx[i * NX + j];
          if (i == j) a += 1.0;
          d.A[GI(t, NX * NX, i * NX + j)] = a;
        }
#pragma unroll
      for (int i = 0; i < NX * NU; ++i) d.Bm[GI(t, NX * NU, i)] = dt * Fu[i];
    }
    ld<NX>(d.F + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, f);
    ld<NX>(d.X + (size_t)cur * d.planeX + GI(t + 1, NX, 0), kLS, x1);
    ld<M>(d.Y + (size_t)cur * d.planeM + GI(t, M, 0), kLS, y);
    ld<M>(d.S + (size_t)cur * d.planeM + GI(t, M, 0), kLS, sv);
    ld<M>(d.G + (size_t)cur * d.planeM + GI(t, M, 0), kLS, g);
    double *out = d.cst + GI(t, L::SIZE, 0);
    double dd[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) dd[i] = f[i] - x1[i];
    st<NX>(out + (size_t)L::oDd * kLS, kLS, dd);
    double Qx[NX], Qu[NU], Gx[MM * NX], Gu[MM * NU];
    Obj::lx(P, xrt, t, x, Qx);
    Obj::lu(P, u, Qu);
#pragma unroll
    for (int i = 0; i < MM * NX; ++i) Gx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < MM * NU; ++i) Gu[i] = 0.0;
    Cons::template jac<NX, NU>(cc, x, u, Gx, Gu);
#pragma unroll
    for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
      for (int r = 0; r < M; ++r) s += Gx[r * NX + i] * y[r];
      Qx[i] = Qx[i] + s; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
      for (int r = 0; r < M; ++r) s += Gu[r * NU + i] * y[r];
      Qu[i] = Qu[i] + s; }
    st<NX>(out + (size_t)L::oQx * kLS, kLS, Qx);
    st<NU>(out + (size_t)L::oQu * kLS, kLS, Qu);
    double YS[MM], pres[MM], cres[MM], rhat[MM], Sir[MM];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      YS[i] = y[i] / sv[i];
      pres[i] = g[i] + sv[i];
      cres[i] = y[i] * sv[i] - mu; rhat[i] = y[i] * pres[i] - cres[i]; Sir[i] = rhat[i] / sv[i];
    }
    st<M>(out + (size_t)L::oYS * kLS, kLS, YS); st<M>(out + (size_t)L::oRhat * kLS, kLS, rhat); st<M>(out + (size_t)L::oPres * kLS, kLS, pres);
    double GuYG[NU * NU], GuYGx[NU * NX], GxYGx[NX * NX], GxYGu[NX * NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
#pragma unroll
      for (int c = 0; c < NU; ++c) { double s = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s += (Gu[r * NU + i] * YS[r]) * Gu[r * NU + c];
        GuYG[i * NU + c] = s; }
#pragma unroll
      for (int c = 0; c < NX; ++c) { double s = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s += (Gu[r * NU + i] * YS[r]) * Gx[r * NX + c];
        GuYGx[i * NX + c] = s; }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int c = 0; c < NX; ++c) { double s = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s += (Gx[r * NX + i] * YS[r]) * Gx[r * NX + c];
        GxYGx[i * NX + c] = s; }
#pragma unroll
      for (int c = 0; c < NU; ++c) { double s = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s += (Gx[r * NX + i] * YS[r]) * Gu[r * NU + c];
        GxYGu[i * NU + c] = s; }
    }
    st<NU * NU>(out + (size_t)L::oGuYG * kLS, kLS, GuYG); st<NU * NX>(out + (size_t)L::oGuYGx * kLS, kLS, GuYGx);
    st<NX * NX>(out + (size_t)L::oGxYGx * kLS, kLS, GxYGx); st<NX * NU>(out + (size_t)L::oGxYGu * kLS, kLS, GxYGu);
    double GuS[NU], GxS[NX];
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
      for (int r = 0; r < M; ++r) s += Gu[r * NU + i] * Sir[r];
      GuS[i] = s; }
#pragma unroll
    for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
      for (int r = 0; r < M; ++r) s += Gx[r * NX + i] * Sir[r];
      GxS[i] = s; }
    st<NU>(out + (size_t)L::oGuS * kLS, kLS, GuS); st<NX>(out + (size_t)L::oGxS * kLS, kLS, GxS);
    double ipr = 0.0, icomp = 0.0;   // the step's share of the residual maxima (the sweep takes the maximum over the steps: order-free)
#pragma unroll
    for (int r = 0; r < M; ++r) { ipr = dmax(ipr, fabs(pres[r])); icomp = dmax(icomp, fabs(cres[r])); }
    out[(size_t)(L::oRes + 0) * kLS] = ipr; out[(size_t)(L::oRes + 1) * kLS] = icomp;
  }
}

template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_ms_post(DevBuf d, const ProblemDev *__restrict__ Pk, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = M > 0 ? M : 1;
  typedef MsCst<Model, Cons> L;
  if constexpr (M > 0) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int t = blockIdx.y;
    if (b >= d.B) return;
    if (force) { if (!d.bwd_ok[b]) return; }
    else if (d.phase[b] != PH_FWD1) return;
    const ProblemDev *__restrict__ P = Pk;
    const int cur = d.cur[b];
    typename Cons::Ctx cc;
    Cons::load(P, cc);
    double x[NX], u[NU], y[MM], sv[MM], kk[NU], KK[NU * NX], YS[MM], rhat[MM], pres[MM];
    ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, x);
    ld<NU>(d.U + (size_t)cur * d.planeU + GI(t, NU, 0), kLS, u);
    ld<M>(d.Y + (size_t)cur * d.planeM + GI(t, M, 0), kLS, y);
    ld<M>(d.S + (size_t)cur * d.planeM + GI(t, M, 0), kLS, sv);
    ld<NU>(d.k + GI(t, NU, 0), kLS, kk);
    ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
    const double *in = d.cst + GI(t, L::SIZE, 0);
    ld<M>(in + (size_t)L::oYS * kLS, kLS, YS); ld<M>(in + (size_t)L::oRhat * kLS, kLS, rhat); ld<M>(in + (size_t)L::oPres * kLS, kLS, pres);
    double Gx[MM * NX], Gu[MM * NU];
#pragma unroll
    for (int i = 0; i < MM * NX; ++i) Gx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < MM * NU; ++i) Gu[i] = 0.0;
    Cons::template jac<NX, NU>(cc, x, u, Gx, Gu);
    double ky[MM], ksv[MM], Ky[MM * NX], Ks[MM * NX];
#pragma unroll
    for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) s += Gu[r * NU + i] * kk[i];
      ky[r] = (rhat[r] + y[r] * s) / sv[r];
      ksv[r] = (0.0 - pres[r]) - s; }
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int c = 0; c < NX; ++c) { double s = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) s += Gu[r * NU + i] * KK[i * NX + c];
        Ky[r * NX + c] = YS[r] * (Gx[r * NX + c] + s);
        Ks[r * NX + c] = (0.0 - Gx[r * NX + c]) - s; }
    st<M>(d.ky + GI(t, M, 0), kLS, ky); st<M>(d.ks + GI(t, M, 0), kLS, ksv);
    st<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky); st<M * NX>(d.Ks + GI(t, M * NX, 0), kLS, Ks);
  }
}

template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_msipddp_lean(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef Objective<NX, NU> Obj;
  typedef MsCst<Model, Cons> L;
  static_assert(M > 0, "the split sweep is the path-constrained branch");
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Lc = d.Lam + (size_t)cur * d.planeX;
  if (count_iter) d.iter[b] += 1;
  double reg = d.reg[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, idu = 0, ipr = 0, icomp = 0, idef = 0, snorm = 0;
  for (;;) {
    ++nb;
    double xN[NX], Vx[NX], Vxx[NX * NX];
    ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
    Obj::final_grad(P, xN, Vx);
    const double *Qf = P->pool + P->off_Qf;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * ((2.0 * Qf[i * NX + c]) + (2.0 * Qf[c * NX + i]));   // :1122
    st<NX>(d.Vx + GI(N, NX, 0), kLS, Vx);
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Vxx);
    dV0 = 0; dV1 = 0; idu = 0; ipr = 0; icomp = 0; idef = 0; snorm = 0;
    bool fail = false;
    struct Rec { double A[NX * NX], Bm[NX * NU], lam[NX], c[L::oSweep]; };
    auto fetch = [&](int tt, Rec &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<NX>(Lc + GI(tt, NX, 0), kLS, r.lam);
      ld<L::oSweep>(d.cst + GI(tt, L::SIZE, 0), kLS, r.c);
    };
    constexpr bool kPF = sizeof(Rec) <= 72 * sizeof(double);
    auto step = [&](const int t, const Rec &c, Rec &n) -> bool {
      if constexpr (kPF) { fetch(t > 0 ? t - 1 : 0, n); PIPELINE_FENCE(); }
      const double (&A)[NX * NX] = c.A; const double (&Bm)[NX * NU] = c.Bm; const double (&lam)[NX] = c.lam;
      const double *dd = c.c + L::oDd, *GuYG = c.c + L::oGuYG, *GuYGx = c.c + L::oGuYGx, *GxYGx = c.c + L::oGxYGx, *GxYGu = c.c + L::oGxYGu,
                   *GuS = c.c + L::oGuS, *GxS = c.c + L::oGxS;
      double Vd[NX], w[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Vxx[i * NX + j] * dd[j];
        Vd[i] = s; w[i] = Vx[i] + s; }
      double Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) Qx[i] = c.c[L::oQx + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] = c.c[L::oQu + i];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += A[k * NX + i] * w[k];
        Qx[i] = Qx[i] + s; }
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bm[k * NU + i] * w[k];
        Qu[i] = Qu[i] + s; }
      q_blocks<NX, NU>(P, A, Bm, Vx, Vxx, Qxx, Qux, Quu);
      double kk[NU], KK[NU * NX];
      {
        double kl[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) kl[i] = ((0.0 - lam[i]) + Vx[i]) + Vd[i];
        st<NX>(d.kl + GI(t, NX, 0), kLS, kl);
      }
      double Qr[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int cc2 = 0; cc2 < NU; ++cc2) Qr[i * NU + cc2] = (0.5 * (Quu[i * NU + cc2] + Quu[cc2 * NU + i])) + GuYG[i * NU + cc2];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
      LDLTs<NU> f;
      f.compute(Qr, NU);
      if (!f.ok) return false;
      double col[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) col[i] = Qu[i] + GuS[i];
      f.solve(col);
#pragma unroll
      for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
      for (int cc2 = 0; cc2 < NX; ++cc2) {
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + cc2] + GuYGx[i * NX + cc2];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) KK[i * NX + cc2] = -col[i];
      }
      // the condensed blocks (:1391-1400), Q_ux with the reference's layout (:1398)
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] = Qu[i] + GuS[i];
#pragma unroll
      for (int i = 0; i < NX; ++i) Qx[i] = Qx[i] + GxS[i];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) Qxx[i] = Qxx[i] + GxYGx[i];
      if constexpr (NU == 1) {
#pragma unroll
        for (int cc2 = 0; cc2 < NX; ++cc2) Qux[cc2] = Qux[cc2] + GxYGu[cc2];
      } else {
        static_assert(NX == NU, "msipddp_solver.cpp:1398 defines the constrained recursion for nu = 1 or nx = nu only");
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) Qux[i] = Qux[i] + GxYGu[i];
      }
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] = Quu[i] + GuYG[i];
      ipr = dmax(ipr, c.c[L::oRes + 0]); icomp = dmax(icomp, c.c[L::oRes + 1]);
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      { double s0 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) s0 += kk[i] * Qu[i];
        dV0 += s0;
        double s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) { double q = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) q += Quu[i * NU + j] * kk[j];
          s1 += kk[i] * q; }
        dV1 += 0.5 * s1; }
      double KtQ[NX * NU];
      mm_tn<NX, NU, NU>(KK, Quu, KtQ);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = 0.0, bb = 0.0, c3 = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KK[j * NX + i] * Qu[j]; bb += Qux[j * NX + i] * kk[j]; c3 += KtQ[i * NU + j] * kk[j]; }
        Vx[i] = ((Qx[i] + a) + bb) + c3;
      }
      double Vn[NX * NX];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int cc2 = 0; cc2 < NX; ++cc2) {
          double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { a += KK[j * NX + i] * Qux[j * NX + cc2]; bb += Qux[j * NX + i] * KK[j * NX + cc2]; e += KtQ[i * NU + j] * KK[j * NX + cc2]; }
          Vn[i * NX + cc2] = ((Qxx[i * NX + cc2] + a) + bb) + e;
        }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int cc2 = 0; cc2 < NX; ++cc2) Vxx[i * NX + cc2] = 0.5 * (Vn[i * NX + cc2] + Vn[cc2 * NX + i]);
      st<NX>(d.Vx + GI(t, NX, 0), kLS, Vx);
      st<NX * NX>(d.Vxx + GI(t, NX * NX, 0), kLS, Vxx);
#pragma unroll
      for (int i = 0; i < NU; ++i) { idu = dmax(idu, fabs(Qu[i])); snorm = dmax(snorm, fabs(kk[i])); }
#pragma unroll
      for (int i = 0; i < NX; ++i) idef = dmax(idef, fabs(dd[i]));
      return true;
    };
    if constexpr (kPF) {
      Rec ra, rb;
      fetch(N - 1, ra);
      int t = N - 1;
      for (; t >= 1; t -= 2) {
        if (!step(t, ra, rb)) { fail = true; break; }
        if (!step(t - 1, rb, ra)) { fail = true; break; }
      }
      if (!fail && t == 0) fail = !step(0, ra, rb);
    } else {
      for (int t = N - 1; t >= 0; --t) { Rec r; fetch(t, r); if (!step(t, r, r)) { fail = true; break; } }
    }
    if (!fail) { ok = true; break; }
    if (force == 2) break;
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = idu; d.step_norm[b] = snorm;
    d.inf_pr[b] = dmax(ipr, idef); d.inf_comp[b] = icomp;
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  d.phase[b] = PH_FWD1;
}

// ================================================================================ K4
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_forward_msipddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int a0, int na, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, NSEG = Cons::NSEG, MM = M > 0 ? M : 1;
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int a = a0 + blockIdx.y;
  (void)na;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const int slot = trial_slot(cur, a);
  const double *Xc = d.X + (size_t)cur * d.planeX, *Uc = d.U + (size_t)cur * d.planeU;
  const double *Fc = d.F + (size_t)cur * d.planeX, *Lc = d.Lam + (size_t)cur * d.planeX;
  double *Xn = d.X + (size_t)slot * d.planeX, *Un = d.U + (size_t)slot * d.planeU;
  double *Fn = d.F + (size_t)slot * d.planeX, *Ln = d.Lam + (size_t)slot * d.planeX;
  const double alpha = P->alphas[a];
  const double mu = d.mu[b];
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  const int seg = o.msipddp_segment_length, rtype = o.msipddp_rollout_type;
  const int n_alphas = d.n_alphas;
  atomicAdd(d.launched, 1ull);
  DynCtx dc;
  dc.load(P->integrator, P->dt, P->mp);
  typename Obj::Ctx oc;
  Obj::load(P, oc);
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  double x[NX];
  ld<NX>(Xc + GI(0, NX, 0), kLS, x);     // r.X[0] = initial state (:1443; X_[0] is the initial state for every iterate)
  st<NX>(Xn + GI(0, NX, 0), kLS, x);
  double cost = 0.0, merit_b = 0.0, cv = 0.0;
  [[maybe_unused]] double r_ipr = 0.0, r_idef = 0.0;   // max |g + s|, max |F_t - x_{t+1}| of the trial (kept for K5's filter reset, ms_replay_filter)
  [[maybe_unused]] double *evn = (M > 0) ? d.ev + (size_t)slot * ms_ev_plane<Cons>(d) : nullptr;
  bool alive = true;
  int steps = N;
  unsigned int ymask = 0xffffffffu;      // dual step sizes of the ladder that keep every row above its fraction-to-boundary bound so far
  // the ladder in registers (one scalar fetch per entry in front of the step loop instead of one per row, entry and step inside it)
  constexpr int kAL = 16;
  [[maybe_unused]] double al[kAL];
#pragma unroll
  for (int q = 0; q < kAL; ++q) al[q] = (q < n_alphas) ? P->alphas[q] : 0.0;
  // One step's inputs: nominal state / control, gains, slack / dual rows and their gains, costate and its gains.  Fetched one step ahead
  // (the rollout is ONE dependent instruction stream per wavefront: a load issued at the top of its own step is a full memory round
  // trip on the chain) when the record fits the register file twice.
  struct Rec {
    double xo[NX], uo[NU], kk[NU], KK[NU * NX], lo[NX], kl[NX], Kl[NX * NX];
    double so[MM], ksv[MM], Ks[MM * NX], yo[MM], ky[MM], Ky[MM * NX];
  };
  constexpr int REC = 3 * NX + 2 * NU + NU * NX + NX * NX + (M > 0 ? 4 * M + 2 * M * NX : 0);
  constexpr bool kPF = REC <= 32;   // cart-pole with its box (58 doubles): two copies took 256 VGPR + 168 AGPR, i.e. accvgpr moves on the chain
  auto fetch = [&](int tt, Rec &r) {
    ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
    ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
    ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
    ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
    ld<NX>(Lc + GI(tt, NX, 0), kLS, r.lo);
    ld<NX>(d.kl + GI(tt, NX, 0), kLS, r.kl);
    ld<NX * NX>(d.Vxx + GI(tt + 1, NX * NX, 0), kLS, r.Kl);
    if constexpr (M > 0) {
      ld<M>(d.S + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.so);
      ld<M>(d.ks + GI(tt, M, 0), kLS, r.ksv);
      ld<M * NX>(d.Ks + GI(tt, M * NX, 0), kLS, r.Ks);
      ld<M>(d.Y + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.yo);
      ld<M>(d.ky + GI(tt, M, 0), kLS, r.ky);
      ld<M * NX>(d.Ky + GI(tt, M * NX, 0), kLS, r.Ky);
    }
  };
  auto step = [&](const int t, const Rec &c, Rec &n) {
    if constexpr (kPF) { fetch(t + 1 < N ? t + 1 : N - 1, n); PIPELINE_FENCE(); }
    double dx[NX], u[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[i] = x[i] - c.xo[i];
    [[maybe_unused]] double sn[MM];
    if constexpr (M > 0) {   // slack trial and its fraction-to-boundary test (:1547-1560): the trial is abandoned at the first violation
#pragma unroll
      for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += c.Ks[r * NX + j] * dx[j];
        sn[r] = (c.so[r] + alpha * c.ksv[r]) + s;
        const bool viol = alive && (sn[r] < (1.0 - tau) * c.so[r]);
        steps = viol ? t : steps; alive = alive && !viol; }
      st<M>(d.S + (size_t)slot * d.planeM + GI(t, M, 0), kLS, sn);
      // dual trials y + a_y k_y + K_y dx for every a_y of the ladder (:1612-1644): feasibility only; the rows are written once a_y is known
#pragma unroll
      for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += c.Ky[r * NX + j] * dx[j];
        const double bound = (1.0 - tau) * c.yo[r];
        // (branch-free: bits of ladder entries beyond n_alphas are set or not, and never read)
        unsigned int bad = 0u;
#pragma unroll
        for (int q = 0; q < kAL; ++q) {
          const double yn = (c.yo[r] + al[q] * c.ky[r]) + s;
          bad |= (yn < bound) ? (1u << q) : 0u;
        }
        ymask &= ~bad;
        for (int q = kAL; q < n_alphas; ++q) {   // a ladder longer than the register copy (CDDP_HIP_MAX_ALPHAS = 32)
          const double yn = (c.yo[r] + P->alphas[q] * c.ky[r]) + s;
          if (yn < bound) ymask &= ~(1u << q);
        } }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += c.KK[i * NX + j] * dx[j];
      u[i] = (c.uo[i] + alpha * c.kk[i]) + s; }
    {   // costate trial (:1466-1467 == :1639-1641): lambda + a k_lambda + K_lambda dx, K_lambda = V_xx(t+1)
      double ln[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += c.Kl[i * NX + j] * dx[j];
        ln[i] = (c.lo[i] + alpha * c.kl[i]) + s; }
      st<NX>(Ln + GI(t, NX, 0), kLS, ln);
    }
    double fn[NX], xn[NX];
    Stepper<Model>::step(dc, x, u, fn);
    // next state: the dynamics value inside a segment, a gap-closing rule at a segment boundary (:1483-1509 == :1575-1601)
    const bool boundary = (seg > 1) && ((t + 1) % seg == 0) && (t + 1 < N);
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = fn[i];
    if (boundary && rtype != 1) {
      double fo[NX], x1[NX];
      ld<NX>(Fc + GI(t, NX, 0), kLS, fo);
      ld<NX>(Xc + GI(t + 1, NX, 0), kLS, x1);
      if (rtype == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = (x1[i] + (fn[i] - fo[i])) + alpha * (fo[i] - x1[i]);
      } else {   // "hybrid": linearised closed-loop step
        double A[NX * NX], Bm[NX * NU];
        ld<NX * NX>(d.A + GI(t, NX * NX, 0), kLS, A);
        ld<NX * NU>(d.Bm + GI(t, NX * NU, 0), kLS, Bm);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double lin = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) { double bk = 0.0;
#pragma unroll
            for (int q = 0; q < NU; ++q) bk += Bm[i * NU + q] * c.KK[q * NX + j];
            lin += (A[i * NX + j] + bk) * dx[j]; }
          double bkk = 0.0;
#pragma unroll
          for (int q = 0; q < NU; ++q) bkk += Bm[i * NU + q] * c.kk[q];
          xn[i] = (x1[i] + lin) + alpha * ((bkk + fo[i]) - x1[i]);
        }
      }
    }
    cost += Obj::running_cost(oc, xrt, t, x, u);
    if constexpr (M > 0) {   // constraint values, barrier and violation terms of the trial (:1650-1672)
      double g[MM];
      Cons::template eval<NX, NU>(cc, x, u, g);
      st<M>(d.G + (size_t)slot * d.planeM + GI(t, M, 0), kLS, g);
#pragma unroll
      for (int cs = 0; cs < NSEG; ++cs) {
        const int off = Cons::seg_off(cs), dim = Cons::seg_dim(cs);
        double lsum = 0.0, l1 = 0.0;
#pragma unroll
        for (int i = 0; i < MM; ++i) if (i < dim) { lsum += solver_log(sn[off + i]); const double pr = g[off + i] + sn[off + i]; l1 += fabs(pr); r_ipr = dmax(r_ipr, fabs(pr)); }
        merit_b -= mu * lsum; cv += l1;
        evn[GI(t, (NSEG > 0 ? NSEG : 1), cs)] = lsum;
      }
      double n1 = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) { const double r = fn[i] - xn[i]; n1 += fabs(r); r_idef = dmax(r_idef, fabs(r)); }
      cv += n1;
    }
    st<NU>(Un + GI(t, NU, 0), kLS, u);
    st<NX>(Fn + GI(t, NX, 0), kLS, fn);
    st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  };
  if constexpr (kPF) {
    Rec ra, rb;
    fetch(0, ra);
    int t = 0;
    for (; t + 1 < N; t += 2) { step(t, ra, rb); step(t + 1, rb, ra); }
    if (t < N) step(t, ra, rb);
  } else {
    for (int t = 0; t < N; ++t) { Rec r; fetch(t, r); step(t, r, r); }
  }
  cost += Obj::terminal_cost(P, x);
  const size_t ti = (size_t)a * d.Bp + b;
  bool success = false;
  double merit = cost, theta = 0.0, adu = 1.0;
  if constexpr (M == 0) {   // :1516-1530: expected-reduction ratio test
    const double dJ = d.cost[b] - cost;
    const double expected = -alpha * (d.dV0[b] + 0.5 * alpha * d.dV1[b]);
    const double ratio = expected > 0.0 ? dJ / expected : sign_of_reduction(dJ);
    success = ratio > 1e-6;
  } else {
    merit = merit_b + cost;
    theta = cv;
    int q_sel = -1;
    for (int q = 0; q < n_alphas; ++q) if (q_sel < 0 && ((ymask >> q) & 1u)) q_sel = q;
    if (alive && q_sel >= 0) {
      adu = P->alphas[q_sel];
      // the dual rows of the accepted dual step (:1627-1637)
      for (int t = 0; t < N; ++t) {
        double xo[NX], xt[NX], dx[NX], yo[MM], ky[MM], Ky[MM * NX], yn[MM];
        ld<NX>(Xc + GI(t, NX, 0), kLS, xo); ld<NX>(Xn + GI(t, NX, 0), kLS, xt);
        ld<M>(d.Y + (size_t)cur * d.planeM + GI(t, M, 0), kLS, yo);
        ld<M>(d.ky + GI(t, M, 0), kLS, ky);
        ld<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky);
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = xt[i] - xo[i];
#pragma unroll
        for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) s += Ky[r * NX + j] * dx[j];
          yn[r] = (yo[r] + adu * ky[r]) + s; }
        st<M>(d.Y + (size_t)slot * d.planeM + GI(t, M, 0), kLS, yn);
      }
      success = ms_filter_acceptable(d, b, o, merit, theta, alpha * d.dV0[b]);
    }
  }
  d.t_steps[ti] = steps;
  d.t_success[ti] = success ? 1 : 0;
  d.t_cost[ti] = cost; d.t_merit[ti] = merit; d.t_theta[ti] = theta; d.t_inf_pr[ti] = theta; d.t_inf_comp[ti] = 0.0;
  d.t_apr[ti] = alpha; d.t_adu[ti] = adu;
  if constexpr (M > 0) { d.t_ysmin[ti] = r_ipr; d.t_ysmax[ti] = r_idef; }
}

// ================================================================================ K4 (two-role) + K4b (dual rows)
// Round 5: the multiple-shooting rollout as a PRODUCER / CONSUMER pair of wavefronts per (64-trajectory tile, alpha), the form the IPDDP
// rollout has (kernels_lean.hpp::k_forward_ipddp_pc).  Only x_{t+1} = F(x_t, u_t(x_t)) (or its gap-closing rule at a segment boundary) is a
// serial chain; one wave per SIMD leaves every dependent f64 operation's latency exposed, so everything else moves to a second wave:
//   wave 0 (producer)  u_t = u + a k + K dx, F_t = f(x_t, u_t), x_{t+1} (inside a segment: F_t; at a boundary: :1483-1509), the defect
//                      1-norm |F_t - x_{t+1}|_1, l_f(x_N); stores X, U, F of the trial                       (msipddp_solver.cpp:1462-1509)
//   wave 1 (consumer)  slack trial + fraction-to-boundary test, dual feasibility of every ladder entry, costate trial, running cost,
//                      g(x_t, u_t), barrier / violation sums, the filter test and the trial record            (:1547-1560, 1612-1724)
// Channel: an LDS ring of kRing steps carrying (x_t, dx_t, u_t, |defect_{t-1}|_1) per lane and two LDS counters polled with s_sleep.
// Every accumulator sees its own terms in the order of the one-wave kernel (the defect norm of step t - 1 is added in front of the
// violation terms of step t, i.e. behind those of step t - 1), so the pair is bitwise the one-wave kernel
// (tests/test_msipddp_device.py::test_two_role_rollout_agrees_bitwise).  The nominal F_t and x_{t+1} rows a boundary step needs ride in the
// producer's prefetched record (the one-wave kernel fetched them at the boundary: a memory round trip on the chain every segment).
// The ROWS of a trial -- slack, dual (:1627-1637), costate (:1639-1641), constraint values, parked log sums -- are no longer written by the
// rollout: nobody reads them unless the trial is the one the selection rule accepts, so k_rows_msipddp forms them at (batch x N) width
// for that trial only, from the trial's stored X / U rows with the rollout's own expressions (same operands, same order: same bits).  The
// consumer keeps what DECIDES: the slack trial (fraction-to-boundary test, barrier and violation sums), the dual step search, cost and
// merit.  Why: the consumer, not the dynamics chain, bounded the pair -- 24 row loads + 7 row stores per step against the ~16 outstanding
// row operations a wavefront can hold (profiles/r05_ms_rollout_roles.md: consumer emptied 153 -> 70 us per launch, producer emptied 152);
// the one-wave kernel even walked the horizon a second time per trial for the dual rows.
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_rows_msipddp(DevBuf d, const ProblemDev *__restrict__ Pk, int a0, int na, int phase_req, int force, int first_only) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = M > 0 ? M : 1, NSEG = Cons::NSEG, NS = NSEG > 0 ? NSEG : 1;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  const ProblemDev *__restrict__ P = Pk;
  const int cur = d.cur[b];
  int a_lo = a0, a_hi = a0 + na;
  if (!force && first_only != 0) {
    const int only = first_only == 2 ? d.cand[b] : first_surviving_trial(d, a0, na, b);
    if (only < 0) return;
    a_lo = only; a_hi = only + 1;
  }
  double xo[NX], lo[NX], kl[NX], Kl[NX * NX];
  ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, xo);
  ld<NX>(d.Lam + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, lo);
  ld<NX>(d.kl + GI(t, NX, 0), kLS, kl);
  ld<NX * NX>(d.Vxx + GI(t + 1, NX * NX, 0), kLS, Kl);
  [[maybe_unused]] double so[MM], ksv[MM], Ks[MM * NX], yo[MM], ky[MM], Ky[MM * NX];
  [[maybe_unused]] typename Cons::Ctx cc;
  if constexpr (M > 0) {
    Cons::load(P, cc);
    ld<M>(d.S + (size_t)cur * d.planeM + GI(t, M, 0), kLS, so);
    ld<M>(d.ks + GI(t, M, 0), kLS, ksv);
    ld<M * NX>(d.Ks + GI(t, M * NX, 0), kLS, Ks);
    ld<M>(d.Y + (size_t)cur * d.planeM + GI(t, M, 0), kLS, yo);
    ld<M>(d.ky + GI(t, M, 0), kLS, ky);
    ld<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky);
  }
  for (int a = a_lo; a < a_hi; ++a) {
    const size_t ti = (size_t)a * d.Bp + b;
    if (d.t_success[ti] != 1) continue;
    const int slot = trial_slot(cur, a);
    const double alpha = d.t_apr[ti];
    double xt[NX], dx[NX];
    ld<NX>(d.X + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, xt);
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[i] = xt[i] - xo[i];
    {   // costate trial (:1466-1467 == :1639-1641): lambda + a k_lambda + K_lambda dx, K_lambda = V_xx(t+1)
      double ln[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Kl[i * NX + j] * dx[j];
        ln[i] = (lo[i] + alpha * kl[i]) + s; }
      st<NX>(d.Lam + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, ln);
    }
    if constexpr (M > 0) {
      const double adu = d.t_adu[ti];
      double ut[NU], sn[MM], yn[MM], g[MM];
      ld<NU>(d.U + (size_t)slot * d.planeU + GI(t, NU, 0), kLS, ut);
#pragma unroll
      for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Ks[r * NX + j] * dx[j];
        sn[r] = (so[r] + alpha * ksv[r]) + s; }
#pragma unroll
      for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Ky[r * NX + j] * dx[j];
        yn[r] = (yo[r] + adu * ky[r]) + s; }
      Cons::template eval<NX, NU>(cc, xt, ut, g);
      st<M>(d.S + (size_t)slot * d.planeM + GI(t, M, 0), kLS, sn);
      st<M>(d.Y + (size_t)slot * d.planeM + GI(t, M, 0), kLS, yn);
      st<M>(d.G + (size_t)slot * d.planeM + GI(t, M, 0), kLS, g);
      double *evn = d.ev + (size_t)slot * ms_ev_plane<Cons>(d);
#pragma unroll
      for (int cs = 0; cs < NSEG; ++cs) {   // the parked log sums K5's filter reset replays (ms_replay_filter): the rollout's own terms
        const int off = Cons::seg_off(cs), dim = Cons::seg_dim(cs);
        double lsum = 0.0;
#pragma unroll
        for (int i = 0; i < MM; ++i) if (i < dim) lsum += solver_log(sn[off + i]);
        evn[GI(t, NS, cs)] = lsum;
      }
    }
  }
}

template <class Model, class Cons>
__global__ __launch_bounds__(128) void k_forward_msipddp_pc(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int a0, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, NSEG = Cons::NSEG, MM = M > 0 ? M : 1;
  typedef Objective<NX, NU> Obj;
  constexpr int RW = 2 * NX + NU + 1;      // doubles per lane per step: x_t, dx_t, u_t, |F_{t-1} - x_t|_1
  constexpr int kRing = RW <= 12 ? 8 : (RW <= 24 ? 4 : 2);
  __shared__ double s_ring[kRing * RW * 64];
  __shared__ int s_prod, s_cons;
  __shared__ double s_pcost[64], s_pn1[64], s_pidef[64];   // the producer lane's l_f(x_N), its last defect 1-norm, its largest defect entry
  __shared__ double s_al[CDDP_HIP_MAX_ALPHAS];  // the ladder, for the per-lane look-ups of the dual step search's slow path
  __shared__ int s_pq[64];                      // the producer lane's dual step: index of the first ladder entry every row of every step accepts, -1 = none
  const int lane = threadIdx.x & 63;
  const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64;
  const int b = blockIdx.x * 64 + lane;
  const int a = a0 + blockIdx.y;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const bool active = (b < d.B) && (force || d.phase[b] == phase_req);
  if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;   // same mask in both waves: both leave
  if (producer && lane == 0) { s_prod = 0; s_cons = 0; }
  if (!producer && lane < CDDP_HIP_MAX_ALPHAS) s_al[lane] = (lane < d.n_alphas) ? P->alphas[lane] : 0.0;
  __syncthreads();
  // inactive lanes run along on their own (scratch) trial rows: unconditional stores, no exec-mask regions (see k_forward_ipddp_pc)
  const int bb = (b < d.B) ? b : 0;
  const int cur = (b < d.B) ? d.cur[b] : 0;
  const int slot = trial_slot(cur, a);
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double alpha = P->alphas[a];
  const int kAbort = 2 * N + kRing;        // s_cons value with which the consumer releases (and stops) the producer
  auto wait_ge = [&](int *ctr, int need) -> int {
    int v;
    while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    return v;
  };

  if (producer) {
    const double *Uc = d.U + (size_t)cur * d.planeU, *Fc = d.F + (size_t)cur * d.planeX;
    double *Xn = d.X + (size_t)slot * d.planeX, *Un = d.U + (size_t)slot * d.planeU, *Fn = d.F + (size_t)slot * d.planeX;
    const int seg = o.msipddp_segment_length, rtype = o.msipddp_rollout_type;
    DynCtx dc;
    dc.load(P->integrator, P->dt, P->mp);
    double x[NX];
    ld<NX>(Xc + GI(0, NX, 0), kLS, x);     // r.X[0] = initial state (:1443)
    st<NX>(Xn + GI(0, NX, 0), kLS, x);
    // Dual step search (:1612-1644), here because the PRODUCER has the slack (profiles/r05_ms_rollout_roles.md): on a strictly decreasing
    // ladder (what cddp_hip_build_alphas makes; any other ladder runs the one-wave kernel, launch.hpp) the dual trial of one row of one
    // step, yn(a) = (y + a k_y) + K_y dx, is a monotone function of a in IEEE arithmetic (a rounded product and two rounded sums with
    // fixed other operands), so the ladder entries it rejects (yn < bound) are a PREFIX or a SUFFIX of the ladder (NaN / Inf cases
    // included: a NaN trial is never "below", and an infinite K_y dx makes the trial one value wherever it is not NaN), and the entries
    // every row of every step so far accepts are an interval [qL, qH).  Only its two ends are probed per row (2 trials instead of
    // n_alphas); a lane whose end is rejected walks it inwards (LDS table).  First accepted entry = qL: the lowest set bit of the
    // one-wave kernel's mask.
    const int n_alphas = d.n_alphas;
    [[maybe_unused]] const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - d.mu[bb]);
    [[maybe_unused]] int qL = 0, qH = n_alphas;
    [[maybe_unused]] double aL = P->alphas[0], aH = P->alphas[n_alphas > 0 ? n_alphas - 1 : 0];
    struct PRec { double xo[NX], uo[NU], kk[NU], KK[NU * NX], fo[NX], yo[MM], ky[MM], Ky[MM * NX]; };
    auto fetch = [&](int tt, PRec &r) {
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
      ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
      ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
      ld<NX>(Fc + GI(tt, NX, 0), kLS, r.fo);
      if constexpr (M > 0) {
        ld<M>(d.Y + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.yo);
        ld<M>(d.ky + GI(tt, M, 0), kLS, r.ky);
        ld<M * NX>(d.Ky + GI(tt, M * NX, 0), kLS, r.Ky);
      }
    };
    constexpr bool kPing = sizeof(PRec) <= 40 * sizeof(double);
    double n1_prev = 0.0, r_idef = 0.0;
    bool stop = false;
    auto step = [&](const int t, PRec &c, PRec &n) {
      // the record of step t + 1 (clamped): its x_old row is also the x1 of this step's gap-closing rule
      if constexpr (kPing) { fetch(t + 1 < N ? t + 1 : N - 1, n); PIPELINE_FENCE(); }
      double dx[NX], u[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = x[i] - c.xo[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += c.KK[i * NX + j] * dx[j];
        u[i] = (c.uo[i] + alpha * c.kk[i]) + s; }
      if (t >= kRing && (t % (kRing / 2)) == 0) { if (wait_ge(&s_cons, t - kRing / 2) >= kAbort) stop = true; }
      {
        double *rs = s_ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
        for (int i = 0; i < NX; ++i) { rs[i * 64] = x[i]; rs[(NX + i) * 64] = dx[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) rs[(2 * NX + i) * 64] = u[i];
        rs[(2 * NX + NU) * 64] = n1_prev;
        RING_FENCE();
        __hip_atomic_store(&s_prod, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if constexpr (M > 0) {   // dual feasibility of the two ends of the accepted interval (off the chain: nothing below reads it)
#pragma unroll
        for (int r = 0; r < M; ++r) { double sy = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) sy += c.Ky[r * NX + j] * dx[j];
          const double bound = (1.0 - tau) * c.yo[r];
          const double ynL = (c.yo[r] + aL * c.ky[r]) + sy, ynH = (c.yo[r] + aH * c.ky[r]) + sy;
          const bool hit = (qL < qH) && ((ynL < bound) || (ynH < bound));
          if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
            if (hit) {
              while (qL < qH && ((c.yo[r] + s_al[qL] * c.ky[r]) + sy) < bound) ++qL;
              while (qH > qL && ((c.yo[r] + s_al[qH - 1] * c.ky[r]) + sy) < bound) --qH;
              aL = s_al[qL < n_alphas ? qL : n_alphas - 1]; aH = s_al[qH > 0 ? qH - 1 : 0];
            }
          } }
      }
      double kk_keep[NU], KK_keep[NU * NX], fo[NX];   // what the "hybrid" rule reads of the record (dead otherwise)
#pragma unroll
      for (int i = 0; i < NU; ++i) kk_keep[i] = c.kk[i];
#pragma unroll
      for (int i = 0; i < NU * NX; ++i) KK_keep[i] = c.KK[i];
#pragma unroll
      for (int i = 0; i < NX; ++i) fo[i] = c.fo[i];
      if constexpr (!kPing) { fetch(t + 1 < N ? t + 1 : t, c); PIPELINE_FENCE(); }   // next record into the (dead) set, behind the integrator
      double fn[NX], xn[NX];
      Stepper<Model>::step(dc, x, u, fn);
      const bool boundary = (seg > 1) && ((t + 1) % seg == 0) && (t + 1 < N);
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = fn[i];
      if (boundary && rtype != 1) {
        const double *x1 = kPing ? n.xo : c.xo;      // X_c[t + 1]
        if (rtype == 0) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xn[i] = (x1[i] + (fn[i] - fo[i])) + alpha * (fo[i] - x1[i]);
        } else {   // "hybrid": linearised closed-loop step
          double A[NX * NX], Bm[NX * NU];
          ld<NX * NX>(d.A + GI(t, NX * NX, 0), kLS, A);
          ld<NX * NU>(d.Bm + GI(t, NX * NU, 0), kLS, Bm);
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            double lin = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) { double bk = 0.0;
#pragma unroll
              for (int q = 0; q < NU; ++q) bk += Bm[i * NU + q] * KK_keep[q * NX + j];
              lin += (A[i * NX + j] + bk) * dx[j]; }
            double bkk = 0.0;
#pragma unroll
            for (int q = 0; q < NU; ++q) bkk += Bm[i * NU + q] * kk_keep[q];
            xn[i] = (x1[i] + lin) + alpha * ((bkk + fo[i]) - x1[i]);
          }
        }
      }
      { double n1 = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) { const double r = fn[i] - xn[i]; n1 += fabs(r); r_idef = dmax(r_idef, fabs(r)); }
        n1_prev = n1; }
      st<NU>(Un + GI(t, NU, 0), kLS, u);
      st<NX>(Fn + GI(t, NX, 0), kLS, fn);
      st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = xn[i];
    };
    auto prime = [&]() {   // the VMEM queue primed with one step's store pattern (see k_forward_ipddp_pc)
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      st<NU>(Un + GI(0, NU, 0), kLS, z);
      st<NX>(Fn + GI(0, NX, 0), kLS, z);
      st<NX>(Xn + GI(1, NX, 0), kLS, z);
    };
    PRec ra, rb;
    fetch(0, ra);
    prime();
    int t = 0;
    if constexpr (kPing) {
      for (; t + 1 < N && !stop; t += 2) { step(t, ra, rb); step(t + 1, rb, ra); }
      if (t < N && !stop) step(t, ra, rb);
    } else {
      for (; t < N && !stop; ++t) step(t, ra, ra);
    }
    s_pcost[lane] = Obj::terminal_cost(P, x);
    s_pn1[lane] = n1_prev; s_pidef[lane] = r_idef;
    if constexpr (M > 0) s_pq[lane] = (qL < qH) ? qL : -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(&s_prod, N + kRing + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }

  // -------------------------------------------------------------------- consumer
  const double mu = d.mu[bb];
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  const int n_alphas = d.n_alphas;
  const size_t ti = (size_t)a * d.Bp + bb;
  if (active) atomicAdd(d.launched, 1ull);
  typename Obj::Ctx oc;
  Obj::load(P, oc);
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  double cost = 0.0, merit_b = 0.0, cv = 0.0;
  [[maybe_unused]] double r_ipr = 0.0;   // max |g + s| of the trial (K5's filter reset)
  int seen_prod = 0;     // last value of the producer's counter this wave saw (wave-uniform)
  bool alive = true;
  int steps = N;
  struct CRec { double so[MM], ksv[MM], Ks[MM * NX]; };
  auto fetch = [&](int tt, CRec &r) {
    if constexpr (M > 0) {
      ld<M>(d.S + (size_t)cur * d.planeM + GI(tt, M, 0), kLS, r.so);
      ld<M>(d.ks + GI(tt, M, 0), kLS, r.ksv);
      ld<M * NX>(d.Ks + GI(tt, M * NX, 0), kLS, r.Ks);
    }
  };
  constexpr bool kPingC = sizeof(CRec) <= 40 * sizeof(double);
  auto step = [&](const int t, CRec &c, CRec &n) {
    if constexpr (kPingC) { fetch(t + 1 < N ? t + 1 : N - 1, n); PIPELINE_FENCE(); }
    if (!CDDP_RING_LAZY_POLL || seen_prod < t + 1) seen_prod = __builtin_amdgcn_readfirstlane(wait_ge(&s_prod, t + 1));
    double x[NX], dx[NX], u[NU], n1_prev;
    {
      const double *rs = s_ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
      for (int i = 0; i < NX; ++i) { x[i] = rs[i * 64]; dx[i] = rs[(NX + i) * 64]; }
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = rs[(2 * NX + i) * 64];
      n1_prev = rs[(2 * NX + NU) * 64];
      RING_FENCE();
      __hip_atomic_store(&s_cons, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    [[maybe_unused]] double sn[MM];
    if constexpr (M > 0) {
      cv += n1_prev;   // |F_{t-1} - x_t|_1: behind the violation terms of step t - 1, where the one-wave kernel adds it
#pragma unroll
      for (int r = 0; r < M; ++r) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += c.Ks[r * NX + j] * dx[j];
        sn[r] = (c.so[r] + alpha * c.ksv[r]) + s;
        const bool viol = alive && (sn[r] < (1.0 - tau) * c.so[r]);
        steps = viol ? t : steps; alive = alive && !viol; }
    }
    if constexpr (!kPingC) { fetch(t + 1 < N ? t + 1 : t, c); PIPELINE_FENCE(); }   // next record into the (dead) set, behind the cost / barrier terms
    cost += Obj::running_cost(oc, xrt, t, x, u);
    if constexpr (M > 0) {
      double g[MM];
      Cons::template eval<NX, NU>(cc, x, u, g);
#pragma unroll
      for (int cs = 0; cs < NSEG; ++cs) {
        const int off = Cons::seg_off(cs), dim = Cons::seg_dim(cs);
        double lsum = 0.0, l1 = 0.0;
#pragma unroll
        for (int i = 0; i < MM; ++i) if (i < dim) { lsum += solver_log(sn[off + i]); const double pr = g[off + i] + sn[off + i]; l1 += fabs(pr); r_ipr = dmax(r_ipr, fabs(pr)); }
        merit_b -= mu * lsum; cv += l1;
      }
    }
  };
  {
    CRec ra, rb;
    fetch(0, ra);
    int t = 0;
    bool all_dead = false;
    if constexpr (kPingC) {
      for (; t + 1 < N; t += 2) {
        step(t, ra, rb); step(t + 1, rb, ra);
        if (__builtin_amdgcn_ballot_w64(alive && active) == 0ull) { all_dead = true; break; }
      }
      if (!all_dead && t < N) step(t, ra, rb);
    } else {
      for (; t < N; ++t) {
        step(t, ra, ra);
        if (__builtin_amdgcn_ballot_w64(alive && active) == 0ull) { all_dead = true; break; }
      }
    }
    if (all_dead) {   // every trial of the tile has been abandoned (M > 0 only: `alive` never drops otherwise): stop the producer
      __hip_atomic_store(&s_cons, kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (active) {
        d.t_steps[ti] = steps; d.t_success[ti] = 0;
        d.t_cost[ti] = cost; d.t_merit[ti] = merit_b + cost; d.t_theta[ti] = cv; d.t_inf_pr[ti] = cv; d.t_inf_comp[ti] = 0.0;
        d.t_apr[ti] = alpha; d.t_adu[ti] = 1.0;
      }
      return;
    }
  }
  wait_ge(&s_prod, N + kRing + 1);
  if (!active) return;
  // (the defect norm of step N - 1 is |F - F|_1: zero, or NaN exactly when F_{N-1} is not finite -- added like every other one)
  if constexpr (M > 0) cv += s_pn1[lane];
  cost += s_pcost[lane];
  bool success = false;
  double merit = cost, theta = 0.0, adu = 1.0;
  if constexpr (M == 0) {   // :1516-1530: expected-reduction ratio test
    const double dJ = d.cost[b] - cost;
    const double expected = -alpha * (d.dV0[b] + 0.5 * alpha * d.dV1[b]);
    const double ratio = expected > 0.0 ? dJ / expected : sign_of_reduction(dJ);
    success = ratio > 1e-6;
  } else {
    merit = merit_b + cost;
    theta = cv;
    const int q_sel = s_pq[lane];   // the producer's dual step search
    if (alive && q_sel >= 0) {
      adu = P->alphas[q_sel];
      success = ms_filter_acceptable(d, b, o, merit, theta, alpha * d.dV0[b]);   // the rows of the accepted trial: k_rows_msipddp
    }
  }
  d.t_steps[ti] = steps;
  d.t_success[ti] = success ? 1 : 0;
  d.t_cost[ti] = cost; d.t_merit[ti] = merit; d.t_theta[ti] = theta; d.t_inf_pr[ti] = theta; d.t_inf_comp[ti] = 0.0;
  d.t_apr[ti] = alpha; d.t_adu[ti] = adu;
  if constexpr (M > 0) { d.t_ysmin[ti] = r_ipr; d.t_ysmax[ti] = s_pidef[lane]; }
}

// ================================================================================ K5
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_update_msipddp(DevBuf d, const ProblemDev *__restrict__ Pk, int stage, int n1, int is_last_iter, int do_count) {
  constexpr int M = Cons::M;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int ph = d.phase[b];
  const int n_alphas = d.n_alphas;
  int ladder_bin = -1;
  if ((stage == 1 && ph == PH_FWD1) || (stage == 2 && ph == PH_FWD2)) {
    const int lo = (stage == 1) ? 0 : 1;
    const int hi = (stage == 1) ? n1 : n_alphas;
    int win = -1;
    if (P->ls_rule == CDDP_HIP_LS_FIRST_SUCCESS) {
      for (int a = lo; a < hi && win < 0; ++a) if (d.t_success[(size_t)a * d.Bp + b] == 1) win = a;
    } else {   // success && merit < best.merit, strict (cddp_solver_base.cpp:280-286)
      double best = INFINITY;
      for (int a = lo; a < hi; ++a) {
        const size_t ti = (size_t)a * d.Bp + b;
        if (d.t_success[ti] != 0 && d.t_merit[ti] < best) { best = d.t_merit[ti]; win = a; }
      }
    }
    if (win < 0 && hi < n_alphas) { d.phase[b] = PH_FWD2; goto count; }
    ladder_bin = (win >= 0) ? win : n_alphas;
    {
      double mu = d.mu[b];
      bool running = true;
      const bool fp_success = win >= 0;
      [[maybe_unused]] double sdu = 0.0, ys_lo = INFINITY, ys_hi = -INFINITY;
      [[maybe_unused]] bool have_sdu = false;
      if (fp_success) {
        const size_t ti = (size_t)win * d.Bp + b;
        const int old_cur = d.cur[b];
        const double w_cost = d.t_cost[ti], w_merit = d.t_merit[ti], w_apr = d.t_apr[ti], w_adu = d.t_adu[ti], w_theta = d.t_theta[ti];
        const double dJ = d.cost[b] - w_cost;
        const int na_walked = (P->ls_rule == CDDP_HIP_LS_FIRST_SUCCESS) ? win + 1 : n_alphas;
        int ns = d.n_fwd_steps[b];
        for (int a = 0; a < na_walked; ++a) ns += d.t_steps[(size_t)a * d.Bp + b];
        d.n_fwd[b] = d.n_fwd[b] + na_walked;
        d.n_fwd_steps[b] = ns;
        const int slot_now = trial_slot(old_cur, win);
        d.cur[b] = slot_now;
        // applyForwardPassResult (:287-304)
        d.cost[b] = w_cost; d.merit[b] = w_merit; d.phi[b] = w_merit; d.alpha_pr[b] = w_apr; d.alpha_du[b] = w_adu;
        ms_filter_add(d, b, w_merit, w_theta);
        hist_push(d, b, mu);
        d.reg[b] = reg_decrease(o, d.reg[b]);
        // checkConvergence (:306-364): the residuals are those of the last backward pass / filter reset, the duals the new iterate's
        const double ipr = d.inf_pr[b], icomp = d.inf_comp[b];
        if constexpr (M > 0) {   // the mu-independent residual pieces of the new iterate (ms_replay_filter)
          d.ms_res[b] = d.t_ysmin[ti]; d.ms_res[(size_t)d.Bp + b] = d.t_ysmax[ti]; d.ms_res[(size_t)2 * d.Bp + b] = w_theta;
        }
        sdu = ms_scaled_inf_du<Model, Cons>(d, b, slot_now, d.inf_du[b], ys_lo, ys_hi); have_sdu = true;
        const double metric = dmax(dmax(sdu, ipr), icomp);
        int st = CDDP_HIP_STATUS_RUNNING;
        const int iter = d.iter[b];
        if (metric <= o.tolerance) st = CDDP_HIP_STATUS_OPTIMAL;
        else if (fabs(dJ) < o.acceptable_tolerance && iter > 10 && ipr < sqrt(o.acceptable_tolerance) && icomp < sqrt(o.acceptable_tolerance)) st = CDDP_HIP_STATUS_ACCEPTABLE;
        else if (iter >= 1 && d.step_norm[b] < o.tolerance * 10.0 && ipr < 1e-4) st = CDDP_HIP_STATUS_ACCEPTABLE;
        if (st != CDDP_HIP_STATUS_RUNNING) { d.status[b] = st; d.phase[b] = PH_DONE; running = false; }
      } else {
        // handleForwardPassFailure (:371-398) with checkAndPerformFilterRestoration (:810-836)
        const int nf0 = d.n_fwd[b]; int ns = d.n_fwd_steps[b];
        for (int a = 0; a < n_alphas; ++a) ns += d.t_steps[(size_t)a * d.Bp + b];
        d.n_fwd[b] = nf0 + n_alphas;
        d.n_fwd_steps[b] = ns;
        const int fn = d.filt_n[b];
        bool needs = fn > 5;
        if (!needs)
          for (int i = 0; i < fn; ++i)
            if (!dfinite(d.filt[(size_t)i * d.Bp + b]) || !dfinite(d.filt[(size_t)(d.filt_cap + i) * d.Bp + b])) { needs = true; break; }
        if (needs && fn > 0) ms_filter_prune(d, b);
        else {
          const double reg = reg_increase(o, d.reg[b]);
          d.reg[b] = reg;
          if (reg >= o.reg_max_value) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; running = false; }
        }
      }
      if (running) {
        // postIterationUpdate -> updateBarrierParameters (:1751-1850)
        if constexpr (M > 0) {
          const int slot_now = d.cur[b];
          bool reset = false;
          if (o.barrier_strategy == CDDP_HIP_BARRIER_MONOTONIC) {
            mu = dmax(o.barrier_mu_min_value, o.barrier_mu_update_factor * mu);
            reset = true;
            if (!have_sdu) { sdu = ms_scaled_inf_du<Model, Cons>(d, b, slot_now, d.inf_du[b], ys_lo, ys_hi); have_sdu = true; }   // (for the extreme products y s)
          } else {
            if (!have_sdu) { sdu = ms_scaled_inf_du<Model, Cons>(d, b, slot_now, d.inf_du[b], ys_lo, ys_hi); have_sdu = true; }   // (after a failed pass: duals and inf_du unchanged)
            const double metric = dmax(dmax(sdu, d.inf_pr[b]), d.inf_comp[b]);
            if (o.barrier_strategy == CDDP_HIP_BARRIER_IPOPT) {
              if (metric <= 10.0 * mu) {
                const double lin = o.barrier_mu_update_factor * mu, sup = solver_pow(mu, o.barrier_mu_update_power);
                mu = dmax(o.tolerance / 10.0, dmin(lin, sup));
                reset = true;
              }
            } else {
              const double threshold = (mu < 1e-5) ? dmax(metric * 10.0, mu * 100.0) : dmax(o.barrier_mu_update_factor * mu, mu * 2.0);
              const bool slow = fp_success && d.alpha_pr[b] > 0 && (metric < 1e-3);
              if (metric <= threshold || slow) {
                double factor = o.barrier_mu_update_factor;
                if (mu > 1e-12) {
                  const double ratio = metric / mu;
                  if (ratio < 0.01) factor = o.barrier_mu_update_factor * 0.1;
                  else if (ratio < 0.1) factor = o.barrier_mu_update_factor * 0.3;
                  else if (ratio < 0.5) factor = o.barrier_mu_update_factor * 0.6;
                }
                const double lin = factor * mu, sup = solver_pow(mu, o.barrier_mu_update_power);
                if (slow && mu > o.tolerance) mu = dmin(lin, sup);
                else mu = dmax(o.tolerance / 100.0, dmin(lin, sup));
                reset = true;
              }
            }
          }
          if (reset) { d.mu[b] = mu; ms_replay_filter<Model, Cons>(d, b, slot_now, mu, d.cost[b], ys_lo, ys_hi); }
        }
        d.phase[b] = PH_ACTIVE;
      }
    }
  }
count:
  if (d.win_hist) {
    for (int a = 0; a <= n_alphas; ++a) {
      const unsigned long long m = __ballot(ladder_bin == a);
      if (m != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(d.win_hist + a, (int)__popcll(m));
    }
  }
  if (do_count) {
    if (is_last_iter && d.phase[b] != PH_DONE) { d.status[b] = CDDP_HIP_STATUS_MAX_ITERATIONS; d.phase[b] = PH_DONE; }
    if (d.phase[b] != PH_DONE) atomicAdd(d.n_active, 1);
  }
}

#undef GI
}  // namespace cddp_dev
