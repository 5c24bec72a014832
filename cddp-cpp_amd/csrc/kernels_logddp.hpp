// LogDDP on the device: the reference's single-shooting relaxed-log-barrier DDP (logddp_solver.cpp:43-707,
// RelaxedLogBarrier barrier.hpp:37-296) for the built-in plants, as K0 / K2 / K4 / K5 variants of the batched core -- the same handle,
// stacks, trial slots, phase machine, ladder shapes and host loop as CLDDP / IPDDP (capi.hip::SolveRun), K1 (k_derivs) unchanged.
// Rounds 2-3 served LogDDP through the plug-in boundary only (cddp_hip_plugin_solve: GPU sweeps on caller-built stacks, host
// rollouts, one trajectory at a time); here the whole solve is resident:
//
//   K0  k_init_logddp      grid (batch)        rollOutNominalTrajectory, cost, mu / delta, evaluateTrajectory + resetFilter
//                                              (logddp_solver.cpp:30-38, 45-205, 317-361)
//   K2  k_backward_logddp  grid (batch)        Riccati sweep with the barrier's gradients / Hessians folded into the Q blocks, LDLT of
//                                              the regularised, symmetrised Q_uu, retry loop  (logddp_solver.cpp:363-590, barrier.hpp:95-213)
//   K4  k_forward_logddp   grid (batch x n_a)  single-shooting rollout, cost, barrier merit, violation, filter test (logddp_solver.cpp:594-707)
//   K5  k_update_logddp    grid (batch)        selection rule, applyForwardPassResult, checkConvergence, regularisation schedule,
//                                              postIterationUpdate (mu) + resetFilter  (cddp_solver_base.cpp:29-186, logddp_solver.cpp:216-277)
//
// One trajectory per lane (the layout of the fused CLDDP kernels); every sum in the reference's order, FMA contraction off, the
// logarithm the shared straight-line routine (dev_trig.hpp::solver_log) -- the CPU checker's LogDDP in its trig_mode 1 runs the same
// operations, so tests/test_logddp_device.py compares iteration traces exactly.
//
// resetFilter (logddp_solver.cpp:333-361) is not a second pass over the trajectory here: with b_tc = sum_i beta(U_i - g_i) of
// constraint c at step t, the merit is cost + sum_t sum_c mu * b_tc and the violation sum_t sum_c sum_i max(g_i, 0).  Neither b_tc
// nor the violation depends on mu; the rollout that produced an iterate evaluated both on the very same (x_t, u_t) in the very
// same order.  K0 / K4 park b_tc per trial slot (d.ev, [slot][N][NSEG]) and K5 replays the chain  merit += mu_new * b_tc  -- the
// idiom of the IPDDP update (kernels.hpp::k_update).
#pragma once
#include "kernels.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

// beta_delta(z), beta', beta'' (barrier.hpp:274-296).  -log(1e-12) is a compile-time constant of the reference's host libm.
DEV void lg_beta(double z, double delta, double &b0, double &b1, double &b2) {
  if (z > delta) {
    if (z <= 1e-12) { b0 = 0x1.ba18a998fffa0p+4; b1 = -1.0 / 1e-12; b2 = 1.0 / (1e-12 * 1e-12); }
    else { b0 = -solver_log(z); b1 = -1.0 / z; b2 = 1.0 / (z * z); }
  } else {
    const double tdd = (z - 2.0 * delta) / delta;
    b0 = 0.5 * (tdd * tdd - 1.0) - solver_log(delta);
    b1 = tdd / delta;
    b2 = 1.0 / (delta * delta);
  }
}
DEV double lg_beta0(double z, double delta) {   // the value alone (RelaxedLogBarrier::evaluate, :61-91)
  if (z > delta) return (z <= 1e-12) ? 0x1.ba18a998fffa0p+4 : -solver_log(z);
  const double tdd = (z - 2.0 * delta) / delta;
  return 0.5 * (tdd * tdd - 1.0) - solver_log(delta);
}

// Per-constraint pieces, in ConList order (= std::map order of the reference's constraint set).
template <int OFF, int CI, class... Cs> struct LgImpl;
template <int OFF, int CI> struct LgImpl<OFF, CI> {
  template <int NX, int NU> DEV static void values(const double *, double, double *, double &) {}
  template <int NX, int NU> DEV static void derivs(const ProblemDev *, const double *, const double *, const double *, const double *, double, double,
                                                   double *, double *, double *, double *, double *) {}
};
template <int OFF, int CI, class C, class... Rest> struct LgImpl<OFF, CI, C, Rest...> {
  typedef LgImpl<OFF + C::DUAL, CI + 1, Rest...> Next;
  // b[CI] = sum_i beta(-g_i) over the rows of this constraint; viol += the positive rows (row order)
  template <int NX, int NU>
  DEV static void values(const double *g, double delta, double *bsum, double &viol) {
    double total = 0.0;
#pragma unroll
    for (int i = 0; i < C::DUAL; ++i) total += lg_beta0(-g[OFF + i], delta);
    bsum[CI] = total;
#pragma unroll
    for (int i = 0; i < C::DUAL; ++i) if (g[OFF + i] > 0.0) viol += g[OFF + i];
    Next::template values<NX, NU>(g, delta, bsum, viol);
  }
  // getGradients + getHessians of this constraint (barrier.hpp:95-213), scaled by mu, added to the Q blocks (logddp_solver.cpp:518-530)
  template <int NX, int NU>
  DEV static void derivs(const ProblemDev *P, const double *g, const double *Gx, const double *Gu, const double *u, double mu, double delta,
                         double *Qx, double *Qu, double *Qxx, double *Quu, double *Qux) {
    double gx[NX], gu[NU], Hxx[NX * NX], Huu[NU * NU], Hux[NU * NX];
#pragma unroll
    for (int a = 0; a < NX; ++a) gx[a] = 0.0;
#pragma unroll
    for (int a = 0; a < NU; ++a) gu[a] = 0.0;
#pragma unroll
    for (int a = 0; a < NX * NX; ++a) Hxx[a] = 0.0;
#pragma unroll
    for (int a = 0; a < NU * NU; ++a) Huu[a] = 0.0;
#pragma unroll
    for (int a = 0; a < NU * NX; ++a) Hux[a] = 0.0;
    const ConDev &cd = P->cons[CI];
    // second derivatives of the rows (Constraint::getHessians): the cone throws (no curvature term at all), the ball and the thrust
    // rows have one; boxes and linear rows keep the base-class zeros (adding t2 * 0 to an entry that is never -0 changes nothing)
    [[maybe_unused]] double Hth[NU * NU];
    if constexpr (C::KIND == CDDP_HIP_CON_THRUST || C::KIND == CDDP_HIP_CON_MAX_THRUST) {   // constraint.hpp:899-920, 1021-1042
      double sq = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) sq += u[i] * u[i];
      const double term = sq + cd.scale, den = solver_pow(term, 1.5);
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < NU; ++j) Hth[i * NU + j] = (den > DBL_MIN) ? ((i == j ? term : 0.0) - u[i] * u[j]) / den : 0.0;
    }
#pragma unroll
    for (int i = 0; i < C::DUAL; ++i) {
      double b0, b1, b2;
      lg_beta(-g[OFF + i], delta, b0, b1, b2);
      const double dCost = 0.0 - b1;            // upper side only: every built-in constraint has lower bound -inf
      const double t1 = 0.0 + b2, t2 = 0.0 - b1;
      const double *gxr = Gx + (OFF + i) * NX, *gur = Gu + (OFF + i) * NU;
#pragma unroll
      for (int a = 0; a < NX; ++a) gx[a] += dCost * gxr[a];
#pragma unroll
      for (int a = 0; a < NU; ++a) gu[a] += dCost * gur[a];
#pragma unroll
      for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) Hxx[a * NX + c] += (t1 * gxr[a]) * gxr[c];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int c = 0; c < NU; ++c) Huu[a * NU + c] += (t1 * gur[a]) * gur[c];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) Hux[a * NX + c] += (t1 * gur[a]) * gxr[c];
      if constexpr (C::KIND == CDDP_HIP_CON_BALL) {            // constraint.hpp:387-396: -2 scale on the first dim diagonal entries
#pragma unroll
        for (int a = 0; a < C::DIM; ++a) Hxx[a * NX + a] = Hxx[a * NX + a] + t2 * (-2.0 * cd.scale);
      }
      if constexpr (C::KIND == CDDP_HIP_CON_THRUST) {          // rows (min - |u|, |u| - max): -H, H
#pragma unroll
        for (int a = 0; a < NU * NU; ++a) Huu[a] = Huu[a] + t2 * (i == 0 ? -1.0 * Hth[a] : Hth[a]);
      }
      if constexpr (C::KIND == CDDP_HIP_CON_MAX_THRUST) {
#pragma unroll
        for (int a = 0; a < NU * NU; ++a) Huu[a] = Huu[a] + t2 * Hth[a];
      }
    }
#pragma unroll
    for (int a = 0; a < NX; ++a) Qx[a] = Qx[a] + mu * gx[a];
#pragma unroll
    for (int a = 0; a < NU; ++a) Qu[a] = Qu[a] + mu * gu[a];
#pragma unroll
    for (int a = 0; a < NX * NX; ++a) Qxx[a] = Qxx[a] + mu * Hxx[a];
#pragma unroll
    for (int a = 0; a < NU * NU; ++a) Quu[a] = Quu[a] + mu * Huu[a];
#pragma unroll
    for (int a = 0; a < NU * NX; ++a) Qux[a] = Qux[a] + mu * Hux[a];
    Next::template derivs<NX, NU>(P, g, Gx, Gu, u, mu, delta, Qx, Qu, Qxx, Quu, Qux);
  }
};
template <class L> struct LgCons;
template <class... Cs> struct LgCons<ConList<Cs...>> : LgImpl<0, 0, Cs...> {};

// Second-order dynamics terms of LogDDP's full-DDP branch (logddp_solver.cpp:505-515):  Q_xx += (timestep * V_x(i)) * f_xx[i]  on the
// continuous-time Hessians -- NOT the association of the base-class helper the IPDDP sweeps follow (V_x(i) * (timestep * f_xx[i]),
// kernels.hpp::ddp_tensor_terms): one rounding apart per entry, so it has its own routine.
template <class Model>
DEV void lg_tensor_terms(const ProblemDev *P, const double *x, const double *u, const double *w, double *Qxx, double *Qux, double *Quu) {
  constexpr int NX = Model::NX, NU = Model::NU;
  static_assert(Model::kHasHess, "the resident LogDDP kernels are instantiated for plants with explicit Hessian tensors");
  double Fxx[NX * NX * NX], Fuu[NX * NU * NU], Fux[NX * NU * NX];
  Model::hess(P->mp, x, u, Fxx, Fuu, Fux);
  const double dt = P->dt;
  for (int i = 0; i < NX; ++i) {
    const double s = dt * w[i];
    for (int e = 0; e < NX * NX; ++e) Qxx[e] = Qxx[e] + s * Fxx[i * NX * NX + e];
    for (int e = 0; e < NU * NX; ++e) Qux[e] = Qux[e] + s * Fux[i * NU * NX + e];
    for (int e = 0; e < NU * NU; ++e) Quu[e] = Quu[e] + s * Fuu[i * NU * NU + e];
  }
}

// parked barrier sums: [slot][N][NSEG] wave-tiled rows
template <class Cons> DEV size_t lg_ev_plane(const DevBuf &d) { return (size_t)d.N * (Cons::NSEG > 0 ? Cons::NSEG : 1) * d.Bp; }

// ================================================================================ K0
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_init_logddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int mode) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, NSEG = Cons::NSEG, MM = M > 0 ? M : 1, NS = NSEG > 0 ? NSEG : 1;
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.n_active = d.B;
  if (b >= d.B) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  (void)mode;   // a warm start keeps nothing the first backward pass does not overwrite (logddp_solver.cpp:64-125)
  d.cur[b] = 0;
  d.iter[b] = 0; d.status[b] = CDDP_HIP_STATUS_RUNNING; d.phase[b] = PH_ACTIVE;
  d.n_bwd[b] = 0; d.n_fwd[b] = 0; d.n_fwd_steps[b] = 0; d.bwd_ok[b] = 0; d.filt_n[b] = 0;
  if (b < d.hist_batch) d.hist_n[b] = 0;
  d.reg[b] = o.reg_initial_value;
  d.dV0[b] = 0.0; d.dV1[b] = 0.0; d.step_norm[b] = 0.0; d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;
  const double mu = o.logddp_mu_initial, delta = o.logddp_relaxed_delta;
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  double *X0 = d.X, *U0 = d.U;
  double x[NX];
  ld<NX>(X0 + GI(0, NX, 0), kLS, x);
  double cost = 0.0, viol = 0.0;
  // rollOutNominalTrajectory (:30-38) fused with evaluateTrajectory (:317-331) and resetFilter (:333-361): separate accumulators, each
  // in its own reference order
  for (int t = 0; t < N; ++t) {
    double u[NU], xn[NX];
    ld<NU>(U0 + GI(t, NU, 0), kLS, u);
    cost += Obj::running_cost(P, xrt, t, x, u);
    if constexpr (M > 0) {
      double g[MM], bs[NS];
      Cons::template eval<NX, NU>(cc, x, u, g);
      LgCons<Cons>::template values<NX, NU>(g, delta, bs, viol);
#pragma unroll
      for (int c = 0; c < NSEG; ++c) d.ev[GI(t, NS, c)] = bs[c];
    }
    Stepper<Model>::step(P->integrator, P->dt, P->mp, x, u, xn);
    st<NX>(X0 + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  }
  cost += Obj::terminal_cost(P, x);
  // merit = cost, then += mu b_tc in (t, c) order: the chain starts from the finished cost (:335), so it is replayed here
  double merit = cost;
  if constexpr (M > 0) {
    for (int t = 0; t < N; ++t)
#pragma unroll
      for (int c = 0; c < NSEG; ++c) merit += mu * d.ev[GI(t, NS, c)];
  }
  d.cost[b] = cost; d.merit[b] = merit; d.phi[b] = merit;
  d.filter_theta[b] = viol; d.theta[b] = viol;      // constraint_violation_
  d.inf_pr[b] = viol; d.inf_du[b] = INFINITY; d.inf_comp[b] = INFINITY;   // cddp_core.cpp:297-301, resetFilter :360
  d.alpha_pr[b] = o.ls_initial_step_size; d.alpha_du[b] = 0.0; d.mu[b] = mu;
  hist_push(d, b, mu);
}

// ================================================================================ K2
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_logddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = M > 0 ? M : 1;
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
  if (count_iter) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b], delta = o.logddp_relaxed_delta;
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, Qu_err = 0;
  for (;;) {
    ++nb;
    double xN[NX], Vx[NX], Vxx[NX * NX];
    ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
    Obj::final_grad(P, xN, Vx);
    const double *Qf = P->pool + P->off_Qf;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * ((2.0 * Qf[i * NX + c]) + (2.0 * Qf[c * NX + i]));   // :478
    st<NX>(d.Vx + GI(N, NX, 0), kLS, Vx);
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Vxx);
    dV0 = 0; dV1 = 0; Qu_err = 0.0;
    bool fail = false;
    struct StepIn { double A[NX * NX], Bm[NX * NU], x[NX], u[NU]; };
    auto load_step = [&](int tt, StepIn &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.x);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.u);
    };
    StepIn nxt;
    load_step(N - 1, nxt);
    for (int t = N - 1; t >= 0; --t) {
      StepIn cs = nxt;
      if (t > 0) load_step(t - 1, nxt);
      PIPELINE_FENCE();
      double (&A)[NX * NX] = cs.A; double (&Bm)[NX * NU] = cs.Bm; double (&x)[NX] = cs.x; double (&u)[NU] = cs.u;
      double Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU];
      Obj::lx(P, xrt, t, x, Qx);
      Obj::lu(P, u, Qu);
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += A[k * NX + i] * Vx[k];
        Qx[i] = Qx[i] + s; }
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bm[k * NU + i] * Vx[k];
        Qu[i] = Qu[i] + s; }
      q_blocks<NX, NU>(P, A, Bm, Vx, Vxx, Qxx, Qux, Quu);
      if constexpr (Model::kHasHess) {   // (plants whose tensors exist only in the blocked dual form: full DDP refused at create)
        if (!o.use_ilqr) lg_tensor_terms<Model>(P, x, u, Vx, Qxx, Qux, Quu);    // :505-515
      }
      if constexpr (M > 0) {   // :518-530
        double g[MM], Gx[MM * NX], Gu[MM * NU];
#pragma unroll
        for (int i = 0; i < MM * NX; ++i) Gx[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MM * NU; ++i) Gu[i] = 0.0;
        Cons::template eval<NX, NU>(cc, x, u, g);
        Cons::template jac<NX, NU>(cc, x, u, Gx, Gu);
        LgCons<Cons>::template derivs<NX, NU>(P, g, Gx, Gu, u, mu, delta, Qx, Qu, Qxx, Quu, Qux);
      }
      // Q_uu_reg = sym(Q_uu + reg I), LDLT (:533-543)
      double Qr[NU * NU], Qs[NU * NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = Quu[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Qr[i * NU + c] + Qr[c * NU + i]);
      LDLTs<NU> f;
      f.compute(Qs, NU);
      if (!f.ok) { fail = true; break; }
      double kk[NU], KK[NU * NX];
      {
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
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      // dV, V_x, V_xx with the un-regularised Q_uu (:566-573), CLDDP's association
      double Quuk[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) s += Quu[i * NU + j] * kk[j];
        Quuk[i] = s; }
      { double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) { s0 += Qu[i] * kk[i]; s1 += kk[i] * Quuk[i]; }
        dV0 += s0; dV1 += 0.5 * s1; }
      double KtQ[NX * NU];
      mm_tn<NX, NU, NU>(KK, Quu, KtQ);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KtQ[i * NU + j] * kk[j]; bb += Qux[j * NX + i] * kk[j]; c += KK[j * NX + i] * Qu[j]; }
        Vx[i] = ((Qx[i] + a) + bb) + c;
      }
      double Vn[NX * NX];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { a += KtQ[i * NU + j] * KK[j * NX + c]; bb += Qux[j * NX + i] * KK[j * NX + c]; e += KK[j * NX + i] * Qux[j * NX + c]; }
          Vn[i * NX + c] = ((Qxx[i * NX + c] + a) + bb) + e;
        }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (Vn[i * NX + c] + Vn[c * NX + i]);
      st<NX>(d.Vx + GI(t, NX, 0), kLS, Vx);
      st<NX * NX>(d.Vxx + GI(t, NX * NX, 0), kLS, Vxx);
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu_err = dmax(Qu_err, fabs(Qu[i]));   // :576
    }
    if (!fail) { ok = true; break; }
    if (force == 2) break;   // single un-retried pass (step-level API)
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  if (ok) { d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = Qu_err; }
  if (force) return;
  // handleBackwardPassRegularizationLimit (:216-222): LogDDP treats an exhausted regularisation as converged
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT_CONVERGED; d.phase[b] = PH_DONE; return; }
  d.phase[b] = PH_FWD1;   // no early convergence test (base-class default)
}

// ================================================================================ K4
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_forward_logddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int a0, int na, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, NSEG = Cons::NSEG, MM = M > 0 ? M : 1, NS = NSEG > 0 ? NSEG : 1;
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
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  double *Xn = d.X + (size_t)slot * d.planeX;
  double *Un = d.U + (size_t)slot * d.planeU;
  double *evn = d.ev + (size_t)slot * lg_ev_plane<Cons>(d);
  const double alpha = P->alphas[a];
  const double mu = d.mu[b], delta = o.logddp_relaxed_delta;
  atomicAdd(d.launched, 1ull);
  DynCtx dc;
  dc.load(P->integrator, P->dt, P->mp);
  typename Obj::Ctx oc;
  Obj::load(P, oc);
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  double x[NX];
  ld<NX>(Xc + GI(0, NX, 0), kLS, x);     // X_[0] == initial state (:607)
  st<NX>(Xn + GI(0, NX, 0), kLS, x);
  double cost = 0.0, merit_b = 0.0, viol = 0.0;
  bool finite = true;
  int steps = N;
  constexpr int REC = NX + 2 * NU + NU * NX;
  constexpr bool kPF = REC <= 32;
  struct Rec { double xo[NX], uo[NU], kk[NU], KK[NU * NX]; };
  auto fetch = [&](int tt, Rec &r) {
    ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
    ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
    ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
    ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
  };
  // The reference rolls the whole horizon out first (abandoning the trial at the first non-finite x_{t+1} or u_t, :617-639) and
  // evaluates cost / barrier / violation in a second loop (:641-666); fused here: every accumulator still sees its own terms in
  // the same order, and the sums of an abandoned trial are never read.
  auto step = [&](const int t, const Rec &c, Rec &n) {
    if constexpr (kPF) { fetch(t + 1 < N ? t + 1 : N - 1, n); PIPELINE_FENCE(); }
    double u[NU], dx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[i] = x[i] - c.xo[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += c.KK[i * NX + j] * dx[j];
      u[i] = (c.uo[i] + alpha * c.kk[i]) + s;
    }
    cost += Obj::running_cost(oc, xrt, t, x, u);
    if constexpr (M > 0) {
      double g[MM], bs[NS];
      Cons::template eval<NX, NU>(cc, x, u, g);
      LgCons<Cons>::template values<NX, NU>(g, delta, bs, viol);
#pragma unroll
      for (int s = 0; s < NSEG; ++s) { merit_b += mu * bs[s]; evn[GI(t, NS, s)] = bs[s]; }
    }
    double xn[NX];
    Stepper<Model>::step(dc, x, u, xn);
    bool fin = true;
#pragma unroll
    for (int i = 0; i < NX; ++i) fin = fin && dfinite(xn[i]);
#pragma unroll
    for (int i = 0; i < NU; ++i) fin = fin && dfinite(u[i]);
    if (finite && !fin) { finite = false; steps = t; }
    st<NU>(Un + GI(t, NU, 0), kLS, u);
    st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  };
  if constexpr (kPF) {
    Rec ra, rb;
    fetch(0, ra);
    {
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      st<NU>(Un + GI(0, NU, 0), kLS, z);
      st<NX>(Xn + GI(1, NX, 0), kLS, z);
    }
    int t = 0;
    for (; t + 1 < N; t += 2) { step(t, ra, rb); step(t + 1, rb, ra); }
    if (t < N) step(t, ra, rb);
  } else {
    for (int t = 0; t < N; ++t) { Rec r; fetch(t, r); step(t, r, r); }
  }
  cost += Obj::terminal_cost(P, x);
  const double merit = merit_b + cost;       // merit_function_new += cost_new (:669)
  // filter-based acceptance (:671-697)
  const double cv_old = d.filter_theta[b], cv_new = viol, merit_old = d.merit[b];
  const double expected = alpha * d.dV0[b];
  bool accept = false;
  if (cv_new > o.filter_max_violation_threshold) {
    if (cv_new < (1.0 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
  } else if (dmax(cv_new, cv_old) < o.filter_min_violation_for_armijo_check && expected < 0) {
    if (merit < merit_old + o.filter_armijo_constant * expected) accept = true;
  } else {
    if (merit < merit_old - o.filter_merit_acceptance_threshold * cv_old || cv_new < (1.0 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
  }
  const size_t ti = (size_t)a * d.Bp + b;
  d.t_steps[ti] = steps;
  d.t_success[ti] = (finite && accept) ? 1 : 0;
  d.t_cost[ti] = cost; d.t_merit[ti] = merit; d.t_theta[ti] = 0.0; d.t_inf_pr[ti] = cv_new; d.t_inf_comp[ti] = 0.0;
  d.t_apr[ti] = alpha; d.t_adu[ti] = 1.0;
}

// ================================================================================ K4 (two-role)
// Round 5: the single-shooting rollout as a PRODUCER / CONSUMER pair of wavefronts per (64-trajectory tile, alpha) -- the form of
// kernels_lean.hpp::k_forward_ipddp_pc.  wave 0: u_t = u + a k + K dx, x_{t+1} = f(x_t, u_t), the non-finite test, X / U stores, l_f(x_N)
// (logddp_solver.cpp:607-639); wave 1: running cost, g(x_t, u_t), the relaxed barrier's values and the violation (:641-666), the
// filter test (:671-697) and the trial record.  An LDS ring carries (x_t, u_t); every accumulator sees its terms in the one-wave
// kernel's order, so the pair is bitwise that kernel (tests/test_logddp_device.py::test_two_role_rollout_agrees_bitwise).
template <class Model, class Cons>
__global__ __launch_bounds__(128) void k_forward_logddp_pc(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int a0, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, NSEG = Cons::NSEG, MM = M > 0 ? M : 1, NS = NSEG > 0 ? NSEG : 1;
  typedef Objective<NX, NU> Obj;
  constexpr int RW = NX + NU;
  constexpr int kRing = RW <= 8 ? 8 : (RW <= 16 ? 4 : 2);
  __shared__ double s_ring[kRing * RW * 64];
  __shared__ int s_prod, s_cons;
  __shared__ int s_psteps[64];    // steps completed before the producer lane went non-finite (N = never)
  __shared__ double s_pcost[64];  // l_f(x_N)
  __shared__ double s_obj[Obj::kStage];
  const int lane = threadIdx.x & 63;
  const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64;
  const int b = blockIdx.x * 64 + lane;
  const int a = a0 + blockIdx.y;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const bool active = (b < d.B) && (force || d.phase[b] == phase_req);
  if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;
  if (producer && lane == 0) { s_prod = 0; s_cons = 0; }
  Obj::stage(P, s_obj, (int)threadIdx.x, 128);
  __syncthreads();
  const int bb = (b < d.B) ? b : 0;
  const int cur = (b < d.B) ? d.cur[b] : 0;
  const int slot = trial_slot(cur, a);
  const double alpha = P->alphas[a];
  auto wait_ge = [&](int *ctr, int need) -> int {
    int v;
    while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    return v;
  };
  if (producer) {
    const double *Xc = d.X + (size_t)cur * d.planeX, *Uc = d.U + (size_t)cur * d.planeU;
    double *Xn = d.X + (size_t)slot * d.planeX, *Un = d.U + (size_t)slot * d.planeU;
    DynCtx dc;
    dc.load(P->integrator, P->dt, P->mp);
    double x[NX];
    ld<NX>(Xc + GI(0, NX, 0), kLS, x);
    st<NX>(Xn + GI(0, NX, 0), kLS, x);
    bool finite = true;
    int steps = N;
    struct Rec { double xo[NX], uo[NU], kk[NU], KK[NU * NX]; };
    auto fetch = [&](int tt, Rec &r) {
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
      ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
      ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
    };
    constexpr bool kPing = sizeof(Rec) <= 40 * sizeof(double);
    auto step = [&](const int t, Rec &c, Rec &n) {
      if constexpr (kPing) { fetch(t + 1 < N ? t + 1 : N - 1, n); PIPELINE_FENCE(); }
      double u[NU], dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = x[i] - c.xo[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += c.KK[i * NX + j] * dx[j];
        u[i] = (c.uo[i] + alpha * c.kk[i]) + s;
      }
      if (t >= kRing && (t % (kRing / 2)) == 0) wait_ge(&s_cons, t - kRing / 2);
      {
        double *rs = s_ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
        for (int i = 0; i < NX; ++i) rs[i * 64] = x[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) rs[(NX + i) * 64] = u[i];
        RING_FENCE();
        __hip_atomic_store(&s_prod, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if constexpr (!kPing) { fetch(t + 1 < N ? t + 1 : t, c); PIPELINE_FENCE(); }
      double xn[NX];
      Stepper<Model>::step(dc, x, u, xn);
      bool fin = true;
#pragma unroll
      for (int i = 0; i < NX; ++i) fin = fin && dfinite(xn[i]);
#pragma unroll
      for (int i = 0; i < NU; ++i) fin = fin && dfinite(u[i]);
      if (finite && !fin) { finite = false; steps = t; }
      st<NU>(Un + GI(t, NU, 0), kLS, u);
      st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = xn[i];
    };
    Rec ra, rb;
    fetch(0, ra);
    {
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      st<NU>(Un + GI(0, NU, 0), kLS, z);
      st<NX>(Xn + GI(1, NX, 0), kLS, z);
    }
    int t = 0;
    if constexpr (kPing) {
      for (; t + 1 < N; t += 2) { step(t, ra, rb); step(t + 1, rb, ra); }
      if (t < N) step(t, ra, rb);
    } else {
      for (; t < N; ++t) step(t, ra, ra);
    }
    s_pcost[lane] = Obj::terminal_cost(P, x);
    s_psteps[lane] = finite ? N + 1 : steps;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(&s_prod, N + kRing + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
  // -------------------------------------------------------------------- consumer
  double *evn = d.ev + (size_t)slot * lg_ev_plane<Cons>(d);
  const double mu = d.mu[bb], delta = o.logddp_relaxed_delta;
  if (active) atomicAdd(d.launched, 1ull);
  typename Obj::Ctx oc;
  Obj::load_staged(P, oc, s_obj);
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  double cost = 0.0, merit_b = 0.0, viol = 0.0;
  int seen_prod = 0;     // last value of the producer's counter this wave saw (wave-uniform)
  if constexpr (M > 0) {   // the VMEM queue primed with one step's store pattern
#pragma unroll
    for (int s = 0; s < NSEG; ++s) evn[GI(0, NS, s)] = 0.0;
  }
  for (int t = 0; t < N; ++t) {
    if (!CDDP_RING_LAZY_POLL || seen_prod < t + 1) seen_prod = __builtin_amdgcn_readfirstlane(wait_ge(&s_prod, t + 1));
    double x[NX], u[NU];
    {
      const double *rs = s_ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = rs[i * 64];
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = rs[(NX + i) * 64];
      RING_FENCE();
      __hip_atomic_store(&s_cons, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    cost += Obj::running_cost(oc, xrt, t, x, u);
    if constexpr (M > 0) {
      double g[MM], bs[NS];
      Cons::template eval<NX, NU>(cc, x, u, g);
      LgCons<Cons>::template values<NX, NU>(g, delta, bs, viol);
#pragma unroll
      for (int s = 0; s < NSEG; ++s) { merit_b += mu * bs[s]; evn[GI(t, NS, s)] = bs[s]; }
    }
  }
  wait_ge(&s_prod, N + kRing + 1);
  if (!active) return;
  cost += s_pcost[lane];
  const int psteps = s_psteps[lane];
  const bool finite = psteps > N;
  const int steps = finite ? N : psteps;
  const double merit = merit_b + cost;       // merit_function_new += cost_new (:669)
  const double cv_old = d.filter_theta[b], cv_new = viol, merit_old = d.merit[b];
  const double expected = alpha * d.dV0[b];
  bool accept = false;
  if (cv_new > o.filter_max_violation_threshold) {
    if (cv_new < (1.0 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
  } else if (dmax(cv_new, cv_old) < o.filter_min_violation_for_armijo_check && expected < 0) {
    if (merit < merit_old + o.filter_armijo_constant * expected) accept = true;
  } else {
    if (merit < merit_old - o.filter_merit_acceptance_threshold * cv_old || cv_new < (1.0 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
  }
  const size_t ti = (size_t)a * d.Bp + b;
  d.t_steps[ti] = steps;
  d.t_success[ti] = (finite && accept) ? 1 : 0;
  d.t_cost[ti] = cost; d.t_merit[ti] = merit; d.t_theta[ti] = 0.0; d.t_inf_pr[ti] = cv_new; d.t_inf_comp[ti] = 0.0;
  d.t_apr[ti] = alpha; d.t_adu[ti] = 1.0;
}

// ================================================================================ K5
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_update_logddp(DevBuf d, const ProblemDev *__restrict__ Pk, int stage, int n1, int is_last_iter, int do_count) {
  constexpr int NSEG = Cons::NSEG, NS = NSEG > 0 ? NSEG : 1;
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
      for (int base = lo; base < hi && win < 0; base += 8) {
        int f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (base + i < hi) ? d.t_success[(size_t)(base + i) * d.Bp + b] : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (win < 0 && f[i] == 1) win = base + i;
      }
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
      int slot_now = d.cur[b];
      if (win >= 0) {
        const size_t ti = (size_t)win * d.Bp + b;
        const int old_cur = d.cur[b];
        const double w_cost = d.t_cost[ti], w_merit = d.t_merit[ti], w_apr = d.t_apr[ti], w_cv = d.t_inf_pr[ti];
        const double dJ = d.cost[b] - w_cost, dL = d.merit[b] - w_merit;
        const double c_reg = d.reg[b], c_ipr = d.inf_pr[b], c_idu = d.inf_du[b];
        const int na_walked = (P->ls_rule == CDDP_HIP_LS_FIRST_SUCCESS) ? win + 1 : n_alphas;
        int ns = d.n_fwd_steps[b];
        for (int a = 0; a < na_walked; ++a) ns += d.t_steps[(size_t)a * d.Bp + b];
        d.n_fwd[b] = d.n_fwd[b] + na_walked;
        d.n_fwd_steps[b] = ns;
        slot_now = trial_slot(old_cur, win);
        d.cur[b] = slot_now;
        d.cost[b] = w_cost; d.merit[b] = w_merit; d.alpha_pr[b] = w_apr; d.alpha_du[b] = 1.0;   // applyForwardPassResult (:224-231)
        d.filter_theta[b] = w_cv; d.theta[b] = w_cv;
        hist_push(d, b, mu);
        d.reg[b] = reg_decrease(o, c_reg);
        // checkConvergence (:233-261): inf_pr is still the value of the last resetFilter
        int st = CDDP_HIP_STATUS_RUNNING;
        if (dmax(c_idu, c_ipr) <= o.tolerance) st = CDDP_HIP_STATUS_OPTIMAL;
        else if (fabs(dJ) < o.acceptable_tolerance && fabs(dL) < o.acceptable_tolerance) st = CDDP_HIP_STATUS_ACCEPTABLE;
        if (st != CDDP_HIP_STATUS_RUNNING) { d.status[b] = st; d.phase[b] = PH_DONE; running = false; }
        else mu = dmax(o.logddp_mu_min_value, mu * o.logddp_mu_update_factor);   // postIterationUpdate (:263-277)
      } else {
        // handleForwardPassFailure (cddp_solver_base.cpp:206-218)
        const int nf0 = d.n_fwd[b]; int ns = d.n_fwd_steps[b]; const double reg0 = d.reg[b];
        for (int a = 0; a < n_alphas; ++a) ns += d.t_steps[(size_t)a * d.Bp + b];
        d.n_fwd[b] = nf0 + n_alphas;
        d.n_fwd_steps[b] = ns;
        const double reg = reg_increase(o, reg0);
        d.reg[b] = reg;
        if (reg >= o.reg_max_value) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; running = false; }
        else mu = dmin(o.logddp_mu_initial, mu * 5.0);
      }
      if (running) {
        // resetFilter (:333-361) under the new mu: the merit chain replayed from the parked barrier sums of the current iterate's slot;
        // the violation is the sum that slot's rollout evaluated (see the header)
        d.mu[b] = mu;
        double mer = d.cost[b];
        if constexpr (NSEG > 0) {
          const int N = d.N;
          const double *evb = d.ev + (size_t)slot_now * lg_ev_plane<Cons>(d);
          constexpr int kRB = 16;
          const int total = N * NSEG;     // rows in (t, c) order; row r = (t, c) = (r / NSEG, r % NSEG)
          int r = 0;
          for (; r + kRB - 1 < total; r += kRB) {   // kRB row loads per round trip, then the ordered chain
            double v[kRB];
#pragma unroll
            for (int k = 0; k < kRB; ++k) v[k] = evb[GI((r + k) / NSEG, NS, (r + k) % NSEG)];
#pragma unroll
            for (int k = 0; k < kRB; ++k) mer += mu * v[k];
          }
          for (; r < total; ++r) mer += mu * evb[GI(r / NSEG, NS, r % NSEG)];
        }
        d.merit[b] = mer; d.phi[b] = mer;
        d.inf_pr[b] = d.filter_theta[b];
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
