// Hand-written HIP kernels (gfx950) of the batched CLDDP / IPDDP backward-forward sweep.
//
//   K1  k_derivs            grid (batch x N)   f_x/f_u -> A_t = I + dt f_x, B_t = dt f_u stacks
//                                              (reference cddp_solver_base.cpp:319-394, clddp_solver.cpp:113-118)
//   K2  k_backward_clddp    grid (batch)       Riccati sweep + BoxQP          (clddp_solver.cpp:79-204, boxqp.cpp)
//       k_backward_ipddp    grid (batch)       Riccati sweep + KKT condensation + linear rollout
//                                              + step-size caps               (ipddp_solver.cpp:960-1569, 2939-2988)
//   K4  k_forward_clddp     grid (batch x n_a) clamped nonlinear rollout       (clddp_solver.cpp:215-262)
//       k_forward_ipddp     grid (batch x n_a) primal-dual rollout + filter    (ipddp_solver.cpp:1571-1876)
//   K5  k_update            grid (batch)       line-search selection, commit, barrier/filter update,
//                                              regularisation schedule, convergence (cddp_solver_base.cpp:29-186,
//                                              ipddp_solver.cpp:1878-2082, 2548-2660)
//   K0  k_init              grid (batch)       ISolverAlgorithm::initialize   (clddp_solver.cpp:28-75, ipddp_solver.cpp:819-913)
//
// One trajectory per lane; time is serial inside a lane (the Riccati recursion is a length-N
// dependency chain), the batch (and the alpha ladder) is the parallel axis.  All stacks are
// batch-minor so each wavefront load/store is one coalesced 512-B transaction.
#pragma once
#include "dev_constraints.hpp"
#include "dev_boxqp.hpp"
#include "dev_models.hpp"
#include "dev_terminal.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))
// hipcc sinks the prefetch loads of the software pipeline down to their first use (next iteration), which
// removes the overlap; a compiler-level memory barrier right after issuing them pins them at the loop top.
#define PIPELINE_FENCE() asm volatile("" ::: "memory")
// Hand-over points of the rollouts' LDS rings (kernels_lean.hpp, kernels_logddp.hpp, kernels_msipddp.hpp): ring rows first, then the
// counter.  Both are LDS operations of ONE wavefront, which the LDS pipeline executes in issue order, so a compiler barrier would do
// (CDDP_RING_FENCE_WAIT=0: built, 257 bitwise / parity tests green, and NO measurable gain on C2 / C3 / LogDDP / MSIPDDP,
// profiles/r05_ring_fence.md): the s_waitcnt lgkmcnt(0) stays -- it does not lean on that property and costs nothing measurable.
#ifndef CDDP_RING_FENCE_WAIT
#define CDDP_RING_FENCE_WAIT 1
#endif
// The consumer remembers the last value of the producer's counter it saw and polls again only when that value does not cover the step it
// is about to take: with the producer ahead by several steps (the ring holds up to 8) a poll -- an LDS round trip on the consumer's
// chain -- is needed once per several steps instead of every step.  CDDP_RING_LAZY_POLL=0: poll every step (rounds 1 - 4).
#ifndef CDDP_RING_LAZY_POLL
#define CDDP_RING_LAZY_POLL 1
#endif
#if CDDP_RING_FENCE_WAIT
#define RING_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define RING_FENCE() asm volatile("" ::: "memory")
#endif
// Sub-tile-minor ("T4") layout of the SWEEP-INPUT stacks (A_t, B_t, condensed terms) of the G = 16 cooperative sweeps (nx > 8:
// kernels_coop.hpp::k_backward_ipddp_coop_big, kernels_te.hpp::k_backward_te_coop).  A single-wave workgroup of those sweeps holds
// 4 trajectories, i.e. 32 B of every 512-B row of the wave-tiled layout: sixteen workgroups share each row (four share each
// 128-B line) and, drifting apart over 150 - 400 dependent steps, fetch it again and again (round 3, after the XCD map: 7.1 GB
// fetched per launch against 3.7 GB at C4, 17.5 against 4.7 at C5 -- profiles/r03_pmc_counters_*.md).  Here element e of step t
// of trajectory b lives at (((t * NB * 16 + b / 4) * E + e) * 4 + b % 4): the step record of one workgroup's 4 trajectories is
// E * 32 B of CONTIGUOUS memory that no other workgroup touches, and a cooperative fetch of 16 consecutive elements x 4
// trajectories is one 512-B coalesced access.  Written by the (batch x N) producers (k_derivs, k_condense, k_te_condense), read
// by the sweeps, k_te_post and the host getter; d.t4 is set per launch by launch.hpp (same rule in derivs() and backward()).
#define GT(t, E, e) (d.t4 ? ((((size_t)(t) * (size_t)d.NB * 16 + (size_t)(b >> 2)) * (size_t)(E) + (size_t)(e)) * 4 + (size_t)(b & 3)) : GI(t, E, e))
// index into a stack that exists ONLY in the sub-tile-minor form (d.Kt: the gains the sweep just wrote, re-read by its own rollouts)
#define G4(t, E, e) ((((size_t)(t) * (size_t)d.NB * 16 + (size_t)(b >> 2)) * (size_t)(E) + (size_t)(e)) * 4 + (size_t)(b & 3))
#define TSTRIDE (d.t4 ? (size_t)4 : (size_t)kLS)   // distance between consecutive elements of one trajectory's record

constexpr double kSlackInteriorOffset = 1e-4;   // ipddp_solver.cpp:35-38
constexpr double kEpsSlack = 1e-10;
constexpr double kMaxBarrierRatio = 1e6;

DEV double clip_pos(double num, double den) { return dclamp(num / den, 0.0, kMaxBarrierRatio); }
DEV double clip_sgn(double num, double den) { return dclamp(num / den, -kMaxBarrierRatio, kMaxBarrierRatio); }

template <int N> DEV void ld(const double *base, size_t stride, double *out) {
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = base[(size_t)i * stride];
}
template <int N> DEV void st(double *base, size_t stride, const double *in) {
#pragma unroll
  for (int i = 0; i < N; ++i) base[(size_t)i * stride] = in[i];
}

// regularisation schedule (cddp_core.cpp:308-346)
// A regularisation of exactly 0 (cddp_hip_set_barrier_state, reg_initial_value = 0) is a fixed point of the reference's
// rule reg = min(reg * f, max): its retry loop (cddp_solver_base.cpp:93-111) would spin forever on the host; here it would
// wedge a GPU queue.  From 0 the step therefore restarts at reg_min_value (the value decreaseRegularization never goes
// below; reg_max_value if that is 0 too, i.e. "limit reached").  Identical to the reference for every reg > 0.
DEV double reg_increase(const cddp_hip_options &o, double r) {
  r *= o.reg_update_factor;
  if (!(r > 0.0)) r = (o.reg_min_value > 0.0) ? o.reg_min_value : o.reg_max_value;
  return dmin(r, o.reg_max_value);
}
DEV double reg_decrease(const cddp_hip_options &o, double r) { r /= o.reg_update_factor; return dmax(r, o.reg_min_value); }

DEV void hist_push(const DevBuf &d, int b, double mu_or_zero) {
  if (b >= d.hist_batch) return;
  int n = d.hist_n[b];
  if (n >= d.hist_cap) return;
  double *row = d.hist + ((size_t)b * d.hist_cap + n) * kHistCols;
  row[0] = d.cost[b]; row[1] = d.merit[b]; row[2] = d.alpha_pr[b]; row[3] = d.alpha_du[b];
  row[4] = d.inf_du[b]; row[5] = d.inf_pr[b]; row[6] = d.inf_comp[b]; row[7] = mu_or_zero; row[8] = d.reg[b];
  d.hist_n[b] = n + 1;
}

// ================================================================================ K1
template <class Model>
__global__ __launch_bounds__(64) void k_derivs(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force) {
  constexpr int NX = Model::NX, NU = Model::NU;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  // first kernel of an outer iteration: reset the "still running" counter K5 adds to (saves a memset node per iteration)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && !force) *d.n_active = 0;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  double x[NX], u[NU], Fx[NX * NX], Fu[NX * NU];
  ld<NX>(Xc + GI(t, NX, 0), kLS, x);
  ld<NU>(Uc + GI(t, NU, 0), kLS, u);
  Model::jac(P->mp, x, u, Fx, Fu);
  const double dt = P->dt;
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double a = dt * Fx[i * NX + j];
      if (i == j) a += 1.0;
      d.A[GT(t, NX * NX, i * NX + j)] = a;
    }
#pragma unroll
  for (int i = 0; i < NX * NU; ++i) d.Bm[GT(t, NX * NU, i)] = dt * Fu[i];
}

// Q-function blocks shared by both solvers:
//   Q_x = l_x [+ Q_yx^T y] + A^T V_x, Q_u = l_u [+ Q_yu^T y] + B^T V_x,
//   Q_xx = l_xx + (A^T V_xx) A, Q_ux = l_ux + (B^T V_xx) A, Q_uu = l_uu + (B^T V_xx) B
template <int NX, int NU>
DEV void q_blocks(const double *Q, const double *R, const double *A, const double *Bm, const double *Vxx,
                  double *Qxx, double *Qux, double *Quu) {
  double T1[NX * NX], T2[NU * NX];
  mm_tn<NX, NX, NX>(A, Vxx, T1);    // A^T V_xx
  mm_tn<NU, NX, NX>(Bm, Vxx, T2);   // B^T V_xx
  mm_nn<NX, NX, NX>(T1, A, Qxx);
  mm_nn<NU, NX, NX>(T2, A, Qux);
  mm_nn<NU, NX, NU>(T2, Bm, Quu);
#pragma unroll
  for (int i = 0; i < NX * NX; ++i) Qxx[i] = (2.0 * Q[i]) + Qxx[i];
#pragma unroll
  for (int i = 0; i < NU * NU; ++i) Quu[i] = (2.0 * R[i]) + Quu[i];
}
template <int NX, int NU>
DEV void q_blocks(const ProblemDev *P, const double *A, const double *Bm, const double *Vx, const double *Vxx,
                  double *Qxx, double *Qux, double *Quu) {
  q_blocks<NX, NU>(P->pool + P->off_Qdt, P->pool + P->off_Rdt, A, Bm, Vxx, Qxx, Qux, Quu);
  (void)Vx;
}

// Second-order dynamics terms of full DDP (options.use_ilqr == 0): with F_xx_[t][i] = dt f_xx[i] etc.
// (cddp_solver_base.cpp:346-356),  for i: Q_xx += w(i) F_xx[i]; Q_ux += w(i) F_ux[i]; Q_uu += w(i) F_uu[i]
// (ipddp_solver.cpp:1070-1082, 1396-1408; w = V_x of step t + 1).
template <class Model>
DEV void ddp_tensor_terms(const ProblemDev *P, const double *x, const double *u, const double *w, double *Qxx, double *Qux, double *Quu) {
  constexpr int NX = Model::NX, NU = Model::NU;
  if constexpr (Model::kHasHess) {
    double Fxx[NX * NX * NX], Fuu[NX * NU * NU], Fux[NX * NU * NX];
    Model::hess(P->mp, x, u, Fxx, Fuu, Fux);
    const double dt = P->dt;
    for (int i = 0; i < NX; ++i) {
      for (int e = 0; e < NX * NX; ++e) Qxx[e] = Qxx[e] + w[i] * (dt * Fxx[i * NX * NX + e]);
      for (int e = 0; e < NU * NX; ++e) Qux[e] = Qux[e] + w[i] * (dt * Fux[i * NU * NX + e]);
      for (int e = 0; e < NU * NU; ++e) Quu[e] = Quu[e] + w[i] * (dt * Fuu[i * NU * NU + e]);
    }
  } else if constexpr (HessBlocked<Model>::value) {   // same sums, same order per entry; the tensors are never stored (dev_models.hpp)
    ad_tensor_terms_blocked<typename Model::HessDyn, NX, NU, 4>(P->mp, x, u, w, P->dt, Model::kHessDiv, Qxx, Qux, Quu);
  }
}

// ================================================================================ K2 (CLDDP)
template <class Model>
__global__ __launch_bounds__(64) void k_backward_clddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  if (count_iter) d.iter[b] += 1;
  double reg = d.reg[b];
  const int box = P->clddp_box;
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0;
  for (;;) {
    ++nb;
    double xN[NX], Vx[NX], Vxx[NX * NX];
    ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
    Obj::final_grad(P, xN, Vx);
    const double *Qf = P->pool + P->off_Qf;
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Vxx[i] = 2.0 * Qf[i];
    st<NX>(d.Vx + GI(N, NX, 0), kLS, Vx);
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Vxx);
    dV0 = 0; dV1 = 0;
    double norm_Vx = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) norm_Vx += fabs(Vx[i]);
    double Qu_error = 0.0;
    bool fail = false;
    struct StepIn { double A[NX * NX], Bm[NX * NU], x[NX], u[NU], k0[NU]; };
    auto load_step = [&](int tt, StepIn &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.x);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.u);
      ld<NU>(d.k + GI(tt, NU, 0), kLS, r.k0);      // BoxQP warm start x0 = k_u_[t] of the previous iteration
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
      double Quu_reg[NU * NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu_reg[i] = Quu[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Quu_reg[i * NU + i] += reg;
      if (min_real_eig<NU>(Quu_reg) <= 0) { fail = true; break; }   // clddp_solver.cpp:133-140
      double kk[NU], KK[NU * NX];
      if (box < 0) {   // clddp_solver.cpp:142-145
        double H[NU * NU];
        inverse_pplu<NU>(Quu_reg, H);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) s += (-H[i * NU + j]) * Qu[j];
          kk[i] = s;
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) s2 += (-H[i * NU + j]) * Qux[j * NX + c];
            KK[i * NX + c] = s2;
          }
        }
      } else {         // clddp_solver.cpp:147-178
        const ConDev &cc = P->cons[box];
        double lb[NU], ub[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { lb[i] = P->pool[cc.off_lower + i] - u[i]; ub[i] = P->pool[cc.off_upper + i] - u[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = cs.k0[i];   // warm start x0 = k_u_[t]
        if constexpr (NU == 1) {   // scalar BoxQP, see dev_boxqp.hpp::boxqp_solve1
          int fr;
          const int stq1 = boxqp_solve1(o, Quu_reg[0], Qu[0], lb[0], ub[0], kk[0], fr);
          if (stq1 == BQ_HESSIAN_NOT_PD || stq1 == BQ_NO_DESCENT) { fail = true; break; }
#pragma unroll
          for (int c = 0; c < NX; ++c) KK[c] = fr ? -ldlt1_solve(Quu_reg[0], Qux[c]) : 0.0;
        } else {
        int free_[NU];
        LDLTd<NU> Hfree;
        int stq = boxqp_solve<NU>(o, Quu_reg, Qu, lb, ub, kk, free_, Hfree);
        if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) { fail = true; break; }
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) KK[i] = 0.0;
        int free_idx[NU]; int nf = 0;
        for (int i = 0; i < NU; ++i) if (free_[i]) free_idx[nf++] = i;
        if (nf > 0) {
          for (int c = 0; c < NX; ++c) {
            double col[NU];
            for (int i = 0; i < nf; ++i) col[i] = Qux[free_idx[i] * NX + c];
            Hfree.solve(col);
            for (int i = 0; i < nf; ++i) KK[free_idx[i] * NX + c] = -col[i];
          }
        }
        }
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      // dV (un-regularised Q_uu, clddp_solver.cpp:184-186)
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
      // V_x = Q_x + (K^T Q_uu) k + Q_ux^T k + K^T Q_u ; V_xx = Q_xx + (K^T Q_uu) K + Q_ux^T K + K^T Q_ux
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
      { double s = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) s += fabs(Vx[i]);
        norm_Vx += s; }
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu_error = dmax(Qu_error, fabs(Qu[i]));
    }
    if (!fail) {
      double scaling = o.termination_scaling_max_factor;
      scaling = dmax(scaling, norm_Vx / (N * NX)) / scaling;
      inf_du = Qu_error / scaling;
      ok = true;
      break;
    }
    if (force == 2) break;   // single un-retried pass (step-level API)
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  if (ok) { d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du; }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  if (inf_du < o.tolerance) {   // checkEarlyConvergence (clddp_solver.cpp:206-213)
    d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, 0.0); return;
  }
  d.phase[b] = PH_FWD1;
}


// rolloutLinearPolicy with dx0 = 0 (ipddp_solver.cpp:368-392, 1511-1520) fused with the slack / dual
// directions dS = k_s + K_s dX, dY = clamp(k_y + K_y dX) (:1522-1532), the terminal-inequality
// directions (:1534-1561) and computeMaxStepSizes (:2939-2988).  Nothing but the two caps is stored.
template <class Model, class Cons, bool TERM>
DEV void lin_rollout_caps(const DevBuf &d, const ProblemDev *__restrict__ P, int b, const double *Sc, const double *Yc,
                          const double *Xc, double mu, double &apr, double &adu) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int mT = TERM ? P->mT : 0;
  apr = 1.0; adu = 1.0;
  if (M == 0 && mT == 0) return;
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  double dx[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) dx[i] = 0.0;
  // software pipeline: gains / slack record of step t+1 in flight while step t is reduced
  struct StepIn { double kk[NU], KK[NU * NX], A[NX * NX], Bm[NX * NU], ksv[MM], ky[MM], Ksm[MM * NX], Ky[MM * NX], s[MM], y[MM]; };
  auto load_step = [&](int tt, StepIn &r) {
    ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
    ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
    if (tt < N - 1 || mT > 0) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
    }
    if constexpr (M > 0) {
      ld<M>(d.ks + GI(tt, M, 0), kLS, r.ksv);
      ld<M>(d.ky + GI(tt, M, 0), kLS, r.ky);
      ld<M * NX>(d.Ks + GI(tt, M * NX, 0), kLS, r.Ksm);
      ld<M * NX>(d.Ky + GI(tt, M * NX, 0), kLS, r.Ky);
      ld<M>(Sc + GI(tt, M, 0), kLS, r.s);
      ld<M>(Yc + GI(tt, M, 0), kLS, r.y);
    }
  };
  StepIn nxt;
  load_step(0, nxt);
  for (int t = 0; t < N; ++t) {
    StepIn cs = nxt;
    if (t + 1 < N) load_step(t + 1, nxt);
    PIPELINE_FENCE();
    if constexpr (M > 0) {
#pragma unroll
      for (int r = 0; r < M; ++r) {
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) { a += cs.Ksm[r * NX + j] * dx[j]; c += cs.Ky[r * NX + j] * dx[j]; }
        double ds = cs.ksv[r] + a;
        double dy = dmin(dmax(cs.ky[r] + c, -kMaxBarrierRatio), kMaxBarrierRatio);
        if (ds < 0.0) apr = dmin(apr, -tau * cs.s[r] / ds);
        if (dy < 0.0) adu = dmin(adu, -tau * cs.y[r] / dy);
      }
    }
    if (t < N - 1 || mT > 0) {
      double du[NU], dxn[NX];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) a += cs.KK[i * NX + j] * dx[j];
        du[i] = cs.kk[i] + a; }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) a += cs.A[i * NX + j] * dx[j];
#pragma unroll
        for (int j = 0; j < NU; ++j) c += cs.Bm[i * NU + j] * du[j];
        dxn[i] = (a + c) + 0.0;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
    }
  }
  if constexpr (TERM) {
    if (mT > 0) {   // terminal-inequality directions from dX_N (ipddp_solver.cpp:1534-1561)
      double xN[NX], gT[kMTMax];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      term_ineq_eval<NX>(P, xN, gT);
      const double fl0 = dmax(mu * 1e-3, kEpsSlack);
      for (int i = 0; i < mT; ++i) {
        const double *row = term_ineq_row(P, i);
        const double sT = d.ST[(size_t)i * d.Bp + b], yT = d.YT[(size_t)i * d.Bp + b];
        const double r_p = gT[i] + sT;
        const double r_d = sT * yT - mu;
        double gd = 0.0;
        for (int j = 0; j < NX; ++j) gd += row[j] * dx[j];
        const double dsT = (-r_p) - gd;
        const double s_safe = dmax(sT, fl0);
        const double dual_ratio = dclamp(yT / s_safe, 0.0, kMaxBarrierRatio);
        const double affine = dclamp(-r_d / s_safe, -kMaxBarrierRatio, kMaxBarrierRatio);
        const double dyT = dclamp(affine - dual_ratio * dsT, -kMaxBarrierRatio, kMaxBarrierRatio);
        d.dST[(size_t)i * d.Bp + b] = dsT; d.dYT[(size_t)i * d.Bp + b] = dyT;
        if (dsT < 0.0) apr = dmin(apr, -tau * sT / dsT);
        if (dyT < 0.0) adu = dmin(adu, -tau * yT / dyT);
      }
    }
  }
  apr = dclamp(apr, 0.0, 1.0); adu = dclamp(adu, 0.0, 1.0);
}


// ---------------------------------------------------------------------------------------------
// Terminal-equality reduced LQR branch of IPDDPSolver::backwardPass (ipddp_solver.cpp:1120-1353):
// (p+1) sequential LQR sweeps (solveSequentialLQR :413-476) + linear rollouts (:368-392), the
// p x p regularised normal-equation solve for the terminal multipliers (solveTerminalEqualityLQR
// :478-639) and the recombination of k, p.  One lane does all variants of its trajectory.
// ---------------------------------------------------------------------------------------------
template <int NMAXP>
DEV void singular_minmax(const double *A, int n, double &smax, double &smin) {   // one-sided Jacobi
  double U[NMAXP * NMAXP];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) U[i * NMAXP + j] = A[i * NMAXP + j];
  for (int sweep = 0; sweep < 80; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < n; ++i) { alpha += U[i * NMAXP + p] * U[i * NMAXP + p]; beta += U[i * NMAXP + q] * U[i * NMAXP + q]; gamma += U[i * NMAXP + p] * U[i * NMAXP + q]; }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < n; ++i) { double up = U[i * NMAXP + p], uq = U[i * NMAXP + q]; U[i * NMAXP + p] = cs * up - sn * uq; U[i * NMAXP + q] = sn * up + cs * uq; }
      }
    if (!rotated) break;
  }
  smax = 0.0; smin = INFINITY;
  for (int j = 0; j < n; ++j) { double s2 = 0; for (int i = 0; i < n; ++i) s2 += U[i * NMAXP + j] * U[i * NMAXP + j]; double sv = sqrt(s2); smax = dmax(smax, sv); smin = dmin(smin, sv); }
  if (n == 0) { smax = 0.0; smin = 0.0; }
}

template <class Model, class Cons>
DEV bool te_backward(const DevBuf &d, int b, const double *Xc, const double *Uc, const double *Sc, const double *Yc,
                     const double *Gc, const double *VxN, const double *VxxN, double reg, double mu,
                     double &inf_pr, double &inf_comp, double &inf_du, double &step_norm) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  typedef Objective<NX, NU> Obj;
  const ProblemDev *P = d.P;
  const cddp_hip_options &o = P->opt;
  const int N = d.N, pT = P->pT;
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  double xN[NX], hT[kPTMax], lam_prev[kPTMax];
  ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
  term_eq_residual<NX>(P, xN, hT);
  for (int r = 0; r < pT; ++r) { inf_pr = dmax(inf_pr, fabs(hT[r])); lam_prev[r] = d.LamT[(size_t)r * d.Bp + b]; }
  // per-step LQ model (:1143-1245)
  auto lq_model = [&](int t, double *Q, double *q, double *R, double *r, double *Mm, double *A, double *Bm, bool track) {
    double x[NX], u[NU];
    ld<NX * NX>(d.A + GI(t, NX * NX, 0), kLS, A);
    ld<NX * NU>(d.Bm + GI(t, NX * NU, 0), kLS, Bm);
    ld<NX>(Xc + GI(t, NX, 0), kLS, x);
    ld<NU>(Uc + GI(t, NU, 0), kLS, u);
    const double *Qd = P->pool + P->off_Qdt, *Rd = P->pool + P->off_Rdt;
    for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) Q[i * NX + c] = 0.5 * ((2.0 * Qd[i * NX + c]) + (2.0 * Qd[c * NX + i]));
    for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) R[i * NU + c] = 0.5 * ((2.0 * Rd[i * NU + c]) + (2.0 * Rd[c * NU + i]));
    Obj::lx(P, d.xref_traj, t, x, q);
    Obj::lu(P, u, r);
    for (int i = 0; i < NX * NU; ++i) Mm[i] = 0.0;
    if (!P->opt.use_ilqr) {
      // full DDP: the current costate iterate stands in for the value gradient (ipddp_solver.cpp:1160-1178):
      // Q += lambda(i) F_xx[i], M += lambda(i) F_ux[i]^T, R += lambda(i) F_uu[i], then Q, R symmetrised
      if constexpr (Model::kHasHess) {
        double lam[NX], Fxx[NX * NX * NX], Fuu[NX * NU * NU], Fux[NX * NU * NX];
        ld<NX>(d.Lam + (size_t)d.cur[b] * d.planeX + GI(t + 1, NX, 0), kLS, lam);
        bool fin = true;
        for (int i = 0; i < NX; ++i) fin = fin && dfinite(lam[i]);
        if (!fin) for (int i = 0; i < NX; ++i) lam[i] = 0.0;
        Model::hess(P->mp, x, u, Fxx, Fuu, Fux);
        const double dt = P->dt;
        for (int i = 0; i < NX; ++i) {
          for (int e = 0; e < NX * NX; ++e) Q[e] = Q[e] + lam[i] * (dt * Fxx[i * NX * NX + e]);
          for (int a = 0; a < NU; ++a) for (int c = 0; c < NX; ++c) Mm[c * NU + a] = Mm[c * NU + a] + lam[i] * (dt * Fux[i * NU * NX + a * NX + c]);
          for (int e = 0; e < NU * NU; ++e) R[e] = R[e] + lam[i] * (dt * Fuu[i * NU * NU + e]);
        }
        double Qs[NX * NX], Rs[NU * NU];
        for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) Qs[i * NX + c] = 0.5 * (Q[i * NX + c] + Q[c * NX + i]);
        for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) Rs[i * NU + c] = 0.5 * (R[i * NU + c] + R[c * NU + i]);
        for (int i = 0; i < NX * NX; ++i) Q[i] = Qs[i];
        for (int i = 0; i < NU * NU; ++i) R[i] = Rs[i];
      } else if constexpr (HessBlocked<Model>::value) {   // blocked second-order duals (dev_models.hpp): M accumulates from zero, transposed afterwards
        double lam[NX], Mt[NU * NX];
        ld<NX>(d.Lam + (size_t)d.cur[b] * d.planeX + GI(t + 1, NX, 0), kLS, lam);
        bool fin = true;
        for (int i = 0; i < NX; ++i) fin = fin && dfinite(lam[i]);
        if (!fin) for (int i = 0; i < NX; ++i) lam[i] = 0.0;
        for (int i = 0; i < NU * NX; ++i) Mt[i] = 0.0;
        ad_tensor_terms_blocked<typename Model::HessDyn, NX, NU, 4>(P->mp, x, u, lam, P->dt, Model::kHessDiv, Q, Mt, R);
        for (int a = 0; a < NU; ++a) for (int c = 0; c < NX; ++c) Mm[c * NU + a] = Mt[a * NX + c];
        double Qs[NX * NX], Rs[NU * NU];
        for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) Qs[i * NX + c] = 0.5 * (Q[i * NX + c] + Q[c * NX + i]);
        for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) Rs[i * NU + c] = 0.5 * (R[i * NU + c] + R[c * NU + i]);
        for (int i = 0; i < NX * NX; ++i) Q[i] = Qs[i];
        for (int i = 0; i < NU * NU; ++i) R[i] = Rs[i];
      }
    }
    if constexpr (M > 0) {
      double y[MM], s[MM], g[MM], Qyx[MM * NX], Qyu[MM * NU], YS[MM], ypS[MM];
      ld<M>(Yc + GI(t, M, 0), kLS, y);
      ld<M>(Sc + GI(t, M, 0), kLS, s);
      ld<M>(Gc + GI(t, M, 0), kLS, g);
      for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
      for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
      Cons::template jac<NX, NU>(P, x, u, Qyx, Qyu);
      for (int i = 0; i < M; ++i) {
        const double ss = dmax(s[i], s_floor);
        YS[i] = clip_pos(y[i], ss);
        const double rp = g[i] + s[i], rc = y[i] * s[i] - mu;
        const double rhat = y[i] * rp - rc;
        ypS[i] = y[i] + clip_sgn(rhat, ss);
        if (track) { inf_pr = dmax(inf_pr, fabs(rp)); inf_comp = dmax(inf_comp, fabs(rc)); }
      }
      for (int i = 0; i < NX; ++i) { double a = 0.0; for (int rr = 0; rr < M; ++rr) a += Qyx[rr * NX + i] * ypS[rr]; q[i] += a; }
      for (int i = 0; i < NU; ++i) { double a = 0.0; for (int rr = 0; rr < M; ++rr) a += Qyu[rr * NU + i] * ypS[rr]; r[i] += a; }
      double Qn[NX * NX], Rn[NU * NU];
      for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) { double a = 0.0; for (int rr = 0; rr < M; ++rr) a += (Qyx[rr * NX + i] * YS[rr]) * Qyx[rr * NX + c]; Qn[i * NX + c] = Q[i * NX + c] + a; }
      for (int i = 0; i < NU; ++i) for (int c = 0; c < NX; ++c) { double a = 0.0; for (int rr = 0; rr < M; ++rr) a += (Qyu[rr * NU + i] * YS[rr]) * Qyx[rr * NX + c]; Mm[c * NU + i] += a; }
      for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) { double a = 0.0; for (int rr = 0; rr < M; ++rr) a += (Qyu[rr * NU + i] * YS[rr]) * Qyu[rr * NU + c]; Rn[i * NU + c] = R[i * NU + c] + a; }
      for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) Q[i * NX + c] = 0.5 * (Qn[i * NX + c] + Qn[c * NX + i]);
      for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) R[i * NU + c] = 0.5 * (Rn[i * NU + c] + Rn[c * NU + i]);
    }
    for (int i = 0; i < NU; ++i) R[i * NU + i] += reg;
  };
  // The (p+1) sequential LQR sweeps of the reference (solveSequentialLQR :413-476, once per unit terminal direction)
  // share every MATRIX quantity -- LQ model, P recursion, Q_uu factor, K -- and differ only in the gradient
  // recursion p_v.  One sweep therefore carries the matrices once and the p+1 gradient variants side by side
  // (their running values stream through the te_p stack); each variant's arithmetic is unchanged.
  double xT[(kPTMax + 1) * NX];
  {
    double Pm[NX * NX];
    for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) Pm[i * NX + c] = 0.5 * (VxxN[i * NX + c] + VxxN[c * NX + i]);
    for (int v = 0; v <= pT; ++v)
      for (int i = 0; i < NX; ++i) {
        double a = VxN[i];
        double add = 0.0;                                   // (H_T^T lambda_prev)_i, k ascending
        for (int r = 0; r < pT; ++r) add += ((term_eq_col(P, r) == i) ? 1.0 : 0.0) * lam_prev[r];
        a += add;
        if (v > 0 && term_eq_col(P, v - 1) == i) a += 1.0;
        d.te_p[(((size_t)v * (N + 1) + N) * NX + i) * d.Bp + b] = a;
      }
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Pm);
    for (int t = N - 1; t >= 0; --t) {
      double Q[NX * NX], q[NX], R[NU * NU], r[NU], Mm[NX * NU], A[NX * NX], Bm[NX * NU];
      lq_model(t, Q, q, R, r, Mm, A, Bm, true);
      double BtP[NU * NX], Quu[NU * NU], Qux[NU * NX];
      mm_tn<NU, NX, NX>(Bm, Pm, BtP);
      {
        double T1[NU * NU], T2[NU * NU];
        mm_nn<NU, NX, NU>(BtP, Bm, T1);                 // BtP * B
        {   // (B^T * P^T) * B with left-to-right association
          double BtPt[NU * NX];
          for (int i = 0; i < NU; ++i) for (int c = 0; c < NX; ++c) { double a = 0.0; for (int k = 0; k < NX; ++k) a += Bm[k * NU + i] * Pm[c * NX + k]; BtPt[i * NX + c] = a; }
          mm_nn<NU, NX, NU>(BtPt, Bm, T2);
        }
        for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) Quu[i * NU + c] = 0.5 * (((R[i * NU + c] + T1[i * NU + c]) + R[c * NU + i]) + T2[i * NU + c]);
      }
      {
        double T3[NU * NX];
        mm_nn<NU, NX, NX>(BtP, A, T3);
        for (int i = 0; i < NU; ++i) for (int c = 0; c < NX; ++c) Qux[i * NX + c] = T3[i * NX + c] + Mm[c * NU + i];
      }
      LDLTd<NU> f;
      f.compute(Quu, NU);
      if (!f.ok) return false;
      double KK[NU * NX], col[NU];
      for (int c = 0; c < NX; ++c) { for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c]; f.solve(col); for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i]; }
      double KtQ[NX * NU];
      mm_tn<NX, NU, NU>(KK, Quu, KtQ);
      bool fin = true;
      for (int i = 0; i < NU * NX; ++i) fin = fin && dfinite(KK[i]);
      // ---- gradient variants (uses P_{t+1} = Pm before it is overwritten: the drift term is P * 0)
      for (int v = 0; v <= pT; ++v) {
        double pv[NX], drift[NX], Qx[NX], Qu[NU], kk[NU], pn[NX];
        for (int i = 0; i < NX; ++i) pv[i] = d.te_p[(((size_t)v * (N + 1) + t + 1) * NX + i) * d.Bp + b];
        for (int i = 0; i < NX; ++i) { double a = 0.0; for (int k = 0; k < NX; ++k) a += Pm[i * NX + k] * 0.0; drift[i] = pv[i] + a; }
        for (int i = 0; i < NX; ++i) { double a = 0.0; for (int k = 0; k < NX; ++k) a += A[k * NX + i] * drift[k]; Qx[i] = q[i] + a; }
        for (int i = 0; i < NU; ++i) { double a = 0.0; for (int k = 0; k < NX; ++k) a += Bm[k * NU + i] * drift[k]; Qu[i] = r[i] + a; }
        for (int i = 0; i < NU; ++i) col[i] = Qu[i];
        f.solve(col);
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        for (int i = 0; i < NX; ++i) {
          double a1 = 0.0, a2 = 0.0, a3 = 0.0;
          for (int j = 0; j < NU; ++j) { a1 += Qux[j * NX + i] * kk[j]; a2 += KK[j * NX + i] * Qu[j]; a3 += KtQ[i * NU + j] * kk[j]; }
          pn[i] = ((Qx[i] + a1) + a2) + a3;
          fin = fin && dfinite(pn[i]);
        }
        for (int i = 0; i < NU; ++i) fin = fin && dfinite(kk[i]);
        for (int i = 0; i < NU; ++i) d.te_k[(((size_t)v * N + t) * NU + i) * d.Bp + b] = kk[i];
        for (int i = 0; i < NX; ++i) d.te_p[(((size_t)v * (N + 1) + t) * NX + i) * d.Bp + b] = pn[i];
      }
      // P = Q + A^T P A + Q_xu K + K^T Q_ux + K^T Q_uu K
      double T1[NX * NX], AtPA[NX * NX], Pn[NX * NX];
      mm_tn<NX, NX, NX>(A, Pm, T1);
      mm_nn<NX, NX, NX>(T1, A, AtPA);
      for (int i = 0; i < NX; ++i)
        for (int c = 0; c < NX; ++c) {
          double a1 = 0.0, a2 = 0.0, a3 = 0.0;
          for (int j = 0; j < NU; ++j) { a1 += Qux[j * NX + i] * KK[j * NX + c]; a2 += KK[j * NX + i] * Qux[j * NX + c]; a3 += KtQ[i * NU + j] * KK[j * NX + c]; }
          Pn[i * NX + c] = (((Q[i * NX + c] + AtPA[i * NX + c]) + a1) + a2) + a3;
        }
      for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) { Pm[i * NX + c] = 0.5 * (Pn[i * NX + c] + Pn[c * NX + i]); fin = fin && dfinite(Pm[i * NX + c]); }
      if (!fin) return false;
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      st<NX * NX>(d.Vxx + GI(t, NX * NX, 0), kLS, Pm);
    }
  }
  for (int v = 0; v <= pT; ++v) {
    // rolloutLinearPolicy for this variant (dx0 = 0)
    double dx[NX];
    for (int i = 0; i < NX; ++i) dx[i] = 0.0;
    for (int t = 0; t < N; ++t) {
      double KK[NU * NX], A[NX * NX], Bm[NX * NU], du[NU], dxn[NX];
      ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      ld<NX * NX>(d.A + GI(t, NX * NX, 0), kLS, A);
      ld<NX * NU>(d.Bm + GI(t, NX * NU, 0), kLS, Bm);
      for (int i = 0; i < NU; ++i) { double a = 0.0; for (int j = 0; j < NX; ++j) a += KK[i * NX + j] * dx[j]; du[i] = d.te_k[(((size_t)v * N + t) * NU + i) * d.Bp + b] + a; }
      for (int i = 0; i < NX; ++i) { double a = 0.0, c = 0.0; for (int j = 0; j < NX; ++j) a += A[i * NX + j] * dx[j]; for (int j = 0; j < NU; ++j) c += Bm[i * NU + j] * du[j]; dxn[i] = (a + c) + 0.0; }
      for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
    }
    for (int i = 0; i < NX; ++i) xT[v * NX + i] = dx[i];
  }
  // ---- reduced terminal system (:550-617)
  double As[kPTMax * kPTMax], rhs[kPTMax], AtA[kPTMax * kPTMax], Atb[kPTMax];
  for (int r = 0; r < pT; ++r) {
    const int cr = term_eq_col(P, r);
    for (int i = 0; i < pT; ++i) {   // (H_T S)(r,i) = sum_k H_T(r,k) S(k,i)
      double a = 0.0;
      for (int k = 0; k < NX; ++k) a += ((k == cr) ? 1.0 : 0.0) * (xT[(i + 1) * NX + k] - xT[k]);
      As[r * kPTMax + i] = a;
    }
    double hx = 0.0;
    for (int k = 0; k < NX; ++k) hx += ((k == cr) ? 1.0 : 0.0) * xT[k];
    rhs[r] = (-hT[r]) - hx;
  }
  double tr = 0.0;
  for (int i = 0; i < pT; ++i) {
    for (int c = 0; c < pT; ++c) { double a = 0.0; for (int k = 0; k < pT; ++k) a += As[k * kPTMax + i] * As[k * kPTMax + c]; AtA[i * kPTMax + c] = a; }
    double a = 0.0; for (int k = 0; k < pT; ++k) a += As[k * kPTMax + i] * rhs[k]; Atb[i] = a;
  }
  for (int i = 0; i < pT; ++i) tr += AtA[i * kPTMax + i];
  const double trace_term = (tr > 1.0 ? tr / (pT > 1 ? pT : 1) : 1.0);
  const double base_floor = dmax(1e-10, o.ipddp_jacobian_regularization_value * solver_pow(dmax(mu, 0.0), o.ipddp_jacobian_regularization_exponent));
  const double regv = dmax(base_floor, 1e-6 * trace_term);
  double smax, smin;
  singular_minmax<kPTMax>(As, pT, smax, smin);
  const double svd_reg = dmax(1e-8 * smax - smin, 0.0);
  const double reg_base = dmax(regv, svd_reg);
  double rn = 0.0; for (int r = 0; r < pT; ++r) rn += rhs[r] * rhs[r];
  const double cap = 100.0 * (1.0 + sqrt(rn));
  const double scales[5] = {1.0, 10.0, 100.0, 1e3, 1e4};
  double best[kPTMax]; for (int i = 0; i < pT; ++i) best[i] = 0.0;
  double best_res = INFINITY; bool found = false;
  for (int sc = 0; sc < 5; ++sc) {
    const double reg_i = dmax(reg_base * scales[sc], 1e-12);
    double Sh[kPTMax * kPTMax];
    for (int i = 0; i < pT; ++i) for (int c = 0; c < pT; ++c) Sh[i * kPTMax + c] = AtA[i * kPTMax + c] + reg_i * ((i == c) ? 1.0 : 0.0);
    LDLTd<kPTMax> f;
    f.compute(Sh, pT);
    if (!f.ok) continue;
    double lam[kPTMax]; for (int i = 0; i < pT; ++i) lam[i] = Atb[i];
    f.solve(lam);
    bool fin = true; double ln = 0.0;
    for (int i = 0; i < pT; ++i) { fin = fin && dfinite(lam[i]); ln += lam[i] * lam[i]; }
    if (!fin) continue;
    ln = sqrt(ln);
    if (ln > cap) { const double f2 = cap / dmax(ln, 1e-12); for (int i = 0; i < pT; ++i) lam[i] = lam[i] * f2; }
    double res = 0.0;
    for (int r = 0; r < pT; ++r) { double a = 0.0; for (int i = 0; i < pT; ++i) a += As[r * kPTMax + i] * lam[i]; const double e = a - rhs[r]; res += e * e; }
    res = sqrt(res);
    if (!dfinite(res)) continue;
    if (!found || res < best_res) { for (int i = 0; i < pT; ++i) best[i] = lam[i]; best_res = res; found = true; }
  }
  if (!found) for (int i = 0; i < pT; ++i) best[i] = 0.0;
  for (int i = 0; i < pT; ++i) d.dLamT[(size_t)i * d.Bp + b] = best[i];   // dLambda_T_eq_ = lambda_delta (:1259)
  // ---- recombination (:619-634), inf_du / step_norm (:1260-1266)
  for (int t = 0; t <= N; ++t) {
    if (t < N) {
      double ko[NU];
      for (int i = 0; i < NU; ++i) ko[i] = d.te_k[(((size_t)0 * N + t) * NU + i) * d.Bp + b];
      for (int v = 0; v < pT; ++v)
        for (int i = 0; i < NU; ++i) ko[i] += best[v] * (d.te_k[(((size_t)(v + 1) * N + t) * NU + i) * d.Bp + b] - d.te_k[(((size_t)0 * N + t) * NU + i) * d.Bp + b]);
      st<NU>(d.k + GI(t, NU, 0), kLS, ko);
      for (int i = 0; i < NU; ++i) step_norm = dmax(step_norm, fabs(ko[i]));
    }
    double po[NX];
    for (int i = 0; i < NX; ++i) po[i] = d.te_p[(((size_t)0 * (N + 1) + t) * NX + i) * d.Bp + b];
    for (int v = 0; v < pT; ++v)
      for (int i = 0; i < NX; ++i) po[i] += best[v] * (d.te_p[(((size_t)(v + 1) * (N + 1) + t) * NX + i) * d.Bp + b] - d.te_p[(((size_t)0 * (N + 1) + t) * NX + i) * d.Bp + b]);
    st<NX>(d.Vx + GI(t, NX, 0), kLS, po);
  }
  for (int t = 0; t < N; ++t) {
    double Q[NX * NX], q[NX], R[NU * NU], r[NU], Mm[NX * NU], A[NX * NX], Bm[NX * NU], pn[NX];
    lq_model(t, Q, q, R, r, Mm, A, Bm, false);
    ld<NX>(d.Vx + GI(t + 1, NX, 0), kLS, pn);
    for (int i = 0; i < NU; ++i) { double a = 0.0; for (int k = 0; k < NX; ++k) a += Bm[k * NU + i] * pn[k]; inf_du = dmax(inf_du, fabs(r[i] + a)); }
    if constexpr (M > 0) {   // slack / dual gains with the final k, K (:1270-1312)
      double x[NX], y[MM], s[MM], g[MM], Qyx[MM * NX], Qyu[MM * NU], kk[NU], KK[NU * NX];
      ld<NX>(Xc + GI(t, NX, 0), kLS, x);
      ld<M>(Yc + GI(t, M, 0), kLS, y); ld<M>(Sc + GI(t, M, 0), kLS, s); ld<M>(Gc + GI(t, M, 0), kLS, g);
      ld<NU>(d.k + GI(t, NU, 0), kLS, kk); ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
      for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
      double uj[NU];
      if constexpr (Cons::NEEDS_U) ld<NU>(Uc + GI(t, NU, 0), kLS, uj);
      Cons::template jac<NX, NU>(P, x, uj, Qyx, Qyu);
      double ky[MM], ksv[MM], Ky[MM * NX], Ksm[MM * NX];
      for (int rr = 0; rr < M; ++rr) {
        const double ss = dmax(s[rr], s_floor);
        const double YSr = clip_pos(y[rr], ss);
        const double rp = g[rr] + s[rr], rc = y[rr] * s[rr] - mu;
        const double rhat = y[rr] * rp - rc;
        double temp = 0.0; for (int i = 0; i < NU; ++i) temp += Qyu[rr * NU + i] * kk[i];
        ky[rr] = clip_sgn(rhat + y[rr] * temp, ss);
        ksv[rr] = (-rp) - temp;
        for (int c = 0; c < NX; ++c) {
          double s2 = 0.0; for (int i = 0; i < NU; ++i) s2 += Qyu[rr * NU + i] * KK[i * NX + c];
          Ky[rr * NX + c] = dmin(dmax(YSr * (Qyx[rr * NX + c] + s2), -kMaxBarrierRatio), kMaxBarrierRatio);
          Ksm[rr * NX + c] = (-Qyx[rr * NX + c]) - s2;
        }
      }
      st<M>(d.ky + GI(t, M, 0), kLS, ky); st<M>(d.ks + GI(t, M, 0), kLS, ksv);
      st<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky); st<M * NX>(d.Ks + GI(t, M * NX, 0), kLS, Ksm);
    }
  }
  return true;
}

// computeScaledDualInfeasibility (ipddp_solver.cpp:2725-2776): max(v, max_t |G_x[t]^T y[t]|_inf) with G_x from the
// X slot of the last backward pass (`xslot`) and Y from the current slot; v = the raw inf_du.
template <class Model, class Cons>
DEV double scaled_inf_du_v(const DevBuf &d, int b, int xslot, double v) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  const ProblemDev *P = d.P;
  if (!P->opt.ipddp_check_state_stationarity || M == 0) return v;
  if (!Cons::HAS_X) return dmax(v, 0.0);   // G_x == 0: every |G_x^T y| entry is exactly 0
  const double *Xs = d.X + (size_t)xslot * d.planeX;
  const double *Yc = d.Y + (size_t)d.cur[b] * d.planeM;
  double ss = 0.0;
  for (int t = 0; t < d.N; ++t) {
    double x[NX], y[MM], Gx[MM * NX], Gu[MM * NU];
    ld<NX>(Xs + GI(t, NX, 0), kLS, x);
    ld<M>(Yc + GI(t, M, 0), kLS, y);
    for (int i = 0; i < M * NX; ++i) Gx[i] = 0.0;
    for (int i = 0; i < M * NU; ++i) Gu[i] = 0.0;
    double uz[NU];
    for (int i = 0; i < NU; ++i) uz[i] = 0.0;   // only G_x is read below; rows whose Jacobian depends on u have G_x = 0
    Cons::template jac<NX, NU>(P, x, uz, Gx, Gu);
    for (int c = 0; c < Cons::NSEG; ++c) {
      const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
      for (int j = 0; j < NX; ++j) {
        double s = 0.0;
        for (int i = 0; i < dim; ++i) s += Gx[(off + i) * NX + j] * y[off + i];
        ss = dmax(ss, fabs(s));
      }
    }
  }
  return dmax(v, ss);
}
template <class Model, class Cons>
DEV double scaled_inf_du(const DevBuf &d, int b, int xslot) { return scaled_inf_du_v<Model, Cons>(d, b, xslot, d.inf_du[b]); }

// ================================================================================ K2 (IPDDP)
// Unconstrained branch (ipddp_solver.cpp:1048-1118) when Cons::M == 0, path-constraint branch
// (:1355-1568) otherwise; followed by the linear-policy rollout (:1511-1532) fused with
// computeMaxStepSizes (:2939-2988).
template <class Model, class Cons, bool TERM = false>
__global__ __launch_bounds__(64) void k_backward_ipddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  const double *Sc = d.S + (size_t)cur * d.planeM;
  const double *Yc = d.Y + (size_t)cur * d.planeM;
  const double *Gc = d.G + (size_t)cur * d.planeM;
  if (count_iter) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  const int mT = TERM ? P->mT : 0, pT = TERM ? P->pT : 0;
  const bool nobar = (M == 0) && mT == 0;
  const bool uncon = nobar && pT == 0;        // the unconstrained branch (ipddp_solver.cpp:1048)
  const bool path_m0 = (M == 0) && !uncon;    // path / terminal-inequality branch with no path rows
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  for (;;) {
    ++nb;
    double xN[NX], Vx[NX], Vxx[NX * NX];
    ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
    Obj::final_grad(P, xN, Vx);
    const double *Qf = P->pool + P->off_Qf;
    {  // V_xx = symmetrize(2 Qf)
      double H2[NX * NX];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) H2[i] = 2.0 * Qf[i];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (H2[i * NX + c] + H2[c * NX + i]);
    }
    dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0;
    if constexpr (TERM) {
      if (mT > 0) {   // terminal-inequality barrier terms folded into V_x, V_xx (ipddp_solver.cpp:1000-1031)
        double gT[kMTMax];
        term_ineq_eval<NX>(P, xN, gT);
        for (int c = 0; c < P->n_term; ++c) {
          const TermDev &td = P->terms[c];
          if (td.kind != CDDP_HIP_TERM_INEQUALITY) continue;
          double sig[kMTMax], bg[kMTMax];
          for (int r = 0; r < td.dim; ++r) {
            const int j = td.offset + r;
            const double sT = d.ST[(size_t)j * d.Bp + b], yT = d.YT[(size_t)j * d.Bp + b];
            d.GT[(size_t)j * d.Bp + b] = gT[j];
            const double s_safe = dmax(sT, s_floor);
            const double y_safe = dmax(yT, 1e-10);
            sig[r] = clip_pos(y_safe, s_safe);
            bg[r] = y_safe + clip_sgn(y_safe * gT[j] + mu, s_safe);
            inf_pr = dmax(inf_pr, fabs(gT[j] + sT));
            inf_comp = dmax(inf_comp, fabs(yT * sT - mu));
          }
          const double *Am = P->pool + td.off_A;
          for (int i = 0; i < NX; ++i) { double a = 0.0; for (int r = 0; r < td.dim; ++r) a += Am[r * NX + i] * bg[r]; Vx[i] += a; }
          double Vn[NX * NX];
          for (int i = 0; i < NX; ++i)
            for (int c2 = 0; c2 < NX; ++c2) { double a = 0.0; for (int r = 0; r < td.dim; ++r) a += (Am[r * NX + i] * sig[r]) * Am[r * NX + c2]; Vn[i * NX + c2] = Vxx[i * NX + c2] + a; }
          for (int i = 0; i < NX; ++i) for (int c2 = 0; c2 < NX; ++c2) Vxx[i * NX + c2] = 0.5 * (Vn[i * NX + c2] + Vn[c2 * NX + i]);
        }
      }
      if (pT > 0) {   // terminal-equality reduced LQR branch (ipddp_solver.cpp:1120-1353)
        const bool okte = te_backward<Model, Cons>(d, b, Xc, Uc, Sc, Yc, Gc, Vx, Vxx, reg, mu, inf_pr, inf_comp, inf_du, step_norm);
        if (okte) { ok = true; break; }
        if (force == 2) break;
        reg = reg_increase(o, reg);
        if (reg >= o.reg_max_value) break;
        continue;
      }
    }
    st<NX>(d.Vx + GI(N, NX, 0), kLS, Vx);
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Vxx);
    bool fail = false;
    // software pipeline: the record of step t-1 is in flight while step t is computed (the only
    // latency hiding available along the serial chain with one wave per SIMD)
    struct StepIn { double A[NX * NX], Bm[NX * NU], x[NX], u[NU], y[MM], s[MM], g[MM]; };
    auto load_step = [&](int tt, StepIn &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.x);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.u);
      if constexpr (M > 0) {
        ld<M>(Yc + GI(tt, M, 0), kLS, r.y);
        ld<M>(Sc + GI(tt, M, 0), kLS, r.s);
        ld<M>(Gc + GI(tt, M, 0), kLS, r.g);
      }
    };
    StepIn nxt;
    load_step(N - 1, nxt);
    for (int t = N - 1; t >= 0; --t) {
      StepIn cs = nxt;
      if (t > 0) load_step(t - 1, nxt);
      PIPELINE_FENCE();
      double (&A)[NX * NX] = cs.A; double (&Bm)[NX * NU] = cs.Bm; double (&x)[NX] = cs.x; double (&u)[NU] = cs.u;
      double (&y)[MM] = cs.y; double (&s)[MM] = cs.s; double (&g)[MM] = cs.g;
      double Qyx[MM * NX], Qyu[MM * NU];
      if constexpr (M > 0) {
#pragma unroll
        for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
        for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
        Cons::template jac<NX, NU>(P, x, u, Qyx, Qyu);
      }
      double Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU];
      Obj::lx(P, xrt, t, x, Qx);
      Obj::lu(P, u, Qu);
      // Q_x = l_x + Q_yx^T y + A^T V_x ; Q_u = l_u + Q_yu^T y + B^T V_x
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double acc = Qx[i];
        if constexpr (M > 0) { double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += Qyx[r * NX + i] * y[r];
          acc = acc + s1; }
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += A[k * NX + i] * Vx[k];
        Qx[i] = acc + s2;
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double acc = Qu[i];
        if constexpr (M > 0) { double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += Qyu[r * NU + i] * y[r];
          acc = acc + s1; }
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Bm[k * NU + i] * Vx[k];
        Qu[i] = acc + s2;
      }
      q_blocks<NX, NU>(P, A, Bm, Vx, Vxx, Qxx, Qux, Quu);
      if (!o.use_ilqr) ddp_tensor_terms<Model>(P, x, u, Vx, Qxx, Qux, Quu);

      double kk[NU], KK[NU * NX];
      double YS[MM], rp[MM], rc[MM], rhat[MM], Sir[MM], s_safe[MM];
      if constexpr (M == 0) {
        // ---- unconstrained: regularisation stays in Q_uu (ipddp_solver.cpp:1084-1107).  With terminal
        // inequalities only (path_m0) the path branch runs with zero path rows: same factor, but the
        // V update keeps the un-symmetrised, un-regularised Q_uu (:1424-1426, 1497-1500).
        double Qs[NU * NU];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]);
#pragma unroll
        for (int i = 0; i < NU; ++i) Qs[i * NU + i] += reg;
        if (!path_m0) {
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) Quu[i] = Qs[i];
        }
        if (NU == 1) {
          kk[0] = -ldlt1_solve(Qs[0], Qu[0]);
#pragma unroll
          for (int c = 0; c < NX; ++c) KK[c] = -ldlt1_solve(Qs[0], Qux[c]);
        } else {
          LDLTd<NU> f;
          f.compute(Qs, NU);
          if (!f.ok) { fail = true; break; }
          double col[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qu[i];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) kk[i] = -col[i];
          for (int c = 0; c < NX; ++c) {
#pragma unroll
            for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c];
            f.solve(col);
#pragma unroll
            for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
          }
        }
      } else {
        // ---- KKT condensation (ipddp_solver.cpp:1410-1492)
#pragma unroll
        for (int i = 0; i < M; ++i) {
          s_safe[i] = dmax(s[i], s_floor);
          YS[i] = clip_pos(y[i], s_safe[i]);
          rp[i] = g[i] + s[i];
          rc[i] = y[i] * s[i] - mu;
          rhat[i] = y[i] * rp[i] - rc[i];
          Sir[i] = clip_sgn(rhat[i], s_safe[i]);
        }
        // W = Q_yu^T YSinv  (NU x M)
        double W[NU * MM];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int r = 0; r < M; ++r) W[i * M + r] = Qyu[r * NU + i] * YS[r];
        double WQyu[NU * NU], WQyx[NU * NX];
        mm_nn<NU, M, NU>(W, Qyu, WQyu);
        mm_nn<NU, M, NX>(W, Qyx, WQyx);
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int c = 0; c < NU; ++c) Qr[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]) + WQyu[i * NU + c];
#pragma unroll
        for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
        double QyuSir[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += Qyu[r * NU + i] * Sir[r];
          QyuSir[i] = s1; }
        if (NU == 1) {
          kk[0] = -ldlt1_solve(Qr[0], Qu[0] + QyuSir[0]);
#pragma unroll
          for (int c = 0; c < NX; ++c) KK[c] = -ldlt1_solve(Qr[0], Qux[c] + WQyx[c]);
        } else {
          LDLTd<NU> f;
          f.compute(Qr, NU);
          if (!f.ok) { fail = true; break; }
          double col[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qu[i] + QyuSir[i];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) kk[i] = -col[i];
          for (int c = 0; c < NX; ++c) {
#pragma unroll
            for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c] + WQyx[i * NX + c];
            f.solve(col);
#pragma unroll
            for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
          }
        }
        // slack / dual gains (:1458-1472)
        double temp[MM], ky[MM], ksv[MM], Ky[MM * NX], Ksm[MM * NX];
#pragma unroll
        for (int r = 0; r < M; ++r) {
          double s1 = 0.0;
#pragma unroll
          for (int i = 0; i < NU; ++i) s1 += Qyu[r * NU + i] * kk[i];
          temp[r] = s1;
          ky[r] = clip_sgn(rhat[r] + y[r] * temp[r], s_safe[r]);
          ksv[r] = (-rp[r]) - temp[r];
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            double s2 = 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) s2 += Qyu[r * NU + i] * KK[i * NX + c];
            double inner = Qyx[r * NX + c] + s2;
            Ky[r * NX + c] = dmin(dmax(YS[r] * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
            Ksm[r * NX + c] = (-Qyx[r * NX + c]) - s2;
          }
        }
        st<M>(d.ky + GI(t, M, 0), kLS, ky);
        st<M>(d.ks + GI(t, M, 0), kLS, ksv);
        st<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky);
        st<M * NX>(d.Ks + GI(t, M * NX, 0), kLS, Ksm);
        // condensed, un-regularised Q blocks (:1488-1492)
#pragma unroll
        for (int i = 0; i < NU; ++i) Qu[i] += QyuSir[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += Qyx[r * NX + i] * Sir[r];
          Qx[i] += s1; }
        double Wx[NX * MM], WxQyx[NX * NX];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
          for (int r = 0; r < M; ++r) Wx[i * M + r] = Qyx[r * NX + i] * YS[r];
        mm_nn<NX, M, NX>(Wx, Qyx, WxQyx);
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) Qxx[i] += WxQyx[i];
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) Qux[i] += WQyx[i];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Quu[i] += WQyu[i];
#pragma unroll
        for (int r = 0; r < M; ++r) { inf_pr = dmax(inf_pr, fabs(rp[r])); inf_comp = dmax(inf_comp, fabs(rc[r])); }
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      // dV, V_x, V_xx (:1494-1503 / :1098-1107)
      double Quuk[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) s1 += Quu[i * NU + j] * kk[j];
        Quuk[i] = s1; }
      { double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) { s0 += kk[i] * Qu[i]; s1 += kk[i] * Quuk[i]; }
        dV0 += s0; dV1 += 0.5 * s1; }
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
      for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(Qu[i])); step_norm = dmax(step_norm, fabs(kk[i])); }
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
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du; d.step_norm[b] = step_norm;
    d.inf_pr[b] = uncon ? 0.0 : inf_pr;
    d.inf_comp[b] = uncon ? 0.0 : inf_comp;
    // ---- linear-policy rollout dX (dx0 = 0) -> dS, dY -> fraction-to-boundary caps
    double apr = 1.0, adu = 1.0;
    lin_rollout_caps<Model, Cons, TERM>(d, P, b, Sc, Yc, Xc, mu, apr, adu);
    d.apr_max[b] = apr; d.adu_max[b] = adu;
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  // checkEarlyConvergence (ipddp_solver.cpp:925-958)
  bool conv;
  const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
  if (nobar) conv = (d.inf_pr[b] < o.tolerance && sdu_early < o.tolerance);
  else {
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
  }
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, mu); return; }
  d.phase[b] = PH_FWD1;
}

// slot that trial `a` of trajectory b writes to: the a-th slot different from the current one
DEV int trial_slot(int cur, int a) { return (a < cur) ? a : a + 1; }

// ================================================================================ K4 (CLDDP)
template <class Model>
__global__ __launch_bounds__(64) void k_forward_clddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int a0, int na, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  // alpha on blockIdx.y: one wavefront = the 64 trajectories of one wave tile at ONE alpha (coalesced 512-B
  // rows, uniform tile base address).  The alternative -- the trials of one trajectory in adjacent lanes -- was
  // measured 1.37x slower at C2 (11-way scattered trial stores, no whole-wave early exit); see DESIGN.md.
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int a = a0 + blockIdx.y;
  (void)na;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const int slot = trial_slot(cur, a);
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  double *Xn = d.X + (size_t)slot * d.planeX;
  double *Un = d.U + (size_t)slot * d.planeU;
  const double alpha = P->alphas[a];
  const int box = P->clddp_box;
  atomicAdd(d.launched, 1ull);
  d.t_steps[(size_t)a * d.Bp + b] = N;   // the CLDDP rollout is never abandoned (clddp_solver.cpp:215-262)
  double x[NX];
  ld<NX>(Xc + GI(0, NX, 0), kLS, x);     // X_[0] == initial state
  st<NX>(Xn + GI(0, NX, 0), kLS, x);
  double J = 0.0;
  DynCtx dc;                       // loop-invariant plant / integrator constants in scalar registers (as the IPDDP producer keeps them)
  dc.load(P->integrator, P->dt, P->mp);
  typename Obj::Ctx oc;            // Q dt | R dt | goal hoisted for small plants, on the pool otherwise
  Obj::load(P, oc);
  // One step of look-ahead on the (x_old, u_old, k, K) record of the current iterate, ping-pong register sets with the loop unrolled
  // by two (the idiom of the IPDDP rollouts): the record of step t + 1 does not depend on the trial, so its
  // HBM latency hides behind the integrator chain of step t (round 3: loads at the top of their own step, 238 us per launch).
  // Records above 32 doubles (nx >= 12) keep the single set: two of them would not fit beside the plant's registers.
  constexpr int REC = NX + 2 * NU + NU * NX;
  constexpr bool kPF = REC <= 32;
  struct Rec { double xo[NX], uo[NU], kk[NU], KK[NU * NX]; };
  auto fetch = [&](int tt, Rec &r) {
    ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
    ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
    ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
    ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
  };
  double lo[NU], hi[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) { lo[i] = box >= 0 ? P->pool[P->cons[box].off_lower + i] : 0.0; hi[i] = box >= 0 ? P->pool[P->cons[box].off_upper + i] : 0.0; }
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
      if (box >= 0) u[i] = dmin(dmax(u[i], lo[i]), hi[i]);
    }
    J += Obj::running_cost(oc, xrt, t, x, u);
    double xn[NX];
    Stepper<Model>::step(dc, x, u, xn);
    st<NU>(Un + GI(t, NU, 0), kLS, u);
    st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  };
  if constexpr (kPF) {
    Rec ra, rb;
    fetch(0, ra);
    {   // prime the VMEM queue with one step's store pattern (rows rewritten by step 0), so that the loop-entry state the waitcnt
        // pass joins with the back edge ends in stores, not loads: otherwise every step waits for vmcnt(0), i.e. for its own stores
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
  J += Obj::terminal_cost(P, x);
  const double dJ = d.cost[b] - J;
  const double expected = -alpha * (d.dV0[b] + 0.5 * alpha * d.dV1[b]);
  const double ratio = expected > 0.0 ? dJ / expected : sign_of_reduction(dJ);
  const size_t ti = (size_t)a * d.Bp + b;
  d.t_success[ti] = (ratio > o.filter_armijo_constant) ? 1 : 0;
  d.t_cost[ti] = J; d.t_merit[ti] = J; d.t_theta[ti] = 0.0; d.t_inf_pr[ti] = 0.0; d.t_inf_comp[ti] = 0.0;
  d.t_apr[ti] = alpha; d.t_adu[ti] = 1.0;
}

// Sums of computeTheta / computeBarrierMerit / computePrimalAndComplementarity over one slot, in the
// reference's order (constraint-major, then t) -- ipddp_solver.cpp:2778-2937.
template <class Cons>
DEV void ip_reductions(const DevBuf &d, int b, int N, const double *S, const double *Y, const double *G,
                       double mu, double cost0, bool l2, double &phi, double &theta, double &inf_pr, double &inf_comp,
                       const TermState *ts = nullptr, int mT = 0, int pT = 0) {
  constexpr int M = Cons::M;
  double total = 0.0, max_entry = 0.0, ipr = 0.0, icomp = 0.0, mer = cost0;
  for (int c = 0; c < Cons::NSEG; ++c) {
    const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
    for (int t = 0; t < N; ++t) {
      double n1 = 0.0, ninf = 0.0;
      for (int i = 0; i < dim; ++i) {
        const size_t j = GI(t, M, off + i);
        const double r = G[j] + S[j];
        n1 += l2 ? r * r : fabs(r);
        ninf = dmax(ninf, fabs(r));
        icomp = dmax(icomp, fabs(Y[j] * S[j] - mu));
      }
      total += n1;
      max_entry = dmax(max_entry, ninf);
      ipr = dmax(ipr, ninf);
    }
  }
  for (int c = 0; c < Cons::NSEG; ++c) {
    const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
    for (int t = 0; t < N; ++t) {
      double ls = 0.0;
      for (int i = 0; i < dim; ++i) ls += solver_log(dmax(S[GI(t, M, off + i)], kEpsSlack));
      mer -= mu * ls;
    }
  }
  if (ts) term_reductions(d.P, *ts, mT, pT, mu, l2, total, max_entry, mer, ipr, icomp);
  const double th = l2 ? sqrt(total) : total;
  theta = dmax(th, max_entry);
  phi = mer; inf_pr = ipr; inf_comp = icomp;
}

// ================================================================================ K4 (IPDDP)
template <class Model, class Cons, bool TERM = false>
__global__ __launch_bounds__(64) void k_forward_ipddp(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int a0, int na, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  typedef Objective<NX, NU> Obj;
  // alpha on blockIdx.y: one wavefront = the 64 trajectories of one wave tile at ONE alpha (coalesced 512-B
  // rows, uniform tile base address).  The alternative -- the trials of one trajectory in adjacent lanes -- was
  // measured 1.37x slower at C2 (11-way scattered trial stores, no whole-wave early exit); see DESIGN.md.
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int a = a0 + blockIdx.y;
  (void)na;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const int slot = trial_slot(cur, a);
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  const double *Sc = d.S + (size_t)cur * d.planeM;
  const double *Yc = d.Y + (size_t)cur * d.planeM;

  double *Xn = d.X + (size_t)slot * d.planeX;
  double *Un = d.U + (size_t)slot * d.planeU;
  double *Sn = d.S + (size_t)slot * d.planeM;
  double *Yn = d.Y + (size_t)slot * d.planeM;
  double *Gn = d.G + (size_t)slot * d.planeM;

  const double alpha = P->alphas[a];
  const double mu = d.mu[b];
  const int mT = TERM ? P->mT : 0, pT = TERM ? P->pT : 0;
  const double tau = (M == 0 && mT == 0) ? 1.0 : dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  TermState tn;   // terminal variables of the trial (TERM only)
  const double a_pr = dmin(alpha, d.apr_max[b]);
  const double a_du = dmin(alpha, d.adu_max[b]);
  const size_t ti = (size_t)a * d.Bp + b;
  atomicAdd(d.launched, 1ull);
  d.t_apr[ti] = a_pr; d.t_adu[ti] = a_du;
  d.t_success[ti] = 0;
  d.t_steps[ti] = N;     // overwritten by the step index where the trial is abandoned
  d.t_cost[ti] = d.cost[b]; d.t_merit[ti] = d.phi[b]; d.t_theta[ti] = d.theta[b];
  d.t_inf_pr[ti] = 0.0; d.t_inf_comp[ti] = 0.0;
  double x[NX];
  ld<NX>(Xc + GI(0, NX, 0), kLS, x);
  st<NX>(Xn + GI(0, NX, 0), kLS, x);
  double cost_new = 0.0;
  double ev_total0 = 0.0, ev_max = 0.0, ev_icomp = 0.0;
  const bool l2norm = o.ipddp_theta_norm_l2 != 0;
  // software pipeline: record of step t+1 (old iterate, gains, value expansion) in flight during step t
  struct StepIn {
    double xo[NX], uo[NU], kk[NU], KK[NU * NX];
    double s[MM], y[MM], ksv[MM], ky[MM], Ksm[MM * NX], Ky[MM * NX];
  };
  auto load_step = [&](int tt, StepIn &r) {
    ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
    // (the costate trial is evaluated by k_costate, kernels_lean.hpp, for the trials that survive the rollout)
    if (tt < N) {
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
      ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
      ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
      if constexpr (M > 0) {
        ld<M>(Sc + GI(tt, M, 0), kLS, r.s);
        ld<M>(Yc + GI(tt, M, 0), kLS, r.y);
        ld<M>(d.ks + GI(tt, M, 0), kLS, r.ksv);
        ld<M>(d.ky + GI(tt, M, 0), kLS, r.ky);
        ld<M * NX>(d.Ks + GI(tt, M * NX, 0), kLS, r.Ksm);
        ld<M * NX>(d.Ky + GI(tt, M * NX, 0), kLS, r.Ky);
      }
    }
  };
  // large records (nx >= 12: K_s, K_y alone are 2 m nx rows) are fetched at the top of their own step into one register
  // set -- a second copy would only add scratch traffic
  constexpr bool kPing = sizeof(StepIn) <= 48 * sizeof(double);
  StepIn nxt;
  if constexpr (kPing) load_step(0, nxt);
  for (int t = 0; t <= N; ++t) {
    StepIn held;
    if constexpr (kPing) { held = nxt; if (t < N) load_step(t + 1, nxt); }
    else load_step(t, nxt);
    StepIn &cs = kPing ? held : nxt;
    PIPELINE_FENCE();
    double dx[NX];
    bool finite = true;
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[i] = x[i] - cs.xo[i];
    if constexpr (TERM) {
      if (t == N) {   // terminal slack / dual / multiplier trial (ipddp_solver.cpp:1667-1723)
        TermState to;
        term_load(d, b, mT, pT, to);
        double g0[kMTMax];
        term_ineq_eval<NX>(P, cs.xo, g0);           // residual at the CURRENT x_N
        const double fl0 = dmax(mu * 1e-3, kEpsSlack);
        for (int i = 0; i < mT; ++i) {
          const double *row = term_ineq_row(P, i);
          const double k_s_T = -(g0[i] + to.s[i]);
          double crow[NX];
          for (int j = 0; j < NX; ++j) crow[j] = -row[j];
          tn.s[i] = affine_2r<NX>(to.s[i], a_pr, k_s_T, crow, dx);
          const double s_safe = dmax(to.s[i], fl0);
          const double r_d = to.y[i] * to.s[i] - mu;
          const double dual_ratio = clip_pos(to.y[i], s_safe);
          const double k_y = clip_sgn(-r_d - to.y[i] * k_s_T, s_safe);
          for (int j = 0; j < NX; ++j) crow[j] = -(dual_ratio * (-row[j]));
          tn.y[i] = affine_2r<NX>(to.y[i], a_du, k_y, crow, dx);
          const double s_floor = dmax((1.0 - tau) * to.s[i], fl0);
          if (tn.s[i] < s_floor || tn.y[i] < (1.0 - tau) * to.y[i]) { d.t_steps[ti] = t < N ? t : N; return; }
          if (!dfinite(tn.s[i]) || !dfinite(tn.y[i])) { d.t_steps[ti] = t < N ? t : N; return; }
        }
        for (int i = 0; i < pT; ++i) {
          tn.lam[i] = to.lam[i] + a_pr * d.dLamT[(size_t)i * d.Bp + b];
          if (!dfinite(tn.lam[i])) { d.t_steps[ti] = t < N ? t : N; return; }
        }
      }
    }
    if (t == N) break;
    double sn[MM], yn[MM];
    if constexpr (M > 0) {
      bool feas = true;
#pragma unroll
      for (int r = 0; r < M; ++r) {
        sn[r] = affine_2r<NX>(cs.s[r], a_pr, cs.ksv[r], cs.Ksm + r * NX, dx);
        yn[r] = affine_2r<NX>(cs.y[r], a_du, cs.ky[r], cs.Ky + r * NX, dx);
        if (sn[r] < (1.0 - tau) * cs.s[r] || yn[r] < (1.0 - tau) * cs.y[r]) feas = false;
        if (!dfinite(sn[r]) || !dfinite(yn[r])) feas = false;
      }
      if (!feas) { d.t_steps[ti] = t < N ? t : N; return; }
      st<M>(Sn + GI(t, M, 0), kLS, sn);
      st<M>(Yn + GI(t, M, 0), kLS, yn);
    }
    double u[NU], xn[NX];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      double s1 = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s1 += cs.KK[i * NX + j] * dx[j];
      u[i] = (cs.uo[i] + a_pr * cs.kk[i]) + s1;
      finite = finite && dfinite(u[i]);
    }
    Stepper<Model>::step(P->integrator, P->dt, P->mp, x, u, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) finite = finite && dfinite(xn[i]);
    if (!finite) { d.t_steps[ti] = t < N ? t : N; return; }
    cost_new += Obj::running_cost(P, xrt, t, x, u);
    if constexpr (M > 0) {
      double g[MM];
      Cons::template eval<NX, NU>(P, x, u, g);
      st<M>(Gn + GI(t, M, 0), kLS, g);
      if constexpr (!TERM) {
        // Per-step terms of computeTheta / computeBarrierMerit / computePrimalAndComplementarity
        // (ipddp_solver.cpp:2778-2937).  The reference sums constraint-major, then t: the first constraint
        // object's |g+s| terms can therefore be accumulated right here in t order; the other objects' terms
        // and every log-barrier term (whose chain starts from the still unknown cost_new) are parked in the
        // ev scratch and added after the rollout in the reference's order -- no second pass over S/Y/G.
        double *ev = d.ev + GI((size_t)a * N + t, 2 * Cons::NSEG, 0);
#pragma unroll
        for (int c = 0; c < Cons::NSEG; ++c) {
          const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
          double n1 = 0.0, ninf = 0.0, ls = 0.0;
          for (int i = 0; i < dim; ++i) {
            const double r = g[off + i] + sn[off + i];
            n1 += l2norm ? r * r : fabs(r);
            ninf = dmax(ninf, fabs(r));
            ev_icomp = dmax(ev_icomp, fabs(yn[off + i] * sn[off + i] - mu));
            ls += solver_log(dmax(sn[off + i], kEpsSlack));
          }
          ev_max = dmax(ev_max, ninf);
          if (c == 0) ev_total0 += n1; else ev[(size_t)(Cons::NSEG + c) * kLS] = n1;
          ev[(size_t)c * kLS] = ls;
        }
      }
    }
    st<NU>(Un + GI(t, NU, 0), kLS, u);
    st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  }
  cost_new += Obj::terminal_cost(P, x);
  double phi_new = cost_new, theta_new = 0.0, ipr = 0.0, icomp = 0.0;
  if constexpr (TERM) {
    term_ineq_eval<NX>(P, x, tn.g);      // G_T_new, h_T_new at the trial's x_N (:1750-1760)
    term_eq_residual<NX>(P, x, tn.h);
    __threadfence_block();
    ip_reductions<Cons>(d, b, N, Sn, Yn, Gn, mu, cost_new, o.ipddp_theta_norm_l2 != 0, phi_new, theta_new, ipr, icomp, &tn, mT, pT);
    for (int i = 0; i < mT; ++i) {
      const size_t k2 = ((size_t)a * kMTMax + i) * d.Bp + b;
      d.STt[k2] = tn.s[i]; d.YTt[k2] = tn.y[i]; d.GTt[k2] = tn.g[i];
    }
    for (int i = 0; i < pT; ++i) d.LamTt[((size_t)a * kPTMax + i) * d.Bp + b] = tn.lam[i];
  } else if constexpr (M > 0) {
    // add the parked terms in the reference's order (loads are independent of the running sums)
    double total = ev_total0, mer = cost_new;
    const double *evb = d.ev + GI((size_t)a * N, 2 * Cons::NSEG, 0);
    const size_t tstride = (size_t)d.NB * (2 * Cons::NSEG) * kLS;
    for (int c = 1; c < Cons::NSEG; ++c) {
      const double *q = evb + (size_t)(Cons::NSEG + c) * kLS;
      int t = 0;
      for (; t + 3 < N; t += 4) {
        const double v0 = q[(size_t)t * tstride], v1 = q[(size_t)(t + 1) * tstride], v2 = q[(size_t)(t + 2) * tstride], v3 = q[(size_t)(t + 3) * tstride];
        total += v0; total += v1; total += v2; total += v3;
      }
      for (; t < N; ++t) total += q[(size_t)t * tstride];
    }
    for (int c = 0; c < Cons::NSEG; ++c) {
      const double *q = evb + (size_t)c * kLS;
      int t = 0;
      for (; t + 3 < N; t += 4) {
        const double v0 = q[(size_t)t * tstride], v1 = q[(size_t)(t + 1) * tstride], v2 = q[(size_t)(t + 2) * tstride], v3 = q[(size_t)(t + 3) * tstride];
        mer -= mu * v0; mer -= mu * v1; mer -= mu * v2; mer -= mu * v3;
      }
      for (; t < N; ++t) mer -= mu * q[(size_t)t * tstride];
    }
    const double th = l2norm ? sqrt(total) : total;
    theta_new = dmax(th, ev_max);
    phi_new = mer; ipr = ev_max; icomp = ev_icomp;
  }
  if (!dfinite(phi_new) || !dfinite(theta_new) || !dfinite(ipr) || !dfinite(icomp)) return;
  bool accept = false;
  const bool uncon = (M == 0) && mT == 0 && pT == 0;
  if (uncon) {   // ipddp_solver.cpp:1785-1792
    const double dJ = d.cost[b] - cost_new;
    const double expected = -a_pr * (d.dV0[b] + 0.5 * a_pr * d.dV1[b]);
    const double ratio = expected > 0.0 ? dJ / expected : sign_of_reduction(dJ);
    accept = ratio > 1e-6;
  } else {        // ipddp_solver.cpp:1793-1834
    const double expected_improvement = a_pr * d.dV0[b];
    const int fn = d.filt_n[b];
    const double cv_old = (fn == 0) ? 0.0 : d.filt[(size_t)(kFilterCap + fn - 1) * d.Bp + b];
    const double high_ref = (fn == 0) ? d.filter_theta[b] : cv_old;
    const double merit_old = d.merit[b];
    if (theta_new > o.filter_max_violation_threshold) {
      if (theta_new < (1 - o.filter_violation_acceptance_threshold) * high_ref) accept = true;
    } else if (dmax(theta_new, cv_old) < o.filter_min_violation_for_armijo_check && expected_improvement < 0) {
      if (phi_new < merit_old + o.filter_armijo_constant * expected_improvement) accept = true;
    } else {
      if (phi_new < merit_old - o.filter_merit_acceptance_threshold * theta_new ||
          theta_new < (1 - o.filter_violation_acceptance_threshold) * cv_old) accept = true;
    }
  }
  d.t_cost[ti] = cost_new; d.t_merit[ti] = phi_new; d.t_theta[ti] = theta_new;
  d.t_inf_pr[ti] = ipr; d.t_inf_comp[ti] = icomp;
  d.t_success[ti] = accept ? 1 : 0;
}


// ---- filter helpers (interior_point_utils.cpp:79-139) on the per-trajectory filter columns
DEV void filter_accept(const DevBuf &d, int b, double mf, double cv) {
  int n = d.filt_n[b];
  for (int i = 0; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(kFilterCap + i) * d.Bp + b];
    if (fm <= mf && fv <= cv) return;   // dominated by an existing point
  }
  int w = 0;
  for (int i = 0; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(kFilterCap + i) * d.Bp + b];
    if (!(mf <= fm && cv <= fv)) {      // keep points the candidate does not dominate
      d.filt[(size_t)w * d.Bp + b] = fm; d.filt[(size_t)(kFilterCap + w) * d.Bp + b] = fv; ++w;
    }
  }
  if (w < kFilterCap) { d.filt[(size_t)w * d.Bp + b] = mf; d.filt[(size_t)(kFilterCap + w) * d.Bp + b] = cv; ++w; }
  d.filt_n[b] = w;
}
DEV void filter_prune(const DevBuf &d, int b) {
  int n = d.filt_n[b];
  if (n == 0) return;
  double bvm = d.filt[b], bvv = d.filt[(size_t)kFilterCap * d.Bp + b];
  double bmm = bvm, bmv = bvv;
  for (int i = 1; i < n; ++i) {
    const double fm = d.filt[(size_t)i * d.Bp + b], fv = d.filt[(size_t)(kFilterCap + i) * d.Bp + b];
    if (fv < bvv) { bvm = fm; bvv = fv; }
    if (fm < bmm) { bmm = fm; bmv = fv; }
  }
  d.filt[b] = bvm; d.filt[(size_t)kFilterCap * d.Bp + b] = bvv;
  int w = 1;
  if (fabs(bmv - bvv) > 1e-12 || fabs(bmm - bvm) > 1e-12) {
    d.filt[(size_t)1 * d.Bp + b] = bmm; d.filt[(size_t)(kFilterCap + 1) * d.Bp + b] = bmv; w = 2;
  }
  d.filt_n[b] = w;
}

// Costate trial of ONE trial, every step, on the lane of trajectory b -- the arithmetic of k_costate (kernels_lean.hpp).
// Only reached when the first-success rule has to move past a trial whose costate was not finite (k_costate stopped at
// that trial, so the later candidates have no costate rows yet).  Returns false when this trial's costate is not finite either.
template <int NX>
DEV bool costate_trial_serial(const DevBuf &d, int b, int cur, int a) {
  const size_t ti = (size_t)a * d.Bp + b;
  const int slot = (a < cur) ? a : a + 1;   // trial_slot
  const double a_pr = d.t_apr[ti];
  if ((d.fail_costate_mask >> a) & 1) return false;   // test hook (DevBuf::fail_costate_mask)
  bool finite = true;
  for (int t = 0; t <= d.N; ++t) {
    double xo[NX], lo[NX], vx[NX], xn[NX], lam[NX];
    ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, xo);
    ld<NX>(d.Lam + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, lo);
    ld<NX>(d.Vx + GI(t, NX, 0), kLS, vx);
    ld<NX>(d.X + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, xn);
    const double *vb = d.Vxx + GI(t, NX * NX, 0);
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
      for (int j = 0; j < NX; ++j) {
        const int lo_ = i < j ? i : j, hi_ = i < j ? j : i;   // upper triangle, as k_costate reads it
        s += vb[(size_t)(lo_ * NX + hi_) * kLS] * (xn[j] - xo[j]);
      }
      lam[i] = (lo[i] + a_pr * vx[i]) + s;
      finite = finite && dfinite(lam[i]);
    }
    if (!finite) return false;
    st<NX>(d.Lam + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, lam);
  }
  return true;
}

// ================================================================================ K5
// stage 1: trials [0, n1) were evaluated for PH_FWD1 trajectories (n1 = 1 for the first-success rule,
//          n1 = n_alphas for the best-merit rule); stage 2: trials [1, n_alphas) for PH_FWD2.
template <class Model, class Cons, bool TERM = false>
__global__ __launch_bounds__(64) void k_update(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int stage, int n1, int is_last_iter, int do_count) {
  constexpr int M = Cons::M;
  constexpr int NXu = Model::NX;
  constexpr int kRB = (NXu <= 6 && !TERM) ? 32 : 16;   // parked log-barrier terms fetched per round trip by the merit replay (registers permitting)
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const cddp_hip_options &o = P->opt;
  const bool ipddp = (P->solver == CDDP_HIP_SOLVER_IPDDP);
  const int ph = d.phase[b];
  const int n_alphas = d.n_alphas;
  int ladder_bin = -1;   // index of the accepted alpha (n_alphas: none worked) for the host's ladder-shape statistics
  const int mT = (TERM && ipddp) ? P->mT : 0, pT = (TERM && ipddp) ? P->pT : 0;
  const bool nobar = (M == 0) && mT == 0;     // no_barrier_needed (ipddp_solver.cpp:2552-2554)
  if ((stage == 1 && ph == PH_FWD1) || (stage == 2 && ph == PH_FWD2)) {
    const int lo = (stage == 1) ? 0 : 1;
    const int hi = (stage == 1) ? n1 : n_alphas;
    int win = -1;
    // t_success: 0 failed, 1 passed, 2 passed every test but its costate trial is not finite (k_costate) = failed
    if (P->ls_rule == CDDP_HIP_LS_FIRST_SUCCESS) {
      // The flags are fetched eight at a time (loads in flight together): walking them one by one is up to n_alpha dependent L2 round
      // trips on this one-lane-per-trajectory kernel, and a wave walks as far as its worst lane -- the whole ladder whenever one of
      // its 64 trajectories accepts nothing.  first1 = the first trial flagged 1, bad = a trial flagged 2 in front of it.
      int first1 = -1; bool bad = false;
      for (int base = lo; base < hi && first1 < 0; base += 8) {
        int f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (base + i < hi) ? d.t_success[(size_t)(base + i) * d.Bp + b] : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (first1 < 0) { if (f[i] == 1) first1 = base + i; else if (f[i] == 2) bad = true; }
      }
      if (first1 < 0 || !(bad && ipddp)) win = first1;
      else {
        bool past_bad = false;   // k_costate stopped at a trial flagged 2: later candidates have no costate rows yet
        for (int a = lo; a < hi; ++a) {
          const int sc = d.t_success[(size_t)a * d.Bp + b];
          if (sc == 0) continue;
          if (sc == 2) { past_bad = ipddp; continue; }
          if (past_bad && !costate_trial_serial<NXu>(d, b, d.cur[b], a)) continue;
          win = a; break;
        }
      }
    } else {
      // best-merit rule: k_costate evaluated the costate of ONE trial per trajectory, the least-merit trial among those that passed
      // every other test (flag 1 or 2).  Flag 1: that trial wins (no other successful trial has less merit).  Flag 2 (its costate was
      // not finite): the remaining successful trials are walked in merit order and their costate is evaluated here.
      double best = INFINITY; int cand = -1;
      for (int a = lo; a < hi; ++a) {
        const size_t ti = (size_t)a * d.Bp + b;
        if (d.t_success[ti] != 0 && d.t_merit[ti] < best) { best = d.t_merit[ti]; cand = a; }
      }
      if (cand >= 0 && (d.t_success[(size_t)cand * d.Bp + b] == 1 || !ipddp)) win = cand;
      else if (cand >= 0) {
        double floor_m = best; int floor_a = cand;   // trials already ruled out: merit < floor, or == floor with index <= floor_a
        for (;;) {
          double nb = INFINITY; int na_ = -1;
          for (int a = lo; a < hi; ++a) {
            const size_t ti = (size_t)a * d.Bp + b;
            const double mt = d.t_merit[ti];
            if (d.t_success[ti] != 1) continue;
            if (mt < floor_m || (mt == floor_m && a <= floor_a)) continue;
            if (mt < nb) { nb = mt; na_ = a; }
          }
          if (na_ < 0) break;
          if (costate_trial_serial<NXu>(d, b, d.cur[b], na_)) { win = na_; break; }
          floor_m = nb; floor_a = na_;
        }
      }
    }
    if (win < 0 && hi < n_alphas) { d.phase[b] = PH_FWD2; goto count; }   // more alphas to try
    ladder_bin = (win >= 0) ? win : n_alphas;
    {
      const int iter = d.iter[b];
      if (win >= 0) {
        // ---- applyForwardPassResult (cddp_solver_base.cpp:190-198, ipddp_solver.cpp:1878-1951)
        const size_t ti = (size_t)win * d.Bp + b;
        // Every word this block reads is fetched BEFORE its first store: the DevBuf arrays may alias as far as the compiler knows, so a
        // load behind a store cannot be hoisted above it, and "d.x[b] = d.t_x[ti]" line by line is one dependent L2 round trip per line
        // on this one-lane-per-trajectory kernel (measured: 10 us of the kernel's 26, profiles/r03_k4_block_times.md).
        const int old_cur = d.cur[b];
        const double w_cost = d.t_cost[ti], w_merit = d.t_merit[ti], w_apr = d.t_apr[ti], w_theta = d.t_theta[ti];
        const double w_adu = ipddp ? d.t_adu[ti] : 1.0, w_ipr = ipddp ? d.t_inf_pr[ti] : 0.0, w_icomp = ipddp ? d.t_inf_comp[ti] : 0.0;
        const double dJ = d.cost[b] - w_cost;
        const double c_reg = d.reg[b], c_sn = d.step_norm[b];   // not written before their use below
        // rollouts the reference runs to get here: the first-success rule stops at the winner, the best-merit rule
        // (one std::async per alpha, cddp_solver_base.cpp:264-286) always evaluates the whole ladder
        const int na_walked = (P->ls_rule == CDDP_HIP_LS_FIRST_SUCCESS) ? win + 1 : n_alphas;
        int ns = d.n_fwd_steps[b];
        for (int a = 0; a < na_walked; ++a) ns += d.t_steps[(size_t)a * d.Bp + b];
        const int nf = d.n_fwd[b] + na_walked;
        d.n_fwd[b] = nf;
        d.n_fwd_steps[b] = ns;
        d.cur[b] = trial_slot(old_cur, win);
        d.cost[b] = w_cost;
        d.merit[b] = w_merit;
        d.alpha_pr[b] = w_apr;
        d.alpha_du[b] = w_adu;
        int st = CDDP_HIP_STATUS_RUNNING;
        bool conv = false;
        if (ipddp) {
          d.inf_pr[b] = w_ipr; d.inf_comp[b] = w_icomp;
          d.phi[b] = w_merit; d.filter_theta[b] = w_theta; d.theta[b] = w_theta;
          if constexpr (TERM) {   // terminal slack / dual / residual / multipliers of the winner (:1900-1941)
            for (int i = 0; i < mT; ++i) {
              const size_t k2 = ((size_t)win * kMTMax + i) * d.Bp + b;
              d.ST[(size_t)i * d.Bp + b] = d.STt[k2]; d.YT[(size_t)i * d.Bp + b] = d.YTt[k2]; d.GT[(size_t)i * d.Bp + b] = d.GTt[k2];
            }
            for (int i = 0; i < pT; ++i) d.LamT[(size_t)i * d.Bp + b] = d.LamTt[((size_t)win * kPTMax + i) * d.Bp + b];
          }
          // ---- updateBarrierParameters(true) (ipddp_solver.cpp:2548-2660)
          const double sdu = scaled_inf_du<Model, Cons>(d, b, old_cur);
          double mu = d.mu[b];
          const double mu_old = mu;
          if (!nobar) {
            if (o.barrier_strategy == CDDP_HIP_BARRIER_ADAPTIVE) {
              const double kkt = dmax(dmax(w_ipr, sdu), w_icomp);   // = d.inf_pr[b], d.inf_comp[b], stored above
              const double threshold = dmax(o.barrier_mu_update_factor * mu, 2.0 * mu);
              if (kkt <= threshold) {
                double factor = o.barrier_mu_update_factor;
                if (mu > 1e-20) {
                  const double ratio = kkt / dmax(mu, 1e-20);
                  if (ratio < 0.01) factor = 0.1 * o.barrier_mu_update_factor;
                  else if (ratio < 0.1) factor = 0.3 * o.barrier_mu_update_factor;
                  else if (ratio < 0.5) factor = 0.6 * o.barrier_mu_update_factor;
                }
                const double linear = factor * mu;
                const double superlinear = solver_pow(mu, o.barrier_mu_update_power);
                mu = dmax(dmin(linear, superlinear), dmax(o.barrier_mu_min_value, o.tolerance / 100.0));
              }
            } else {
              const double kkt = dmax(dmax(w_ipr, sdu * o.ipddp_barrier_update_dual_weight), w_icomp);
              if (kkt <= o.ipddp_mu_kappa_epsilon * mu) {
                const double linear = o.barrier_mu_update_factor * mu;
                const double superlinear = solver_pow(mu, o.barrier_mu_update_power);
                mu = dmax(o.barrier_mu_min_value, dmin(linear, superlinear));
              }
            }
          }
          d.mu[b] = mu;
          const int cs = d.cur[b];
          // computeTheta / computeBarrierMerit / computePrimalAndComplementarity on the accepted iterate
          // (ipddp_solver.cpp:2622-2656).  With an unchanged mu they are the very sums the winning trial
          // already evaluated (same routine, same order), so the pass over S/Y/G is only repeated when mu moved.
          double phi_n = w_merit, theta_n = w_theta, ipr = w_ipr, icomp = w_icomp;
          if constexpr (TERM) {
            if (mu != mu_old) {
              TermState ts;
              term_load(d, b, mT, pT, ts);
              double xN[NXu];
              ld<NXu>(d.X + (size_t)cs * d.planeX + GI(d.N, NXu, 0), kLS, xN);
              term_eq_residual<NXu>(P, xN, ts.h);
              bool replayed = false;
              if constexpr (M > 0) {
                if (d.ev_valid && mT == 0) {
                  // terminal equality only, trials evaluated by the two-role rollout: theta and the primal residual do
                  // not depend on mu; the merit chain is replayed from the parked log-barrier terms and the multiplier
                  // term lambda^T h is appended, exactly the order of ip_reductions + term_reductions (see the
                  // branch below for layouts without a terminal set)
                  const int N = d.N;
                  const double *evb = d.ev + GI((size_t)win * N, 2 * Cons::NSEG, 0);
                  const size_t tstride = (size_t)d.NB * (2 * Cons::NSEG) * kLS;
                  double mer = d.cost[b];
                  for (int c = 0; c < Cons::NSEG; ++c) {
                    const double *q = evb + (size_t)c * kLS;
                    int t = 0;
                    for (; t + kRB - 1 < N; t += kRB) {
                      double v[kRB];
#pragma unroll
                      for (int k = 0; k < kRB; ++k) v[k] = q[(size_t)(t + k) * tstride];
#pragma unroll
                      for (int k = 0; k < kRB; ++k) mer -= mu * v[k];
                    }
                    for (; t < N; ++t) mer -= mu * q[(size_t)t * tstride];
                  }
                  double dp = 0.0;
                  for (int r = 0; r < pT; ++r) dp += ts.lam[r] * ts.h[r];
                  mer += dp;
                  phi_n = mer;
                  icomp = dmax(fabs(d.t_ysmax[ti] - mu), fabs(d.t_ysmin[ti] - mu));
                  replayed = true;
                }
              }
              if (!replayed)
                ip_reductions<Cons>(d, b, d.N, d.S + (size_t)cs * d.planeM, d.Y + (size_t)cs * d.planeM,
                                    d.G + (size_t)cs * d.planeM, mu, d.cost[b], o.ipddp_theta_norm_l2 != 0, phi_n, theta_n, ipr, icomp, &ts, mT, pT);
            }
          } else if constexpr (M > 0) {
            if (mu != mu_old) {
              // theta and the primal residual do not depend on mu; the barrier merit and the complementarity
              // residual do.  The log-barrier terms ls(c, t) of the accepted trial are still parked in the ev
              // scratch (same values, K4), so the merit chain  mer -= mu * ls  is replayed in the reference's
              // order without re-evaluating 2 N M logarithms on one lane.
              const int N = d.N;
              const double *evb = d.ev + GI((size_t)win * N, 2 * Cons::NSEG, 0);
              const size_t tstride = (size_t)d.NB * (2 * Cons::NSEG) * kLS;
              double mer = d.cost[b];
              for (int c = 0; c < Cons::NSEG; ++c) {
                const double *q = evb + (size_t)c * kLS;
                int t = 0;
                for (; t + kRB - 1 < N; t += kRB) {   // kRB row loads per round trip, then the ordered chain
                  double v[kRB];
#pragma unroll
                  for (int k = 0; k < kRB; ++k) v[k] = q[(size_t)(t + k) * tstride];
#pragma unroll
                  for (int k = 0; k < kRB; ++k) mer -= mu * v[k];
                }
                for (; t < N; ++t) mer -= mu * q[(size_t)t * tstride];
              }
              // max |y s - mu| over the iterate: |v - mu| is monotone in v on either side of mu (rounded
              // subtraction is monotone), so the extreme products recorded by the trial give the same maximum
              const double ic = dmax(fabs(d.t_ysmax[ti] - mu), fabs(d.t_ysmin[ti] - mu));
              phi_n = mer; icomp = ic;
            }
          }
          const double ftheta = dmax(theta_n, 1e-8);
          const bool reset = (mu < mu_old) && (mu > 0.0);
          if (reset) {   // filter cleared; re-seeded only when terminal constraints exist (:2629-2637)
            d.filt_n[b] = 0;
            if (mT > 0 || pT > 0) filter_accept(d, b, w_merit, ftheta);   // (w_merit = d.phi[b], stored above)
          }
          else { filter_accept(d, b, w_merit, ftheta); if (d.filt_n[b] > o.ipddp_max_filter_size) filter_prune(d, b); }
          d.inf_pr[b] = ipr; d.inf_comp[b] = icomp;
          d.merit[b] = phi_n; d.phi[b] = phi_n; d.filter_theta[b] = ftheta;
          d.theta[b] = dmax(ftheta, dmax(o.ipddp_theta_0_floor, 1e-8));
          hist_push(d, b, mu);
          d.reg[b] = reg_decrease(o, c_reg);
          // ---- checkConvergence (ipddp_solver.cpp:1953-2025)
          const double sdu2 = scaled_inf_du<Model, Cons>(d, b, old_cur);
          const double scomp = icomp, pr = ipr, sn = c_sn;   // the values just stored / fetched at the top
          if (nobar) {
            if (pr < o.tolerance && sdu2 < o.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; conv = true; }
            else if (o.acceptable_tolerance > 0.0) {
              const double sq = sqrt(o.acceptable_tolerance);
              bool acc = (pr < sq && sdu2 < sq && iter > 50);
              if (dJ > 0.0) acc = acc || (dJ < o.acceptable_tolerance && iter > 50 && pr < sq && sdu2 < sq);
              if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; conv = true; }
            }
          } else {
            const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
            if (pr < tol && sdu2 < tol && scomp < tol && sn < o.tolerance * 10.0) { st = CDDP_HIP_STATUS_OPTIMAL; conv = true; }
            else if (o.acceptable_tolerance > 0.0) {
              const double at = sqrt(o.acceptable_tolerance);
              const double bat = dmax(o.barrier_mu_min_value * 100.0, o.tolerance / 10.0);
              const bool akkt = pr < at && sdu2 < at && scomp < at;
              const bool bpc = mu <= bat;
              bool acc = akkt && bpc && iter > 10 && fabs(dJ) < o.acceptable_tolerance;
              acc = acc || (akkt && bpc && iter >= 1 && sn < o.tolerance * 10.0 && pr < 1e-4);
              if (acc) { st = CDDP_HIP_STATUS_ACCEPTABLE; conv = true; }
            }
          }
        } else {
          hist_push(d, b, 0.0);
          d.reg[b] = reg_decrease(o, d.reg[b]);
          // checkConvergence (clddp_solver.cpp:264-277)
          if (d.inf_du[b] < o.tolerance) { st = CDDP_HIP_STATUS_OPTIMAL; conv = true; }
          else if (dJ > 0.0 && dJ < o.acceptable_tolerance) { st = CDDP_HIP_STATUS_ACCEPTABLE; conv = true; }
        }
        if (conv) { d.status[b] = st; d.phase[b] = PH_DONE; }
        else d.phase[b] = PH_ACTIVE;
      } else {
        // ---- handleForwardPassFailure (cddp_solver_base.cpp:206-218, ipddp_solver.cpp:2037-2082)
        const int nf0 = d.n_fwd[b]; int ns = d.n_fwd_steps[b]; const double reg0 = d.reg[b];   // fetched before the first store (see above)
        for (int a = 0; a < n_alphas; ++a) ns += d.t_steps[(size_t)a * d.Bp + b];
        d.n_fwd[b] = nf0 + n_alphas;
        d.n_fwd_steps[b] = ns;
        double reg = reg_increase(o, reg0);
        if (ipddp && !nobar && pT > 0) reg = reg_increase(o, reg);   // extra bump for terminal-equality problems (:2043-2050)
        d.reg[b] = reg;
        if (reg >= o.reg_max_value) {
          int st = CDDP_HIP_STATUS_REG_LIMIT;
          if (ipddp) {
            const double sdu = scaled_inf_du<Model, Cons>(d, b, d.cur[b]);
            const double base = sqrt(dmax(o.acceptable_tolerance, o.tolerance));
            const double at = nobar ? base : dmax(base, o.ipddp_barrier_tol_mult * d.mu[b]);
            const bool acc = o.acceptable_tolerance > 0.0 && d.inf_pr[b] < at && sdu < at && (nobar || d.inf_comp[b] < at);
            if (acc) st = CDDP_HIP_STATUS_ACCEPTABLE;
          }
          d.status[b] = st; d.phase[b] = PH_DONE;
        } else d.phase[b] = PH_ACTIVE;
      }
    }
  }
count:
  if (d.win_hist) {   // one atomic per populated bin and wavefront
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

// ================================================================================ K0
// ISolverAlgorithm::initialize.  CLDDP: cost of the given (X,U) (clddp_solver.cpp:68-74).
// IPDDP cold start (ipddp_solver.cpp:819-913): re-rollout X from U, mu, g, s/y initialisation,
// cost, filter reset.
// mode: kInitCold = fresh solver object (clddp_solver.cpp:62-74, ipddp_solver.cpp:819-913);
//       kInitWarmProvided = options.warm_start with NO solver state ("warm start with provided trajectory",
//                           ipddp_solver.cpp:733-816; CLDDP falls back to the cold start, clddp_solver.cpp:61-66);
//       kInitWarmExisting = options.warm_start on a handle that already holds gains / duals ("existing solver state",
//                           ipddp_solver.cpp:675-731, clddp_solver.cpp:51-60): the staged slack / dual / costate /
//                           terminal variables are kept (per-constraint re-initialisation test + interior repair,
//                           :264-292, 2345-2426), context scalars (regularisation, step lengths, inf_du) persist.
enum { kInitCold = 0, kInitWarmProvided = 1, kInitWarmExisting = 2 };

// repairWarmstartInterior (ipddp_solver.cpp:233-262) on one constraint object's slack / dual vector
DEV void repair_interior(const cddp_hip_options &o, double *s, double *y, int dim) {
  if (!o.ipddp_warmstart_repair) return;
  double mn = INFINITY, mny = INFINITY;
  for (int i = 0; i < dim; ++i) { s[i] = dmax(s[i], o.ipddp_warmstart_s_min); mn = dmin(mn, s[i]); }
  if (mn < o.ipddp_warmstart_s_min * o.ipddp_warmstart_interior_factor) for (int i = 0; i < dim; ++i) s[i] *= o.ipddp_warmstart_interior_factor;
  for (int i = 0; i < dim; ++i) { y[i] = dmax(y[i], o.ipddp_warmstart_y_min); mny = dmin(mny, y[i]); }
  if (mny < o.ipddp_warmstart_y_min * o.ipddp_warmstart_interior_factor) for (int i = 0; i < dim; ++i) y[i] *= o.ipddp_warmstart_interior_factor;
}
// warmstartNeedsReinit (ipddp_solver.cpp:264-292)
DEV bool needs_reinit(const cddp_hip_options &o, const double *y, const double *s, const double *g, int dim) {
  for (int i = 0; i < dim; ++i) if (!dfinite(y[i]) || !dfinite(s[i])) return true;
  for (int i = 0; i < dim; ++i) {
    if (y[i] <= 1e-10 || s[i] <= kEpsSlack) return true;
    const double required = dmax(o.ipddp_slack_var_init_scale, -g[i] + kSlackInteriorOffset);
    if (s[i] < 0.1 * required) return true;
  }
  return false;
}

// Stage the live iterate of every trajectory into slot 0 before a warm re-initialisation: slack / dual / costate rows
// always, X / U rows unless the caller supplied a new initial trajectory (copy_xu = 0).
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_stage(DevBuf d, int copy_xu, int ipddp) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  const int cur = d.cur[b];
  if (cur == 0) return;
  if (copy_xu) {
    for (int i = 0; i < NX; ++i) d.X[GI(t, NX, i)] = d.X[(size_t)cur * d.planeX + GI(t, NX, i)];
    if (t < d.N) for (int i = 0; i < NU; ++i) d.U[GI(t, NU, i)] = d.U[(size_t)cur * d.planeU + GI(t, NU, i)];
  }
  if (ipddp) {
    for (int i = 0; i < NX; ++i) d.Lam[GI(t, NX, i)] = d.Lam[(size_t)cur * d.planeX + GI(t, NX, i)];
    if constexpr (M > 0) {
      if (t < d.N)
        for (int i = 0; i < M; ++i) {
          d.S[GI(t, M, i)] = d.S[(size_t)cur * d.planeM + GI(t, M, i)];
          d.Y[GI(t, M, i)] = d.Y[(size_t)cur * d.planeM + GI(t, M, i)];
        }
    }
  }
}

template <class Model, class Cons, bool TERM = false>
__global__ __launch_bounds__(64) void k_init(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int mode) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  typedef Objective<NX, NU> Obj;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  const ProblemDev *__restrict__ P = Pk;   // direct kernel argument: scalar (SMEM) loads, no vmcnt traffic
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const bool ipddp = (P->solver == CDDP_HIP_SOLVER_IPDDP);
  const bool existing = (mode == kInitWarmExisting);
  d.cur[b] = 0;
  double *X0 = d.X, *U0 = d.U;
  d.iter[b] = 0; d.status[b] = CDDP_HIP_STATUS_RUNNING; d.phase[b] = PH_ACTIVE;
  d.n_bwd[b] = 0; d.n_fwd[b] = 0; d.n_fwd_steps[b] = 0; d.bwd_ok[b] = 0; d.filt_n[b] = 0;
  if (b < d.hist_batch) d.hist_n[b] = 0;
  if (!existing) d.reg[b] = o.reg_initial_value;
  d.dV0[b] = 0.0; d.dV1[b] = 0.0;
  if (!(existing && !ipddp)) d.step_norm[b] = 0.0;
  d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;
  double x[NX];
  ld<NX>(X0 + GI(0, NX, 0), kLS, x);
  double cost = 0.0;
  if (!ipddp) {
    for (int t = 0; t < N; ++t) {
      double xt[NX], u[NU];
      ld<NX>(X0 + GI(t, NX, 0), kLS, xt);
      ld<NU>(U0 + GI(t, NU, 0), kLS, u);
      cost += Obj::running_cost(P, xrt, t, xt, u);
      if (!existing) {
        double z[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) z[i] = 0.0;
        st<NU>(d.k + GI(t, NU, 0), kLS, z);     // initializeGains: k_u_ = 0 (BoxQP warm start)
      }
    }
    double xN[NX];
    ld<NX>(X0 + GI(N, NX, 0), kLS, xN);
    cost += Obj::terminal_cost(P, xN);
    d.cost[b] = cost; d.merit[b] = cost;
    if (!existing) {
      d.inf_pr[b] = INFINITY; d.inf_du[b] = INFINITY; d.inf_comp[b] = INFINITY;   // cddp_core.cpp:297-301
      d.alpha_pr[b] = o.ls_initial_step_size; d.alpha_du[b] = 0.0; d.mu[b] = 0.0;
    }
    d.phi[b] = cost; d.theta[b] = 0.0; d.filter_theta[b] = 0.0;
    hist_push(d, b, 0.0);
    return;
  }
  const int mT = TERM ? P->mT : 0, pT = TERM ? P->pT : 0;
  const bool unconstrained = (M == 0 && !(TERM && P->n_term > 0));
  // mu_ = constraint_set.empty() && terminal set empty ? max(tol/10, mu_min) : mu_initial   (ipddp_solver.cpp:876-879);
  // existing solver state: 0.1 mu_initial (:684); provided trajectory: from the largest violation (:777-803, below)
  double mu = unconstrained ? dmax(o.tolerance / 10.0, o.barrier_mu_min_value) : o.barrier_mu_initial;
  if (existing) mu = o.barrier_mu_initial * 0.1;
  if (!existing) { d.alpha_pr[b] = 1.0; d.alpha_du[b] = 1.0; }
  double *S0 = d.S, *Y0 = d.Y, *G0 = d.G, *L0 = d.Lam;
  if (mode != kInitCold) {
    // ---- warm start: rollout + evaluateTrajectoryWarmStart (:2296-2343) first, mu next, the interior after that
    double maxviol = 0.0;
    for (int t = 0; t < N; ++t) {
      double u[NU], xn[NX];
      ld<NU>(U0 + GI(t, NU, 0), kLS, u);
      cost += Obj::running_cost(P, xrt, t, x, u);
      if constexpr (M > 0) {
        double g[MM];
        Cons::template eval<NX, NU>(P, x, u, g);
#pragma unroll
        for (int i = 0; i < M; ++i) maxviol = dmax(maxviol, g[i]);
        st<M>(G0 + GI(t, M, 0), kLS, g);
      }
      if (!existing) {
        double z[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) z[i] = 0.0;
        st<NX>(L0 + GI(t, NX, 0), kLS, z);
      }
      Stepper<Model>::step(P->integrator, P->dt, P->mp, x, u, xn);
      st<NX>(X0 + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = xn[i];
    }
    if (!existing) {
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      st<NX>(L0 + GI(N, NX, 0), kLS, z);
    }
    cost += Obj::terminal_cost(P, x);
    if constexpr (TERM) {
      double gT[kMTMax];
      term_ineq_eval<NX>(P, x, gT);
      for (int i = 0; i < mT; ++i) maxviol = dmax(maxviol, gT[i]);
    }
    if (mode == kInitWarmProvided && !unconstrained) {   // :786-803
      if (maxviol <= o.tolerance) mu = dmax(o.tolerance, o.barrier_mu_min_value);
      else if (maxviol <= 0.1) mu = dmax(o.tolerance * 10.0, o.barrier_mu_initial * 0.01);
      else mu = o.barrier_mu_initial * 0.1;
    }
    // the provided-trajectory branch never evaluates the cost of an unconstrained problem (:777-781): it stays +inf
    if (mode == kInitWarmProvided && unconstrained) cost = INFINITY;
    if constexpr (M > 0) {
      __threadfence_block();
      for (int t = 0; t < N; ++t) {   // initializeDualSlackVariablesWarmStart (:2345-2426)
        double g[MM], s[MM], y[MM];
        ld<M>(G0 + GI(t, M, 0), kLS, g);
        if (existing) { ld<M>(S0 + GI(t, M, 0), kLS, s); ld<M>(Y0 + GI(t, M, 0), kLS, y); }
        for (int c = 0; c < Cons::NSEG; ++c) {
          const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
          const bool need = !existing || needs_reinit(o, y + off, s + off, g + off, dim);
          if (need)
            for (int i = 0; i < dim; ++i) {
              s[off + i] = dmax(o.ipddp_slack_var_init_scale, -g[off + i] + kSlackInteriorOffset);
              y[off + i] = (mu * o.ipddp_dual_var_init_scale) / dmax(s[off + i], kEpsSlack);
            }
          repair_interior(o, s + off, y + off, dim);
        }
        st<M>(S0 + GI(t, M, 0), kLS, s);
        st<M>(Y0 + GI(t, M, 0), kLS, y);
      }
    }
  } else {
  for (int t = 0; t < N; ++t) {
    double u[NU], xn[NX];
    ld<NU>(U0 + GI(t, NU, 0), kLS, u);
    cost += Obj::running_cost(P, xrt, t, x, u);
    if constexpr (M > 0) {
      double g[MM], s[MM], y[MM];
      Cons::template eval<NX, NU>(P, x, u, g);
#pragma unroll
      for (int i = 0; i < M; ++i) {   // initializeDualSlackVariables (:2456-2468)
        s[i] = dmax(o.ipddp_slack_var_init_scale, -g[i] + kSlackInteriorOffset);
        y[i] = (mu * o.ipddp_dual_var_init_scale) / dmax(s[i], kEpsSlack);
      }
      for (int c = 0; c < Cons::NSEG; ++c) repair_interior(o, s + Cons::seg_off(c), y + Cons::seg_off(c), Cons::seg_dim(c));
      st<M>(G0 + GI(t, M, 0), kLS, g);
      st<M>(S0 + GI(t, M, 0), kLS, s);
      st<M>(Y0 + GI(t, M, 0), kLS, y);
    }
    double z[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = 0.0;
    st<NX>(L0 + GI(t, NX, 0), kLS, z);
    Stepper<Model>::step(P->integrator, P->dt, P->mp, x, u, xn);
    st<NX>(X0 + GI(t + 1, NX, 0), kLS, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  }
  {
    double z[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = 0.0;
    st<NX>(L0 + GI(N, NX, 0), kLS, z);
  }
  cost += Obj::terminal_cost(P, x);
  }
  d.mu[b] = mu;
  d.cost[b] = cost;
  // resetFilter (ipddp_solver.cpp:2484-2519)
  double phi = cost, theta = 0.0, ipr = 0.0, icomp = 0.0;
  if constexpr (TERM) {
    // terminal slack / dual initialisation (ipddp_solver.cpp:889-908) and multipliers (:829-830)
    TermState ts;
    term_ineq_eval<NX>(P, x, ts.g);
    if (existing) for (int i = 0; i < mT; ++i) { ts.s[i] = d.ST[(size_t)i * d.Bp + b]; ts.y[i] = d.YT[(size_t)i * d.Bp + b]; }
    for (int c = 0; c < P->n_term; ++c) {   // cold :889-908 / initializeTerminalWarmstartDualSlack :294-353
      const TermDev &td = P->terms[c];
      if (td.kind != CDDP_HIP_TERM_INEQUALITY) continue;
      const bool need = !existing || needs_reinit(o, ts.y + td.offset, ts.s + td.offset, ts.g + td.offset, td.dim);
      if (need)
        for (int r = 0; r < td.dim; ++r) {
          const int j = td.offset + r;
          ts.s[j] = dmax(o.ipddp_slack_var_init_scale, -ts.g[j] + kSlackInteriorOffset);
          ts.y[j] = (mu * o.ipddp_dual_var_init_scale) / dmax(ts.s[j], kEpsSlack);
        }
      repair_interior(o, ts.s + td.offset, ts.y + td.offset, td.dim);
    }
    for (int i = 0; i < mT; ++i) {
      d.GT[(size_t)i * d.Bp + b] = ts.g[i]; d.ST[(size_t)i * d.Bp + b] = ts.s[i]; d.YT[(size_t)i * d.Bp + b] = ts.y[i];
      d.dST[(size_t)i * d.Bp + b] = 0.0; d.dYT[(size_t)i * d.Bp + b] = 0.0;
    }
    for (int i = 0; i < pT; ++i) {   // multipliers: zero, or kept when finite under a warm start with solver state (:355-366)
      double lam = existing ? d.LamT[(size_t)i * d.Bp + b] : 0.0;
      ts.lam[i] = lam;
    }
    if (existing) { bool fin = true; for (int i = 0; i < pT; ++i) fin = fin && dfinite(ts.lam[i]); if (!fin) for (int i = 0; i < pT; ++i) ts.lam[i] = 0.0; }
    for (int i = 0; i < pT; ++i) { d.LamT[(size_t)i * d.Bp + b] = ts.lam[i]; d.dLamT[(size_t)i * d.Bp + b] = 0.0; }
    term_eq_residual<NX>(P, x, ts.h);
    __threadfence_block();
    ip_reductions<Cons>(d, b, N, S0, Y0, G0, mu, cost, o.ipddp_theta_norm_l2 != 0, phi, theta, ipr, icomp, &ts, mT, pT);
  } else {
    if constexpr (M > 0) { __threadfence_block(); ip_reductions<Cons>(d, b, N, S0, Y0, G0, mu, cost, o.ipddp_theta_norm_l2 != 0, phi, theta, ipr, icomp); }
  }
  d.merit[b] = phi; d.phi[b] = phi; d.inf_pr[b] = ipr; d.inf_comp[b] = icomp;
  const double ft = dmax(theta, 1e-8);
  d.filter_theta[b] = ft;
  d.theta[b] = dmax(ft, dmax(o.ipddp_theta_0_floor, 1e-8));
  if (!existing) d.inf_du[b] = 0.0;
  if constexpr (TERM) { if (mT > 0 || pT > 0) filter_accept(d, b, phi, ft); }   // resetFilter seeds the filter (:2513-2516)
  hist_push(d, b, mu);
}

#undef GI
}  // namespace cddp_dev
