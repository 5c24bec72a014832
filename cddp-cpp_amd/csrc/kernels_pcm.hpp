// K4 for the small path-constrained layouts, one workgroup per (64-trajectory tile, GROUP of NA step sizes):
//   waves 0 .. NA-1  producers, one per step size   u_t = u + a k + K dx, x_{t+1} = f(x_t, u_t)   (the dependent chain)
//   wave NA          ONE consumer for the NA trials  slack / dual trials, fraction-to-boundary test, cost, g(x, u), barrier / theta
//                                                    terms, filter test, trial records                (everything off the chain)
//
// Why (round 4): the two-wave form k_forward_ipddp_pc (kernels_lean.hpp) launches 2 waves per (tile, step
// size) -- 1408 wavefronts for the 64 x 11 trials of BASELINE config[1] on 1024 SIMDs.  A producer that has its SIMD to itself walks
// the 100 steps in 143 us (its instruction stream: ~650 un-fused f64 instructions per RK4 cart-pole step at ~5 cycles each); on the
// 192 CUs that host three workgroups the waves share SIMDs and the launch lasts 237 - 300 us (profiles/r03_k4_block_times.md).  The
// consumer's per-step work splits into a part that does NOT depend on the step size -- the record of the current iterate (s, y, k_s,
// k_y, K, Y S^-1: 14 row loads), the rows of K_s / K_y rebuilt from it -- and a part that does (the trial itself).  One consumer
// serving NA producers loads and rebuilds the shared part once, and the launch needs (NA + 1) / NA waves per trial instead of 2:
// with NA = 3 the 11 step sizes of a tile are 4 workgroups of 4 + 4 + 4 + 3 waves, 256 workgroups for the 64 tiles of config[1] --
// one per CU, one wave per SIMD.  Same arithmetic per trial, in the same order (bitwise: tests/test_gpu_parity.py runs both forms).
//
// Channel per producer: an LDS ring (x_t, dx_t, u_t per lane), a produced-step and a retired-step counter, as in the two-wave form.
#pragma once
#include <type_traits>
#include "kernels_lean.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

// which layouts have the multi-alpha form: both roles' step records small enough for the ping-pong register sets
template <class Model, class Cons>
struct PcmTraits {
  static constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  static constexpr int PREC = NX + 2 * NU + NU * NX;                                                  // producer record (doubles)
  static constexpr int CREC = (Cons::HAS_X ? NX : 1) + (Cons::NEEDS_U ? NU : 1) + 5 * M + NU * NX;      // consumer record
  static constexpr bool kOk = M > 0 && PREC <= 40 && CREC <= 40 && (2 * NX + NU) <= 12;
};

template <class Model, class Cons, int NA>
__global__ __launch_bounds__(64 * (NA + 1)) void k_forward_ipddp_pcm(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                                     int a0, int na, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef Objective<NX, NU> Obj;
  static_assert(PcmTraits<Model, Cons>::kOk, "multi-alpha rollout is for the small path-constrained layouts");
  constexpr int RW = 2 * NX + NU;          // doubles per lane per step: x_t, dx_t, u_t
  constexpr int kRing = 4;                 // steps in flight per producer (NA rings of <= 4 * 12 * 64 doubles: <= 73 KB at NA = 3)
  __shared__ double s_ring[NA][kRing * RW * 64];
  __shared__ int s_prod[NA];        // steps published by producer j
  __shared__ int s_cons[NA];        // steps of producer j retired by the consumer
  __shared__ int s_pstat[NA][64];   // first step at which the producer lane went non-finite (N + 2 = never)
  __shared__ double s_pcost[NA][64];
  __shared__ double s_obj[Obj::kStage];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
  const bool producer = wave < NA;
  const int b = blockIdx.x * 64 + lane;
  const int a_first = a0 + (int)blockIdx.y * NA;
  const int n_here = (a0 + na - a_first) < NA ? (a0 + na - a_first) : NA;   // step sizes of this group (the last group of a ladder may be short)
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const bool active = (b < d.B) && (force || d.phase[b] == phase_req);
  if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;   // same mask in every wave of the workgroup: all leave
  if (producer) s_pstat[wave][lane] = N + 2;
  if (threadIdx.x < NA) { s_prod[threadIdx.x] = 0; s_cons[threadIdx.x] = 0; }
  Obj::stage(P, s_obj, (int)threadIdx.x, 64 * (NA + 1));
  __syncthreads();
  const int bb = (b < d.B) ? b : 0;
  const int cur = (b < d.B) ? d.cur[b] : 0;
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double apr_max = d.apr_max[bb];
  auto wait_ge = [&](int *ctr, int need) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  if (producer) {
    // ------------------------------------------------------------------ producer `wave`: the dynamics chain of one step size
    if (wave >= n_here) return;
    const int a = a_first + wave;
    const int slot = trial_slot(cur, a);
    const double alpha = P->alphas[a];
    const double a_pr = dmin(alpha, apr_max);
    double *ring = s_ring[wave];
    int *my_prod = &s_prod[wave], *my_cons = &s_cons[wave];
    double *Xn = d.X + (size_t)slot * d.planeX;
    double *Un = d.U + (size_t)slot * d.planeU;
    const double *Uc = d.U + (size_t)cur * d.planeU;
    bool alive = active;
    double x[NX];
    ld<NX>(Xc + GI(0, NX, 0), kLS, x);
    st<NX>(Xn + GI(0, NX, 0), kLS, x);
    struct StepIn { double xo[NX], uo[NU], kk[NU], KK[NU * NX]; };
    auto load_step = [&](int tt, StepIn &r) {
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.uo);
      ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
      ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
    };
    DynCtx dc;
    dc.load(P->integrator, P->dt, P->mp);
    auto prime = [&]() {   // one step's store pattern ahead of the loop (see k_forward_ipddp_pc)
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      st<NU>(Un + GI(0, NU, 0), kLS, z);
      st<NX>(Xn + GI(1, NX, 0), kLS, z);
    };
    auto step = [&](const int t, StepIn &cs, StepIn &nxt) {
      load_step(t + 1 < N ? t + 1 : N - 1, nxt);   // unconditional (clamped) prefetch
      PIPELINE_FENCE();
      double dx[NX], u[NU], xn[NX];
      bool finite = true;
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = x[i] - cs.xo[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s1 += cs.KK[i * NX + j] * dx[j];
        u[i] = (cs.uo[i] + a_pr * cs.kk[i]) + s1;
        finite = finite && dfinite(u[i]);
      }
      if (t >= kRing && (t % (kRing / 2)) == 0) wait_ge(my_cons, t - kRing / 2);
      {
        double *rs = ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
        for (int i = 0; i < NX; ++i) { rs[i * 64] = x[i]; rs[(NX + i) * 64] = dx[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) rs[(2 * NX + i) * 64] = u[i];
        if (alive && !finite) { s_pstat[wave][lane] = t; alive = false; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(my_prod, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      Stepper<Model>::step(dc, x, u, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) finite = finite && dfinite(xn[i]);
      if (alive && !finite) { s_pstat[wave][lane] = t; alive = false; }
      st<NU>(Un + GI(t, NU, 0), kLS, u);
      st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
      if (alive) {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
      }
    };
    StepIn R[2];
    load_step(0, R[0]);
    prime();
    int t = 0;
    for (; t + 1 < N; t += 2) {
      step(t, R[0], R[1]);
      step(t + 1, R[1], R[0]);
      if (__builtin_amdgcn_ballot_w64(alive) == 0ull) { t = N; break; }   // nothing downstream reads the rows any more
      // the consumer abandoned this trial on every lane (fraction-to-boundary rule): its rows are never read -- stop integrating
      if (__hip_atomic_load(my_cons, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 2 * N) { t = N; break; }
    }
    if (t < N) step(t, R[0], R[1]);
    if (alive) s_pcost[wave][lane] = Obj::terminal_cost(P, x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(my_prod, N + kRing + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }

  // -------------------------------------------------------------------- the consumer of the group's n_here trials
  const double *Sc = d.S + (size_t)cur * d.planeM;
  const double *Yc = d.Y + (size_t)cur * d.planeM;
  const double mu = d.mu[bb];
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  const double adu_max = d.adu_max[bb];
  const bool l2norm = o.ipddp_theta_norm_l2 != 0;
  // per-step record of the CURRENT iterate: the same for every step size of the group -- loaded ONCE per step
  struct StepIn { double xo[Cons::HAS_X ? NX : 1], uo[Cons::NEEDS_U ? NU : 1], s[M], y[M], ksv[M], ky[M], KK[NU * NX], ys[M]; };
  auto load_step = [&](int tt, StepIn &r) {
    if constexpr (Cons::HAS_X) ld<NX>(Xc + GI(tt, NX, 0), kLS, r.xo);
    if constexpr (Cons::NEEDS_U) ld<NU>(d.U + (size_t)cur * d.planeU + GI(tt, NU, 0), kLS, r.uo);
    ld<M>(Sc + GI(tt, M, 0), kLS, r.s);
    ld<M>(Yc + GI(tt, M, 0), kLS, r.y);
    ld<M>(d.ks + GI(tt, M, 0), kLS, r.ksv);
    ld<M>(d.ky + GI(tt, M, 0), kLS, r.ky);
    ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
    ld<M>(d.ys + GI(tt, M, 0), kLS, r.ys);
  };
  typename Cons::Ctx cc;
  Cons::load(P, cc);
  typename Obj::Ctx oc;
  Obj::load_staged(P, oc, s_obj);
  // The body is instantiated per group size NE (the last group of a ladder may be short): inside it every trial index is a
  // compile-time constant and the per-step work of the NE trials is ONE straight-line block -- the NE independent dependency
  // chains (affine slack / dual trials, the logarithms of the barrier terms, the running cost) interleave in the issue slots of the
  // single wave instead of each waiting out its own f64 latency.
  auto run = [&](auto ne_tag) {
    constexpr int NE = decltype(ne_tag)::value;
    double a_pr[NE], a_du[NE], ev_total0[NE], ev_max[NE], ev_icomp[NE], ys_lo[NE], ys_hi[NE], run_cost[NE];
    bool alive[NE], released[NE];
    int fail_t[NE];
    double *Sn[NE], *Yn[NE], *Gn[NE], *Ev[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int a = a_first + j;
      const int slot = trial_slot(cur, a);
      Sn[j] = d.S + (size_t)slot * d.planeM; Yn[j] = d.Y + (size_t)slot * d.planeM; Gn[j] = d.G + (size_t)slot * d.planeM;
      Ev[j] = d.ev + GI((size_t)a * N, 2 * Cons::NSEG, 0);
      const double alpha = P->alphas[a];
      a_pr[j] = dmin(alpha, apr_max); a_du[j] = dmin(alpha, adu_max);
      ev_total0[j] = 0.0; ev_max[j] = 0.0; ev_icomp[j] = 0.0; ys_lo[j] = INFINITY; ys_hi[j] = -INFINITY; run_cost[j] = 0.0;
      alive[j] = active; fail_t[j] = N; released[j] = false;
      if (active) {
        const size_t ti = (size_t)a * d.Bp + bb;
        atomicAdd(d.launched, 1ull);
        d.t_apr[ti] = a_pr[j]; d.t_adu[ti] = a_du[j];
        d.t_success[ti] = 0;
        d.t_cost[ti] = d.cost[b]; d.t_merit[ti] = d.phi[b]; d.t_theta[ti] = d.theta[b];
        d.t_inf_pr[ti] = 0.0; d.t_inf_comp[ti] = 0.0;
      }
    }
    const size_t ev_tstride = (size_t)d.NB * (2 * Cons::NSEG) * kLS;
    {   // prime the VMEM queue with one step's store pattern (see k_forward_ipddp_pc)
      double z[M];
#pragma unroll
      for (int i = 0; i < M; ++i) z[i] = 0.0;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        st<M>(Sn[j] + GI(0, M, 0), kLS, z);
        st<M>(Yn[j] + GI(0, M, 0), kLS, z);
        st<M>(Gn[j] + GI(0, M, 0), kLS, z);
#pragma unroll
        for (int c = 0; c < Cons::NSEG; ++c) { if (c > 0) Ev[j][(size_t)(Cons::NSEG + c) * kLS] = 0.0; Ev[j][(size_t)c * kLS] = 0.0; }
      }
    }
    auto step = [&](const int t, StepIn &cs, StepIn &nxt) {
      load_step(t + 1 < N ? t + 1 : N - 1, nxt);
      PIPELINE_FENCE();
      // ---- phase A: take step t of every producer from its ring, hand the slots back
      double rx[NE][NX], dx[NE][NX], u[NE][NU];
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        wait_ge(&s_prod[j], t + 1);
        const double *rs = s_ring[j] + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
        for (int i = 0; i < NX; ++i) { rx[j][i] = rs[i * 64]; dx[j][i] = rs[(NX + i) * 64]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) u[j][i] = rs[(2 * NX + i) * 64];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (!released[j]) __hip_atomic_store(&s_cons[j], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (alive[j] && s_pstat[j][lane] <= t) { alive[j] = false; fail_t[j] = t; }
      }
      // ---- phase B: one straight-line block for the NE trials
      // rows of K_s, K_y rebuilt from K and YS exactly as k_post forms them (ipddp_solver.cpp:1465-1472): step-size independent
      double Ksr[M * NX], Kyr[M * NX];
      if constexpr (UDiag<Cons>::value) {
#pragma unroll
        for (int r = 0; r < M; ++r) {
          const int ic = UDiag<Cons>::col(r);
          const double gv = UDiag<Cons>::val(cc, r);
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            const double s2 = 0.0 + gv * cs.KK[ic * NX + c];
            const double inner = 0.0 + s2;
            Kyr[r * NX + c] = dmin(dmax(cs.ys[r] * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
            Ksr[r * NX + c] = (-0.0) - s2;
          }
        }
      } else {
        double Gx[M * NX], Gu[M * NU];
#pragma unroll
        for (int i = 0; i < M * NX; ++i) Gx[i] = 0.0;
#pragma unroll
        for (int i = 0; i < M * NU; ++i) Gu[i] = 0.0;
        Cons::template jac<NX, NU>(cc, cs.xo, cs.uo, Gx, Gu);
#pragma unroll
        for (int r = 0; r < M; ++r) {
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            double s2 = 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) s2 += Gu[r * NU + i] * cs.KK[i * NX + c];
            const double inner = Gx[r * NX + c] + s2;
            Kyr[r * NX + c] = dmin(dmax(cs.ys[r] * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
            Ksr[r * NX + c] = (-Gx[r * NX + c]) - s2;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        double sn[M], yn[M];
        bool feas = true;
#pragma unroll
        for (int r = 0; r < M; ++r) {
          sn[r] = affine_2r<NX>(cs.s[r], a_pr[j], cs.ksv[r], Ksr + r * NX, dx[j]);
          yn[r] = affine_2r<NX>(cs.y[r], a_du[j], cs.ky[r], Kyr + r * NX, dx[j]);
          if (sn[r] < (1.0 - tau) * cs.s[r] || yn[r] < (1.0 - tau) * cs.y[r]) feas = false;
          if (!dfinite(sn[r]) || !dfinite(yn[r])) feas = false;
        }
        fail_t[j] = (alive[j] && !feas) ? t : fail_t[j];
        alive[j] = alive[j] && feas;
        st<M>(Sn[j] + GI(t, M, 0), kLS, sn);
        st<M>(Yn[j] + GI(t, M, 0), kLS, yn);
        double g[M];
        Cons::template eval<NX, NU>(cc, rx[j], u[j], g);
        run_cost[j] += Obj::running_cost(oc, xrt, t, rx[j], u[j]);   // same t-ordered sum as the fused rollout (:1726-1748)
        st<M>(Gn[j] + GI(t, M, 0), kLS, g);
        double *ev = Ev[j] + (size_t)t * ev_tstride;
#pragma unroll
        for (int c = 0; c < Cons::NSEG; ++c) {
          const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
          double n1 = 0.0, ninf = 0.0, ls = 0.0;
          for (int i = 0; i < dim; ++i) {
            const double r = g[off + i] + sn[off + i];
            n1 += l2norm ? r * r : fabs(r);
            ninf = dmax(ninf, fabs(r));
            const double ysp = yn[off + i] * sn[off + i];
            ev_icomp[j] = dmax(ev_icomp[j], fabs(ysp - mu));
            ys_lo[j] = dmin(ys_lo[j], ysp); ys_hi[j] = dmax(ys_hi[j], ysp);
            ls += solver_log(dmax(sn[off + i], kEpsSlack));
          }
          ev_max[j] = dmax(ev_max[j], ninf);
          if (c == 0) ev_total0[j] += n1; else ev[(size_t)(Cons::NSEG + c) * kLS] = n1;
          ev[(size_t)c * kLS] = ls;
        }
      }
    };
    // a trial that has failed on every lane of the tile: release (and thereby stop) its producer; when every trial of the group has,
    // keep the step counts and leave
    auto retire_dead = [&]() {
      bool all = true;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (!released[j] && __builtin_amdgcn_ballot_w64(alive[j]) == 0ull) {
          __hip_atomic_store(&s_cons[j], 2 * N + kRing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          released[j] = true;
        }
        all = all && released[j];
      }
      return all;
    };
    {
      StepIn R[2];
      load_step(0, R[0]);
      int t = 0;
      for (; t + 1 < N; t += 2) {
        step(t, R[0], R[1]);
        step(t + 1, R[1], R[0]);
        if (retire_dead()) {
#pragma unroll
          for (int j = 0; j < NE; ++j) if (active) d.t_steps[(size_t)(a_first + j) * d.Bp + bb] = fail_t[j];
          return;
        }
      }
      if (t < N) step(t, R[0], R[1]);
    }
    // ---- after the rollouts: the reductions and the filter test of each trial, in the reference's order
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int a = a_first + j;
      const size_t ti = (size_t)a * d.Bp + bb;
      wait_ge(&s_prod[j], N + kRing + 1);
      if (active) d.t_steps[ti] = fail_t[j];
      bool ok = alive[j];
      if (ok && s_pstat[j][lane] <= N) ok = false;
      // (no early return: the other trials of the group still have their tails to run; a dead lane stores nothing)
      const double cost_new = run_cost[j] + s_pcost[j][lane];   // + l_f(x_N)
      double total = ev_total0[j], mer = cost_new;
      const double *evb = Ev[j];
      if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
        for (int c = 1; c < Cons::NSEG; ++c) {
          const double *q = evb + (size_t)(Cons::NSEG + c) * kLS;
          int t = 0;
          for (; t + 3 < N; t += 4) {
            const double v0 = q[(size_t)t * ev_tstride], v1 = q[(size_t)(t + 1) * ev_tstride], v2 = q[(size_t)(t + 2) * ev_tstride], v3 = q[(size_t)(t + 3) * ev_tstride];
            total += v0; total += v1; total += v2; total += v3;
          }
          for (; t < N; ++t) total += q[(size_t)t * ev_tstride];
        }
        for (int c = 0; c < Cons::NSEG; ++c) {
          const double *q = evb + (size_t)c * kLS;
          int t = 0;
          for (; t + 3 < N; t += 4) {
            const double v0 = q[(size_t)t * ev_tstride], v1 = q[(size_t)(t + 1) * ev_tstride], v2 = q[(size_t)(t + 2) * ev_tstride], v3 = q[(size_t)(t + 3) * ev_tstride];
            mer -= mu * v0; mer -= mu * v1; mer -= mu * v2; mer -= mu * v3;
          }
          for (; t < N; ++t) mer -= mu * q[(size_t)t * ev_tstride];
        }
      }
      const double th = l2norm ? sqrt(total) : total;
      const double theta_new = dmax(th, ev_max[j]), phi_new = mer, ipr = ev_max[j], icomp = ev_icomp[j];
      if (!dfinite(phi_new) || !dfinite(theta_new) || !dfinite(ipr) || !dfinite(icomp)) ok = false;
      if (ok) {
        bool accept = false;
        {   // filter acceptance, ipddp_solver.cpp:1793-1834
          const double expected_improvement = a_pr[j] * d.dV0[b];
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
        d.t_ysmin[ti] = ys_lo[j]; d.t_ysmax[ti] = ys_hi[j];
        d.t_success[ti] = accept ? 1 : 0;
      }
    }
  };
  if (n_here == NA) run(std::integral_constant<int, NA>{});
  else if constexpr (NA >= 2) {
    if (n_here == NA - 1) run(std::integral_constant<int, NA - 1>{});
    else if constexpr (NA >= 3) { if (n_here == NA - 2) run(std::integral_constant<int, NA - 2>{}); }
  }
}

#undef GI
}  // namespace cddp_dev
