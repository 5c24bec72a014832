// Lane-cooperative IPDDP Riccati sweep: G lanes per trajectory, lane q owns COLUMN q of V_xx.
//
// The per-lane sweep (k_backward_ipddp_lean) runs ~740 f64 instructions per step on ONE wavefront per 64
// trajectories: 64 wavefronts for the 4096-trajectory C2 batch, 6 % of the SIMDs, each bound by its own
// instruction stream (a dependent f64 op costs ~9 cycles, an independent one ~5.4 on gfx950 -- profiles/ubench).
// Here the small dense products of one step are split across the G lanes of a trajectory group:
//
//   round 1  T1[:,q] = A^T V_xx[:,q], T2[:,q] = B^T V_xx[:,q], Q_x[q]                  -> LDS
//   round 2  Q_xx[:,q] = l_xx[:,q] + T1 A[:,q], Q_ux[:,q] = T2 A[:,q], Q_uu (replicated), factor (replicated),
//            k (replicated), K[:,q]                                                   -> LDS
//   round 3  V_x[q], Vn[:,q] = Q_xx[:,q] + K^T Q_ux[:,q] + Q_ux^T K[:,q] + K^T Q_uu K[:,q] -> LDS
//            V_xx[:,q] = (Vn[:,q] + Vn[q,:]^T) / 2
//
// Every output element is still accumulated by ONE lane in the reference's order (same sums, same association
// as k_backward_ipddp_lean / ipddp_solver.cpp:1392-1508), so the sweep stays bit-identical; only the column
// loops are spread over lanes.  The wavefront is its own workgroup, so an LDS round trip needs no barrier,
// only lgkmcnt(0).  Lanes q >= NX (when G > NX) shadow column NX-1 and write the same values to the same places.
#pragma once
#include "kernels_lean.hpp"
#include "kernels_logddp.hpp"   // LgCons: the relaxed barrier's gradients / Hessians (the LogDDP mode of the plain cooperative sweep)
#include <type_traits>
#include <utility>

#ifndef CDDP_DX_PREFETCH_DEEP
#define CDDP_DX_PREFETCH_DEEP 1
#endif

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

template <class Model>
struct CoopCfg {
  static constexpr int NX = Model::NX, NU = Model::NU;
  static constexpr int G = NX <= 4 ? 4 : (NX <= 8 ? 8 : 16);   // lanes per trajectory
  static constexpr int TPW = 64 / G;                            // trajectories per wavefront
  static constexpr int oT1 = 0, oT2 = oT1 + NX * NX, oKK = oT2 + NU * NX, oQux = oKK + NU * NX, oVn = oQux + NU * NX,
                       oVx = oVn + NX * NX, oDx = oVx + NX, RAW = oDx + NX;
  static constexpr int STRIDE = (RAW + 31) / 32 * 32 + 4;       // consecutive trajectories start 8 banks apart
};

DEV void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the value `v` holds in lane P of the caller's quad (DPP quad_perm broadcast: a VALU operand modifier, no LDS)
template <int P> DEV double quad_bcast(double v) {
  constexpr int ctrl = P | (P << 2) | (P << 4) | (P << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int N> DEV void quad_gather(double v, double *out) {   // out[k] = v of quad lane k, k < N <= 4
  out[0] = quad_bcast<0>(v);
  if constexpr (N > 1) out[1] = quad_bcast<1>(v);
  if constexpr (N > 2) out[2] = quad_bcast<2>(v);
  if constexpr (N > 3) out[3] = quad_bcast<3>(v);
}

// XCD-aware block -> trajectory-group map of the cooperative sweeps.
// A single-wave workgroup with G lanes per trajectory holds TPW = 64 / G trajectories, i.e. it touches TPW * 8 bytes of every
// 512-B row of the wave-tiled stacks: 32 B at G = 16.  The BPT = 64 / TPW workgroups that share the rows of one 64-trajectory
// tile are consecutive block indices, and consecutive blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8,
// MI355X_MICROARCH.md "Workgroup dispatch"): every 128-B line was then fetched by up to four different L2s (round 2's counters:
// 28 GB of fabric traffic per launch of the C5 sweep against 8.5 GB algorithmic).  Here block bid of XCD x = bid % 8 takes the
// groups of tile (8 * super + x), so that all sharers of a line sit behind ONE L2 and run at the same time (same dispatch
// round).  The grid is rounded up to whole super-groups of 8 * BPT blocks; surplus blocks see b >= B and leave.  Pure
// placement: which block computes a trajectory does not enter its arithmetic (bitwise tests unchanged).
template <int TPW> DEV int coop_group(int bid, int enable) {
  constexpr int BPT = 64 / TPW;            // blocks per 64-trajectory tile
  if (BPT < 8 || !enable) return bid;      // >= 128 B per block and row already (G <= 4): lines are not shared
  constexpr int SUPER = 8 * BPT;
  const int sup = bid / SUPER, r = bid - sup * SUPER;
  const int xcd = r & 7, j = r >> 3;
  return (sup * 8 + xcd) * BPT + j;
}
template <int TPW> inline unsigned coop_grid(int B, int enable) {
  constexpr int BPT = 64 / TPW;
  const unsigned n = (unsigned)((B + TPW - 1) / TPW);
  if (BPT < 8 || !enable) return n;
  constexpr unsigned SUPER = 8 * BPT;
  return (n + SUPER - 1) / SUPER * SUPER;
}

// Role-split form (round 6), NH > 0: the workgroup is the recursion wavefront (wave 0: the code of the single-wave kernel, unchanged
// arithmetic and order) plus NH HELPER wavefronts on the otherwise idle SIMDs of the CU.
//   phase 1 (backward in t)  helpers evaluate what k_condense<.., true> evaluates -- A_t, B_t and every V-independent condensation term
//            (condense_eval: the same device function) -- for blocks of SPB = 64 / TPW steps x TPW trajectories (one lane per
//            (trajectory, step)), and hand the records to the recursion through an LDS ring of RB blocks; A_t, B_t also go to global
//            memory (the dX rollout and the host getter read them); the condensed-term stack is never written or re-read.
//   phase 2 (forward in t)   the recursion wave runs the linear-policy rollout dX as before.
// Protocol (LDS words, polled with s_sleep): s_ready[slot] = global block number + 1 once a block is in the ring; s_ret = steps the
// recursion has taken into registers (a block's slot is rewritten only after that block is retired); s_verdict = passes requested so
// far (a failed factorisation restarts the sweep with a larger regularisation: the helpers produce the blocks again) or kDone;
// s_hdone counts helpers whose global stores are released.  Every lane of the recursion wave walks all N steps of a pass (a lane whose
// factorisation failed idles through the rest of the pass), so block accounting is wave-uniform.
#ifdef CDDP_ROLES_TIMING   // experiment: stamps of the LAST role-split sweep launch (profiles/scripts/roles_times.py), 16 words per workgroup
__device__ unsigned long long g_roles_times[4096 * 16];
#define ROLES_STAMP(i) do { if (lane == 0 && blockIdx.x < 4096) g_roles_times[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#define ROLES_ACC(i, dt) do { if (lane == 0 && blockIdx.x < 4096) g_roles_times[(size_t)blockIdx.x * 16 + (i)] += (dt); } while (0)
#define ROLES_ZERO(i) do { if (lane == 0 && blockIdx.x < 4096) g_roles_times[(size_t)blockIdx.x * 16 + (i)] = 0ull; } while (0)
#else
#define ROLES_STAMP(i)
#define ROLES_ACC(i, dt)
#define ROLES_ZERO(i)
#endif
template <class Model, class Cons, int RB>
struct RoleCfg {
  typedef CoopCfg<Model> C;
  typedef CstLayout<Model, Cons> L;
  static constexpr int NX = Model::NX, NU = Model::NU;
  static constexpr int oA = 0, oB = oA + NX * NX, oC = oB + NX * NU, REC = oC + L::SIZE;
  static constexpr int RECP = REC | 1;          // odd record stride: the 16 / 8 trajectories of a step start in distinct LDS banks
  static constexpr int SPB = 64 / C::TPW;       // steps per block (= helper lanes per trajectory)
  static constexpr int R = SPB * RB;            // ring capacity in steps
  static constexpr int RING = R * C::TPW * RECP;
  static constexpr int kDone = 1 << 30;
};

template <class Model, class Cons, int NH = 0, int RB = 2>
__global__ __launch_bounds__(64 * (1 + NH)) void k_backward_ipddp_coop(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                            int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  typedef CoopCfg<Model> C;
  typedef RoleCfg<Model, Cons, RB> RC;
  constexpr int CST = L::SIZE;
  constexpr bool ROLES = NH > 0;
  // kQuad: at G = 4 the lanes of a trajectory are a DPP quad, so rounds 1, 2 and the rollout's dx exchange can be quad broadcasts
  // instead of LDS rounds (bitwise equal; tests/test_gpu_parity.py ran green with it).  MEASURED on MI355X (round 3,
  // profiles/r03_element_sweep.md): no gain -- sweep class 16.5 ms vs 16.2 ms per C2 solve, C3 44.5 vs 42.8, CLDDP 19.5 vs 18.5.
  // The kernel is bound by VALU issue (SQ_ACTIVE_INST_ANY 56 % of the wave cycles, 5.7 cycles per instruction, un-fused f64
  // multiplies and adds), and the ~30 v_mov_dpp that replace ~20 LDS operations are more instructions, not fewer.  Off.
  constexpr bool kQuad = false;
  __shared__ double lds[C::TPW * C::STRIDE];
  [[maybe_unused]] __shared__ double s_ring[ROLES ? RC::RING : 1];
  [[maybe_unused]] __shared__ int s_ready[ROLES ? RB : 1];
  [[maybe_unused]] __shared__ int s_ret, s_verdict, s_hdone;
  // phase 2: which trajectories take the linear-policy rollout, the dX ring's counters, the step caps (IEEE bit patterns of values >= 0
  // order like unsigned integers: ds_min_u64)
  [[maybe_unused]] __shared__ int s_need[ROLES ? C::TPW : 1], s_phase2, s_dxprod, s_cur[ROLES ? NH : 1], s_pdone, s_take;
  [[maybe_unused]] __shared__ unsigned long long s_aprv[ROLES ? C::TPW : 1], s_aduv[ROLES ? C::TPW : 1];
  // the phase-1 ring is dead by then: its memory carries dx_t for RD steps
  [[maybe_unused]] constexpr int RD = ROLES ? (RC::RING / (C::TPW * NX)) / RC::SPB * RC::SPB : 1;
  static_assert(!ROLES || RD >= 2 * RC::SPB, "dX ring shorter than two blocks");
  const int lane = threadIdx.x & 63;
  [[maybe_unused]] const int wave = ROLES ? (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6) : 0;
  const bool helper = ROLES && wave > 0;
  const int q = lane % C::G, tl = helper ? lane % C::TPW : lane / C::G;
  const int qc = q < NX ? q : NX - 1;
  const int b = coop_group<C::TPW>((int)blockIdx.x, d.xcd_map) * C::TPW + tl;
  const ProblemDev *__restrict__ P = Pk;
  const int N = d.N;
  [[maybe_unused]] auto wait_ge = [&](int *ctr, int need) -> int {
    int v;
    while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    return v;
  };
  // Phase 2 of the role-split form: what k_post evaluates, in blocks of SPB steps x TPW trajectories (lane = (step offset so, trajectory)).
  // Blocks are handed out through ONE packed LDS counter (low half: next block from the front, high half: next block from the back): the
  // helpers take from the front, behind the dX rollout; the recursion wave, once its rollout is finished, takes from the back as long as
  // the dX ring still holds the block (steps >= N - RD).  A block is taken exactly once (the add / subtract returns the old pair; the take
  // is valid while front <= back).  The next block is taken -- and its rows requested -- before the current one is reduced.
  [[maybe_unused]] auto post_phase = [&](const int so, const bool need, const double mu, const int cur, const bool front, const int hidx) {
    if constexpr (ROLES) {
      constexpr int M = Cons::M;
      constexpr int kNone = 1 << 29;
      const int nblk = (N + RC::SPB - 1) / RC::SPB;
      const double *Xc = d.X + (size_t)cur * d.planeX;
      const double *Uc = d.U + (size_t)cur * d.planeU;
      const double *Sc = d.S + (size_t)cur * d.planeM;
      const double *Yc = d.Y + (size_t)cur * d.planeM;
      const double *Gc = d.G + (size_t)cur * d.planeM;
      const double tau = dmax(P->opt.barrier_min_fraction_to_boundary, 1.0 - mu);
      unsigned long long run_apr = (unsigned long long)__double_as_longlong(1.0), run_adu = run_apr;
      const int i_min = (front || RD >= N) ? 0 : (N - RD + RC::SPB - 1) / RC::SPB;
      int expect = nblk - 1;   // (back) the block the next take would return
      const bool leader = lane == (int)__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1;
      auto take = [&]() -> int {
        if (!front && expect < i_min) return -1;
        int v = 0;
        if (leader) v = front ? __hip_atomic_fetch_add(&s_take, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                              : __hip_atomic_fetch_sub(&s_take, 1 << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v = __builtin_amdgcn_readfirstlane(v);
        const int f = v & 0xffff, bk = v >> 16;
        if (f > bk) return -1;
        expect = bk - 1;
        return front ? f : bk;
      };
      struct PIn { double x[NX], y[M], sv[M], g[M], kk[NU], KK[NU * NX], uj[NU], ys[M]; };
      auto pload = [&](int i, PIn &r) {
        const int t = i * RC::SPB + so;
        if (need && i >= 0 && t < N) {
          ld<NX>(Xc + GI(t, NX, 0), kLS, r.x);
          ld<M>(Yc + GI(t, M, 0), kLS, r.y);
          ld<M>(Sc + GI(t, M, 0), kLS, r.sv);
          ld<M>(Gc + GI(t, M, 0), kLS, r.g);
          ld<NU>(d.k + GI(t, NU, 0), kLS, r.kk);
          ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, r.KK);
          if constexpr (Cons::NEEDS_U) ld<NU>(Uc + GI(t, NU, 0), kLS, r.uj);
          ld<M>(d.ys + GI(t, M, 0), kLS, r.ys);
        }
      };
      auto pproc = [&](const int i, const PIn &in, const int inext) {
        const int t = i * RC::SPB + so;
        const bool valid = need && t < N;
        double ky[M], ksv[M], Ky[M * NX], Ksm[M * NX], ysv[M];
        if (valid) {
#pragma unroll
          for (int r = 0; r < M; ++r) ysv[r] = in.ys[r];
          post_rows<Model, Cons, true>(P, in.x, in.uj, in.y, in.sv, in.g, in.kk, in.KK, mu, ky, ksv, ysv, Ky, Ksm);
          st<M>(d.ky + GI(t, M, 0), kLS, ky);
          st<M>(d.ks + GI(t, M, 0), kLS, ksv);
        }
        const int last = (i + 1) * RC::SPB < N ? (i + 1) * RC::SPB : N;   // dx_0 .. dx_{last - 1} cover the block
        wait_ge(&s_dxprod, last);
        if (valid) {
          const double *slot = s_ring + ((size_t)(t % RD) * C::TPW + tl) * NX;
          double dx[NX];
#pragma unroll
          for (int k = 0; k < NX; ++k) dx[k] = slot[k];
          double apr = 1.0, adu = 1.0;
          post_caps<NX, M>(ky, ksv, Ky, Ksm, in.sv, in.y, dx, tau, apr, adu);
          // (k_post: if (apr < 1.0) atomic_min_pos(...): a NaN cap is dropped, a negative one counts as 0)
          if (apr < 1.0) { const unsigned long long v = (unsigned long long)__double_as_longlong(apr >= 0.0 ? apr : 0.0); run_apr = v < run_apr ? v : run_apr; }
          if (adu < 1.0) { const unsigned long long v = (unsigned long long)__double_as_longlong(adu >= 0.0 ? adu : 0.0); run_adu = v < run_adu ? v : run_adu; }
        }
        if (front) {   // the dX rollout may rewrite the slots of every block below the one this helper works on next
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(&s_cur[hidx], inext >= 0 ? inext : kNone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      PIn pa, pb;
      int i0 = take();
      if (front) __hip_atomic_store(&s_cur[hidx], i0 >= 0 ? i0 : kNone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      pload(i0, pa);
      while (i0 >= 0) {
        const int i1 = take();
        pload(i1, pb);
        PIPELINE_FENCE();
        pproc(i0, pa, i1);
        if (i1 < 0) break;
        const int i2 = take();
        pload(i2, pa);
        PIPELINE_FENCE();
        pproc(i1, pb, i2);
        i0 = i2;
      }
      if (need) {
        __hip_atomic_fetch_min(&s_aprv[tl], run_apr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min(&s_aduv[tl], run_adu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  if constexpr (ROLES) {
    // (this launch replaces k_condense<.., true>, the first kernel of an outer iteration: see k_derivs)
    if (blockIdx.x == 0 && threadIdx.x == 0 && !force) *d.n_active = 0;
    const bool act = (b < d.B) && (force || d.phase[b] == PH_ACTIVE);
    if (__builtin_amdgcn_ballot_w64(act) == 0ull) return;   // the same trajectories in every wave of the workgroup: all leave
    if (threadIdx.x == 0) {
      s_ret = 0; s_verdict = 1; s_hdone = 0; s_phase2 = 0; s_dxprod = 0; s_pdone = 0;
      s_take = (((N + RC::SPB - 1) / RC::SPB - 1) << 16);     // front = 0, back = last block
#pragma unroll
      for (int i = 0; i < NH; ++i) s_cur[i] = 0;              // block a helper works on (nothing below it is still needed by that helper)
#pragma unroll
      for (int i = 0; i < RB; ++i) s_ready[i] = 0;
    }
    __syncthreads();
    const int nblk = (N + RC::SPB - 1) / RC::SPB;
    if (helper) {
      // ---------------------------------------------------------------- helper wavefront: lane = (step offset so, trajectory tl)
      const int hidx = wave - 1;
      const int so = lane / C::TPW;
      if (hidx == 0) { ROLES_STAMP(8); ROLES_ZERO(13); }
      const int cur = act ? d.cur[b] : 0;
      const double *Xc = d.X + (size_t)cur * d.planeX;
      const double *Uc = d.U + (size_t)cur * d.planeU;
      const double *Sc = d.S + (size_t)cur * d.planeM;
      const double *Yc = d.Y + (size_t)cur * d.planeM;
      const double *Gc = d.G + (size_t)cur * d.planeM;
      const double mu = act ? d.mu[b] : 1.0;
      constexpr int M = Cons::M;
      // (the rows of a block are requested one block ahead: a pass of this wave is otherwise one memory round trip + the arithmetic)
      struct HIn { double x[NX], u[NU], y[M], sv[M], g[M]; };
      auto hload = [&](int j, HIn &r) {
        const int t = N - 1 - (j * RC::SPB + so);
        if (act && j < nblk && t >= 0) {
          ld<NX>(Xc + GI(t, NX, 0), kLS, r.x);
          ld<NU>(Uc + GI(t, NU, 0), kLS, r.u);
          ld<M>(Yc + GI(t, M, 0), kLS, r.y);
          ld<M>(Sc + GI(t, M, 0), kLS, r.sv);
          ld<M>(Gc + GI(t, M, 0), kLS, r.g);
        }
      };
      for (int pass = 0;; ++pass) {
        auto hstep = [&](const int j, const HIn &in, HIn &nxt) {
          if (j >= nblk) return;
          hload(j + NH, nxt);
          PIPELINE_FENCE();
          const int gj = pass * nblk + j;
          const int t = N - 1 - (j * RC::SPB + so);
          const bool valid = act && t >= 0;
          double A[NX * NX], Bq[NX * NU], c[CST], ysr[M];
          if (valid) condense_eval<Model, Cons, true>(P, xrt, t, in.x, in.u, in.y, in.sv, in.g, mu, A, Bq, c, ysr);
          if (gj >= RB) wait_ge(&s_ret, (gj - RB + 1) * RC::SPB);   // the slot's previous block has been taken into registers
          if (valid) {
            double *r = s_ring + ((size_t)((gj % RB) * RC::SPB + so) * C::TPW + tl) * RC::RECP;
#pragma unroll
            for (int i = 0; i < NX * NX; ++i) r[RC::oA + i] = A[i];
#pragma unroll
            for (int i = 0; i < NX * NU; ++i) r[RC::oB + i] = Bq[i];
#pragma unroll
            for (int i = 0; i < CST; ++i) r[RC::oC + i] = c[i];
            if (pass == 0) {
#pragma unroll
              for (int i = 0; i < NX * NX; ++i) d.A[GI(t, NX * NX, i)] = A[i];
#pragma unroll
              for (int i = 0; i < NX * NU; ++i) d.Bm[GI(t, NX * NU, i)] = Bq[i];
              st<M>(d.ys + GI(t, M, 0), kLS, ysr);   // Y S^-1 (K3's third row set): mu-, y- and s-dependent only, phase 2 reads it back
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(&s_ready[gj % RB], gj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (hidx == 0 && gj == 0) ROLES_STAMP(9);
        };
        HIn ha, hb;
        hload(hidx, ha);
        for (int j = hidx; j < nblk; j += 2 * NH) { hstep(j, ha, hb); hstep(j + NH, hb, ha); }
        if (hidx == 0 && pass == 0) ROLES_STAMP(10);
        if (wait_ge(&s_verdict, pass + 2) >= RC::kDone) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // A_t, B_t visible to the recursion wave's dX rollout
      if (lane == 0) __hip_atomic_fetch_add(&s_hdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // ---- phase 2 (forward in t): what k_post evaluates, block by block behind the recursion wave's dX rollout
      if (wait_ge(&s_phase2, 1) != 1) return;                  // nobody takes a step (converged / failed): nothing to post
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // K, k of the sweep
      if (hidx == 0) ROLES_STAMP(11);
      post_phase(so, act && s_need[tl] != 0, mu, cur, true, hidx);
      if (hidx == 0) ROLES_STAMP(12);
      if (lane == 0) __hip_atomic_fetch_add(&s_pdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
  }
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  double *Ls = lds + tl * C::STRIDE;
  const cddp_hip_options &o = P->opt;
  if constexpr (ROLES) { ROLES_STAMP(0); ROLES_ZERO(6); ROLES_ZERO(7); }
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  if (count_iter && q == 0) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  // loop-invariant per-lane constants: column qc of l_xx / 2 = Q dt, and R dt
  double Qq[NX], Rr[NU * NU];
  {
    const double *Qp = P->pool + P->off_Qdt, *Rp = P->pool + P->off_Rdt;
#pragma unroll
    for (int i = 0; i < NX; ++i) Qq[i] = Qp[i * NX + qc];
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Rr[i] = Rp[i];
  }
  for (;;) {
    ++nb;
    double Vx[NX], Vc[NX];   // V_x (replicated), V_xx[:, qc]
    {
      double xN[NX];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      Obj::final_grad(P, xN, Vx);
      const double *Qf = P->pool + P->off_Qf;
#pragma unroll
      for (int i = 0; i < NX; ++i) Vc[i] = 0.5 * ((2.0 * Qf[i * NX + qc]) + (2.0 * Qf[qc * NX + i]));
    }
    dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0;
    {
      // V_x(N)[qc]: dynamic element of a replicated register array -> route through LDS
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oVx + i] = Vx[i];
      lds_sync();
      d.Vx[GI(N, NX, qc)] = Ls[C::oVx + qc];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) d.Vxx[GI(N, NX * NX, i * NX + qc)] = Vc[i];
    bool fail = false;
    struct In1 { double A[NX * NX], Aq[NX]; };
    struct In2 { double Bm[NX * NU], cu[NU], WQyu[NU * NU], QyuSir[NU], ipr, icomp, cxq, WQyxq[NU], QyxSirq, WxQyxq[NX]; };
    // ROLES: the step records come from the helpers' LDS ring (slot of step index k = N - 1 - t in this pass)
    [[maybe_unused]] const int pbase = ROLES ? (((nb - 1) * ((N + RC::SPB - 1) / RC::SPB)) % RB) : 0;
    [[maybe_unused]] auto ring_rec = [&](int tt) -> const double * {
      const int k = N - 1 - tt;
      const int slot = ((pbase + k / RC::SPB) % RB) * RC::SPB + (k % RC::SPB);
      return s_ring + ((size_t)slot * C::TPW + tl) * RC::RECP;
    };
    auto load1 = [&](int tt, In1 &r) {
      if constexpr (ROLES) {
        const double *rr = ring_rec(tt) + RC::oA;
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) r.A[i] = rr[i];
#pragma unroll
        for (int j = 0; j < NX; ++j) r.Aq[j] = rr[j * NX + qc];
        return;
      }
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
#pragma unroll
      for (int j = 0; j < NX; ++j) r.Aq[j] = d.A[GI(tt, NX * NX, j * NX + qc)];
    };
    auto load2 = [&](int tt, In2 &r) {
      if constexpr (ROLES) {
        const double *rr = ring_rec(tt);
#pragma unroll
        for (int i = 0; i < NX * NU; ++i) r.Bm[i] = rr[RC::oB + i];
        const double *c = rr + RC::oC;
#pragma unroll
        for (int i = 0; i < NU; ++i) r.cu[i] = c[L::CU + i];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) r.WQyu[i] = c[L::WQYU + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) r.QyuSir[i] = c[L::QYUSIR + i];
        r.ipr = c[L::IPR]; r.icomp = c[L::ICOMP];
        r.cxq = c[L::CX + qc];
        if constexpr (Cons::HAS_X) {
#pragma unroll
          for (int u = 0; u < NU; ++u) r.WQyxq[u] = c[L::WQYX + u * NX + qc];
          r.QyxSirq = c[L::QYXSIR + qc];
#pragma unroll
          for (int i = 0; i < NX; ++i) r.WxQyxq[i] = c[L::WXQYX + i * NX + qc];
        }
        return;
      }
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      const double *c = d.cst + GI(tt, CST, 0);
      ld<NU>(c + (size_t)L::CU * kLS, kLS, r.cu);
      ld<NU * NU>(c + (size_t)L::WQYU * kLS, kLS, r.WQyu);
      ld<NU>(c + (size_t)L::QYUSIR * kLS, kLS, r.QyuSir);
      r.ipr = c[(size_t)L::IPR * kLS]; r.icomp = c[(size_t)L::ICOMP * kLS];
      r.cxq = c[(size_t)(L::CX + qc) * kLS];
      if constexpr (Cons::HAS_X) {
#pragma unroll
        for (int u = 0; u < NU; ++u) r.WQyxq[u] = c[(size_t)(L::WQYX + u * NX + qc) * kLS];
        r.QyxSirq = c[(size_t)(L::QYXSIR + qc) * kLS];
#pragma unroll
        for (int i = 0; i < NX; ++i) r.WxQyxq[i] = c[(size_t)(L::WXQYX + i * NX + qc) * kLS];
      }
    };
    auto step = [&](const int t, const In1 &c1, const In2 &c2, In1 &n1, In2 &n2) -> bool {
      const int tp = t > 0 ? t - 1 : 0;
      load1(tp, n1);
      PIPELINE_FENCE();
      const double (&A)[NX * NX] = c1.A; const double (&Bm)[NX * NU] = c2.Bm;
      // ---- round 1: column qc of T1 = A^T V_xx and T2 = B^T V_xx; row qc of Q_x; Q_u (replicated)
      double T1c[NX], T2c[NU], Qu[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += A[k * NX + i] * Vc[k];
        T1c[i] = s; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bm[k * NU + u] * Vc[k];
        T2c[u] = s; }
      double Qxq;
      { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += c1.Aq[k] * Vx[k];
        Qxq = c2.cxq + s2; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Bm[k * NU + u] * Vx[k];
        Qu[u] = c2.cu[u] + s2; }
      if constexpr (!kQuad) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Ls[C::oT1 + i * NX + qc] = T1c[i];
#pragma unroll
        for (int u = 0; u < NU; ++u) Ls[C::oT2 + u * NX + qc] = T2c[u];
        lds_sync();
      }
      __builtin_amdgcn_sched_barrier(0);
      load2(tp, n2);
      PIPELINE_FENCE();
      __builtin_amdgcn_sched_barrier(0);
      // ---- round 2: column qc of Q_xx, Q_ux; Q_uu, factor, k replicated; column qc of K
      double T1[NX * NX], T2[NU * NX];
      if constexpr (kQuad) {   // G = 4: the lanes of a trajectory are a quad -- column j of T1, T2 is lane j's, fetched by DPP broadcasts
#pragma unroll
        for (int i = 0; i < NX; ++i) quad_gather<NX>(T1c[i], T1 + i * NX);
#pragma unroll
        for (int u = 0; u < NU; ++u) quad_gather<NX>(T2c[u], T2 + u * NX);
      } else {
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) T1[i] = Ls[C::oT1 + i];
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) T2[i] = Ls[C::oT2 + i];
      }
      double Qxxc[NX], Quxc[NU], Quu[NU * NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += T1[i * NX + j] * c1.Aq[j];
        Qxxc[i] = (2.0 * Qq[i]) + s; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += T2[u * NX + j] * c1.Aq[j];
        Quxc[u] = s; }
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int v = 0; v < NU; ++v) { double s = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) s += T2[u * NX + j] * Bm[j * NU + v];
          Quu[u * NU + v] = (2.0 * Rr[u * NU + v]) + s; }
      double Qr[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qr[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]) + c2.WQyu[i * NU + c];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
      double kk[NU], KKc[NU];
      // condensed Q_ux column: the right-hand side of the K solve IS the condensed value (:1453, :1491)
      double Quxq[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double rhs = Quxc[u];
        if constexpr (Cons::HAS_X) rhs = rhs + c2.WQyxq[u];
        Quxq[u] = rhs;
      }
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Qr[0], Qu[0] + c2.QyuSir[0]);
        KKc[0] = -ldlt1_solve(Qr[0], Quxq[0]);
      } else {
        LDLTs<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i] + c2.QyuSir[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Quxq[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) KKc[i] = -col[i];
      }
      if constexpr (!kQuad) {
#pragma unroll
        for (int u = 0; u < NU; ++u) { Ls[C::oKK + u * NX + qc] = KKc[u]; Ls[C::oQux + u * NX + qc] = Quxq[u]; }
        lds_sync();
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
#pragma unroll
      for (int u = 0; u < NU; ++u) d.K[GI(t, NU * NX, u * NX + qc)] = KKc[u];
      // ---- round 3: value update
      double KK[NU * NX], Qux[NU * NX];
      if constexpr (kQuad) {
#pragma unroll
        for (int u = 0; u < NU; ++u) { quad_gather<NX>(KKc[u], KK + u * NX); quad_gather<NX>(Quxq[u], Qux + u * NX); }
      } else {
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) { KK[i] = Ls[C::oKK + i]; Qux[i] = Ls[C::oQux + i]; }
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] += c2.QyuSir[i];
      if constexpr (Cons::HAS_X) {
        Qxq += c2.QyxSirq;
#pragma unroll
        for (int i = 0; i < NX; ++i) Qxxc[i] += c2.WxQyxq[i];
      }
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] += c2.WQyu[i];
      inf_pr = dmax(inf_pr, c2.ipr); inf_comp = dmax(inf_comp, c2.icomp);
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
      double Vxq;
      {
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KKc[j] * Qu[j]; bb += Quxq[j] * kk[j]; }
        // KtQ[qc, :] = sum_u K[u, qc] Q_uu[u, :]: recomputed from the lane's own column (same expression as mm_tn)
#pragma unroll
        for (int j = 0; j < NU; ++j) { double s = 0.0;
#pragma unroll
          for (int u = 0; u < NU; ++u) s += KKc[u] * Quu[u * NU + j];
          c += s * kk[j]; }
        Vxq = ((Qxq + a) + bb) + c;
      }
      double Vnc[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KK[j * NX + i] * Quxq[j]; bb += Qux[j * NX + i] * KKc[j]; e += KtQ[i * NU + j] * KKc[j]; }
        Vnc[i] = ((Qxxc[i] + a) + bb) + e;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oVn + i * NX + qc] = Vnc[i];
      Ls[C::oVx + qc] = Vxq;
      lds_sync();
#pragma unroll
      for (int i = 0; i < NX; ++i) Vc[i] = 0.5 * (Vnc[i] + Ls[C::oVn + qc * NX + i]);
#pragma unroll
      for (int i = 0; i < NX; ++i) Vx[i] = Ls[C::oVx + i];
      lds_sync();   // reads done before the next step overwrites the regions
      d.Vx[GI(t, NX, qc)] = Vxq;
#pragma unroll
      for (int i = 0; i < NX; ++i) d.Vxx[GI(t, NX * NX, i * NX + qc)] = Vc[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(Qu[i])); step_norm = dmax(step_norm, fabs(kk[i])); }
      return true;
    };
    In1 a1, b1;
    In2 a2, b2;
    if constexpr (ROLES) {
      // Block accounting at wave level: before the records of step index k + 1 are fetched (inside step k) their block must be in
      // the ring; once the last step of a block has run, the block is retired.  Every lane still in the pass loop walks all N steps.
      const int nblk = (N + RC::SPB - 1) / RC::SPB;
      const int gb0 = (nb - 1) * nblk;
      auto need_block = [&](int k) {   // the block of step index k (this pass)
        const int j = k / RC::SPB;
#ifdef CDDP_ROLES_TIMING
        const unsigned long long w0 = wall_clock64();
#endif
        wait_ge(&s_ready[(gb0 + j) % RB], gb0 + j + 1);
#ifdef CDDP_ROLES_TIMING
        if (k > 0) ROLES_ACC(6, wall_clock64() - w0);
#endif
      };
      auto retire = [&](int k) {       // step index k has run: its record and the prefetched one are in registers
        if ((k + 1) % RC::SPB == 0 || k == N - 1) {
          lds_sync();
          const int done = (k == N - 1) ? (gb0 + nblk) * RC::SPB : gb0 * RC::SPB + k + 1;
          __hip_atomic_store(&s_ret, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      need_block(0);
      if (nb == 1) ROLES_STAMP(1);
      load1(N - 1, a1);
      load2(N - 1, a2);
      int t = N - 1;
      for (; t >= 1; t -= 2) {
        const int k = N - 1 - t;
        if ((k + 1) % RC::SPB == 0) need_block(k + 1);
        if (!fail) fail = !step(t, a1, a2, b1, b2);
        retire(k);
        if ((k + 2) % RC::SPB == 0 && k + 2 < N) need_block(k + 2);
        if (!fail) fail = !step(t - 1, b1, b2, a1, a2);
        retire(k + 1);
      }
      if (t == 0) {
        if (!fail) fail = !step(0, a1, a2, b1, b2);
        retire(N - 1);
      }
      bool cont = false;
      if (!fail) ok = true;
      else if (force != 2) { reg = reg_increase(o, reg); cont = reg < o.reg_max_value; }
      const bool any_cont = __builtin_amdgcn_ballot_w64(cont) != 0ull;   // (the lanes still in the pass loop)
      __hip_atomic_store(&s_verdict, any_cont ? nb + 1 : RC::kDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!cont) break;
    } else {
    load1(N - 1, a1);
    load2(N - 1, a2);
    int t = N - 1;
    for (; t >= 1; t -= 2) {
      if (!step(t, a1, a2, b1, b2)) { fail = true; break; }
      if (!step(t - 1, b1, b2, a1, a2)) { fail = true; break; }
    }
    if (!fail && t == 0) fail = !step(0, a1, a2, b1, b2);
    if (!fail) { ok = true; break; }
    if (force == 2) break;
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
    }
  }
  if constexpr (ROLES) {   // the helpers' A_t, B_t stores, before the dX rollout reads them
    ROLES_STAMP(2);
    wait_ge(&s_hdone, NH);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  bool conv = false;
  [[maybe_unused]] double apr_cap = 1.0, adu_cap = 1.0;
  if constexpr (ROLES) {
    if (ok) {
      const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
      const double asn = fabs(d.alpha_pr[b]) * step_norm;
      const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
      conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
    }
    const bool need = ok && (!conv || force);   // the trajectories k_post serves: phase FWD1 / (force) a successful sweep
    const bool any_need = __builtin_amdgcn_ballot_w64(need) != 0ull;
    if (q == 0) {
      s_need[tl] = need ? 1 : 0;
      s_aprv[tl] = (unsigned long long)__double_as_longlong(1.0); s_aduv[tl] = (unsigned long long)__double_as_longlong(1.0);
    }
    double *dxr = s_ring;
    dxr[((size_t)0 * C::TPW + tl) * NX + qc] = 0.0;              // dx_0 = 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // K, k of the sweep visible to the helpers
    lds_sync();
    __hip_atomic_store(&s_dxprod, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&s_phase2, any_need ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ROLES_STAMP(3);
    if (any_need) {
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): lane qc computes row qc of dx_{t+1}; every lane of the wave walks the
      // horizon (a trajectory that takes no step computes on its stale gains: nobody reads its rows), so the ring counters are wave-uniform
      double dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = 0.0;
      // One wave sustains ~16 row loads per memory round trip (profiles/ubench/vmem.hip), and that -- not the ~90-cycle arithmetic of a
      // step -- paces this loop (round 6 stamps: 0.19 us per step at C2 with NU + NU NX + NX + NU = 10 loads per step, 0.5 at C3 with 13).
      // At G = 4 the gains, which every lane of a trajectory needs whole, are therefore fetched ONE COLUMN PER LANE (lane q: K[:, q] and
      // k[q]) and spread over the quad by DPP broadcasts: NU + 1 + NX + NU loads per step instead of NU + NU NX + NX + NU.
      constexpr bool kQ = C::G == 4;
      constexpr bool kQL = kQ && NU <= 4;
      struct RIn { double kk[kQL ? 1 : NU], KK[kQL ? NU : NU * NX], Aq[NX], Bq[NU]; };
      const int qk = q < NU ? q : NU - 1;   // (kQL) the entry of k this lane fetches
      auto load_r = [&](int tt, RIn &r) {
        if constexpr (kQL) {
          r.kk[0] = d.k[GI(tt, NU, qk)];
#pragma unroll
          for (int i = 0; i < NU; ++i) r.KK[i] = d.K[GI(tt, NU * NX, i * NX + qc)];
        } else {
          ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
          ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) r.Aq[j] = d.A[GI(tt, NX * NX, qc * NX + j)];
#pragma unroll
        for (int j = 0; j < NU; ++j) r.Bq[j] = d.Bm[GI(tt, NX * NU, qc * NU + j)];
      };
      auto clampt = [&](int tt) { return tt < N - 1 ? tt : (N - 2 > 0 ? N - 2 : 0); };
      // A step is ~150 cycles of dependent arithmetic; the rows of step t + D - 1 are requested while step t runs (D register sets), so
      // that a memory round trip (0.4 - 1 us under load) is covered; at G = 4 the lanes of a trajectory are a DPP quad and exchange the
      // rows of dx by quad broadcast -- the LDS write only feeds the ring and is published one step late, when it has long landed
      // (round 6 stamps, profiles/r06_sweep_roles.md: 0.37 us per step with D = 4 and two LDS round trips on the chain).
      // (D - 1 steps of rows in flight: at most ~56 loads, the vmcnt counter holds 63; the main loop has NO branch around a load --
      //  the waitcnt pass merges the states at every join and would otherwise wait for all but the newest step's rows)
      constexpr int kRows = (int)(sizeof(RIn) / sizeof(double));
#ifdef CDDP_ROLES_D
      constexpr int D = CDDP_ROLES_D;
#else
      constexpr int D = kRows > 28 ? 2 : (1 + 56 / kRows > 8 ? 8 : 1 + 56 / kRows);
#endif
      auto dstep = [&](const int t, const RIn &rc, RIn &rl, const bool fetch) {
        if (fetch) { load_r(clampt(t + D - 1), rl); PIPELINE_FENCE(); }
        lds_sync();                                                    // dx_t went into the ring one step ago
        __hip_atomic_store(&s_dxprod, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        double du[NU];
        if constexpr (kQL) {
          double kv[4];
          quad_gather<NU>(rc.kk[0], kv);
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double Kr[4];
            quad_gather<NX>(rc.KK[i], Kr);
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += Kr[j] * dx[j];
            du[i] = kv[i] + a;
          }
        } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) a += rc.KK[i * NX + j] * dx[j];
          du[i] = rc.kk[i] + a; }
        }
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) a += rc.Aq[j] * dx[j];
#pragma unroll
        for (int j = 0; j < NU; ++j) c += rc.Bq[j] * du[j];
        const double dxq = (a + c) + 0.0;
        const int w = t + 1;
        if (w % RC::SPB == 0 && w >= RD) {   // the slot's previous tenant (step w - RD and its block) has been read by every helper
#pragma unroll
          for (int hh = 0; hh < NH; ++hh) wait_ge(&s_cur[hh], (w - RD) / RC::SPB + 1);
        }
        double *slot = dxr + ((size_t)(w % RD) * C::TPW + tl) * NX;
        slot[qc] = dxq;
        if constexpr (kQ) quad_gather<NX>(dxq, dx);
        else {
          lds_sync();
#pragma unroll
          for (int i = 0; i < NX; ++i) dx[i] = slot[i];
        }
      };
      {
        RIn R[D];
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load_r(clampt(j), R[j]);
        int t = 0;
        for (; t + D <= N - 1; t += D) {
#pragma unroll
          for (int j = 0; j < D; ++j) dstep(t + j, R[j], R[(j + D - 1) % D], true);
        }
#pragma unroll
        for (int j = 0; j < D - 1; ++j) if (t + j < N - 1) dstep(t + j, R[j], R[j], false);   // (their rows are in the sets already)
      }
      lds_sync();
      __hip_atomic_store(&s_dxprod, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ROLES_STAMP(4);
      post_phase(q, need, mu, cur, false, -1);   // the rollout is done: this wave takes blocks from the back
      ROLES_STAMP(14);
      wait_ge(&s_pdone, NH);
      ROLES_STAMP(5);
      lds_sync();
      if (need) { apr_cap = __longlong_as_double((long long)s_aprv[tl]); adu_cap = __longlong_as_double((long long)s_aduv[tl]); }
    }
  } else
  if (ok) {
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
    if (!conv || force) {
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): lane qc computes row qc of dx_{t+1}
      double dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = 0.0;
      struct RIn { double kk[NU], KK[NU * NX], Aq[NX], Bq[NU]; };
      auto load_r = [&](int tt, RIn &r) {
        ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
        ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
#pragma unroll
        for (int j = 0; j < NX; ++j) r.Aq[j] = d.A[GI(tt, NX * NX, qc * NX + j)];
#pragma unroll
        for (int j = 0; j < NU; ++j) r.Bq[j] = d.Bm[GI(tt, NX * NU, qc * NU + j)];
      };
      double dxown = 0.0;   // (kQuad) this lane's row of dx
      auto rstep = [&](const int t, const RIn &rc, RIn &rn) {
        const int tn = t + 1 < N - 1 ? t + 1 : t;
        load_r(tn, rn);
        PIPELINE_FENCE();
        if constexpr (kQuad) d.dX[GI(t, NX, qc)] = dxown; else d.dX[GI(t, NX, qc)] = Ls[C::oDx + qc];
        if (t < N - 1) {
          double du[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += rc.KK[i * NX + j] * dx[j];
            du[i] = rc.kk[i] + a; }
          double a = 0.0, c = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) a += rc.Aq[j] * dx[j];
#pragma unroll
          for (int j = 0; j < NU; ++j) c += rc.Bq[j] * du[j];
          const double dxq = (a + c) + 0.0;
          if constexpr (kQuad) {   // rows of dx by quad broadcast: no LDS on this chain
            dxown = dxq;
            quad_gather<NX>(dxq, dx);
          } else {
            lds_sync();              // every lane has read the old dx row
            Ls[C::oDx + qc] = dxq;
            lds_sync();
#pragma unroll
            for (int i = 0; i < NX; ++i) dx[i] = Ls[C::oDx + i];
          }
        }
      };
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oDx + i] = 0.0;
      lds_sync();
      // (round 5) the rows of step t + 3 are requested while step t runs: a step is ~150 cycles of dependent arithmetic, a fetch one
      // memory round trip -- with the rows of step t + 1 requested at the top of step t, every step waited for its own round trip
      constexpr bool kDeep = CDDP_DX_PREFETCH_DEEP && sizeof(RIn) <= 24 * sizeof(double);
      if constexpr (kDeep) {
        auto dstep = [&](const int t, const RIn &rc, RIn &rl) {
          if (t >= N) return;
          const int tn = t + 3 < N - 1 ? t + 3 : (N - 2 > 0 ? N - 2 : 0);
          load_r(tn, rl);
          PIPELINE_FENCE();
          if constexpr (kQuad) d.dX[GI(t, NX, qc)] = dxown; else d.dX[GI(t, NX, qc)] = Ls[C::oDx + qc];
          if (t < N - 1) {
            double du[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
              for (int j = 0; j < NX; ++j) a += rc.KK[i * NX + j] * dx[j];
              du[i] = rc.kk[i] + a; }
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += rc.Aq[j] * dx[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) c += rc.Bq[j] * du[j];
            const double dxq = (a + c) + 0.0;
            if constexpr (kQuad) {
              dxown = dxq;
              quad_gather<NX>(dxq, dx);
            } else {
              lds_sync();
              Ls[C::oDx + qc] = dxq;
              lds_sync();
#pragma unroll
              for (int i = 0; i < NX; ++i) dx[i] = Ls[C::oDx + i];
            }
          }
        };
        auto clampt = [&](int tt) { return tt < N - 1 ? tt : (N - 2 > 0 ? N - 2 : 0); };
        RIn r0, r1, r2, r3;
        load_r(0, r0); load_r(clampt(1), r1); load_r(clampt(2), r2);
        for (int t = 0; t < N; t += 4) { dstep(t, r0, r3); dstep(t + 1, r1, r0); dstep(t + 2, r2, r1); dstep(t + 3, r3, r2); }
      } else {
      RIn ra, rb;
      load_r(0, ra);
      int t = 0;
      for (; t + 1 < N; t += 2) { rstep(t, ra, rb); rstep(t + 1, rb, ra); }
      if (t < N) rstep(t, ra, rb);
      }
    }
  }
  if (q != 0) return;
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  if constexpr (ROLES) { d.apr_max[b] = apr_cap; d.adu_max[b] = adu_cap; }   // the helpers' minima over the horizon (1.0: no step taken)
  else { d.apr_max[b] = 1.0; d.adu_max[b] = 1.0; }        // K3 lowers them by atomic min
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du; d.step_norm[b] = step_norm;
    d.inf_pr[b] = inf_pr; d.inf_comp[b] = inf_comp;
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, mu); return; }
  d.phase[b] = PH_FWD1;
}

// ================================================================================ cooperative sweep, no path duals
// The same column-per-lane decomposition for the two sweeps that carry no slack / dual blocks:
//   CLDDP = true   clddp_solver.cpp:79-204 (regularisation only in the factorised matrix, EigenSolver PD test,
//                  PartialPivLU inverse or BoxQP + LDLT of the free block, un-symmetrised terminal V_xx)
//   CLDDP = false  the unconstrained IPDDP branch, ipddp_solver.cpp:1048-1118 (regularised, symmetrised Q_uu kept
//                  in the value update)
// Replicated per trajectory group: Q_u, Q_uu, the PD test / factor / BoxQP, k, dV.  Per lane: column qc of T1, T2,
// Q_xx, Q_ux, K, V_xx and row qc of l_x, Q_x, V_x.  Same sums, same association as the fused per-lane kernels.
//   LG != void     LogDDP (round 4; CLDDP = true selects the value-update association it shares with CLDDP): logddp_solver.cpp:363-590 --
//                  the relaxed barrier's gradients / Hessians of the constraint list LG folded into Q_x, Q_u, Q_xx, Q_ux, Q_uu in
//                  the reference's order (each lane evaluates the rows; it keeps its own column), LDLT of the regularised,
//                  symmetrised Q_uu, un-regularised Q_uu in the value update, raw max |Q_u|, RegularizationLimitReached_Converged;
//                  same sums, same association as the one-lane k_backward_logddp (tests/test_logddp_device.py compares them bitwise)
template <class Model, bool CLDDP, class LG = void>
__global__ __launch_bounds__(64) void k_backward_coop_plain(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                            int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr bool LOGDDP = !std::is_void<LG>::value;
  static_assert(!LOGDDP || CLDDP, "the LogDDP mode shares the CLDDP branch of the value update");
  typedef typename std::conditional<LOGDDP, LG, ConList<>>::type LCons;
  constexpr int LM = LCons::M, LMM = LM > 0 ? LM : 1;
  typedef Objective<NX, NU> Obj;
  typedef CoopCfg<Model> C;
  constexpr bool kQuad = false;   // see k_backward_ipddp_coop (measured: no gain)
  __shared__ double lds[C::TPW * C::STRIDE];
  const int lane = threadIdx.x;
  const int q = lane % C::G, tl = lane / C::G;
  const int qc = q < NX ? q : NX - 1;
  const int b = coop_group<C::TPW>((int)blockIdx.x, d.xcd_map) * C::TPW + tl;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  double *Ls = lds + tl * C::STRIDE;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  if (count_iter && q == 0) d.iter[b] += 1;
  double reg = d.reg[b];
  const int box = (CLDDP && !LOGDDP) ? P->clddp_box : -1;
  [[maybe_unused]] const double lg_mu = LOGDDP ? d.mu[b] : 0.0, lg_delta = o.logddp_relaxed_delta;
  typename LCons::Ctx lcc;
  LCons::load(P, lcc);
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, step_norm = 0;
  // loop-invariant scalars of the control-limited step, fetched ONCE: inside the step they are scalar loads in conditionally executed
  // code, i.e. dependent load + s_waitcnt round trips on the chain of every step (dev_boxqp.hpp::BoxQPConst)
  BoxQPConst qpc; qpc.load(o);
  [[maybe_unused]] double box_lo[NU], box_hi[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) { box_lo[i] = box >= 0 ? P->pool[P->cons[box].off_lower + i] : 0.0; box_hi[i] = box >= 0 ? P->pool[P->cons[box].off_upper + i] : 0.0; }
  // loop-invariant per-lane constants: column / row qc of Q dt, R dt, the goal state
  double Qq[NX], Qrow[NX], Rr[NU * NU], xg[NX];
  {
    const double *Qp = P->pool + P->off_Qdt, *Rp = P->pool + P->off_Rdt, *xp = P->pool + P->off_xref;
#pragma unroll
    for (int i = 0; i < NX; ++i) { Qq[i] = Qp[i * NX + qc]; Qrow[i] = Qp[qc * NX + i]; xg[i] = xp[i]; }
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Rr[i] = Rp[i];
  }
  for (;;) {
    ++nb;
    double Vx[NX], Vc[NX];   // V_x (replicated), V_xx[:, qc]
    {
      double xN[NX];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      Obj::final_grad(P, xN, Vx);
      const double *Qf = P->pool + P->off_Qf;
#pragma unroll
      for (int i = 0; i < NX; ++i)
        Vc[i] = (CLDDP && !LOGDDP) ? 2.0 * Qf[i * NX + qc] : 0.5 * ((2.0 * Qf[i * NX + qc]) + (2.0 * Qf[qc * NX + i]));
    }
    dV0 = 0; dV1 = 0; inf_du = 0; step_norm = 0;
    {
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oVx + i] = Vx[i];
      lds_sync();
      d.Vx[GI(N, NX, qc)] = Ls[C::oVx + qc];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) d.Vxx[GI(N, NX * NX, i * NX + qc)] = Vc[i];
    double norm_Vx = 0.0, Qu_error = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) norm_Vx += fabs(Vx[i]);
    bool fail = false;
    struct In1 { double A[NX * NX], Aq[NX]; };
    struct In2 { double Bm[NX * NU], x[NX], u[NU], k0[NU]; };
    auto load1 = [&](int tt, In1 &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
#pragma unroll
      for (int j = 0; j < NX; ++j) r.Aq[j] = d.A[GI(tt, NX * NX, j * NX + qc)];
    };
    auto load2 = [&](int tt, In2 &r) {
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<NX>(Xc + GI(tt, NX, 0), kLS, r.x);
      ld<NU>(Uc + GI(tt, NU, 0), kLS, r.u);
      if constexpr (CLDDP && !LOGDDP) ld<NU>(d.k + GI(tt, NU, 0), kLS, r.k0);   // BoxQP warm start x0 = k_u_[t] of the previous iteration
    };
    auto step = [&](const int t, const In1 &c1, const In2 &c2, In1 &n1, In2 &n2) -> bool {
      const int tp = t > 0 ? t - 1 : 0;
      load1(tp, n1);
      PIPELINE_FENCE();
      const double (&A)[NX * NX] = c1.A; const double (&Bm)[NX * NU] = c2.Bm;
      // ---- round 1
      double T1c[NX], T2c[NU], Qu[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += A[k * NX + i] * Vc[k];
        T1c[i] = s; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bm[k * NU + u] * Vc[k];
        T2c[u] = s; }
      double Qxq;
      {   // l_x[qc] = sum_j (2 Q dt)[qc, j] e[j]  (objective.cpp:115-131), then + (A^T V_x)[qc]
        double e[NX];
        if (xrt) {
          cptr_t r = uniform_ptr(xrt + (size_t)t * NX);
#pragma unroll
          for (int i = 0; i < NX; ++i) e[i] = c2.x[i] - r[i];
        } else {
#pragma unroll
          for (int i = 0; i < NX; ++i) e[i] = c2.x[i] - xg[i];
        }
        double lxq = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) lxq += (2.0 * Qrow[j]) * e[j];
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += c1.Aq[k] * Vx[k];
        Qxq = lxq + s2;
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s = 0.0;   // Objective::lu on the hoisted R dt (same expression)
#pragma unroll
        for (int j = 0; j < NU; ++j) s += (2.0 * Rr[i * NU + j]) * c2.u[j];
        Qu[i] = s; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Bm[k * NU + u] * Vx[k];
        Qu[u] = Qu[u] + s2; }
      if constexpr (!kQuad) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Ls[C::oT1 + i * NX + qc] = T1c[i];
#pragma unroll
        for (int u = 0; u < NU; ++u) Ls[C::oT2 + u * NX + qc] = T2c[u];
        lds_sync();
      }
      __builtin_amdgcn_sched_barrier(0);
      load2(tp, n2);
      PIPELINE_FENCE();
      __builtin_amdgcn_sched_barrier(0);
      // ---- round 2
      // kEarlyQP (single-input CLDDP: pendulum, cart-pole -- BASELINE config[1] read literally): Q_uu, the PD test and the BoxQP
      // (a data-dependent loop of dependent divisions / square roots, replicated by the lanes of the group) run BEFORE the sixteen
      // T1 entries are fetched from LDS and Q_xx / Q_ux are formed: nothing of those is live across the loop, which is what had the
      // round-3 kernel at 256 VGPR + 124 AGPR with ~250 accvgpr moves per step.  Pure reordering of independent statements.
#ifndef CDDP_EARLYQP
#define CDDP_EARLYQP 1
#endif
      constexpr bool kEarlyQP = CLDDP && !LOGDDP && NU == 1 && !kQuad && CDDP_EARLYQP;
      double T1[NX * NX], T2[NU * NX];
      double Qxxc[NX], Quxc[NU], Quu[NU * NU];
      double kk[NU], KKc[NU];
      [[maybe_unused]] double qp_h = 0.0;      // kEarlyQP: Q_uu + reg
      [[maybe_unused]] int qp_free = 1;        // kEarlyQP: 0 when the BoxQP solution sits on a bound (its gain row is zero)
      if constexpr (kEarlyQP) {
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) T2[i] = Ls[C::oT2 + i];
        { double s = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) s += T2[j] * Bm[j];
          Quu[0] = (2.0 * Rr[0]) + s; }
        qp_h = Quu[0] + reg;
        if (min_real_eig<1>(&qp_h) <= 0) return false;   // clddp_solver.cpp:133-140
        // Q_ux[0, qc] needs T2 and the lane's column of A only: formed here, so the gain column's quotient Q_ux / (Q_uu + reg) is in
        // flight beside the BoxQP's own division instead of behind its loop (two dependent ~150-cycle divisions per step otherwise)
        { double s = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) s += T2[j] * c1.Aq[j];
          Quxc[0] = s; }
        const double kq_free = -ldlt1_solve(qp_h, Quxc[0]);
        if (box < 0) {                                    // clddp_solver.cpp:142-145: k = -Q_uu_reg^-1 Q_u
          double H[1];
          inverse_pplu<1>(&qp_h, H);
          kk[0] = 0.0 + (-H[0]) * Qu[0];
          qp_h = H[0];                                    // the inverse, for the gain column below
        } else {                                          // clddp_solver.cpp:147-178
          const double lb = box_lo[0] - c2.u[0], ub = box_hi[0] - c2.u[0];
          kk[0] = c2.k0[0];
          const int stq = boxqp_solve1_fast(qpc, qp_h, Qu[0], lb, ub, kk[0], qp_free);   // straight-line common traces, the loop otherwise (dev_boxqp.hpp)
          if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
          KKc[0] = qp_free ? kq_free : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) T1[i] = Ls[C::oT1 + i];
      } else if constexpr (kQuad) {   // G = 4: the lanes of a trajectory are a quad -- column j of T1, T2 is lane j's, fetched by DPP broadcasts
#pragma unroll
        for (int i = 0; i < NX; ++i) quad_gather<NX>(T1c[i], T1 + i * NX);
#pragma unroll
        for (int u = 0; u < NU; ++u) quad_gather<NX>(T2c[u], T2 + u * NX);
      } else {
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) T1[i] = Ls[C::oT1 + i];
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) T2[i] = Ls[C::oT2 + i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += T1[i * NX + j] * c1.Aq[j];
        Qxxc[i] = (2.0 * Qq[i]) + s; }
      if constexpr (!kEarlyQP) {
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += T2[u * NX + j] * c1.Aq[j];
        Quxc[u] = s; }
      }
      if constexpr (!kEarlyQP) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int v = 0; v < NU; ++v) { double s = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) s += T2[u * NX + j] * Bm[j * NU + v];
            Quu[u * NU + v] = (2.0 * Rr[u * NU + v]) + s; }
      }
      if constexpr (LOGDDP) {
        // logddp_solver.cpp:518-530: Q_x += mu g_x, Q_u += mu g_u, Q_xx += mu H_xx, Q_uu += mu H_uu, Q_ux += mu H_ux, constraint by
        // constraint (LgCons::derivs, kernels_logddp.hpp).  The lane's column sits in full-size operands whose other columns are
        // never read back; selects, not indexed stores (qc is a lane value)
        if constexpr (LM > 0) {
          double g[LMM], Gx[LMM * NX], Gu[LMM * NU];
#pragma unroll
          for (int i = 0; i < LMM * NX; ++i) Gx[i] = 0.0;
#pragma unroll
          for (int i = 0; i < LMM * NU; ++i) Gu[i] = 0.0;
          LCons::template eval<NX, NU>(lcc, c2.x, c2.u, g);
          LCons::template jac<NX, NU>(lcc, c2.x, c2.u, Gx, Gu);
          double fQx[NX], fQxx[NX * NX], fQux[NU * NX];
#pragma unroll
          for (int a = 0; a < NX; ++a) {
            fQx[a] = (a == qc) ? Qxq : 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) fQxx[i * NX + a] = (a == qc) ? Qxxc[i] : 0.0;
#pragma unroll
            for (int u = 0; u < NU; ++u) fQux[u * NX + a] = (a == qc) ? Quxc[u] : 0.0;
          }
          LgCons<LCons>::template derivs<NX, NU>(P, g, Gx, Gu, c2.u, lg_mu, lg_delta, fQx, Qu, fQxx, Quu, fQux);
#pragma unroll
          for (int a = 0; a < NX; ++a) {
            Qxq = (a == qc) ? fQx[a] : Qxq;
#pragma unroll
            for (int i = 0; i < NX; ++i) Qxxc[i] = (a == qc) ? fQxx[i * NX + a] : Qxxc[i];
#pragma unroll
            for (int u = 0; u < NU; ++u) Quxc[u] = (a == qc) ? fQux[u * NX + a] : Quxc[u];
          }
        }
        // Q_uu_reg = sym(Q_uu + reg I), LDLT (:533-543); the un-regularised Q_uu stays in the value update
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
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Quxc[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) KKc[i] = -col[i];
      } else if constexpr (kEarlyQP) {
        if (box < 0) KKc[0] = 0.0 + (-qp_h) * Quxc[0];   // (box >= 0: set beside the BoxQP above)
      } else if constexpr (CLDDP) {
        double Quu_reg[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Quu_reg[i] = Quu[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) Quu_reg[i * NU + i] += reg;
        if (min_real_eig<NU>(Quu_reg) <= 0) return false;   // clddp_solver.cpp:133-140
        if (box < 0) {   // clddp_solver.cpp:142-145
          double H[NU * NU];
          inverse_pplu<NU>(Quu_reg, H);
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double s = 0.0, s2 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) { s += (-H[i * NU + j]) * Qu[j]; s2 += (-H[i * NU + j]) * Quxc[j]; }
            kk[i] = s; KKc[i] = s2;
          }
        } else {         // clddp_solver.cpp:147-178
          double lb[NU], ub[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) { lb[i] = box_lo[i] - c2.u[i]; ub[i] = box_hi[i] - c2.u[i]; }
#pragma unroll
          for (int i = 0; i < NU; ++i) kk[i] = c2.k0[i];
          if constexpr (NU == 1) {   // scalar BoxQP (dev_boxqp.hpp::boxqp_solve1): the N = 1 trace of the generic solver, nothing indexed
            int fr;
            const int stq = boxqp_solve1_fast(qpc, Quu_reg[0], Qu[0], lb[0], ub[0], kk[0], fr);
            if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
            KKc[0] = fr ? -ldlt1_solve(Quu_reg[0], Quxc[0]) : 0.0;
          } else {
          int free_[NU];
          LDLTd<NU> Hfree;
          const int stq = boxqp_solve<NU>(o, Quu_reg, Qu, lb, ub, kk, free_, Hfree);
          if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
#pragma unroll
          for (int i = 0; i < NU; ++i) KKc[i] = 0.0;
          int free_idx[NU]; int nf = 0;
          for (int i = 0; i < NU; ++i) if (free_[i]) free_idx[nf++] = i;
          if (nf > 0) {
            double col[NU];
            for (int i = 0; i < nf; ++i) col[i] = Quxc[free_idx[i]];
            Hfree.solve(col);
            for (int i = 0; i < nf; ++i) KKc[free_idx[i]] = -col[i];
          }
          }
        }
      } else {
        // regularisation stays in Q_uu (ipddp_solver.cpp:1084-1107)
        double Qs[NU * NU];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]);
#pragma unroll
        for (int i = 0; i < NU; ++i) Qs[i * NU + i] += reg;
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Quu[i] = Qs[i];
        if (NU == 1) {
          kk[0] = -ldlt1_solve(Qs[0], Qu[0]);
          KKc[0] = -ldlt1_solve(Qs[0], Quxc[0]);
        } else {
          LDLTs<NU> f;
          f.compute(Qs, NU);
          if (!f.ok) return false;
          double col[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qu[i];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Quxc[i];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) KKc[i] = -col[i];
        }
      }
      if constexpr (!kQuad) {
#pragma unroll
        for (int u = 0; u < NU; ++u) { Ls[C::oKK + u * NX + qc] = KKc[u]; Ls[C::oQux + u * NX + qc] = Quxc[u]; }
        lds_sync();
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
#pragma unroll
      for (int u = 0; u < NU; ++u) d.K[GI(t, NU * NX, u * NX + qc)] = KKc[u];
      // ---- round 3: value update
      double KK[NU * NX], Qux[NU * NX];
      if constexpr (kQuad) {
#pragma unroll
        for (int u = 0; u < NU; ++u) { quad_gather<NX>(KKc[u], KK + u * NX); quad_gather<NX>(Quxc[u], Qux + u * NX); }
      } else {
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) { KK[i] = Ls[C::oKK + i]; Qux[i] = Ls[C::oQux + i]; }
      }
      double Quuk[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) s1 += Quu[i * NU + j] * kk[j];
        Quuk[i] = s1; }
      { double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) { s0 += CLDDP ? Qu[i] * kk[i] : kk[i] * Qu[i]; s1 += kk[i] * Quuk[i]; }
        dV0 += s0; dV1 += 0.5 * s1; }
      double KtQ[NX * NU];
      mm_tn<NX, NU, NU>(KK, Quu, KtQ);
      double KtQq[NU];   // row qc of K^T Q_uu from the lane's own column (same expression as mm_tn)
#pragma unroll
      for (int j = 0; j < NU; ++j) { double s = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) s += KKc[u] * Quu[u * NU + j];
        KtQq[j] = s; }
      double Vxq, Vnc[NX];
      if constexpr (CLDDP) {   // clddp_solver.cpp:187-191
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KtQq[j] * kk[j]; bb += Quxc[j] * kk[j]; c += KKc[j] * Qu[j]; }
        Vxq = ((Qxq + a) + bb) + c;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double a2 = 0.0, b2 = 0.0, e2 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { a2 += KtQ[i * NU + j] * KKc[j]; b2 += Qux[j * NX + i] * KKc[j]; e2 += KK[j * NX + i] * Quxc[j]; }
          Vnc[i] = ((Qxxc[i] + a2) + b2) + e2;
        }
      } else {                 // ipddp_solver.cpp:1098-1107
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KKc[j] * Qu[j]; bb += Quxc[j] * kk[j]; c += KtQq[j] * kk[j]; }
        Vxq = ((Qxq + a) + bb) + c;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double a2 = 0.0, b2 = 0.0, e2 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { a2 += KK[j * NX + i] * Quxc[j]; b2 += Qux[j * NX + i] * KKc[j]; e2 += KtQ[i * NU + j] * KKc[j]; }
          Vnc[i] = ((Qxxc[i] + a2) + b2) + e2;
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oVn + i * NX + qc] = Vnc[i];
      Ls[C::oVx + qc] = Vxq;
      lds_sync();
#pragma unroll
      for (int i = 0; i < NX; ++i) Vc[i] = 0.5 * (Vnc[i] + Ls[C::oVn + qc * NX + i]);
#pragma unroll
      for (int i = 0; i < NX; ++i) Vx[i] = Ls[C::oVx + i];
      lds_sync();
      d.Vx[GI(t, NX, qc)] = Vxq;
#pragma unroll
      for (int i = 0; i < NX; ++i) d.Vxx[GI(t, NX * NX, i * NX + qc)] = Vc[i];
      if constexpr (LOGDDP) {
#pragma unroll
        for (int i = 0; i < NU; ++i) Qu_error = dmax(Qu_error, fabs(Qu[i]));   // :576
      } else if constexpr (CLDDP) {
        double sN = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) sN += fabs(Vx[i]);
        norm_Vx += sN;
#pragma unroll
        for (int i = 0; i < NU; ++i) Qu_error = dmax(Qu_error, fabs(Qu[i]));
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(Qu[i])); step_norm = dmax(step_norm, fabs(kk[i])); }
      }
      return true;
    };
    In1 a1, b1;
    In2 a2, b2;
    load1(N - 1, a1);
    load2(N - 1, a2);
    int t = N - 1;
    for (; t >= 1; t -= 2) {
      if (!step(t, a1, a2, b1, b2)) { fail = true; break; }
      if (!step(t - 1, b1, b2, a1, a2)) { fail = true; break; }
    }
    if (!fail && t == 0) fail = !step(0, a1, a2, b1, b2);
    if (!fail) {
      if constexpr (LOGDDP) inf_du = Qu_error;
      else if constexpr (CLDDP) {
        double scaling = o.termination_scaling_max_factor;
        scaling = dmax(scaling, norm_Vx / (N * NX)) / scaling;
        inf_du = Qu_error / scaling;
      }
      ok = true;
      break;
    }
    if (force == 2) break;
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  if (q != 0) return;
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du;
    if constexpr (!CLDDP) { d.step_norm[b] = step_norm; d.inf_pr[b] = 0.0; d.inf_comp[b] = 0.0; d.apr_max[b] = 1.0; d.adu_max[b] = 1.0; }
  }
  if (force) return;
  if constexpr (LOGDDP) {   // handleBackwardPassRegularizationLimit (:216-222); no early convergence test (base-class default)
    if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT_CONVERGED; d.phase[b] = PH_DONE; return; }
    d.phase[b] = PH_FWD1;
    return;
  }
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  const bool conv = CLDDP ? (inf_du < o.tolerance)                      // clddp_solver.cpp:206-213
                          : (0.0 < o.tolerance && inf_du < o.tolerance);  // ipddp_solver.cpp:925-958, no barrier
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, CLDDP ? 0.0 : d.mu[b]); return; }
  d.phase[b] = PH_FWD1;
}

// ================================================================================ cooperative sweep, nx > 8
// Column ownership as above, but the step's dense operands live in LDS instead of registers: A_t and B_t are
// fetched cooperatively (each lane of the group a 1/G slice, double-buffered), T1, T2, K, Q_ux, K^T Q_uu are
// streamed from LDS inside the inner products.  With nx = 12..14 the register-resident form needs A (nx^2) twice
// (ping-pong) plus T1 (nx^2) per lane -- > 512 VGPRs, i.e. scratch spills on every step (measured: 178 us per
// step for the C4 quadrotor).  Same sums, same association.
// H lanes per COLUMN (H = 1: a lane owns a whole column; H = 2: the rows of a column are split between two lanes, GG = 32
// lanes per trajectory -- chosen by the launcher when the batch would otherwise leave more than half of the SIMDs without a
// wavefront, e.g. the 2048-trajectory share of C4: 512 -> 1024 wavefronts, ~40 % fewer multiply-adds per lane).
template <class Model, bool HAS_X, int H = 1>
struct CoopBigCfg {
  static constexpr int NX = Model::NX, NU = Model::NU, GC = CoopCfg<Model>::G, G = GC * H, TPW = 64 / G;
  static constexpr int RH = (NX + H - 1) / H;                                         // rows of a column per lane
  static constexpr int UH = (NU + H - 1) / H;                                         // rows of B^T V_xx per lane
  static constexpr int NA = (NX * NX + G - 1) / G, NB = (NX * NU + G - 1) / G;     // per-lane slices of A, B
  static constexpr int RC = NU + NU * NU + NU + 2, NC = (RC + G - 1) / G;          // replicated condensed terms c_u .. icomp
  static constexpr int RK = NU * NX, NK = (RK + G - 1) / G;                         // gain block (rollout epilogue)
  // oM holds T1, then (in place, row by row) Q_xx, then (in place, element by element) Vn
  static constexpr int oA = 0, oB = oA + 2 * NX * NX, oM = oB + 2 * NX * NU, oT2 = oM + NX * NX, oKK = oT2 + NU * NX,
                       oQux = oKK + NU * NX, oKtQ = oQux + NU * NX, oVx = oKtQ + NX * NU, oDx = oVx + NX,
                       oC = oDx + NX, oWx = oC + 2 * RC, oQuu = oWx + (HAS_X ? NX * NX : 0), RAW = oQuu + NU * NU;
  static constexpr int NQ = (NU * NU + G - 1) / G;                                   // Q_uu entries per lane
  static constexpr int STRIDE = (RAW + 31) / 32 * 32 + 4;
};

template <class Model, class Cons, int H = 1>
__global__ __launch_bounds__(64) void k_backward_ipddp_coop_big(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                                int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  typedef CoopBigCfg<Model, Cons::HAS_X, H> C;
  constexpr int CST = L::SIZE, G = C::G, GC = C::GC, RH = C::RH, UH = C::UH;
  __shared__ double lds[C::TPW * C::STRIDE];
  __shared__ double ldsQ[NX * NX];   // Q dt (loop-invariant, shared by the trajectories of the wave)
  __shared__ double ldsR[NU * NU];   // R dt
  const int lane = threadIdx.x;
  {   // every lane of the wavefront takes part, BEFORE the per-trajectory early exits
    const double *Qp = Pk->pool + Pk->off_Qdt, *Rp = Pk->pool + Pk->off_Rdt;
    for (int e = lane; e < NX * NX; e += 64) ldsQ[e] = Qp[e];
    for (int e = lane; e < NU * NU; e += 64) ldsR[e] = Rp[e];
    lds_sync();
  }
  const int q = lane % G, tl = lane / G;       // q: slice index of the cooperative fetches
  const int col = q % GC, hh = q / GC;         // owned column, half of its rows
  const int qc = col < NX ? col : NX - 1;
  const int r0 = hh * RH, u0 = hh * UH;        // first owned row of a column / of B^T V_xx
  const int b = coop_group<C::TPW>((int)blockIdx.x, d.xcd_map) * C::TPW + tl;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  double *Ls = lds + tl * C::STRIDE;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  if (count_iter && q == 0) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  struct InAB { double a[C::NA], bm[C::NB], c[C::NC]; };
  struct In2 { double cxq, WQyxq[NU], QyxSirq; };   // the replicated block (c_u .. icomp) is read from its LDS copy where it is used
  static_assert(L::WQYU == L::CU + NU && L::QYUSIR == L::WQYU + NU * NU && L::IPR == L::QYUSIR + NU && L::ICOMP == L::IPR + 1, "contiguous replicated block");
  auto loadAB = [&](int tt, InAB &r) {   // this lane's slice of A_t, B_t (element e = q + G j; clamped past the end)
#pragma unroll
    for (int j = 0; j < C::NA; ++j) { const int e = q + G * j; r.a[j] = d.A[GT(tt, NX * NX, e < NX * NX ? e : NX * NX - 1)]; }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) { const int e = q + G * j; r.bm[j] = d.Bm[GT(tt, NX * NU, e < NX * NU ? e : NX * NU - 1)]; }
    // the terms every lane of the group needs (c_u, G_u^T YS^-1 G_u, G_u^T S^-1 rhat, residual maxima): one slice per lane
#pragma unroll
    for (int j = 0; j < C::NC; ++j) { const int e = q + G * j; r.c[j] = d.cst[GT(tt, CST, L::CU + (e < C::RC ? e : C::RC - 1))]; }
  };
  auto storeAB = [&](int buf, const InAB &r) {
    double *La = Ls + C::oA + buf * NX * NX, *Lb = Ls + C::oB + buf * NX * NU;
#pragma unroll
    for (int j = 0; j < C::NA; ++j) { const int e = q + G * j; if (e < NX * NX) La[e] = r.a[j]; }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) { const int e = q + G * j; if (e < NX * NU) Lb[e] = r.bm[j]; }
    double *Lc = Ls + C::oC + buf * C::RC;
#pragma unroll
    for (int j = 0; j < C::NC; ++j) { const int e = q + G * j; if (e < C::RC) Lc[e] = r.c[j]; }
  };
  auto load2 = [&](int tt, In2 &r) {
    const double *c = d.cst + GT(tt, CST, 0);
    const size_t ts = TSTRIDE;
    r.cxq = c[(size_t)(L::CX + qc) * ts];
    if constexpr (Cons::HAS_X) {
#pragma unroll
      for (int u = 0; u < NU; ++u) r.WQyxq[u] = c[(size_t)(L::WQYX + u * NX + qc) * ts];
      r.QyxSirq = c[(size_t)(L::QYXSIR + qc) * ts];
      // column qc of G_x^T YS^-1 G_x goes straight to LDS (indexed by a rolled loop below)
#pragma unroll 4
      for (int i = 0; i < NX; ++i) Ls[C::oWx + i * NX + qc] = c[(size_t)(L::WXQYX + i * NX + qc) * ts];
    }
  };
  for (;;) {
    ++nb;
    double Vx[NX], Vc[NX];
    {
      double xN[NX];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      Obj::final_grad(P, xN, Vx);
      const double *Qf = P->pool + P->off_Qf;
#pragma unroll
      for (int i = 0; i < NX; ++i) Vc[i] = 0.5 * ((2.0 * Qf[i * NX + qc]) + (2.0 * Qf[qc * NX + i]));
    }
    dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0;
    {
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oVx + i] = Vx[i];
      lds_sync();
      d.Vx[GI(N, NX, qc)] = Ls[C::oVx + qc];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) d.Vxx[GI(N, NX * NX, i * NX + qc)] = Vc[i];
    bool fail = false;
    // buf = parity of the step whose A, B sit in the LDS buffer
    // (the condensed-term record is fetched at the top of its own step: round 1 covers the latency, and a second
    //  register copy of it would push the kernel into scratch)
    auto step = [&](const int t, In2 &c2, InAB &nab) -> bool {
      const int tp = t > 0 ? t - 1 : 0;
      load2(t, c2);
      const double *La = Ls + C::oA + (t & 1) * NX * NX, *Lb = Ls + C::oB + (t & 1) * NX * NU;
      const double *Lc = Ls + C::oC + (t & 1) * C::RC;   // c_u | G_u^T YS^-1 G_u | G_u^T S^-1 rhat | ipr | icomp
      loadAB(tp, nab);
      PIPELINE_FENCE();
      double Aq[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) Aq[j] = La[j * NX + qc];
      // ---- round 1
      // (outer loops stay rolled and write to LDS: fully unrolled, the 3 nx^2-term products keep hundreds of LDS
      //  operands live and spill)
      double Qu[NU];
      // (rows in pairs -- pairs measured best: 553 ms of sweep class at C4 against 565 / 607 / 1007 for groups of 3 / 4 / 6 --
      //  the operands of group g + 1 fetched from LDS before group g is reduced: one LDS
      //  round trip is covered by the multiply-adds of a group instead of being waited for group by group)
      // owned rows: r0 .. r0 + RH - 1 (all of them for H = 1); a row index past the end repeats row NX - 1 (same value)
      {
        constexpr int GR = 2, NGRP = (RH + GR - 1) / GR;
        double b0[GR * NX], b1[GR * NX];
        auto ldg = [&](const int g, double (&buf)[GR * NX]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int li = g * GR + r; if (li < RH) { const int i = (r0 + li < NX) ? r0 + li : NX - 1;
#pragma unroll
            for (int k = 0; k < NX; ++k) buf[r * NX + k] = La[k * NX + i]; } }
        };
        auto cmp = [&](const int g, const double (&buf)[GR * NX]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int li = g * GR + r; if (li < RH) { const int i = (r0 + li < NX) ? r0 + li : NX - 1; double s = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) s += buf[r * NX + k] * Vc[k];
            Ls[C::oM + i * NX + qc] = s; } }
        };
        ldg(0, b0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          if (g + 1 < NGRP) { if ((g & 1) == 0) ldg(g + 1, b1); else ldg(g + 1, b0); }
          __builtin_amdgcn_sched_barrier(0);
          if ((g & 1) == 0) cmp(g, b0); else cmp(g, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int ul = 0; ul < UH; ++ul) { const int u = (u0 + ul < NU) ? u0 + ul : NU - 1; double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Lb[k * NU + u] * Vc[k];
        Ls[C::oT2 + u * NX + qc] = s; }
      double Qxq;
      { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Aq[k] * Vx[k];
        Qxq = c2.cxq + s2; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Lb[k * NU + u] * Vx[k];
        Qu[u] = Lc[u] + s2; }
      lds_sync();
      // ---- round 2: Q_xx[i, qc] replaces T1[i, qc] in place (row i of T1 is dead once every lane has used it,
      // and the lanes of a wavefront run this loop in lockstep)
      double Quxc[NU], Quu[NU * NU];
      {
        constexpr int GR = 2, NGRP = (RH + GR - 1) / GR;
        double b0[GR * NX], b1[GR * NX];
        auto ldg = [&](const int g, double (&buf)[GR * NX]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int li = g * GR + r; if (li < RH) { const int i = (r0 + li < NX) ? r0 + li : NX - 1;
#pragma unroll
            for (int j = 0; j < NX; ++j) buf[r * NX + j] = Ls[C::oM + i * NX + j]; } }
        };
        auto cmp = [&](const int g, const double (&buf)[GR * NX]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int li = g * GR + r; if (li < RH) { const int i = (r0 + li < NX) ? r0 + li : NX - 1; double s = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) s += buf[r * NX + j] * Aq[j];
            Ls[C::oM + i * NX + qc] = (2.0 * ldsQ[i * NX + qc]) + s; } }
        };
        ldg(0, b0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          if (g + 1 < NGRP) { if ((g & 1) == 0) ldg(g + 1, b1); else ldg(g + 1, b0); }
          __builtin_amdgcn_sched_barrier(0);
          if ((g & 1) == 0) cmp(g, b0); else cmp(g, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Ls[C::oT2 + u * NX + j] * Aq[j];
        Quxc[u] = s; }
      // Q_uu = 2 R dt + T2 B: its entries are spread over the lanes of the group (one LDS round) instead of every lane
      // repeating the nu^2 nx-term products
#pragma unroll
      for (int jq = 0; jq < C::NQ; ++jq) {
        const int e = q + G * jq;
        const int ee = e < NU * NU ? e : NU * NU - 1;
        const int u = ee / NU, v = ee - u * NU;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Ls[C::oT2 + u * NX + j] * Lb[j * NU + v];
        if (e < NU * NU) Ls[C::oQuu + e] = (2.0 * ldsR[ee]) + s;
      }
      lds_sync();
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] = Ls[C::oQuu + i];
      double Qr[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qr[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]) + Lc[NU + i * NU + c];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
      double kk[NU], KKc[NU], Quxq[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double rhs = Quxc[u];
        if constexpr (Cons::HAS_X) rhs = rhs + c2.WQyxq[u];
        Quxq[u] = rhs;
      }
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Qr[0], Qu[0] + Lc[NU + NU * NU]);
        KKc[0] = -ldlt1_solve(Qr[0], Quxq[0]);
      } else {
        LDLTs<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i] + Lc[NU + NU * NU + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Quxq[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) KKc[i] = -col[i];
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] += Lc[NU + NU * NU + i];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] += Lc[NU + i];
      double KtQq[NU];   // row qc of K^T Q_uu (condensed Q_uu), mm_tn's expression
#pragma unroll
      for (int j = 0; j < NU; ++j) { double s = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) s += KKc[u] * Quu[u * NU + j];
        KtQq[j] = s; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { Ls[C::oKK + u * NX + qc] = KKc[u]; Ls[C::oQux + u * NX + qc] = Quxq[u]; Ls[C::oKtQ + qc * NU + u] = KtQq[u]; }
      lds_sync();
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
#pragma unroll
      for (int u = 0; u < NU; ++u) d.K[GI(t, NU * NX, u * NX + qc)] = KKc[u];
      if (d.t4) {   // the copy the dX rollout below re-reads (wave-tiled K / k are for the wide kernels)
#pragma unroll
        for (int u = 0; u < NU; ++u) { d.Kt[G4(t, NU * NX + NU, u * NX + qc)] = KKc[u]; if (q == u) d.Kt[G4(t, NU * NX + NU, NU * NX + u)] = kk[u]; }
      }
      // ---- round 3
      if constexpr (Cons::HAS_X) Qxq += c2.QyxSirq;
      inf_pr = dmax(inf_pr, Lc[NU + NU * NU + NU]); inf_comp = dmax(inf_comp, Lc[NU + NU * NU + NU + 1]);
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
      double Vxq;
      {
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KKc[j] * Qu[j]; bb += Quxq[j] * kk[j]; c += KtQq[j] * kk[j]; }
        Vxq = ((Qxq + a) + bb) + c;
      }
      {   // Vn[i, qc] replaces the lane's own Q_xx[i, qc] in place; row groups pipelined as in rounds 1 and 2
        constexpr int GR = 2, NGRP = (RH + GR - 1) / GR, RW3 = 3 * NU + 2;
        double b0[GR * RW3], b1[GR * RW3];
        auto ldg = [&](const int g, double (&buf)[GR * RW3]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int li = g * GR + r; if (li < RH) { const int i = (r0 + li < NX) ? r0 + li : NX - 1;
#pragma unroll
            for (int j = 0; j < NU; ++j) { buf[r * RW3 + j] = Ls[C::oKK + j * NX + i]; buf[r * RW3 + NU + j] = Ls[C::oQux + j * NX + i]; buf[r * RW3 + 2 * NU + j] = Ls[C::oKtQ + i * NU + j]; }
            buf[r * RW3 + 3 * NU] = Ls[C::oM + i * NX + qc];
            if constexpr (Cons::HAS_X) buf[r * RW3 + 3 * NU + 1] = Ls[C::oWx + i * NX + qc]; } }
        };
        auto cmp = [&](const int g, const double (&buf)[GR * RW3]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int li = g * GR + r; if (li < RH) { const int i = (r0 + li < NX) ? r0 + li : NX - 1;
            double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) { a += buf[r * RW3 + j] * Quxq[j]; bb += buf[r * RW3 + NU + j] * KKc[j]; e += buf[r * RW3 + 2 * NU + j] * KKc[j]; }
            double qxx = buf[r * RW3 + 3 * NU];
            if constexpr (Cons::HAS_X) qxx += buf[r * RW3 + 3 * NU + 1];
            Ls[C::oM + i * NX + qc] = ((qxx + a) + bb) + e; } }
        };
        ldg(0, b0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          if (g + 1 < NGRP) { if ((g & 1) == 0) ldg(g + 1, b1); else ldg(g + 1, b0); }
          __builtin_amdgcn_sched_barrier(0);
          if ((g & 1) == 0) cmp(g, b0); else cmp(g, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Ls[C::oVx + qc] = Vxq;
      storeAB((t & 1) ^ 1, nab);     // next step's A, B into the other LDS buffer
      lds_sync();
#pragma unroll
      for (int i = 0; i < NX; ++i) Vc[i] = 0.5 * (Ls[C::oM + i * NX + qc] + Ls[C::oM + qc * NX + i]);
#pragma unroll
      for (int i = 0; i < NX; ++i) Vx[i] = Ls[C::oVx + i];
      lds_sync();
      d.Vx[GI(t, NX, qc)] = Vxq;
      if constexpr (H == 1) {
#pragma unroll
        for (int i = 0; i < NX; ++i) d.Vxx[GI(t, NX * NX, i * NX + qc)] = Vc[i];
      } else {   // each half stores its own rows (value selects: no dynamic register index)
#pragma unroll
        for (int li = 0; li < RH; ++li) {
          double v = Vc[li];
#pragma unroll
          for (int hx = 1; hx < H; ++hx) { const int ii = hx * RH + li; if (hh == hx) v = Vc[ii < NX ? ii : NX - 1]; }
          const int i = (r0 + li < NX) ? r0 + li : NX - 1;
          d.Vxx[GI(t, NX * NX, i * NX + qc)] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(Qu[i])); step_norm = dmax(step_norm, fabs(kk[i])); }
      return true;
    };
    In2 a2;
    InAB nab;
    loadAB(N - 1, nab);
    storeAB((N - 1) & 1, nab);
    lds_sync();
    for (int t = N - 1; t >= 0; --t)
      if (!step(t, a2, nab)) { fail = true; break; }
    if (!fail) { ok = true; break; }
    if (force == 2) break;
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  bool conv = false;
  if (ok) {
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
    if (!conv || force) {
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): lane qc computes row qc of dx_{t+1}
      double dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = 0.0;
      // the gain block [k | K] of a step is fetched cooperatively (one slice per lane) into the LDS area that held
      // A during the sweep (double-buffered); each lane fetches its own row of A and B
      struct RIn { double ks[C::NK], kq, Aq[NX], Bq[NU]; };
      constexpr int GS = NU + NU * NX;   // LDS doubles per gain buffer (2 GS <= 2 nx^2)
      static_assert(2 * GS <= 2 * NX * NX, "gain buffers fit the A area");
      auto load_r = [&](int tt, RIn &r) {
#pragma unroll
        for (int j = 0; j < C::NK; ++j) { const int e = q + G * j; const int ee = e < C::RK ? e : C::RK - 1; r.ks[j] = d.t4 ? d.Kt[G4(tt, NU * NX + NU, ee)] : d.K[GI(tt, NU * NX, ee)]; }
        r.kq = d.t4 ? d.Kt[G4(tt, NU * NX + NU, NU * NX + (q < NU ? q : NU - 1))] : d.k[GI(tt, NU, q < NU ? q : NU - 1)];
#pragma unroll
        for (int j = 0; j < NX; ++j) r.Aq[j] = d.A[GT(tt, NX * NX, qc * NX + j)];
#pragma unroll
        for (int j = 0; j < NU; ++j) r.Bq[j] = d.Bm[GT(tt, NX * NU, qc * NU + j)];
      };
      auto store_r = [&](int buf, const RIn &r) {
        double *Lg = Ls + C::oA + buf * GS;
        if (q < NU) Lg[q] = r.kq;
#pragma unroll
        for (int j = 0; j < C::NK; ++j) { const int e = q + G * j; if (e < C::RK) Lg[NU + e] = r.ks[j]; }
      };
      RIn rc, rn;
      auto rstep = [&](const int t) {
        const int tn = t + 1 < N - 1 ? t + 1 : t;
        load_r(tn, rn);
        PIPELINE_FENCE();
        d.dX[GI(t, NX, qc)] = Ls[C::oDx + qc];
        if (t < N - 1) {
          const double *Lg = Ls + C::oA + (t & 1) * GS;
          double du[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += Lg[NU + i * NX + j] * dx[j];
            du[i] = Lg[i] + a; }
          double a = 0.0, c = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) a += rc.Aq[j] * dx[j];
#pragma unroll
          for (int j = 0; j < NU; ++j) c += rc.Bq[j] * du[j];
          const double dxq = (a + c) + 0.0;
          lds_sync();
          Ls[C::oDx + qc] = dxq;
          store_r((t & 1) ^ 1, rn);
          lds_sync();
#pragma unroll
          for (int i = 0; i < NX; ++i) dx[i] = Ls[C::oDx + i];
        }
        rc = rn;
      };
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oDx + i] = 0.0;
      load_r(0, rc);
      store_r(0, rc);
      lds_sync();
      for (int t = 0; t < N; ++t) rstep(t);
    }
  }
  if (q != 0) return;
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du; d.step_norm[b] = step_norm;
    d.inf_pr[b] = inf_pr; d.inf_comp[b] = inf_comp;
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, mu); return; }
  d.phase[b] = PH_FWD1;
}

// ================================================================================ cooperative sweep, nx > 8, TWO wavefronts per group
// (round 5)  k_backward_ipddp_coop_big runs ~2 340 VALU instructions per step on ONE wavefront per four
// trajectories (512 wavefronts for the 2048-trajectory C4 share on 1024 SIMDs), and the second half of them -- Q_uu, its pivoted
// factorisation, the two solves -- does not depend on the first half (T1 = A^T V_xx, Q_xx = l_xx + T1 A).  Here a workgroup is two
// wavefronts that own the SAME four trajectories and the same columns:
//   wave 0 (the A side)     T1[:, q], Q_xx[:, q] (in place), rows [0, R0) of the value update, rows [0, R0) of the V_xx store
//   wave 1 (the gain side)  T2[:, q], Q_x[q], Q_u, Q_ux[:, q], Q_uu, factor, k, K[:, q], K^T Q_uu, V_x[q], the step statistics,
//                           rows [R0, nx) of the value update and of the V_xx store, the gain stores, the dX rollout epilogue
// with two s_barriers per step (gain side -> A side: K, Q_ux, K^T Q_uu and the factorisation's verdict; both -> both: the new V_xx).
// The T1 / Q_xx / V_xx area is double-buffered by step parity, so the symmetrised read of V_xx[t + 1] needs no third barrier.
// A barrier must be reached by both wavefronts the same number of times, so nothing leaves the loop early: every trajectory of the
// group runs its own little state machine (its own t; t = N is the pseudo-step that (re)starts a pass from the terminal cost after
// a failed factorisation raised the regularisation), the other lanes' work is predicated, and the loop ends when no trajectory of
// the group is sweeping.  Every element is still accumulated by ONE lane in the reference's order: bitwise the one-wave kernel
// (tests/test_gpu_parity.py::test_two_wave_sweep_agrees_bitwise).
// the value `v` holds in lane K of the caller's ROW of 16 lanes (DPP row_newbcast, gfx90a+: the only DPP control 64-bit moves take).
// With 16 lanes per trajectory a row IS a trajectory group, and lane l of it owns column l: an operand that lives in another
// column's registers costs one v_mov_b64_dpp instead of an LDS write, a wait and a read.  Every lane of a row must be active.
template <int K> DEV double row_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true); }
template <class F, int... Is> DEV void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> DEV void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#ifndef BIG2_EXP
#define BIG2_EXP 0   // timing experiments only (profiles/r05_big2_roles.md): the product is 0
#endif
template <class Model, class Cons>
struct CoopBig2Cfg {
  static constexpr int NX = Model::NX, NU = Model::NU, G = CoopCfg<Model>::G, TPW = 64 / G, G2 = 2 * G;
  static constexpr int NA = (NX * NX + G2 - 1) / G2, NB = (NX * NU + G2 - 1) / G2;   // per-lane slices of A, B (both wavefronts fetch)
  static constexpr int RC = NU + NU * NU + NU + 2, NC = (RC + G2 - 1) / G2;          // replicated condensed terms c_u .. icomp
  static constexpr int RK = NU * NX, NK = (RK + G - 1) / G;                          // gain block (rollout epilogue: one wavefront)
  static constexpr int NE = NU * NU + NU, NQ = (NE + G - 1) / G;                     // Q_uu entries and Q_u, spread over the gain side
  static constexpr int oA = 0, oB = oA + 2 * NX * NX, oM = oB + 2 * NX * NU, oT2 = oM + 2 * NX * NX, oKK = oT2 + NU * NX,
                       oQux = oKK + NU * NX, oKtQ = oQux + NU * NX, oVx = oKtQ + NX * NU, oDx = oVx + NX,
                       oC = oDx + NX, oQuu = oC + 2 * RC, oFlag = oQuu + NE, oKv = oFlag + 1, RAW = oKv + NU;
  static constexpr int STRIDE = (RAW + 31) / 32 * 32 + 4;
#ifndef CDDP_BIG2_R0
#define CDDP_BIG2_R0 ((NX * 3 + 3) / 4)
#endif
  static constexpr int R0 = CDDP_BIG2_R0;                                              // value-update rows of the A side (it has the slack)
};

DEV void wg_sync() { if (BIG2_EXP == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class Model, class Cons>
__global__ __launch_bounds__(128) void k_backward_ipddp_coop_big2(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                                  int force, int count_iter) {
  static_assert(!Cons::HAS_X, "state-constrained shapes run the one-wave kernel");
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  typedef CoopBig2Cfg<Model, Cons> C;
  constexpr int CST = L::SIZE, G = C::G, G2 = C::G2, R0 = C::R0, RC = C::RC;
  __shared__ double lds[C::TPW * C::STRIDE];
  __shared__ double ldsQ[NX * NX];   // Q dt (loop-invariant, shared by the trajectories of the group)
  __shared__ double ldsR[NU * NU];   // R dt
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    const double *Qp = Pk->pool + Pk->off_Qdt, *Rp = Pk->pool + Pk->off_Rdt;
    for (int e = threadIdx.x; e < NX * NX; e += 128) ldsQ[e] = Qp[e];
    for (int e = threadIdx.x; e < NU * NU; e += 128) ldsR[e] = Rp[e];
  }
  const int q = lane % G, tl = lane / G;
  const int qc = q < NX ? q : NX - 1;          // owned column (lanes past nx shadow the last one)
  unsigned long long tk_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_last = 0;
  auto tick = [&](const int k) {
    if (BIG2_EXP == 9) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now = __builtin_readcyclecounter(); tk_acc[k] += now - tk_last; tk_last = now; __builtin_amdgcn_sched_barrier(0); }
  };
  const int q2 = q + G * w;                    // slice index of the cooperative fetches
  const int b_raw = coop_group<C::TPW>((int)blockIdx.x, d.xcd_map) * C::TPW + tl;
  const int b = b_raw < d.B ? b_raw : d.B - 1; // (addresses only; a trajectory past the end never sweeps)
  bool act = b_raw < d.B && (force || d.phase[b] == PH_ACTIVE);
  const bool was_act = act;
  double *Ls = lds + tl * C::STRIDE;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  static_assert(L::WQYU == L::CU + NU && L::QYUSIR == L::WQYU + NU * NU && L::IPR == L::QYUSIR + NU && L::ICOMP == L::IPR + 1, "contiguous replicated block");
  // Every stack is laid out step-major with NB * 64 * E doubles per step (wave-tiled and four-wide layouts alike): the element
  // offsets of this lane are fixed, a step adds t * tstr * E.
  const size_t tstr = (size_t)d.NB * 64;
  struct InAB { double a[C::NA], bm[C::NB], c[C::NC]; };
  size_t offA[C::NA], offB[C::NB], offC[C::NC];
#pragma unroll
  for (int j = 0; j < C::NA; ++j) { const int e = q2 + G2 * j; offA[j] = GT(0, NX * NX, e < NX * NX ? e : NX * NX - 1); }
#pragma unroll
  for (int j = 0; j < C::NB; ++j) { const int e = q2 + G2 * j; offB[j] = GT(0, NX * NU, e < NX * NU ? e : NX * NU - 1); }
#pragma unroll
  for (int j = 0; j < C::NC; ++j) { const int e = q2 + G2 * j; offC[j] = GT(0, CST, L::CU + (e < RC ? e : RC - 1)); }
  auto loadAB = [&](const size_t tb, InAB &r) {   // this lane's slice of A_t, B_t and of the replicated condensed terms (element e = q2 + G2 j)
    const double *pa = d.A + tb * (NX * NX), *pb = d.Bm + tb * (NX * NU), *pc = d.cst + tb * CST;
#pragma unroll
    for (int j = 0; j < C::NA; ++j) r.a[j] = pa[offA[j]];
#pragma unroll
    for (int j = 0; j < C::NB; ++j) r.bm[j] = pb[offB[j]];
#pragma unroll
    for (int j = 0; j < C::NC; ++j) r.c[j] = pc[offC[j]];
  };
  auto storeAB = [&](int buf, const InAB &r) {
    double *La = Ls + C::oA + buf * NX * NX, *Lb = Ls + C::oB + buf * NX * NU, *Lc = Ls + C::oC + buf * RC;
#pragma unroll
    for (int j = 0; j < C::NA; ++j) { const int e = q2 + G2 * j; if (e < NX * NX) La[e] = r.a[j]; }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) { const int e = q2 + G2 * j; if (e < NX * NU) Lb[e] = r.bm[j]; }
#pragma unroll
    for (int j = 0; j < C::NC; ++j) { const int e = q2 + G2 * j; if (e < RC) Lc[e] = r.c[j]; }
  };
  const size_t offVxx = GI(0, NX * NX, qc);
  int t = N;                       // the step this trajectory runs next (N: start of a pass)
  wg_sync();                       // ldsQ / ldsR
  static_assert(G == 16, "a trajectory group is a DPP row");
  // Operands that another column's lane holds come by row broadcast (row_bcast<K>), not through LDS: A[k, i] is lane i's A[k, qc],
  // T1[i, j] lane j's T1[i, qc], K[j, i] lane i's K[j, qc], ...  LDS carries what crosses the two wavefronts (K, Q_ux, K^T Q_uu one
  // way, Q_xx / Vn rows the other), the transposed read of the value update, and the few dynamically indexed operands of Q_uu.
  if (w == 0) {
    // ------------------------------------------------------------------------------------------------ the A side
    double Vc[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) Vc[i] = 0.0;
    // EVERY global store of the sweep is issued here, one step late, in the window this side would otherwise spend waiting at barrier 1
    // (a store of 64 lanes scattered over 16 lines takes ~100 cycles to issue: 2 100 cycles per step on the gain side's critical path,
    //  profiles/r05_big2_roles.md): the step's V_xx column is this side's Vc, its V_x / k / K come through LDS
    const size_t offVx = GI(0, NX, qc), offk = GI(0, NU, q < NU ? q : NU - 1), offK = GI(0, NU * NX, qc);
    bool pv_ok = false, pg_ok = false;      // pending: the value rows (index pt_v), the gains (index pt_g)
    int pt_v = 0, pt_g = 0;
    double pVxq = 0.0, pkq = 0.0, pK[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) pK[u] = 0.0;
    auto flush = [&]() {
      if (pv_ok) {
        const size_t tb = (size_t)pt_v * tstr;
        d.Vx[tb * NX + offVx] = pVxq;
        double *pv = d.Vxx + tb * (NX * NX) + offVxx;
#pragma unroll
        for (int i = 0; i < NX; ++i) pv[(size_t)i * NX * 64] = Vc[i];
      }
      if (pg_ok && !(BIG2_EXP == 9 && pt_g < 10)) {
        const size_t tb = (size_t)pt_g * tstr;
        if (q < NU) d.k[tb * NU + offk] = pkq;
        double *pKp = d.K + tb * (NU * NX) + offK;
#pragma unroll
        for (int u = 0; u < NU; ++u) pKp[(size_t)u * NX * 64] = pK[u];
        if (d.t4) {   // the copy the dX rollout below re-reads (wave-tiled K / k are for the wide kernels)
#pragma unroll
          for (int u = 0; u < NU; ++u) d.Kt[G4(pt_g, NU * NX + NU, u * NX + qc)] = pK[u];
          if (q < NU) d.Kt[G4(pt_g, NU * NX + NU, NU * NX + q)] = pkq;
        }
      }
      pv_ok = false; pg_ok = false;
    };
    if (BIG2_EXP == 9) tk_last = __builtin_readcyclecounter();
    while (__builtin_amdgcn_ballot_w64(act) != 0) {
      const bool isN = t >= N;
      const int ts = isN ? N - 1 : t, tp = isN ? N - 1 : (t > 0 ? t - 1 : 0);
      InAB nab;
      loadAB((size_t)tp * tstr, nab);
      PIPELINE_FENCE();
      const double *La = Ls + C::oA + (ts & 1) * NX * NX;
      double *Mb = Ls + C::oM + (t & 1) * NX * NX;
      double Aq[NX], lq[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) { Aq[j] = La[j * NX + qc]; lq[j] = ldsQ[j * NX + qc]; }
      tick(0);
      // round 1: T1[i, qc] = sum_k A[k, i] V[k, qc]
      double T1[NX];
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += row_bcast<i>(Aq[k]) * Vc[k];
        T1[i] = s;
      });
      tick(1);
      // round 2: Q_xx[i, qc] = 2 Q dt [i, qc] + sum_j T1[i, j] A[j, qc]
      double Qxx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double s = 0.0;
        static_for<NX>([&](auto J) { constexpr int j = decltype(J)::value; s += row_bcast<j>(T1[i]) * Aq[j]; });
        Qxx[i] = (2.0 * lq[i]) + s;
      }
#pragma unroll
      for (int i = R0; i < NX; ++i) Mb[i * NX + qc] = Qxx[i];          // the gain side's rows of the value update
      flush();                                                          // the previous step's stores
      tick(2);
      wg_sync();                                                        // ---- barrier 1: K, Q_ux, K^T Q_uu, the verdict
      tick(3);
      // rows [0, R0) of Vn[i, qc] = ((Q_xx[i, qc] + K[:, i] . Q_ux[:, qc]) + Q_ux[:, i] . K[:, qc]) + (K^T Q_uu)[i, :] . K[:, qc]
      double Quxq[NU], KKc[NU], KtQq[NU];
      const int verdict = (int)Ls[C::oFlag];                            // 0 step done, 1 restart the pass, 2 give up
#pragma unroll
      for (int u = 0; u < NU; ++u) { Quxq[u] = Ls[C::oQux + u * NX + qc]; KKc[u] = Ls[C::oKK + u * NX + qc]; KtQq[u] = Ls[C::oKtQ + qc * NU + u]; }
      const double kq = Ls[C::oKv + (q < NU ? q : NU - 1)];
      double Vn[R0];
      static_for<R0>([&](auto I) {
        constexpr int i = decltype(I)::value;
        double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += row_bcast<i>(KKc[j]) * Quxq[j]; bb += row_bcast<i>(Quxq[j]) * KKc[j]; e += row_bcast<i>(KtQq[j]) * KKc[j]; }
        Vn[i] = ((Qxx[i] + a) + bb) + e;
      });
      if (__builtin_amdgcn_ballot_w64(isN) != 0) {                     // start of a pass: terminal cost, Vn[i, qc] = 2 Qf[i, qc]
        const double *Qf = P->pool + P->off_Qf;
#pragma unroll
        for (int i = 0; i < R0; ++i) { const double v = 2.0 * Qf[i * NX + qc]; Vn[i] = isN ? v : Vn[i]; }
      }
#pragma unroll
      for (int i = 0; i < R0; ++i) Mb[i * NX + qc] = Vn[i];
      tick(4);
      storeAB(tp & 1, nab);
      if (BIG2_EXP == 9) lds_sync();
      tick(5);
      wg_sync();                                                        // ---- barrier 2: Vn, V_x
      tick(6);
      {
        double mr[NX], mc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { mr[i] = i < R0 ? Vn[i < R0 ? i : 0] : Mb[i * NX + qc]; mc[i] = Mb[qc * NX + i]; }
        __builtin_amdgcn_sched_barrier(0);
        const double vxq = Ls[C::oVx + qc];
        const bool good = act && (isN || verdict == 0);
        if (good) {
#pragma unroll
          for (int i = 0; i < NX; ++i) Vc[i] = 0.5 * (mr[i] + mc[i]);
          pv_ok = true; pt_v = isN ? N : t; pVxq = vxq;
          if (!isN) {
            pg_ok = true; pt_g = t; pkq = kq;
#pragma unroll
            for (int u = 0; u < NU; ++u) pK[u] = KKc[u];
          }
        }
      }
      // next state of this trajectory (the gain side takes the same decisions)
      if (act) {
        if (isN) t = N - 1;
        else if (verdict == 1) t = N;
        else if (verdict == 2) act = false;
        else if (t == 0) act = false;
        else t = t - 1;
      }
      tick(7);
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wg_sync();                                                          // the rollout epilogue of the gain side re-reads K, k
    if (BIG2_EXP == 9 && q == 0 && was_act) {
#pragma unroll
      for (int j = 0; j < 10; ++j) d.k[GI(j, NU, 1)] = (double)tk_acc[j];
    }
    return;
  }
  // -------------------------------------------------------------------------------------------------- the gain side
  if (count_iter && q == 0 && act) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  {
    const size_t offCx = GT(0, CST, L::CX + qc);
    double Vc[NX], Vx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { Vc[i] = 0.0; Vx[i] = 0.0; }
    if (BIG2_EXP == 9) tk_last = __builtin_readcyclecounter();
    while (__builtin_amdgcn_ballot_w64(act) != 0) {
      const bool isN = t >= N;
      const int ts = isN ? N - 1 : t, tp = isN ? N - 1 : (t > 0 ? t - 1 : 0);
      const double cxq = d.cst[(size_t)ts * tstr * CST + offCx];
      InAB nab;
      loadAB((size_t)tp * tstr, nab);
      PIPELINE_FENCE();
      const double *La = Ls + C::oA + (ts & 1) * NX * NX, *Lb = Ls + C::oB + (ts & 1) * NX * NU;
      const double *Lc = Ls + C::oC + (ts & 1) * RC;   // c_u | G_u^T YS^-1 G_u | G_u^T S^-1 rhat | ipr | icomp
      double *Mb = Ls + C::oM + (t & 1) * NX * NX;
      double Aq[NX], Br[NU], Lcr[RC];
#pragma unroll
      for (int u = 0; u < NU; ++u) Br[u] = Lb[qc * NU + u];             // row qc of B: B[k, u] is lane k's Br[u]
#pragma unroll
      for (int j = 0; j < NX; ++j) Aq[j] = La[j * NX + qc];
#pragma unroll
      for (int e = 0; e < RC; ++e) Lcr[e] = Lc[e];
      __builtin_amdgcn_sched_barrier(0);
      tick(0);
      // ---- round 1: T2[u, qc] = sum_k B[k, u] V[k, qc], Q_x[qc]
      double T2[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double s = 0.0;
        static_for<NX>([&](auto K) { constexpr int k = decltype(K)::value; s += row_bcast<k>(Br[u]) * Vc[k]; });
        T2[u] = s;
        Ls[C::oT2 + u * NX + qc] = s;      // (the dynamically indexed operands of Q_uu below)
      }
      double Qxq;
      { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Aq[k] * Vx[k];
        Qxq = cxq + s2; }
      tick(1);
      // ---- round 2: Q_ux[:, qc] = T2 A[:, qc]; Q_uu = 2 R dt + T2 B and Q_u = c_u + B^T V_x, one entry (or a few) per lane, one LDS round
      double Quxq[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double s = 0.0;
        static_for<NX>([&](auto J) { constexpr int j = decltype(J)::value; s += row_bcast<j>(T2[u]) * Aq[j]; });
        Quxq[u] = s;
      }
      lds_sync();
      double lt[C::NQ * NX], cb[C::NQ * NX], addend[C::NQ];
#pragma unroll
      for (int jq = 0; jq < C::NQ; ++jq) {
        const int e = q + G * jq;
        const int ee = e < C::NE ? e : C::NE - 1;
        const bool isQuu = ee < NU * NU;
        const int u = ee / NU, v = ee - u * NU;
        const int rT = isQuu ? u : 0, cB = isQuu ? v : ee - NU * NU;
#pragma unroll
        for (int j = 0; j < NX; ++j) { lt[jq * NX + j] = Ls[C::oT2 + rT * NX + j]; cb[jq * NX + j] = Lb[j * NU + cB]; }
        const double a0 = ldsR[isQuu ? ee : 0], a1 = Lc[cB];
        addend[jq] = isQuu ? 2.0 * a0 : a1;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jq = 0; jq < C::NQ; ++jq) {
        const int e = q + G * jq;
        const bool isQuu = (e < C::NE ? e : C::NE - 1) < NU * NU;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += (isQuu ? lt[jq * NX + j] : Vx[j]) * cb[jq * NX + j];
        if (e < C::NE) Ls[C::oQuu + e] = addend[jq] + s;
      }
      lds_sync();
      tick(2);
      double Quu[NU * NU], Qu[NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] = Ls[C::oQuu + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] = Ls[C::oQuu + NU * NU + i];
      __builtin_amdgcn_sched_barrier(0);
      double Qr[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qr[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]) + Lcr[NU + i * NU + c];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
      double kk[NU], KKc[NU];
      bool fok = true;
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Qr[0], Qu[0] + Lcr[NU + NU * NU]);
        KKc[0] = -ldlt1_solve(Qr[0], Quxq[0]);
      } else {
        LDLTs<NU> f;
        f.compute(Qr, NU);
        fok = f.ok;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i] + Lcr[NU + NU * NU + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Quxq[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) KKc[i] = -col[i];
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] += Lcr[NU + NU * NU + i];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] += Lcr[NU + i];
      double KtQq[NU];   // row qc of K^T Q_uu (condensed Q_uu), mm_tn's expression
#pragma unroll
      for (int j = 0; j < NU; ++j) { double s = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) s += KKc[u] * Quu[u * NU + j];
        KtQq[j] = s; }
#pragma unroll
      for (int u = 0; u < NU; ++u) { Ls[C::oKK + u * NX + qc] = KKc[u]; Ls[C::oQux + u * NX + qc] = Quxq[u]; Ls[C::oKtQ + qc * NU + u] = KtQq[u]; }
      // the verdict of this step: 0 done, 1 the factorisation failed and the pass restarts with more regularisation, 2 it failed for good
      int verdict = 0;
      double reg_next = reg;
      if (!isN && !fok) {
        verdict = 2;
        if (force != 2) { reg_next = reg_increase(o, reg); if (!(reg_next >= o.reg_max_value)) verdict = 1; }
      }
      if (q == 0) {
        Ls[C::oFlag] = (double)verdict;
#pragma unroll
        for (int u = 0; u < NU; ++u) Ls[C::oKv + u] = kk[u];
      }
      if (BIG2_EXP == 9) lds_sync();
      tick(3);
      wg_sync();                                                        // ---- barrier 1
      tick(4);
      constexpr int R1 = NX - R0;
      double qx[R1 > 0 ? R1 : 1];
#pragma unroll
      for (int il = 0; il < R1; ++il) qx[il] = Mb[(R0 + il) * NX + qc];
      const bool good = act && !isN && verdict == 0;
      // ---- round 3 (computed for every lane; what a pass start or a failed step must not keep is discarded below)
      double Quuk[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) s1 += Quu[i * NU + j] * kk[j];
        Quuk[i] = s1; }
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) { s0 += kk[i] * Qu[i]; s1 += kk[i] * Quuk[i]; }
      double Vxq;
      {
        double a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += KKc[j] * Qu[j]; bb += Quxq[j] * kk[j]; c += KtQq[j] * kk[j]; }
        Vxq = ((Qxq + a) + bb) + c;
      }
      if (good) {
        inf_pr = dmax(inf_pr, Lcr[NU + NU * NU + NU]); inf_comp = dmax(inf_comp, Lcr[NU + NU * NU + NU + 1]);
        dV0 += s0; dV1 += 0.5 * s1;
#pragma unroll
        for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(Qu[i])); step_norm = dmax(step_norm, fabs(kk[i])); }
      }
      double Vn[R1 > 0 ? R1 : 1];
      static_for<R1>([&](auto IL) {
        constexpr int il = decltype(IL)::value, i = R0 + il;
        double a = 0.0, bb = 0.0, e = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { a += row_bcast<i>(KKc[j]) * Quxq[j]; bb += row_bcast<i>(Quxq[j]) * KKc[j]; e += row_bcast<i>(KtQq[j]) * KKc[j]; }
        Vn[il] = ((qx[il] + a) + bb) + e;
      });
      if (__builtin_amdgcn_ballot_w64(isN) != 0) {   // start of a pass: terminal cost, statistics cleared
        double xN[NX], g[NX];
        ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
        Obj::final_grad(P, xN, g);
        double gq = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) if (i == qc) gq = g[i];
        Vxq = isN ? gq : Vxq;
        const double *Qf = P->pool + P->off_Qf;
#pragma unroll
        for (int il = 0; il < R1; ++il) { const double v = 2.0 * Qf[(R0 + il) * NX + qc]; Vn[il] = isN ? v : Vn[il]; }
        if (isN && act) { ++nb; dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0; }
      }
#pragma unroll
      for (int il = 0; il < R1; ++il) Mb[(R0 + il) * NX + qc] = Vn[il];
      Ls[C::oVx + qc] = Vxq;
      if (BIG2_EXP == 9) lds_sync();
      tick(5);
      storeAB(tp & 1, nab);
      if (BIG2_EXP == 9) lds_sync();
      tick(6);
      wg_sync();                                                        // ---- barrier 2
      tick(7);
      {
        double mr[NX], mc[NX], vxr[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { mr[i] = i >= R0 ? Vn[i >= R0 ? i - R0 : 0] : Mb[i * NX + qc]; mc[i] = Mb[qc * NX + i]; }
        static_for<NX>([&](auto I) { constexpr int i = decltype(I)::value; vxr[i] = row_bcast<i>(Vxq); });
        __builtin_amdgcn_sched_barrier(0);
        const bool good2 = act && (isN || verdict == 0);
        if (good2) {
#pragma unroll
          for (int i = 0; i < NX; ++i) { Vc[i] = 0.5 * (mr[i] + mc[i]); Vx[i] = vxr[i]; }
        }
        tick(8);
      }
      if (act) {
        if (isN) t = N - 1;
        else if (verdict == 1) { t = N; reg = reg_next; }
        else if (verdict == 2) { act = false; if (force != 2) reg = reg_next; }
        else if (t == 0) { act = false; ok = true; }
        else t = t - 1;
      }
      tick(9);
    }
  }
  wg_sync();                                                            // the A side's last stores (K, k of step 0) are out
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (BIG2_EXP == 9 && q == 0 && was_act) {
#pragma unroll
    for (int j = 0; j < 10; ++j) d.k[GI(j, NU, 0)] = (double)tk_acc[j];
  }
  if (!was_act) return;
  bool conv = false;
  if (ok) {
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
    if ((!conv || force) && BIG2_EXP != 1) {
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): lane qc computes row qc of dx_{t+1}, lane u < nu the control
      // row du[u] = k[u] + K[u, :] dx; dx and du travel by row broadcast, so a step is one short dependent chain with no LDS round
      // trip, and the rows of K, A, B of the next steps (independent of dx) are fetched three steps ahead -- the one-wave kernel
      // waits for an HBM round trip every step (2.5 us of it, 1 ms of a 3.8 ms launch at C4).  Whole rows (trajectories) are
      // active or not: the broadcasts see every lane of their row.
      struct RIn { double Kr[NX], kq, Ar[NX], Br[NU]; };
      const int ui = q < NU ? q : NU - 1;
      constexpr int EK = NU * NX + NU;
      const double *baseK = d.t4 ? d.Kt + G4(0, EK, ui * NX) : d.K + GI(0, NU * NX, ui * NX);
      const double *basek = d.t4 ? d.Kt + G4(0, EK, NU * NX + ui) : d.k + GI(0, NU, ui);
      const size_t sK = d.t4 ? tstr * EK : tstr * (NU * NX), sk = d.t4 ? tstr * EK : tstr * NU, se = d.t4 ? 4 : 64;
      const double *baseA = d.A + GT(0, NX * NX, qc * NX), *baseB = d.Bm + GT(0, NX * NU, qc * NU);
      const size_t sa = TSTRIDE;
      auto load_r = [&](int tt, RIn &r) {
        tt = tt < N - 1 ? tt : N - 2;
        tt = tt > 0 ? tt : 0;
        const double *pK = baseK + (size_t)tt * sK, *pA = baseA + (size_t)tt * tstr * (NX * NX), *pB = baseB + (size_t)tt * tstr * (NX * NU);
#pragma unroll
        for (int jj = 0; jj < NX; ++jj) r.Kr[jj] = pK[(size_t)jj * se];
        r.kq = basek[(size_t)tt * sk];
#pragma unroll
        for (int jj = 0; jj < NX; ++jj) r.Ar[jj] = pA[(size_t)jj * sa];
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) r.Br[jj] = pB[(size_t)jj * sa];
      };
      double dxq = 0.0;
      double *pdX = d.dX + GI(0, NX, qc);
      auto rstep = [&](const int t, const RIn &rc, RIn &rl) {   // rc: the rows of step t; rl: free buffer, takes the rows of step t + 3
        if (t >= N) return;
        load_r(t + 3, rl);
        PIPELINE_FENCE();
        pdX[(size_t)t * tstr * NX] = dxq;
        if (t < N - 1) {
          double dx[NX], du[NU];
          static_for<NX>([&](auto J) { constexpr int jj = decltype(J)::value; dx[jj] = row_bcast<jj>(dxq); });
          double au = 0.0;
#pragma unroll
          for (int jj = 0; jj < NX; ++jj) au += rc.Kr[jj] * dx[jj];
          const double du_own = rc.kq + au;
          static_for<NU>([&](auto U) { constexpr int u = decltype(U)::value; du[u] = row_bcast<u>(du_own); });
          double a = 0.0, c = 0.0;
#pragma unroll
          for (int jj = 0; jj < NX; ++jj) a += rc.Ar[jj] * dx[jj];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) c += rc.Br[jj] * du[jj];
          dxq = (a + c) + 0.0;
        }
      };
      RIn r0, r1, r2, r3;
      load_r(0, r0); load_r(1, r1); load_r(2, r2);
      for (int t = 0; t < N; t += 4) { rstep(t, r0, r3); rstep(t + 1, r1, r0); rstep(t + 2, r2, r1); rstep(t + 3, r3, r2); }
    }
  }
  if (q != 0) return;
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du; d.step_norm[b] = step_norm;
    d.inf_pr[b] = inf_pr; d.inf_comp[b] = inf_comp;
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, mu); return; }
  d.phase[b] = PH_FWD1;
}

#undef GI

}  // namespace cddp_dev
