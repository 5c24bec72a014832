// Host-side launch table: one KernelSet per (plant, constraint layout) instantiation.
#pragma once
#include <vector>
#include <cstdlib>
#include <cstring>
#include "kernels_te.hpp"
#include "kernels_pcm.hpp"
#include "kernels_elem.hpp"
#include "kernels_mfma.hpp"
#include "kernels_logddp.hpp"
#include "kernels_msipddp.hpp"

namespace cddp_dev {

// CU-partitioned streams (round 5, capi.hip::CuPlan): when the calling thread has set a hop, the SERIAL sweep kernel of backward() is
// enqueued on the hop's stream (a stream created with hipExtStreamCreateWithCUMask) instead of the caller's, bracketed by two events:
// the wide (batch x N) kernels before / after it stay on the caller's stream.  nullptr (the default): everything on one stream.
struct SweepHop { hipStream_t s; hipEvent_t ev_in, ev_out; };
inline thread_local const SweepHop *tl_sweep_hop = nullptr;
// A failed event record / stream wait would leave two streams of one group unordered (kernels racing on K, k, V and the trial slots) without
// any launch failing: the first such error is latched here and cddp_hip_solve / backward return it (capi.hip::take_order_error).
inline thread_local hipError_t tl_order_error = hipSuccess;
inline void order_check(hipError_t e) { if (e != hipSuccess && tl_order_error == hipSuccess) tl_order_error = e; }
inline hipStream_t sweep_hop_in(hipStream_t s) {
  const SweepHop *h = tl_sweep_hop;
  if (!h || h->s == s) return s;
  order_check(hipEventRecord(h->ev_in, s));
  order_check(hipStreamWaitEvent(h->s, h->ev_in, 0));
  return h->s;
}
inline void sweep_hop_out(hipStream_t s) {
  const SweepHop *h = tl_sweep_hop;
  if (!h || h->s == s) return;
  order_check(hipEventRecord(h->ev_out, h->s));
  order_check(hipStreamWaitEvent(s, h->ev_out, 0));
}

struct KernelSet {
  int model, nx, nu, m;
  int cst_size;   // per-step doubles of the condensed-term stack (lean IPDDP backward), 0 = fused sweep
  int te_rec_size, te_group;   // cooperative terminal-equality sweep: per-step record doubles (0 = none), lanes per trajectory
  const char *name;
  bool (*matches)(const ProblemDev &);
  void (*derivs)(const DevBuf &, int force, hipStream_t);
  void (*backward)(const DevBuf &, int solver, int force, int count_iter, hipStream_t);
  void (*forward)(const DevBuf &, int solver, int a0, int na, int phase_req, int force, int first_only, hipStream_t);
  void (*costate)(const DevBuf &, int solver, int a0, int na, int phase_req, int force, int first_only, hipStream_t);
  void (*update)(const DevBuf &, int stage, int n1, int is_last, int do_count, hipStream_t);
  void (*init)(const DevBuf &, int mode, hipStream_t);
  void (*stage)(const DevBuf &, int copy_xu, int ipddp, hipStream_t);
  bool logddp_ddp;   // LogDDP with use_ilqr = 0: the plant has explicit Hessian tensors (Model::kHasHess)
  bool has_msipddp;  // the MSIPDDP kernels (kernels_msipddp.hpp) are instantiated for this layout: nx <= 8, no terminal set, and -- with path
                     // constraints -- nu = 1 or nx = nu (the shapes msipddp_solver.cpp:1398 defines)
  bool has_logddp;   // the LogDDP kernels (kernels_logddp.hpp) are instantiated for this layout: one lane per trajectory, nx <= 8, no terminal set
  int k4_waves;      // wavefronts per (tile, alpha) of the IPDDP rollout launch: 2 (producer + consumer), 3 where two consumers run (PcTwoConsumers), 1 for the one-wave kernels
  int ms_cst_size;   // MSIPDDP, path-constrained: per-step doubles of the condensed-term stack of the split sweep (kernels_msipddp.hpp::MsCst), 0 = none
  int (*t4_layout)(const DevBuf &);   // 1 when derivs() / backward() of this handle use the sub-tile-minor stacks (kernels.hpp::GT) under the current environment
};

// Layouts whose two-role rollout runs TWO consumer waves (k_forward_ipddp_pc<.., NC = 2>): the consumer, not the dynamics chain, sets the
// pace there (profiles/r05_rollout_roles.md: unicycle with box + ball rows, Euler producer idle 60 % of the launch).  Opt-in (k4_consumers).
template <class Model, class Cons> struct PcTwoConsumers { static constexpr bool value = false; };
template <class Cons> struct PcTwoConsumers<UnicycleModel, Cons> {
  // (the consumer's per-step record must be the ping-pong kind with one step of look-ahead: 16 < doubles <= 40, kernels_lean.hpp::StepIn)
  static constexpr int kRec = 5 * Cons::M + UnicycleModel::NU * UnicycleModel::NX + (Cons::HAS_X ? UnicycleModel::NX : 1) + (Cons::NEEDS_U ? UnicycleModel::NU : 1);
  static constexpr bool value = Cons::M >= 4 && kRec > 16 && kRec <= 40;
};

template <class Model, class Cons, bool TERM = false>
struct Launcher {
  static bool matches(const ProblemDev &P) {
    return P.model == Model::ID && P.nx == Model::NX && P.nu == Model::NU && Cons::matches(P) && (TERM ? P.n_term > 0 : P.n_term == 0);
  }
  // path-constrained, no terminal set: condense -> lean sweep -> post (kernels_lean.hpp)
  static constexpr bool kLean = !TERM && Cons::M > 0;
  static constexpr int cst_size() { if constexpr (kLean) return CstLayout<Model, Cons>::SIZE; else return 0; }
  // terminal equality, no state-dependent path rows: cooperative reduced-LQR sweep (kernels_te.hpp) when the host
  // allocated its record stack (pT > 0, no terminal inequality, pT + 1 <= lanes per trajectory)
  static constexpr bool kTeCoop = TERM && !Cons::HAS_X;
  static constexpr int te_rec_size() { if constexpr (kTeCoop) return TeCfg<Model, Cons>::REC; else return 0; }
  // LogDDP on the device (kernels_logddp.hpp): one-lane kernels for every plant without a terminal set.  Register-resident up to
  // nx = 8; the nx >= 12 plants work through scratch in the sweep (12 KB per lane at the quadrotor: correct, slow -- the cooperative
  // LDS-operand form the IPDDP sweeps have is not built for LogDDP).  Full DDP (use_ilqr = 0) needs the explicit Hessian tensors.
#ifndef CDDP_LOGDDP_MAX_NX
#define CDDP_LOGDDP_MAX_NX 16
#endif
  static constexpr bool kLog = !TERM && Model::NX <= CDDP_LOGDDP_MAX_NX;
  // MSIPDDP on the device (kernels_msipddp.hpp); with path constraints only the shapes for which the reference's recursion is defined
  // (nu = 1 or nx = nu, msipddp_solver.cpp:1398).  Round 5: up to nx = 13, i.e. the unconstrained quadrotor runs resident (one-lane sweep
  // through scratch: correct, slow -- 9 KB per lane); the reference's own MSIPDDPTest.SolveQuadrotor (test_msipddp_solver.cpp:565) adds a
  // control box with nu = 4, nx = 13 -- a shape :1398 does not define -- and stays refused on both routes.
#ifndef CDDP_MSIPDDP_MAX_NX
#define CDDP_MSIPDDP_MAX_NX 13
#endif
  static constexpr bool kMs = !TERM && Model::NX <= CDDP_MSIPDDP_MAX_NX && (Cons::M == 0 || Model::NU == 1 || Model::NX == Model::NU);
  static constexpr int ms_cst_size() { if constexpr (kMs && Cons::M > 0) return MsCst<Model, Cons>::SIZE; else return 0; }
  static dim3 gridB(const DevBuf &d) { return dim3((d.B + 63) / 64); }
  static bool lane_sweep_requested() {   // read per launch (the tests switch it between solves of one process)
    const char *e = std::getenv("CDDP_HIP_SWEEP");
    return e && !std::strcmp(e, "lane");
  }
  static bool ms_lane_rollout_requested() {   // CDDP_HIP_MS_ROLLOUT=lane: the one-wave MSIPDDP rollout (it writes the dual rows itself)
    const char *e = std::getenv("CDDP_HIP_MS_ROLLOUT");
    return e && !std::strcmp(e, "lane");
  }
  // MSIPDDP: the path-constrained iLQR sweep runs split (condense -> recursion -> post) unless full DDP or CDDP_HIP_SWEEP=lane ask for the fused
  // one-lane kernel; derivs() and backward() of one iteration see the same answer
  static bool ms_split_sweep(const DevBuf &d) { return d.ms && d.cst && !d.ddp && !lane_sweep_requested(); }
  static bool lg_lane_rollout_requested() {   // CDDP_HIP_LG_ROLLOUT=lane: the one-wave LogDDP rollout
    const char *e = std::getenv("CDDP_HIP_LG_ROLLOUT");
    return e && !std::strcmp(e, "lane");
  }
  static bool elem_sweep_requested() {   // CDDP_HIP_SWEEP=elem: the element-ownership sweep (kernels_elem.hpp), where instantiated
    const char *e = std::getenv("CDDP_HIP_SWEEP");
    return e && !std::strcmp(e, "elem");
  }
  static bool mfma_sweep_requested() {   // CDDP_HIP_SWEEP=mfma | coop (default: see the note at the launch site)
    const char *e = std::getenv("CDDP_HIP_SWEEP");
    return e && !std::strcmp(e, "mfma");
  }
  // Sub-tile-minor ("T4", kernels.hpp::GT) sweep-input stacks: exactly when the G = 16 cooperative sweep of this layout is the
  // consumer -- the one-lane, matrix-core and full-DDP sweeps read the wave-tiled form.  Evaluated per launch (the tests switch
  // CDDP_HIP_SWEEP between solves of one process); derivs() and backward() of one iteration see the same answer.
  static int t4_layout(const DevBuf &d) {
    if constexpr (CoopCfg<Model>::G != 16) return 0;
    else {
      if (lane_sweep_requested() || mfma_sweep_requested() || d.ddp) return 0;
      if (const char *e = std::getenv("CDDP_HIP_T4")) { if (e[0] == '0') return 0; }
      if constexpr (kLean) return (d.cst && !d.ms && !d.lg) ? 1 : 0;   // (an MSIPDDP handle keeps its own condensed-term stack in d.cst: plain layout)
      else if constexpr (kTeCoop) return d.te_cst ? 1 : 0;
      else return 0;
    }
  }
  // Step sizes per rollout workgroup of the small path-constrained layouts: 1 = the two-wave form (default), CDDP_HIP_K4_NA = 2 | 3 =
  // NA producers + ONE consumer per (tile, group of NA step sizes) (kernels_pcm.hpp).  Built to cut the wavefront count of the
  // rollout launch (1408 -> 1056 / 939 at BASELINE config[1]) after round 3 traced its length to SIMD sharing; bitwise equal
  // (tests/test_gpu_parity.py::test_multi_alpha_rollout_groups_agree_bitwise) and MEASURED SLOWER on MI355X
  // (profiles/r04_k4_groups.md): C2 22.6 -> 25.5 / 30.3 ms of rollout class per solve with the trials of a group served one after
  // the other, 30.1 / 28.7 with their per-step work in one straight-line block; pendulum 3.4 -> 5.8 / 6.0; C3 33 -> 55 / 80.  The
  // consumer's per-trial work (~500 instructions per step: two logarithms, the slack / dual trials, cost, residual terms) is as long
  // as the producer's RK4 step (~650), so one consumer behind two or three producers is the chain the launch waits for.  Opt-in.
  // Consumer waves of the two-role rollout where both forms are instantiated (PcTwoConsumers).  Default ONE: the two-consumer kernel is
  // bitwise equal (tests/test_gpu_parity.py::test_two_consumer_rollout_agrees_bitwise) and 18 % faster as a kernel on the whole chip
  // (C3 ladder of 4 / 11 step sizes: 446 -> 386 / 1499 -> 1215 us), but inside the solve its third wavefront per (tile, step size) breaks the
  // one-wavefront-per-SIMD fit of the adaptive ladder on a group's half of the chip (64 tiles x 4 step sizes x 3 = 768 on 512 SIMDs): C3
  // 78.8 -> 79.7 ms per solve (profiles/r05_rollout_roles.md).  CDDP_HIP_K4_CONSUMERS=2 opts in.
  static int k4_consumers() {
    if (const char *e = std::getenv("CDDP_HIP_K4_CONSUMERS")) { if (e[0] == '2') return 2; }
    return 1;
  }
  static int k4_group() {
    if (const char *e = std::getenv("CDDP_HIP_K4_NA")) { const int v = std::atoi(e); if (v >= 1 && v <= 3) return v; }
    return 1;
  }
  // Role-split sweep (round 6, kernels_coop.hpp::k_backward_ipddp_coop<.., NH > 0>): helper wavefronts in the sweep workgroup evaluate
  // what k_condense<.., true> evaluates and feed the recursion wave through an LDS ring -- one launch instead of two, no condensed-term
  // stack.  CDDP_HIP_SWEEP_ROLES = 0 (the separate kernels; comparison side of the bitwise test) | 1 | 2 helpers; read per launch, derivs()
  // and backward() of one iteration see the same answer.
  template <int RB> static constexpr bool roles_fit() {
    if constexpr (!kLean || Model::NX > 8) return false;
    else return (size_t)RoleCfg<Model, Cons, RB>::RING * 8 + (size_t)CoopCfg<Model>::TPW * CoopCfg<Model>::STRIDE * 8 <= (size_t)72 * 1024;   // two workgroups per CU
  }
  static constexpr bool kRoles = roles_fit<2>();
  static int roles_nh(const DevBuf &d) {
    if constexpr (!kRoles) return 0;
    else {
      if (!(d.cst && !d.ms && !d.lg) || d.ddp) return 0;
      if (lane_sweep_requested() || elem_sweep_requested()) return 0;
      int nh = 1;   // measured (profiles/r06_sweep_roles.md): one helper 43.1 -> 39.9 ms at C2, 77.5 -> 67.5 at C3; two / three helpers: 40.4 / 41.2 and 77.9 / 80.1
      if (const char *e = std::getenv("CDDP_HIP_SWEEP_ROLES")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) nh = v; }
      return nh;
    }
  }
  static void derivs(const DevBuf &d0, int force, hipStream_t s) {
    DevBuf d = d0;
    d.t4 = t4_layout(d0);
    if constexpr (kLean) {
      if (d.cst && !d.ms && !d.lg) {   // IPDDP with path constraints: derivative fill fused into the condensation pass (small plants;
                     // for nx > 8 the two register sets together would spill)
        if (roles_nh(d0) > 0) return;   // ... or evaluated by the helper wavefronts of the sweep itself (backward())
        if constexpr (Model::NX <= 8) {
          hipLaunchKernelGGL((k_condense<Model, Cons, true>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, d.xref_traj, force);
          return;
        } else {
          hipLaunchKernelGGL((k_derivs<Model>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, d.xref_traj, force);
          hipLaunchKernelGGL((k_condense<Model, Cons, false>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, d.xref_traj, force);
          return;
        }
      }
    }
    if constexpr (kMs && Cons::M > 0 && Model::NX <= 8) {   // MSIPDDP, split sweep: the derivative fill rides in k_ms_condense<.., true> (backward())
      if (ms_split_sweep(d)) return;
    }
    hipLaunchKernelGGL((k_derivs<Model>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, d.xref_traj, force);
    if constexpr (kTeCoop)
      if (d.te_cst) hipLaunchKernelGGL((k_te_condense<Model, Cons>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, d.xref_traj, force);
  }
  static void backward(const DevBuf &d0, int solver, int force, int count_iter, hipStream_t s) {
    DevBuf d = d0;
    d.t4 = (solver == CDDP_HIP_SOLVER_IPDDP) ? t4_layout(d0) : 0;
    // lane-cooperative sweeps (kernels_coop.hpp) wherever a layout has one; CDDP_HIP_SWEEP=lane selects the
    // one-lane-per-trajectory kernels instead (comparison / experiments)
    // full DDP (use_ilqr = 0): the one-lane IPDDP kernels carry the tensor terms; CLDDP's backward pass has none
    // (clddp_solver.cpp:79-204 ignores use_ilqr), so it keeps the cooperative sweep
    if (solver == CDDP_HIP_SOLVER_LOGDDP) {
      // lane-cooperative sweep (the LogDDP mode of k_backward_coop_plain) for the register-resident shapes; the one-lane kernel carries
      // the tensor terms of full DDP, serves nx > 8 (scratch-backed) and CDDP_HIP_SWEEP=lane (comparison)
      if constexpr (kLog && Model::NX <= 8) {
        if (!lane_sweep_requested() && !d.ddp) {
          hipLaunchKernelGGL((k_backward_coop_plain<Model, true, Cons>), dim3(coop_grid<CoopCfg<Model>::TPW>(d.B, d.xcd_map)), dim3(64), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
          sweep_hop_out(s);
          return;
        }
      }
      if constexpr (kLog) hipLaunchKernelGGL((k_backward_logddp<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
      return;
    }
    if (solver == CDDP_HIP_SOLVER_MSIPDDP) {
      if constexpr (kMs) {
        // path-constrained iLQR sweeps: condense (batch x N) -> value recursion -> post (batch x N) (round 5); the fused one-lane kernel
        // carries the tensor terms of full DDP, the unconstrained branch with its factor cache, and CDDP_HIP_SWEEP=lane (comparison)
        if constexpr (Cons::M > 0) {
          if (ms_split_sweep(d)) {
            const dim3 gridW((d.B + 63) / 64, d.N);
            if constexpr (Model::NX <= 8) hipLaunchKernelGGL((k_ms_condense<Model, Cons, true>), gridW, dim3(64), 0, s, d, d.P, d.xref_traj, force);   // + A_t, B_t (derivs() skipped K1)
            else hipLaunchKernelGGL((k_ms_condense<Model, Cons, false>), gridW, dim3(64), 0, s, d, d.P, d.xref_traj, force);
            hipLaunchKernelGGL((k_backward_msipddp_lean<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
            hipLaunchKernelGGL((k_ms_post<Model, Cons>), gridW, dim3(64), 0, s, d, d.P, force);
            return;
          }
        }
        hipLaunchKernelGGL((k_backward_msipddp<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
      }
      return;
    }
    const bool lane_sweep = lane_sweep_requested() || (d.ddp && solver == CDDP_HIP_SOLVER_IPDDP);
    const dim3 gridC(coop_grid<CoopCfg<Model>::TPW>(d.B, d.xcd_map));   // whole XCD super-groups when lines are shared between blocks
    if (solver == CDDP_HIP_SOLVER_CLDDP) {
      if (lane_sweep)
        hipLaunchKernelGGL((k_backward_clddp<Model>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
      else {
        hipLaunchKernelGGL((k_backward_coop_plain<Model, true>), gridC, dim3(64), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
        sweep_hop_out(s);
      }
    } else if constexpr (kLean) {
      if (lane_sweep)
        hipLaunchKernelGGL((k_backward_ipddp_lean<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
      else if constexpr (Model::NX > 8) {
        bool launched = false;
        if constexpr (Model::NX <= 15 && Model::NU <= 8) {
          if (mfma_sweep_requested()) {   // one wavefront per trajectory on the f64 matrix core (kernels_mfma.hpp); grid = whole tiles per XCD
            const dim3 gridW(((d.NB + 7) / 8) * 8 * 64);
            hipLaunchKernelGGL((k_backward_ipddp_mfma<Model, Cons>), gridW, dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
            hipLaunchKernelGGL((k_dx_rollout_wave<Model>), gridW, dim3(64), 0, s, d, force);
            launched = true;
          }
        }
        if (!launched) {   // operands in LDS: the register-resident form spills for nx = 12..14
          // CDDP_HIP_COOP_H=2: two lanes per column (32 lanes per trajectory, twice the wavefronts).  Bit-identical
          // (tests/test_gpu_parity.py::test_row_split_sweep_agrees_bitwise) but measured SLOWER where it was meant to help --
          // C4 share, 2048 trajectories: 512 -> 1024 wavefronts, 22 % fewer VALU instructions per wavefront, sweep class
          // 553 -> 637 ms -- the step is a chain of LDS round trips and a replicated factorisation, not issue slots.  Opt-in.
          constexpr int TPW1 = CoopCfg<Model>::TPW, TPW2 = TPW1 / 2;
          int hsel = 1;
          if (const char *e = std::getenv("CDDP_HIP_COOP_H")) { if (e[0] == '2') hsel = 2; }
          if (hsel == 2)
            hipLaunchKernelGGL((k_backward_ipddp_coop_big<Model, Cons, 2>), dim3(coop_grid<TPW2>(d.B, d.xcd_map)), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
          else
          {
            // default (round 5): two wavefronts per group of trajectories -- the A side and the gain side of a step run side by side
            // (k_backward_ipddp_coop_big2); CDDP_HIP_COOP_W=1 keeps the one-wave kernel (bitwise the same results)
            bool two = !Cons::HAS_X;
            if (const char *e = std::getenv("CDDP_HIP_COOP_W")) { if (e[0] == '1') two = false; }
            if constexpr (!Cons::HAS_X) {
              if (two) hipLaunchKernelGGL((k_backward_ipddp_coop_big2<Model, Cons>), gridC, dim3(128), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
            }
            if (!two) hipLaunchKernelGGL((k_backward_ipddp_coop_big<Model, Cons, 1>), gridC, dim3(64), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
            sweep_hop_out(s);
          }
        }
      }
      else {
        // CDDP_HIP_SWEEP=elem (nx <= 4, nu <= 2): element-ownership sweep (kernels_elem.hpp: 16 lanes per trajectory, four times the
        // wavefronts of the column-ownership form).  Bitwise equal, measured slower (165 vs 126 us at C2): opt-in.
        bool launched = false;
        if constexpr (Model::NX <= 4 && Model::NU <= 2) {
          if (elem_sweep_requested()) {
            hipLaunchKernelGGL((k_backward_ipddp_elem<Model, Cons>), dim3(coop_grid<4>(d.B, d.xcd_map)), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
            launched = true;
          }
        }
        if constexpr (kRoles) {
          const int nh = launched ? 0 : roles_nh(d0);
          if (nh > 0) {
            const hipStream_t ss = sweep_hop_in(s);
            // ring depth: helpers + the block being consumed + one of slack, while two workgroups still fit a CU
            if (nh == 1) {
              if constexpr (roles_fit<3>()) hipLaunchKernelGGL((k_backward_ipddp_coop<Model, Cons, 1, 3>), gridC, dim3(128), 0, ss, d, d.P, d.xref_traj, force, count_iter);
              else hipLaunchKernelGGL((k_backward_ipddp_coop<Model, Cons, 1, 2>), gridC, dim3(128), 0, ss, d, d.P, d.xref_traj, force, count_iter);
            } else {
              if constexpr (roles_fit<4>()) hipLaunchKernelGGL((k_backward_ipddp_coop<Model, Cons, 2, 4>), gridC, dim3(192), 0, ss, d, d.P, d.xref_traj, force, count_iter);
              else if constexpr (roles_fit<3>()) hipLaunchKernelGGL((k_backward_ipddp_coop<Model, Cons, 2, 3>), gridC, dim3(192), 0, ss, d, d.P, d.xref_traj, force, count_iter);
              else hipLaunchKernelGGL((k_backward_ipddp_coop<Model, Cons, 2, 2>), gridC, dim3(192), 0, ss, d, d.P, d.xref_traj, force, count_iter);
            }
            sweep_hop_out(s);
            launched = true;
          }
        }
        if (!launched) {
          hipLaunchKernelGGL((k_backward_ipddp_coop<Model, Cons>), gridC, dim3(64), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
          sweep_hop_out(s);
        }
      }
      if constexpr (kRoles) { if (roles_nh(d0) > 0 && !lane_sweep && !elem_sweep_requested()) return; }   // the role-split sweep's helpers did K3's work
      hipLaunchKernelGGL((k_post<Model, Cons>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, force);
    } else if constexpr (!TERM && Cons::M == 0) {
      if (lane_sweep)
        hipLaunchKernelGGL((k_backward_ipddp<Model, Cons, TERM>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
      else {
        hipLaunchKernelGGL((k_backward_coop_plain<Model, false>), gridC, dim3(64), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
        sweep_hop_out(s);
      }
    } else {
      if constexpr (kTeCoop) {
        if (d.te_cst && !lane_sweep) {
          hipLaunchKernelGGL((k_backward_te_coop<Model, Cons>), gridC, dim3(64), 0, sweep_hop_in(s), d, d.P, d.xref_traj, force, count_iter);
          sweep_hop_out(s);
          // steps per block: as many as keep >= ~1024 waves in the grid (1 for small batches)
          const int tiles = (d.B + 63) / 64;
          int tstep = (int)(((long long)tiles * d.N) / 1024); if (tstep < 1) tstep = 1; if (tstep > 16) tstep = 16;
          hipLaunchKernelGGL((k_te_post<Model, Cons>), dim3(tiles, (d.N + tstep - 1) / tstep), dim3(64), 0, s, d, d.P, force, tstep);
          return;
        }
      }
      hipLaunchKernelGGL((k_backward_ipddp<Model, Cons, TERM>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, force, count_iter);
    }
  }
  static void forward(const DevBuf &d, int solver, int a0, int na, int phase_req, int force, int first_only, hipStream_t s) {
    if (na <= 0) return;
    const dim3 grid((d.B + 63) / 64, na);
    if (solver == CDDP_HIP_SOLVER_LOGDDP) {
      // producer / consumer wave pair per (tile, alpha) (round 5); CDDP_HIP_LG_ROLLOUT=lane: the one-wave kernel (comparison: bitwise equal)
      if constexpr (kLog) {
        if (lg_lane_rollout_requested()) hipLaunchKernelGGL((k_forward_logddp<Model, Cons>), grid, dim3(64), 0, s, d, d.P, d.xref_traj, a0, 0, phase_req, force);
        else hipLaunchKernelGGL((k_forward_logddp_pc<Model, Cons>), grid, dim3(128), 0, s, d, d.P, d.xref_traj, a0, phase_req, force);
      }
      return;
    }
    if (solver == CDDP_HIP_SOLVER_MSIPDDP) {
      // producer / consumer wave pair per (tile, alpha) (round 5); CDDP_HIP_MS_ROLLOUT=lane: the one-wave kernel (comparison: bitwise equal)
      if constexpr (kMs) {
        if (ms_lane_rollout_requested() || !d.ladder_sorted) hipLaunchKernelGGL((k_forward_msipddp<Model, Cons>), grid, dim3(64), 0, s, d, d.P, d.xref_traj, a0, 0, phase_req, force);   // (a ladder that is not strictly decreasing: the one-wave kernel's mask form of the dual step search)
        else hipLaunchKernelGGL((k_forward_msipddp_pc<Model, Cons>), grid, dim3(128), 0, s, d, d.P, d.xref_traj, a0, phase_req, force);
      }
      return;
    }
    if (solver == CDDP_HIP_SOLVER_CLDDP)
      hipLaunchKernelGGL((k_forward_clddp<Model>), grid, dim3(64), 0, s, d, d.P, d.xref_traj, a0, 0, phase_req, force);
    else
    {
      if constexpr (kLean) {
        if constexpr (PcmTraits<Model, Cons>::kOk) {   // small layouts: NA producers + one consumer per (tile, group of NA step sizes), kernels_pcm.hpp
          const int ng = k4_group();
          if (ng == 3) { hipLaunchKernelGGL((k_forward_ipddp_pcm<Model, Cons, 3>), dim3((d.B + 63) / 64, (na + 2) / 3), dim3(256), 0, s, d, d.P, d.xref_traj, a0, na, phase_req, force); return; }
          if (ng == 2) { hipLaunchKernelGGL((k_forward_ipddp_pcm<Model, Cons, 2>), dim3((d.B + 63) / 64, (na + 1) / 2), dim3(192), 0, s, d, d.P, d.xref_traj, a0, na, phase_req, force); return; }
        }
        // producer / consumer wave pair per (tile, alpha); CDDP_HIP_K4_CONSUMERS=2: two consumers where instantiated (opt-in, k4_consumers)
        if constexpr (PcTwoConsumers<Model, Cons>::value) {
          if (k4_consumers() == 2) { hipLaunchKernelGGL((k_forward_ipddp_pc<Model, Cons, false, 2>), grid, dim3(192), 0, s, d, d.P, d.xref_traj, a0, phase_req, force); return; }
        }
        hipLaunchKernelGGL((k_forward_ipddp_pc<Model, Cons>), grid, dim3(128), 0, s, d, d.P, d.xref_traj, a0, phase_req, force);
      } else {
        if constexpr (kTeCoop && Cons::M > 0) {   // same rollout after the cooperative terminal-equality sweep
          if (d.te_cst && !(lane_sweep_requested() || d.ddp)) {
            hipLaunchKernelGGL((k_forward_ipddp_pc<Model, Cons, true>), grid, dim3(128), 0, s, d, d.P, d.xref_traj, a0, phase_req, force);
            return;
          }
        }
        hipLaunchKernelGGL((k_forward_ipddp<Model, Cons, TERM>), grid, dim3(64), 0, s, d, d.P, d.xref_traj, a0, 0, phase_req, force);
      }
    }
    (void)first_only;
  }
  // K4b: costate trial of the surviving trials (kernels_lean.hpp)
  static void costate(const DevBuf &d, int solver, int a0, int na, int phase_req, int force, int first_only, hipStream_t s) {
    if (solver == CDDP_HIP_SOLVER_MSIPDDP) {   // slack / dual / costate / constraint rows of the trial the selection rule will take (the two-role rollout leaves them out)
      if constexpr (kMs) {
        if (na <= 0 || ms_lane_rollout_requested() || !d.ladder_sorted) return;
        if (!force && first_only == 2) hipLaunchKernelGGL((k_pick_candidate<0>), dim3((d.B + 63) / 64), dim3(64), 0, s, d, a0, na, phase_req, force);
        hipLaunchKernelGGL((k_rows_msipddp<Model, Cons>), dim3((d.B + 63) / 64, d.N), dim3(64), 0, s, d, d.P, a0, na, phase_req, force, first_only);
      }
      return;
    }
    if (na <= 0 || solver == CDDP_HIP_SOLVER_CLDDP || solver == CDDP_HIP_SOLVER_LOGDDP) return;
    if (!force && first_only == 2) hipLaunchKernelGGL((k_pick_candidate<0>), dim3((d.B + 63) / 64), dim3(64), 0, s, d, a0, na, phase_req, force);
    if constexpr (Model::NX > 8) {
      if (!force && first_only != 0) {   // one trial per trajectory: the streaming kernel (2 - 4 waves per SIMD instead of one)
        hipLaunchKernelGGL((k_costate_one<Model>), dim3((d.B + 63) / 64, d.N + 1), dim3(64), 0, s, d, a0, na, phase_req, first_only);
        return;
      }
    }
    hipLaunchKernelGGL((k_costate<Model>), dim3((d.B + 63) / 64, d.N + 1), dim3(64), 0, s, d, a0, na, phase_req, force, force ? 0 : first_only);
  }
  static void update(const DevBuf &d, int stage, int n1, int is_last, int do_count, hipStream_t s) {
    if (d.lg) {
      if constexpr (kLog) hipLaunchKernelGGL((k_update_logddp<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, stage, n1, is_last, do_count);
      return;
    }
    if (d.ms) {
      if constexpr (kMs) hipLaunchKernelGGL((k_update_msipddp<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, stage, n1, is_last, do_count);
      return;
    }
    DevBuf dd = d;
    if constexpr (kTeCoop && Cons::M > 0) dd.ev_valid = (d.te_cst && !(lane_sweep_requested() || d.ddp)) ? 1 : 0;
    hipLaunchKernelGGL((k_update<Model, Cons, TERM>), gridB(d), dim3(64), 0, s, dd, d.P, d.xref_traj, stage, n1, is_last, do_count);
  }
  static void init(const DevBuf &d, int mode, hipStream_t s) {
    if (d.lg) {
      if constexpr (kLog) hipLaunchKernelGGL((k_init_logddp<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, mode);
      return;
    }
    if (d.ms) {
      if constexpr (kMs) hipLaunchKernelGGL((k_init_msipddp<Model, Cons>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, mode);
      return;
    }
    hipLaunchKernelGGL((k_init<Model, Cons, TERM>), gridB(d), dim3(64), 0, s, d, d.P, d.xref_traj, mode);
  }
  static void stage(const DevBuf &d, int copy_xu, int ipddp, hipStream_t s) {
    hipLaunchKernelGGL((k_stage<Model, Cons>), dim3((d.B + 63) / 64, d.N + 1), dim3(64), 0, s, d, copy_xu, ipddp);
  }
  static KernelSet set(const char *name) {
    KernelSet k;
    k.model = Model::ID; k.nx = Model::NX; k.nu = Model::NU; k.m = Cons::M; k.name = name; k.cst_size = cst_size(); k.te_rec_size = te_rec_size(); k.te_group = CoopCfg<Model>::G;
    k.matches = &matches; k.derivs = &derivs; k.backward = &backward; k.forward = &forward;
    k.costate = &costate; k.update = &update; k.init = &init; k.stage = &stage; k.t4_layout = &t4_layout; k.has_logddp = kLog; k.logddp_ddp = Model::kHasHess; k.has_msipddp = kMs; k.ms_cst_size = ms_cst_size(); k.k4_waves = (kLean && PcTwoConsumers<Model, Cons>::value) ? 3 : 2;
    return k;
  }
};

// one registration function per instantiation translation unit
void register_pendulum(std::vector<KernelSet> &);
void register_cartpole(std::vector<KernelSet> &);
void register_unicycle(std::vector<KernelSet> &);
void register_vehicles(std::vector<KernelSet> &);
void register_lti(std::vector<KernelSet> &);
void register_quadrotor(std::vector<KernelSet> &);
void register_quad12(std::vector<KernelSet> &);
void register_manipulator(std::vector<KernelSet> &);
void register_manip7(std::vector<KernelSet> &);
void register_terminal(std::vector<KernelSet> &);
void register_statebox(std::vector<KernelSet> &);

}  // namespace cddp_dev
