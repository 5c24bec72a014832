// Lean IPDDP backward pipeline for path-constrained problems without terminal constraints.
//
// The fused sweep (k_backward_ipddp) keeps ~1300 instructions per step in ONE in-order wavefront, yet only
// the value recursion V_x, V_xx -> Q blocks -> gains -> V_x, V_xx is a true serial chain.  Everything that does
// not depend on V is hoisted into wide (batch x N) kernels that run at machine width:
//
//   K1b k_condense  (batch x N)  per step: c_x = l_x + G_x^T y, c_u = l_u + G_u^T y, Y S^-1, r_p, r_c, rhat, S^-1 rhat,
//                                G_u^T YS^-1 G_u, G_u^T S^-1 rhat [, G_u^T YS^-1 G_x, G_x^T S^-1 rhat, G_x^T YS^-1 G_x]
//                                (ipddp_solver.cpp:1386-1391, 1410-1444, 1488-1492) -> `cst` stack
//   K2  k_backward_ipddp_lean (batch)  the serial chain only (:1392-1394, 1424-1456, 1494-1508) + linear-policy
//                                rollout dX (:1511-1520), stored
//   K3  k_post      (batch x N)  slack / dual gains k_y, K_y, k_s, K_s (:1458-1486), directions dS, dY (:1522-1532)
//                                and the fraction-to-boundary caps (:2939-2988) by atomic min
//
// Every intermediate keeps the association order of the fused kernel / the reference, so results are bit-identical
// to the fused path (checked by tests/test_gpu_parity.py against the oracle).
#pragma once
#include "kernels.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

template <class Model, class Cons>
struct CstLayout {
  static constexpr int NX = Model::NX, NU = Model::NU;
  static constexpr int CX = 0, CU = CX + NX, WQYU = CU + NU, QYUSIR = WQYU + NU * NU, IPR = QYUSIR + NU, ICOMP = IPR + 1;
  static constexpr int WQYX = ICOMP + 1, QYXSIR = WQYX + NU * NX, WXQYX = QYXSIR + NX;
  static constexpr int SIZE = Cons::HAS_X ? (WXQYX + NX * NX) : WQYX;
};

// ================================================================================ K1b
// The per-(trajectory, step) arithmetic of K1b as ONE device function, shared by the wide kernel below and by the helper
// wavefronts of the role-split sweep (kernels_coop.hpp::k_backward_ipddp_coop<.., NH > 0>): same expressions, same order.
// c[] is the condensed-term record in CstLayout order; WITH_DERIVS also fills A_t = I + dt f_x, B_t = dt f_u (K1,
// cddp_solver_base.cpp:319-394).
template <class Model, class Cons, bool WITH_DERIVS>
DEV void condense_eval(const ProblemDev *__restrict__ P, const double *__restrict__ xrt, const int t, const double *x, const double *u,
                       const double *y, const double *s, const double *g, const double mu, double *A, double *Bq, double *c,
                       double *ys_out = nullptr) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  double Qyx[M * NX], Qyu[M * NU];
  if constexpr (WITH_DERIVS) {   // identical to k_derivs
    double Fx[NX * NX], Fu[NX * NU];
    Model::jac(P->mp, x, u, Fx, Fu);
    const double dt = P->dt;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        double a = dt * Fx[i * NX + j];
        if (i == j) a += 1.0;
        A[i * NX + j] = a;
      }
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Bq[i] = dt * Fu[i];
  }
#pragma unroll
  for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
  Cons::template jac<NX, NU>(P, x, u, Qyx, Qyu);
  double cx[NX], cu[NU];
  Obj::lx(P, xrt, t, x, cx);
  Obj::lu(P, u, cu);
#pragma unroll
  for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) s1 += Qyx[r * NX + i] * y[r];
    cx[i] = cx[i] + s1; }
#pragma unroll
  for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) s1 += Qyu[r * NU + i] * y[r];
    cu[i] = cu[i] + s1; }
  double YS[M], Sir[M], ipr = 0.0, icomp = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const double ss = dmax(s[i], s_floor);
    YS[i] = clip_pos(y[i], ss);
    const double rp = g[i] + s[i];
    const double rc = y[i] * s[i] - mu;
    const double rhat = y[i] * rp - rc;
    Sir[i] = clip_sgn(rhat, ss);
    ipr = dmax(ipr, fabs(rp)); icomp = dmax(icomp, fabs(rc));
  }
  if (ys_out) {   // (role-split sweep) the ratios Y S^-1 k_post would form again from the same y, s, mu
#pragma unroll
    for (int i = 0; i < M; ++i) ys_out[i] = YS[i];
  }
  double W[NU * M], WQyu[NU * NU], QyuSir[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i)
#pragma unroll
    for (int r = 0; r < M; ++r) W[i * M + r] = Qyu[r * NU + i] * YS[r];
  mm_nn<NU, M, NU>(W, Qyu, WQyu);
#pragma unroll
  for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) s1 += Qyu[r * NU + i] * Sir[r];
    QyuSir[i] = s1; }
#pragma unroll
  for (int i = 0; i < NX; ++i) c[L::CX + i] = cx[i];
#pragma unroll
  for (int i = 0; i < NU; ++i) c[L::CU + i] = cu[i];
#pragma unroll
  for (int i = 0; i < NU * NU; ++i) c[L::WQYU + i] = WQyu[i];
#pragma unroll
  for (int i = 0; i < NU; ++i) c[L::QYUSIR + i] = QyuSir[i];
  c[L::IPR] = ipr;
  c[L::ICOMP] = icomp;
  if constexpr (Cons::HAS_X) {
    double WQyx[NU * NX], QyxSir[NX], Wx[NX * M], WxQyx[NX * NX];
    mm_nn<NU, M, NX>(W, Qyx, WQyx);
#pragma unroll
    for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
      for (int r = 0; r < M; ++r) s1 += Qyx[r * NX + i] * Sir[r];
      QyxSir[i] = s1; }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int r = 0; r < M; ++r) Wx[i * M + r] = Qyx[r * NX + i] * YS[r];
    mm_nn<NX, M, NX>(Wx, Qyx, WxQyx);
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) c[L::WQYX + i] = WQyx[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) c[L::QYXSIR + i] = QyxSir[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) c[L::WXQYX + i] = WxQyx[i];
  }
}

// WITH_DERIVS: the same (batch x N) pass also writes A_t = I + dt f_x, B_t = dt f_u (K1, cddp_solver_base.cpp:319-394):
// x, u are read once and one launch is saved per iteration.
template <class Model, class Cons, bool WITH_DERIVS = false>
__global__ __launch_bounds__(64) void k_condense(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef CstLayout<Model, Cons> L;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if constexpr (WITH_DERIVS) {   // first kernel of an outer iteration (see k_derivs)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && !force) *d.n_active = 0;
  }
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  const double *Sc = d.S + (size_t)cur * d.planeM;
  const double *Yc = d.Y + (size_t)cur * d.planeM;
  const double *Gc = d.G + (size_t)cur * d.planeM;
  const double mu = d.mu[b];
  double x[NX], u[NU], y[M], s[M], g[M];
  ld<NX>(Xc + GI(t, NX, 0), kLS, x);
  ld<NU>(Uc + GI(t, NU, 0), kLS, u);
  ld<M>(Yc + GI(t, M, 0), kLS, y);
  ld<M>(Sc + GI(t, M, 0), kLS, s);
  ld<M>(Gc + GI(t, M, 0), kLS, g);
  double A[WITH_DERIVS ? NX * NX : 1], Bq[WITH_DERIVS ? NX * NU : 1], c[L::SIZE];
  condense_eval<Model, Cons, WITH_DERIVS>(P, xrt, t, x, u, y, s, g, mu, A, Bq, c);
  if constexpr (WITH_DERIVS) {
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) d.A[GI(t, NX * NX, i)] = A[i];
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) d.Bm[GI(t, NX * NU, i)] = Bq[i];
  }
  double *o = d.cst + GT(t, L::SIZE, 0);
  const size_t ts = TSTRIDE;
  st<L::SIZE>(o, ts, c);
}

// ================================================================================ K2 (lean)
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_ipddp_lean(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                            int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  constexpr int CST = L::SIZE;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  if (count_iter) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  typename Obj::Ctx oc;   // l_xx = 2 Q dt, l_uu = 2 R dt: loop-invariant, kept in scalar registers
  Obj::load(P, oc);
  const double *Qc = Obj::kHoist ? oc.Q : oc.Qp;
  const double *Rc = Obj::kHoist ? oc.R : oc.Rp;
  for (;;) {
    ++nb;
    double xN[NX], Vx[NX], Vxx[NX * NX];
    ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
    Obj::final_grad(P, xN, Vx);
    const double *Qf = P->pool + P->off_Qf;
    {
      double H2[NX * NX];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) H2[i] = 2.0 * Qf[i];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (H2[i * NX + c] + H2[c * NX + i]);
    }
    dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0;
    st<NX>(d.Vx + GI(N, NX, 0), kLS, Vx);
    st<NX * NX>(d.Vxx + GI(N, NX * NX, 0), kLS, Vxx);
    bool fail = false;
    // two prefetch groups: a wave stalls at VMEM issue beyond ~sixteen outstanding 512-B rows (see K4)
    struct StepIn { double A[NX * NX]; };
    struct StepIn2 { double Bm[NX * NU], c[CST]; };
    auto load_step = [&](int tt, StepIn &r) { ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A); };
    auto load_step2 = [&](int tt, StepIn2 &r) {
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<CST>(d.cst + GI(tt, CST, 0), kLS, r.c);
    };
    // Ping-pong record buffers, loop unrolled by two: a `cur = next` register copy per step makes the compiler
    // wait for the freshly issued prefetch right where the copy lands (measured: vmcnt(0) mid-step).
    auto step = [&](const int t, const StepIn &cs1, const StepIn2 &cs, StepIn &nxt, StepIn2 &nxt2) -> bool {
      const int tp = t > 0 ? t - 1 : 0;   // unconditional (clamped) prefetch: no branch for the optimiser to merge
      load_step(tp, nxt);
      PIPELINE_FENCE();
      const double (&A)[NX * NX] = cs1.A; const double (&Bm)[NX * NU] = cs.Bm;
      const double *cx = cs.c + L::CX, *cu = cs.c + L::CU, *WQyu = cs.c + L::WQYU, *QyuSir = cs.c + L::QYUSIR;
      double Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += A[k * NX + i] * Vx[k];
        Qx[i] = cx[i] + s2; }
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += Bm[k * NU + i] * Vx[k];
        Qu[i] = cu[i] + s2; }
      q_blocks<NX, NU>(Qc, Rc, A, Bm, Vxx, Qxx, Qux, Quu);
      if (!o.use_ilqr) {   // full DDP: second-order dynamics terms at (x_t, u_t) (fetched only in this mode)
        double x[NX], u[NU];
        ld<NX>(Xc + GI(t, NX, 0), kLS, x);
        ld<NU>(d.U + (size_t)cur * d.planeU + GI(t, NU, 0), kLS, u);
        ddp_tensor_terms<Model>(P, x, u, Vx, Qxx, Qux, Quu);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the second group behind the Q-block arithmetic
      load_step2(tp, nxt2);
      PIPELINE_FENCE();
      __builtin_amdgcn_sched_barrier(0);
      double Qr[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qr[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]) + WQyu[i * NU + c];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
      double kk[NU], KK[NU * NX];
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Qr[0], Qu[0] + QyuSir[0]);
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double rhs = Qux[c];
          if constexpr (Cons::HAS_X) rhs = rhs + cs.c[L::WQYX + c];
          KK[c] = -ldlt1_solve(Qr[0], rhs);
        }
      } else {
        LDLTd<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i] + QyuSir[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        for (int c = 0; c < NX; ++c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double rhs = Qux[i * NX + c];
            if constexpr (Cons::HAS_X) rhs = rhs + cs.c[L::WQYX + i * NX + c];
            col[i] = rhs;
          }
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
        }
      }
      st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      st<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
      // condensed, un-regularised Q blocks (ipddp_solver.cpp:1488-1492)
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] += QyuSir[i];
      if constexpr (Cons::HAS_X) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Qx[i] += cs.c[L::QYXSIR + i];
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) Qxx[i] += cs.c[L::WXQYX + i];
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) Qux[i] += cs.c[L::WQYX + i];
      }
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] += WQyu[i];
      inf_pr = dmax(inf_pr, cs.c[L::IPR]); inf_comp = dmax(inf_comp, cs.c[L::ICOMP]);
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
      return true;
    };
    StepIn a1, b1;
    StepIn2 a2, b2;
    load_step(N - 1, a1);
    load_step2(N - 1, a2);
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
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;        // K3 lowers them by atomic min
  bool conv = false;
  if (ok) {
    d.dV0[b] = dV0; d.dV1[b] = dV1; d.inf_du[b] = inf_du; d.step_norm[b] = step_norm;
    d.inf_pr[b] = inf_pr; d.inf_comp[b] = inf_comp;
    // checkEarlyConvergence (ipddp_solver.cpp:925-958), barrier problem
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
    if (!conv || force) {
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): dX stack for K3
      double dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = 0.0;
      struct RIn { double kk[NU], KK[NU * NX], A[NX * NX], Bm[NX * NU]; };
      auto load_r = [&](int tt, RIn &r) {
        ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
        ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
        ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
        ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      };
      RIn rn;
      load_r(0, rn);
      for (int t = 0; t < N; ++t) {
        RIn rc = rn;
        if (t + 1 < N - 1) load_r(t + 1, rn);
        PIPELINE_FENCE();
        st<NX>(d.dX + GI(t, NX, 0), kLS, dx);
        if (t < N - 1) {
          double du[NU], dxn[NX];
#pragma unroll
          for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += rc.KK[i * NX + j] * dx[j];
            du[i] = rc.kk[i] + a; }
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += rc.A[i * NX + j] * dx[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) c += rc.Bm[i * NU + j] * du[j];
            dxn[i] = (a + c) + 0.0;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
        }
      }
    }
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; return; }
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, mu); return; }
  d.phase[b] = PH_FWD1;
}

DEV void atomic_min_pos(double *addr, double v) {   // v >= 0: the IEEE bit pattern orders like an unsigned integer
  if (!(v >= 0.0)) v = 0.0;
  atomicMin((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// ================================================================================ K3
// The per-(trajectory, step) arithmetic of K3 in two device functions, shared by the wide kernel below and by the helper wavefronts of the
// role-split sweep (kernels_coop.hpp): post_rows = everything that does not need dx_t -- k_y, k_s, Y S^-1 and the rows of K_y, K_s
// (ipddp_solver.cpp:1458-1486); post_caps = dS, dY of the linear-policy rollout (:1522-1532) and the fraction-to-boundary caps (:2939-2988).
// HAVE_YS: ysv[] comes in (the role-split sweep's helpers stored the ratios when they condensed the step) instead of being formed here.
template <class Model, class Cons, bool HAVE_YS = false>
DEV void post_rows(const ProblemDev *__restrict__ P, const double *x, const double *uj, const double *y, const double *s, const double *g,
                   const double *kk, const double *KK, const double mu, double *ky, double *ksv, double *ysv, double *Ky, double *Ksm) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  double Qyx[M * NX], Qyu[M * NU];
#pragma unroll
  for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
  Cons::template jac<NX, NU>(P, x, uj, Qyx, Qyu);
#pragma unroll
  for (int r = 0; r < M; ++r) {
    const double ss = dmax(s[r], s_floor);
    double YSr;
    if constexpr (HAVE_YS) YSr = ysv[r]; else { YSr = clip_pos(y[r], ss); ysv[r] = YSr; }
    const double rp = g[r] + s[r];
    const double rc = y[r] * s[r] - mu;
    const double rhat = y[r] * rp - rc;
    double temp = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) temp += Qyu[r * NU + i] * kk[i];
    ky[r] = clip_sgn(rhat + y[r] * temp, ss);
    ksv[r] = (-rp) - temp;
#pragma unroll
    for (int cc = 0; cc < NX; ++cc) {
      double s2 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) s2 += Qyu[r * NU + i] * KK[i * NX + cc];
      const double inner = Qyx[r * NX + cc] + s2;
      Ky[r * NX + cc] = dmin(dmax(YSr * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
      Ksm[r * NX + cc] = (-Qyx[r * NX + cc]) - s2;
    }
  }
}
template <int NX, int M>
DEV void post_caps(const double *ky, const double *ksv, const double *Ky, const double *Ksm, const double *s, const double *y, const double *dx,
                   const double tau, double &apr, double &adu) {
#pragma unroll
  for (int r = 0; r < M; ++r) {
    double a = 0.0, c = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) { a += Ksm[r * NX + j] * dx[j]; c += Ky[r * NX + j] * dx[j]; }
    const double ds = ksv[r] + a;
    const double dy = dmin(dmax(ky[r] + c, -kMaxBarrierRatio), kMaxBarrierRatio);
    // apr, adu start at 1 and only fall, so a quotient >= 1 changes nothing; with a negative denominator the exact quotient num / den
    // is >= 1 exactly when num <= den, and rounding to nearest keeps it >= 1 (1 is representable): the division is evaluated only by
    // wavefronts in which some lane's cap can bind -- same bits, most steps of a solve skip both divisions
    const double nps = -tau * s[r], npy = -tau * y[r];
    if (__builtin_amdgcn_ballot_w64(ds < 0.0 && !(nps <= ds)) != 0ull) { if (ds < 0.0) apr = dmin(apr, nps / ds); }
    if (__builtin_amdgcn_ballot_w64(dy < 0.0 && !(npy <= dy)) != 0ull) { if (dy < 0.0) adu = dmin(adu, npy / dy); }
  }
}

template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_post(DevBuf d, const ProblemDev *__restrict__ Pk, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  if (force) { if (!d.bwd_ok[b]) return; }
  else if (d.phase[b] != PH_FWD1) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Sc = d.S + (size_t)cur * d.planeM;
  const double *Yc = d.Y + (size_t)cur * d.planeM;
  const double *Gc = d.G + (size_t)cur * d.planeM;
  const double mu = d.mu[b];
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  double x[NX], y[M], s[M], g[M], kk[NU], KK[NU * NX], dx[NX];
  ld<NX>(Xc + GI(t, NX, 0), kLS, x);
  ld<M>(Yc + GI(t, M, 0), kLS, y);
  ld<M>(Sc + GI(t, M, 0), kLS, s);
  ld<M>(Gc + GI(t, M, 0), kLS, g);
  ld<NU>(d.k + GI(t, NU, 0), kLS, kk);
  ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
  ld<NX>(d.dX + GI(t, NX, 0), kLS, dx);
  double uj[NU];
  if constexpr (Cons::NEEDS_U) ld<NU>(d.U + (size_t)cur * d.planeU + GI(t, NU, 0), kLS, uj);
  double ky[M], ksv[M], Ky[M * NX], Ksm[M * NX], ysv[M];
  double apr = 1.0, adu = 1.0;
  post_rows<Model, Cons>(P, x, uj, y, s, g, kk, KK, mu, ky, ksv, ysv, Ky, Ksm);
  post_caps<NX, M>(ky, ksv, Ky, Ksm, s, y, dx, tau, apr, adu);
  // The feedback blocks K_s = -(G_x + G_u K), K_y = clamp(YS (G_x + G_u K)) are NOT stored: the rollout consumer
  // rebuilds the rows it needs from K and YS with this very arithmetic (2 M NX fewer rows per step to write here
  // and to read there, where VMEM issue is the scarce resource).
  st<M>(d.ky + GI(t, M, 0), kLS, ky);
  st<M>(d.ks + GI(t, M, 0), kLS, ksv);
  st<M>(d.ys + GI(t, M, 0), kLS, ysv);
  if (apr < 1.0) atomic_min_pos(d.apr_max + b, apr);
  if (adu < 1.0) atomic_min_pos(d.adu_max + b, adu);
}

// ================================================================================ K4b
// Costate trial Lambda_new[t] = Lambda[t] + alpha_pr V_x[t] + V_xx[t] (x_new[t] - x[t]) (ipddp_solver.cpp:1613-1616,
// 1660-1663) for problems without terminal constraints, where nothing reads it back during the solve: it is taken
// off the serial rollout chain (24 of K4's 58 loads per step at nx = 4) and evaluated at (batch x N+1) width for the
// trials that passed every other test.  A trial whose costate is not finite is failed here exactly as the reference
// fails it inside forwardPass, before k_update applies the acceptance rule; under the first-success rule only the
// first surviving trial is evaluated.
// Best-merit rule: the candidate winner of every trajectory -- least merit among the trials that passed every other test (flag 1 or 2),
// first of equals, the order and comparison k_update uses.  One lane per trajectory, launched in front of k_costate.
// First trial of [a0, a0 + na) that passed every other test (flag 1 or 2): the one the first-success rule evaluates.  The flags are
// fetched eight at a time so that the loads are in flight together -- walking them one by one puts up to n_alpha dependent L2 round
// trips in front of every (trajectory, step) lane of K4b.
DEV int first_surviving_trial(const DevBuf &d, int a0, int na, int b) {
  for (int base = 0; base < na; base += 8) {
    int f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (base + i < na) ? d.t_success[(size_t)(a0 + base + i) * d.Bp + b] : 0;
    int hit = -1;
#pragma unroll
    for (int i = 7; i >= 0; --i) if (f[i] != 0) hit = i;
    if (hit >= 0) return a0 + base + hit;
  }
  return -1;
}

template <int kUnused = 0>
__global__ __launch_bounds__(64) void k_pick_candidate(DevBuf d, int a0, int na, int phase_req, int force) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  double best = INFINITY; int only = -1;
  for (int a = a0; a < a0 + na; ++a) {
    const size_t ti = (size_t)a * d.Bp + b;
    const int sc = d.t_success[ti]; const double mt = d.t_merit[ti];
    if (sc != 0 && mt < best) { best = mt; only = a; }
  }
  d.cand[b] = only;
}

// K4b for large states when ONE trial per trajectory is evaluated (every launch of the solve loop: the first surviving trial under the
// first-success rule, the candidate winner under the best-merit rule).  A separate kernel because register allocation is per kernel: the
// general form below holds V_xx as a register-resident triangle (105 doubles at nx = 14: 494 registers plus scratch, one wave per SIMD).
template <class Model>
__global__ __launch_bounds__(64) void k_costate_one(DevBuf d, int a0, int na, int phase_req, int first_only) {
  constexpr int NX = Model::NX;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  if (d.phase[b] != phase_req) return;
  const int cur = d.cur[b];
  // V_xx is STREAMED row by row against dx = x_new - x_old.  Same products in the same order as the general kernel (V_xx is stored
  // exactly symmetric, so row i of the full matrix holds the values the triangle supplies there); the rows of a trial whose costate
  // turns out non-finite may be partly written before the flag is set -- nobody reads them.
  const int only = first_only == 2 ? d.cand[b] : first_surviving_trial(d, a0, na, b);
  if (only < 0) return;
  const size_t ti = (size_t)only * d.Bp + b;
  const int slot = trial_slot(cur, only);
  const double a_pr = d.t_apr[ti];
  double dx[NX];
  {
    double xo[NX], xn[NX];
    ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, xo);
    ld<NX>(d.X + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, xn);
#pragma unroll
    for (int j = 0; j < NX; ++j) dx[j] = xn[j] - xo[j];
  }
  const double *vb = d.Vxx + GI(t, NX * NX, 0);
  const double *lop = d.Lam + (size_t)cur * d.planeX + GI(t, NX, 0), *vxp = d.Vx + GI(t, NX, 0);
  double *lamp = d.Lam + (size_t)slot * d.planeX + GI(t, NX, 0);
  bool finite = true;
#pragma unroll 2
  for (int i = 0; i < NX; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) s += vb[(size_t)(i * NX + j) * kLS] * dx[j];
    const double lam = (lop[(size_t)i * kLS] + a_pr * vxp[(size_t)i * kLS]) + s;
    finite = finite && dfinite(lam);
    lamp[(size_t)i * kLS] = lam;
  }
  if (!finite || ((d.fail_costate_mask >> only) & 1)) d.t_success[ti] = 2;
}

template <class Model>
__global__ __launch_bounds__(64) void k_costate(DevBuf d, int a0, int na, int phase_req, int force, int first_only) {
  constexpr int NX = Model::NX;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  const int cur = d.cur[b];
  // V_xx is stored exactly symmetric (every sweep writes 0.5 (M + M^T), the terminal block is sym(2 Q_f) [+ folded
  // terminal terms, symmetrised the same way]): only the upper triangle is fetched -- nx (nx - 1) / 2 rows less.
  constexpr int NT = NX * (NX + 1) / 2;
  double xo[NX], lo[NX], vx[NX], vt[NT];
  ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, xo);
  ld<NX>(d.Lam + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, lo);
  ld<NX>(d.Vx + GI(t, NX, 0), kLS, vx);
  {
    const double *vb = d.Vxx + GI(t, NX * NX, 0);
    int k = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = i; j < NX; ++j) vt[k++] = vb[(size_t)(i * NX + j) * kLS];
  }
  // first_only == 2 (best-merit rule, cddp_solver_base.cpp:264-317): the only costate rows anybody reads are the WINNER's, and
  // the winner is the successful trial of least merit.  Its candidate is picked here exactly as k_update picks it (same order, strict
  // <) among the trials that passed every other test -- flag 1 or 2, so that a block which already sees another block's "costate not
  // finite" mark still picks the same one -- and only that trial is evaluated; should its costate turn out non-finite, k_update moves
  // to the next-best trial and evaluates that one itself (costate_trial_serial).  With all sixteen trials of the C5 share passing,
  // evaluating every one wrote 1.8 GB of costate rows per launch of which 1 / 16 was ever read.
  int a_lo = a0, a_hi = a0 + na;
  if (first_only != 0) {   // best-merit: picked once per trajectory by k_pick_candidate (the same launch sequence); first-success: here
    const int only = first_only == 2 ? d.cand[b] : first_surviving_trial(d, a0, na, b);
    if (only < 0) return;
    a_lo = only; a_hi = only + 1;
  }
  for (int a = a_lo; a < a_hi; ++a) {
    const size_t ti = (size_t)a * d.Bp + b;
    if (!d.t_success[ti]) continue;
    const int slot = trial_slot(cur, a);
    const double a_pr = d.t_apr[ti];
    double xn[NX], lam[NX];
    ld<NX>(d.X + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, xn);
    bool finite = true;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int lo_ = i < j ? i : j, hi_ = i < j ? j : i;   // V_xx[i][j] from the stored upper triangle
        s += vt[lo_ * NX - lo_ * (lo_ - 1) / 2 + (hi_ - lo_)] * (xn[j] - xo[j]);
      }
      lam[i] = (lo[i] + a_pr * vx[i]) + s;
      finite = finite && dfinite(lam[i]);
    }
    // A non-finite costate fails the trial (ipddp_solver.cpp:1613-1616).  The flag value 2 ("passed every other
    // test, costate not finite") is the ONLY mutation of t_success in this grid-wide kernel and is still non-zero:
    // the blocks of the other steps pick the same trial whether or not they have seen it, so the launch is free of
    // ordering effects; k_update treats 2 as a failed trial and evaluates the costate of the next candidate itself
    // (costate_trial_serial) when the first-success rule stopped this kernel at the failed one.
    if (!finite || ((d.fail_costate_mask >> a) & 1)) d.t_success[ti] = 2;
    else st<NX>(d.Lam + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, lam);
  }
}

// ================================================================================ K4 (two-role)
// Line-searched IPDDP rollout for path-constrained problems without terminal constraints, as a PRODUCER /
// CONSUMER pair of wavefronts per (64-trajectory tile, alpha).  Only x_{t+1} = f(x_t, u_t(x_t)) is a true serial
// chain, and one wave per SIMD leaves every dependent f64 operation's latency exposed, so the work is split:
//   wave 0 (producer)  u_t = u + a k + K dx, x_{t+1} = f(x_t, u_t), l_f(x_N), stores X, U of the trial
//                      (ipddp_solver.cpp:1618-1627);
//   wave 1 (consumer)  slack / dual trial + fraction-to-boundary test (:1629-1658), running cost and g(x_t, u_t)
//                      (:1726-1748), the theta / barrier-merit / residual terms (:2778-2937), the filter test
//                      (:1785-1834) and the trial record.
// Channel: an LDS ring of kRing steps carrying (x_t, dx_t, u_t) per lane, a produced-step counter and a
// consumed-step counter (both LDS words, polled with s_sleep).  The producer publishes step t as soon as u_t is
// known -- before it integrates -- so the consumer works on step t while the producer is inside the RK4 of step t;
// the running cost is summed by the consumer (same t order), the producer hands over l_f(x_N) at the end.
// A lane whose rollout went non-finite is published through s_pstat.
// Why the consumer reads nothing of the trial from global memory: a wave stalls at VMEM issue once ~sixteen 512-B
// row loads are outstanding (profiles/ubench/vmem.hip), so row loads, not arithmetic, set the consumer's pace; the
// ring removes 2 NX + NU of them per step and K_s / K_y are rebuilt from K and YS (see k_post).
// Every lane runs straight-line code with UNCONDITIONAL stores (a dead lane keeps re-evaluating its frozen state;
// rows of a failed trial are never read): with stores inside divergent branches the waitcnt pass cannot count the
// operations behind the prefetch and falls back to vmcnt(0), i.e. it waits for a store acknowledge every step.
// TERM = true: terminal-equality layouts solved by the cooperative reduced-LQR sweep (kernels_te.hpp; no terminal
// inequality): the producer hands x_N over, the consumer appends the multiplier trial and the terminal terms of
// computeTheta / computeBarrierMerit (ipddp_solver.cpp:1711-1723, 2812-2845, 2866-2878) in the reference's order.
#ifdef CDDP_K4_TIMING   // experiment: per-wave duration and steps walked of the LAST rollout launch (profiles/scripts/k4_block_times.py)
__device__ unsigned long long g_k4_times[16384 * 4];
#define K4_TIME_BEGIN const unsigned long long k4_t0 = wall_clock64();
#define K4_TIME_END(role, steps) do { if (lane == 0) { const size_t k4_i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (role) * 2; \
    g_k4_times[k4_i] = wall_clock64() - k4_t0; g_k4_times[k4_i + 1] = (unsigned long long)(steps); } } while (0)
#else
#define K4_TIME_BEGIN
#define K4_TIME_END(role, steps)
#endif
// NC = 2 (round 5): TWO consumer waves per (tile, alpha), consumer c taking the steps t = c (mod 2).  For layouts whose consumer, not the
// dynamics chain, sets the pace (unicycle with its box + ball rows: the Euler producer idles 60 % of the launch, profiles/r05_rollout_roles.md)
// the per-step row work is independent from step to step; what is ORDERED across steps is kept ordered: the running cost is summed by the
// producer (it has x_t, u_t and the slack), the first constraint object's |g + s| terms are parked per step like the other objects' and summed
// in t order after the rollout by consumer 0, which also merges the two waves' maxima / minima / first failing step (order-free) and applies
// the filter test.  A consumer whose 64 trials have all failed posts the step at which it saw that (s_abort); the other finishes its own steps
// below that step (an earlier failure it alone can see decides the step count of the trial) and stops.  Bitwise the one-consumer kernel
// (tests/test_gpu_parity.py::test_two_consumer_rollout_agrees_bitwise); NC = 1 is unchanged code.
template <class Model, class Cons, bool TERM = false, int NC = 1>
__global__ __launch_bounds__(64 * (1 + NC)) void k_forward_ipddp_pc(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                                    int a0, int phase_req, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef Objective<NX, NU> Obj;
  static_assert(M > 0, "two-role rollout is for path-constrained problems");
  constexpr int RW = 2 * NX + NU;          // doubles per lane per step: x_t, dx_t, u_t
  constexpr int kRing = RW <= 10 ? 8 : (RW <= 20 ? 4 : 2);   // steps in flight between the two waves (<= 40 KB of LDS)
  __shared__ double s_ring[kRing * RW * 64];
  static_assert(NC == 1 || NC == 2, "one or two consumer waves");
  __shared__ int s_prod;          // steps published by the producer
  __shared__ int s_consv[2];      // steps retired by consumer c (its own steps: c, c + NC, ...), + 1
  [[maybe_unused]] __shared__ int s_abort;      // NC = 2: the step at which a consumer saw all 64 trials failed (INT_MAX: none)
  [[maybe_unused]] __shared__ int s_done1;      // NC = 2: consumer 1 has published its partial results
  constexpr int kL2 = NC > 1 ? 64 : 1;   // (no LDS for these in the one-consumer kernel: the nx >= 12 rings sit at an occupancy edge)
  [[maybe_unused]] __shared__ int s_pfail[kL2], s_palive[kL2];   // NC = 2: consumer 1's first failing step / alive flag per lane
  [[maybe_unused]] __shared__ double s_pmax[4 * kL2];            // NC = 2: consumer 1's ev_max, ev_icomp, ys_lo, ys_hi per lane
  [[maybe_unused]] __shared__ double s_prun[kL2];                // NC = 2: the producer lane's running cost
  __shared__ int s_pstat[64];     // first step at which the producer lane went non-finite (N + 2 = never)
  __shared__ double s_pcost[64];  // the producer lane's terminal cost l_f(x_N)
  __shared__ double s_xN[TERM ? NX * 64 : 1];   // the producer lane's x_N (terminal residual of the trial)
  __shared__ double s_obj[Obj::kStage];         // Q dt | R dt | x_ref of a large plant (Objective::stage)
  const int lane = threadIdx.x & 63;
  K4_TIME_BEGIN
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
  const bool producer = wave == 0;
  [[maybe_unused]] const int ci = NC > 1 ? wave - 1 : 0;   // consumer index (NC = 2)
  int *const s_cons = &s_consv[0];
  const int b = blockIdx.x * 64 + lane;
  const int a = a0 + blockIdx.y;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const bool active = (b < d.B) && (force || d.phase[b] == phase_req);
  if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;   // same mask in every wave: all leave
  if (producer) { s_pstat[lane] = N + 2; if (lane == 0) { s_prod = 0; s_consv[0] = 0; s_consv[1] = 0; if constexpr (NC > 1) { s_abort = 0x7fffffff; s_done1 = 0; } } }
  Obj::stage(P, s_obj, (int)threadIdx.x, 64 * (1 + NC));
  __syncthreads();
  // Inactive lanes (padding, or a trajectory in another phase) run along on their OWN rows: their trial slots are
  // scratch (trial_slot never returns the current slot), so unconditional stores need no exec-mask branches.
  const int bb = (b < d.B) ? b : 0;
  const int cur = (b < d.B) ? d.cur[b] : 0;
  const int slot = trial_slot(cur, a);
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double alpha = P->alphas[a];
  const double a_pr = dmin(alpha, d.apr_max[bb]);
  auto wait_ge = [&](int *ctr, int need) -> int {
    int v;
    while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    return v;
  };
  const int kAbort = 2 * N + kRing;   // s_cons value with which a consumer whose 64 trials have all failed releases AND stops the producer

  if (producer) {
    // ------------------------------------------------------------------ producer: the dynamics chain
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
    DynCtx dc;   // loop-invariant constants in scalar registers
    dc.load(P->integrator, P->dt, P->mp);
    [[maybe_unused]] typename Obj::Ctx oc_p;   // NC = 2: the producer sums the running cost
    [[maybe_unused]] double run_cost_p = 0.0;
    if constexpr (NC > 1) Obj::load_staged(P, oc_p, s_obj);
    // Prime the VMEM queue with the store pattern of one step (rows of step 0, rewritten by iteration 0): the
    // waitcnt pass joins the loop-entry state with the back-edge state, and an entry state whose newest
    // operations are the loads would make every iteration wait for vmcnt(0), i.e. for its own last stores.
    auto prime = [&]() {
      double z[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = 0.0;
      st<NU>(Un + GI(0, NU, 0), kLS, z);
      st<NX>(Xn + GI(1, NX, 0), kLS, z);
    };
    // Large records (nx >= 12: the gain block alone is nu*nx rows) live in ONE register set (a ping-pong copy would push the
    // kernel into scratch).  The record of step t + 1 is fetched into that same set as soon as u_t is formed -- its
    // registers are dead from there on -- so the loads fly behind the integrator stages instead of being waited for at
    // the top of the next step (round 2 counters: the rollout waves of the 7-joint arm waited 52 % of their cycles).
    // (Only while the record is moderate: the 126-double record of the 7-joint arm, kept live across the integrator, pushes
    //  the kernel into scratch -- measured 680 -> 1039 ms of rollout class at C5 -- so the largest records are still fetched
    //  at the top of their own step.)
    constexpr bool kPing = sizeof(StepIn) <= 40 * sizeof(double);
    constexpr bool kEarly = !kPing && sizeof(StepIn) <= 96 * sizeof(double);
    // Prefetch distance of the ping-pong form, in steps (kDepth + 1 rotating register sets).  Measured at C2 on one box
    // (profiles/r03_ladder_sweep.md): producer / consumer distance 1 / 1: 22.7 ms of rollout class per solve, 3 / 1: 23.5, 1 / 2: 23.1,
    // 5 / 3: 25.0 -- one step of look-ahead already covers the loaded memory latency; more sets only cost registers.
#ifndef CDDP_K4_DEPTH_P
#define CDDP_K4_DEPTH_P 1
#endif
#ifndef CDDP_K4_DEPTH_C
#define CDDP_K4_DEPTH_C 1
#endif
    constexpr int kDepthP = (kPing && sizeof(StepIn) <= 12 * sizeof(double)) ? CDDP_K4_DEPTH_P : 1;
    // The largest records (7-joint arm: 126 doubles) are streamed: the old state row and, per control i, the gain row with
    // u_old[i], k[i] -- two chunk buffers; x_old of step t + 1 and its first chunk are fetched behind the integrator (42
    // doubles live there instead of 126).  Same sums.  With the consumer's chunks (below) the kernel fits two wavefronts per SIMD.
    constexpr bool kChunkP = !kPing && !kEarly;
    struct PChunk { double K[NX], uo, kk; };
    auto load_pchunk = [&](int tt, const int i, PChunk &c) {
      ld<NX>(d.K + GI(tt, NU * NX, i * NX), kLS, c.K);
      c.uo = Uc[GI(tt, NU, i)]; c.kk = d.k[GI(tt, NU, i)];
    };
    PChunk pk0, pk1;
    double xo_c[NX];
    auto step = [&](const int t, StepIn &cs, StepIn &nxt) {
      if constexpr (kPing) {
        const int tn = t + kDepthP < N ? t + kDepthP : N - 1;   // unconditional (clamped) prefetch
        load_step(tn, nxt);
      }
      PIPELINE_FENCE();
      double dx[NX], u[NU], xn[NX];
      bool finite = true;
      if constexpr (kChunkP) {
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = x[i] - xo_c[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          if (i + 1 < NU) { if ((i & 1) == 0) load_pchunk(t, i + 1, pk1); else load_pchunk(t, i + 1, pk0); }
          __builtin_amdgcn_sched_barrier(0);
          const PChunk &c = (i & 1) == 0 ? pk0 : pk1;
          double s1 = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) s1 += c.K[j] * dx[j];
          u[i] = (c.uo + a_pr * c.kk) + s1;
          finite = finite && dfinite(u[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
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
      }
      // publish step t.  Ring slots free up as the consumer retires steps; the counter is polled once every
      // kRing/2 steps for the next kRing/2 slots (an LDS round trip on the chain otherwise).
      // (round 5: a consumer that has given the whole tile up -- every trial failed its fraction-to-boundary test -- also STOPS the
      //  producer here: nobody reads the rows of those trials, and the wave otherwise kept its SIMD busy to the end of the horizon)
      if (t >= kRing && (t % (kRing / 2)) == 0) {   // kRing / 2 >= 1
        if (wait_ge(s_cons, t - kRing / 2) >= kAbort) alive = false;
        if constexpr (NC > 1) { if (wait_ge(&s_consv[1], t - kRing / 2) >= kAbort) alive = false; }
      }
      {
        double *rs = s_ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
        for (int i = 0; i < NX; ++i) { rs[i * 64] = x[i]; rs[(NX + i) * 64] = dx[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) rs[(2 * NX + i) * 64] = u[i];
        if (alive && !finite) { s_pstat[lane] = t; alive = false; }
        RING_FENCE();
        __hip_atomic_store(&s_prod, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if constexpr (NC > 1) run_cost_p += Obj::running_cost(oc_p, xrt, t, x, u);   // t-ordered sum (the consumers take every other step)
      if constexpr (kEarly) {   // next step's record into the (now dead) register set, behind the integrator
        load_step(t + 1 < N ? t + 1 : t, cs);
        PIPELINE_FENCE();
      }
      if constexpr (kChunkP) {   // x_old and the first chunk of the next step, behind the integrator
        const int tn = t + 1 < N ? t + 1 : t;
        ld<NX>(Xc + GI(tn, NX, 0), kLS, xo_c);
        load_pchunk(tn, 0, pk0);
        PIPELINE_FENCE();
      }
      Stepper<Model>::step(dc, x, u, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) finite = finite && dfinite(xn[i]);
      if (alive && !finite) { s_pstat[lane] = t; alive = false; }
      st<NU>(Un + GI(t, NU, 0), kLS, u);
      st<NX>(Xn + GI(t + 1, NX, 0), kLS, xn);
      if (alive) {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
      }
    };
    StepIn ra;
    int t = 0;
    if constexpr (kPing) {
      StepIn R[kDepthP + 1];   // step t lives in R[t % (kDepthP + 1)]; every index below is a compile-time constant
#pragma unroll
      for (int j = 0; j < kDepthP; ++j) load_step(j < N ? j : N - 1, R[j]);
      prime();
      for (; t + kDepthP < N; t += kDepthP + 1) {
#pragma unroll
        for (int j = 0; j <= kDepthP; ++j) step(t + j, R[j], R[(j + kDepthP) % (kDepthP + 1)]);
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) { t = N; break; }   // nothing downstream reads the rows any more
      }
#pragma unroll
      for (int j = 0; j <= kDepthP; ++j) if (t + j < N) step(t + j, R[j], R[(j + kDepthP) % (kDepthP + 1)]);
    } else {
      if constexpr (kChunkP) { ld<NX>(Xc + GI(0, NX, 0), kLS, xo_c); load_pchunk(0, 0, pk0); } else load_step(0, ra);
      prime();
      for (; t < N; ++t) {
        step(t, ra, ra);
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
      }
    }
    if (alive) s_pcost[lane] = Obj::terminal_cost(P, x);
    if constexpr (NC > 1) s_prun[lane] = run_cost_p;
    if constexpr (TERM) {
#pragma unroll
      for (int i = 0; i < NX; ++i) s_xN[i * 64 + lane] = x[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(&s_prod, N + kRing + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    K4_TIME_END(0, t);
    return;
  }

  // -------------------------------------------------------------------- consumer: everything off the chain
  const double *Sc = d.S + (size_t)cur * d.planeM;
  const double *Yc = d.Y + (size_t)cur * d.planeM;
  double *Sn = d.S + (size_t)slot * d.planeM;
  double *Yn = d.Y + (size_t)slot * d.planeM;
  double *Gn = d.G + (size_t)slot * d.planeM;
  const double mu = d.mu[bb];
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  const double a_du = dmin(alpha, d.adu_max[bb]);
  const size_t ti = (size_t)a * d.Bp + bb;
  bool alive = active;
  int seen_prod = 0;     // last value of the producer's counter this wave saw (wave-uniform)
  int fail_t = N;        // steps completed before the trial was abandoned (kept in a register, stored at the exits)
  if (alive && ci == 0) {
    atomicAdd(d.launched, 1ull);
    d.t_apr[ti] = a_pr; d.t_adu[ti] = a_du;
    d.t_success[ti] = 0;
    d.t_cost[ti] = d.cost[b]; d.t_merit[ti] = d.phi[b]; d.t_theta[ti] = d.theta[b];
    d.t_inf_pr[ti] = 0.0; d.t_inf_comp[ti] = 0.0;
  }
  double ev_total0 = 0.0, ev_max = 0.0, ev_icomp = 0.0, ys_lo = INFINITY, ys_hi = -INFINITY, run_cost = 0.0;
  const bool l2norm = o.ipddp_theta_norm_l2 != 0;
  // per-step record of the CURRENT iterate (one prefetch group, <= 16 rows for the C2 layout)
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
  typename Cons::Ctx cc;   // bounds / centres / scales in scalar registers
  Cons::load(P, cc);
  typename Obj::Ctx oc;    // running-cost matrices (scalar registers for small plants, LDS for large ones)
  Obj::load_staged(P, oc, s_obj);
  auto prime = [&]() {   // prime the VMEM queue with one step's store pattern (see the producer): the rows of this wave's FIRST step
    double z[M];
#pragma unroll
    for (int i = 0; i < M; ++i) z[i] = 0.0;
    const int tp = (NC > 1 && ci < N) ? ci : 0;
    st<M>(Sn + GI(tp, M, 0), kLS, z);
    st<M>(Yn + GI(tp, M, 0), kLS, z);
    st<M>(Gn + GI(tp, M, 0), kLS, z);
    double *ev = d.ev + GI((size_t)a * N + tp, 2 * Cons::NSEG, 0);
#pragma unroll
    for (int c = 0; c < Cons::NSEG; ++c) { if (c > 0 || NC > 1) ev[(size_t)(Cons::NSEG + c) * kLS] = 0.0; ev[(size_t)c * kLS] = 0.0; }
  };
  constexpr bool kPing = sizeof(StepIn) <= 40 * sizeof(double);   // see the producer
  constexpr bool kEarly = !kPing && sizeof(StepIn) <= 96 * sizeof(double);
  constexpr int kDepthC = (kPing && sizeof(StepIn) <= 16 * sizeof(double)) ? CDDP_K4_DEPTH_C : 1;
  // The largest records of control-box layouts (7-joint arm: 169 doubles per lane and step) are streamed in NU chunks
  // instead: chunk i = gain row i and the slack / dual entries of the two constraint rows that read it (upper and lower
  // bound of control i), two chunk buffers, chunk i + 1 in flight while chunk i is reduced, chunk 0 of the next step behind
  // the cost / barrier terms.  Same products, same sums; the row pairs are visited as (0, NU), (1, NU + 1), ...
  constexpr bool kChunk = !kPing && !kEarly && UDiag<Cons>::value && M == 2 * NU;
  struct Chunk { double K[NX], s[2], y[2], ksv[2], ky[2], ys[2]; };
  auto load_chunk = [&](int tt, const int i, Chunk &c) {
    ld<NX>(d.K + GI(tt, NU * NX, i * NX), kLS, c.K);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = i + h * NU;
      c.s[h] = Sc[GI(tt, M, r)]; c.y[h] = Yc[GI(tt, M, r)];
      c.ksv[h] = d.ks[GI(tt, M, r)]; c.ky[h] = d.ky[GI(tt, M, r)]; c.ys[h] = d.ys[GI(tt, M, r)];
    }
  };
  Chunk ck0, ck1;
  auto step = [&](const int t, StepIn &cs, StepIn &nxt) {
    if constexpr (kPing) {
      const int tn = t + NC * kDepthC < N ? t + NC * kDepthC : N - 1;   // unconditional (clamped) prefetch: this consumer's next step
      load_step(tn, nxt);
    } else if constexpr (!kEarly && !kChunk) load_step(t, cs);
    PIPELINE_FENCE();
    // take step t from the ring, then hand the slot back
    if (!CDDP_RING_LAZY_POLL || seen_prod < t + 1) seen_prod = __builtin_amdgcn_readfirstlane(wait_ge(&s_prod, t + 1));
    double rx[NX], dx[NX], u[NU];
    {
      const double *rs = s_ring + (size_t)(t % kRing) * RW * 64 + lane;
#pragma unroll
      for (int i = 0; i < NX; ++i) { rx[i] = rs[i * 64]; dx[i] = rs[(NX + i) * 64]; }
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = rs[(2 * NX + i) * 64];
      RING_FENCE();
      __hip_atomic_store(&s_consv[ci], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (alive && s_pstat[lane] <= t) { alive = false; fail_t = t; }
    double sn[M], yn[M];
    bool feas = true;
    // rows of K_s, K_y rebuilt from K and YS exactly as k_post forms them (ipddp_solver.cpp:1465-1472)
    if constexpr (kChunk) {
      auto rows = [&](const int i, const Chunk &c) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = i + h * NU;
          const double gv = UDiag<Cons>::val(cc, r);
          double Ksr[NX], Kyr[NX];
#pragma unroll
          for (int cix = 0; cix < NX; ++cix) {
            const double s2 = 0.0 + gv * c.K[cix];
            const double inner = 0.0 + s2;
            Kyr[cix] = dmin(dmax(c.ys[h] * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
            Ksr[cix] = (-0.0) - s2;
          }
          sn[r] = affine_2r<NX>(c.s[h], a_pr, c.ksv[h], Ksr, dx);
          yn[r] = affine_2r<NX>(c.y[h], a_du, c.ky[h], Kyr, dx);
          if (sn[r] < (1.0 - tau) * c.s[h] || yn[r] < (1.0 - tau) * c.y[h]) feas = false;
          if (!dfinite(sn[r]) || !dfinite(yn[r])) feas = false;
        }
      };
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (i + 1 < NU) { if ((i & 1) == 0) load_chunk(t, i + 1, ck1); else load_chunk(t, i + 1, ck0); }
        __builtin_amdgcn_sched_barrier(0);
        if ((i & 1) == 0) rows(i, ck0); else rows(i, ck1);
        __builtin_amdgcn_sched_barrier(0);
      }
      load_chunk(t + 1 < N ? t + 1 : t, 0, ck0);   // the next step's first chunk, behind the cost / barrier terms
      PIPELINE_FENCE();
    } else if constexpr (UDiag<Cons>::value) {
      // control box only: row r of G_u K is g_r K[col(r), :].  The dense sum adds products with exact zeros around
      // that term (K is finite: the sweep checks it), which leaves it unchanged except that a -0 becomes +0 --
      // hence the explicit 0.0 + ...; G_x = 0 enters as the same +0 / -0 the dense form adds.
#pragma unroll
      for (int r = 0; r < M; ++r) {
        const int ic = UDiag<Cons>::col(r);
        const double gv = UDiag<Cons>::val(cc, r);
        double Ksr[NX], Kyr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          const double s2 = 0.0 + gv * cs.KK[ic * NX + c];
          const double inner = 0.0 + s2;
          Kyr[c] = dmin(dmax(cs.ys[r] * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
          Ksr[c] = (-0.0) - s2;
        }
        sn[r] = affine_2r<NX>(cs.s[r], a_pr, cs.ksv[r], Ksr, dx);
        yn[r] = affine_2r<NX>(cs.y[r], a_du, cs.ky[r], Kyr, dx);
        if (sn[r] < (1.0 - tau) * cs.s[r] || yn[r] < (1.0 - tau) * cs.y[r]) feas = false;
        if (!dfinite(sn[r]) || !dfinite(yn[r])) feas = false;
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
      double Ksr[NX], Kyr[NX];
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) s2 += Gu[r * NU + i] * cs.KK[i * NX + c];
        const double inner = Gx[r * NX + c] + s2;
        Kyr[c] = dmin(dmax(cs.ys[r] * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
        Ksr[c] = (-Gx[r * NX + c]) - s2;
      }
      sn[r] = affine_2r<NX>(cs.s[r], a_pr, cs.ksv[r], Ksr, dx);
      yn[r] = affine_2r<NX>(cs.y[r], a_du, cs.ky[r], Kyr, dx);
      if (sn[r] < (1.0 - tau) * cs.s[r] || yn[r] < (1.0 - tau) * cs.y[r]) feas = false;
      if (!dfinite(sn[r]) || !dfinite(yn[r])) feas = false;
    }
    }
    fail_t = (alive && !feas) ? t : fail_t;
    if (!feas) alive = false;
    st<M>(Sn + GI(t, M, 0), kLS, sn);
    st<M>(Yn + GI(t, M, 0), kLS, yn);
    if constexpr (kEarly) {   // next step's record into the (now dead) register set, behind the cost / barrier terms
      load_step(t + 1 < N ? t + 1 : t, cs);
      PIPELINE_FENCE();
    }
    double g[M];
    Cons::template eval<NX, NU>(cc, rx, u, g);
    if constexpr (NC == 1) run_cost += Obj::running_cost(oc, xrt, t, rx, u);   // same t-ordered sum the fused rollout keeps (:1726-1748); NC = 2: the producer's
    st<M>(Gn + GI(t, M, 0), kLS, g);
    // Per-step terms of computeTheta / computeBarrierMerit / computePrimalAndComplementarity, parked exactly as
    // in k_forward_ipddp: the first constraint object's |g+s| terms accumulate in t order right here, the other
    // objects' terms and every log-barrier term are added after the rollout in the reference's order.
    double *ev = d.ev + GI((size_t)a * N + t, 2 * Cons::NSEG, 0);
#pragma unroll
    for (int c = 0; c < Cons::NSEG; ++c) {
      const int off = Cons::seg_off(c), dim = Cons::seg_dim(c);
      double n1 = 0.0, ninf = 0.0, ls = 0.0;
      for (int i = 0; i < dim; ++i) {
        const double r = g[off + i] + sn[off + i];
        n1 += l2norm ? r * r : fabs(r);
        ninf = dmax(ninf, fabs(r));
        const double ysp = yn[off + i] * sn[off + i];
        ev_icomp = dmax(ev_icomp, fabs(ysp - mu));
        ys_lo = dmin(ys_lo, ysp); ys_hi = dmax(ys_hi, ysp);   // for the residual under an updated mu (k_update)
        ls += solver_log(dmax(sn[off + i], kEpsSlack));
      }
      ev_max = dmax(ev_max, ninf);
      if (c == 0 && NC == 1) ev_total0 += n1; else ev[(size_t)(Cons::NSEG + c) * kLS] = n1;   // (NC = 2: the first object's terms are parked too)
      ev[(size_t)c * kLS] = ls;
    }
  };
  StepIn ra;
  if constexpr (NC > 1) {
    static_assert(NC == 1 || (kPing && kDepthC == 1), "two consumers: ping-pong records with one step of look-ahead only");
    // consumer ci walks t = ci, ci + NC, ...; two register sets, the record of its NEXT step in flight while the current one is reduced
    StepIn R0, R1;
    load_step(ci < N ? ci : N - 1, R0);
    prime();
    auto stop_at = [&](int tt) { return tt > __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    int t = ci, last = -1;
    bool stopped = false;
    for (; t + NC < N; t += 2 * NC) {
      if (stop_at(t)) { stopped = true; break; }
      step(t, R0, R1); last = t;
      if (stop_at(t + NC)) { stopped = true; break; }
      step(t + NC, R1, R0); last = t + NC;
      if (__builtin_amdgcn_ballot_w64(alive) == 0ull) {   // every trial of the tile has failed in this wave's steps: post the step, stop the producer
        __hip_atomic_fetch_min(&s_abort, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        stopped = true; break;
      }
    }
    if (!stopped && t < N && !stop_at(t)) step(t, R0, R1);
    const bool aborted = __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0x7fffffff;
    if (aborted) __hip_atomic_store(&s_consv[ci], kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // releases and stops the producer
    if (ci == 1) {   // hand the partial results to consumer 0 (its parked rows first: workgroup-scope release)
      s_pfail[lane] = fail_t; s_palive[lane] = alive ? 1 : 0;
      s_pmax[lane] = ev_max; s_pmax[64 + lane] = ev_icomp; s_pmax[128 + lane] = ys_lo; s_pmax[192 + lane] = ys_hi;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __hip_atomic_store(&s_done1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
    wait_ge(&s_done1, 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    fail_t = s_pfail[lane] < fail_t ? s_pfail[lane] : fail_t;
    alive = alive && (s_palive[lane] != 0);
    ev_max = dmax(ev_max, s_pmax[lane]); ev_icomp = dmax(ev_icomp, s_pmax[64 + lane]);
    ys_lo = dmin(ys_lo, s_pmax[128 + lane]); ys_hi = dmax(ys_hi, s_pmax[192 + lane]);
    if (__hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0x7fffffff) {   // (either wave may have posted it meanwhile)
      __hip_atomic_store(&s_consv[0], kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (active) d.t_steps[ti] = fail_t;
      return;
    }
  } else if constexpr (kPing) {
    StepIn R[kDepthC + 1];   // see the producer
#pragma unroll
    for (int j = 0; j < kDepthC; ++j) load_step(j < N ? j : N - 1, R[j]);
    prime();
    int t = 0;
    for (; t + kDepthC < N; t += kDepthC + 1) {
#pragma unroll
      for (int j = 0; j <= kDepthC; ++j) step(t + j, R[j], R[(j + kDepthC) % (kDepthC + 1)]);
      if (__builtin_amdgcn_ballot_w64(alive) == 0ull) {   // every trial of the tile has failed: release the producer
        __hip_atomic_store(s_cons, kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (active) d.t_steps[ti] = fail_t;
        K4_TIME_END(1, t);
        return;
      }
    }
#pragma unroll
    for (int j = 0; j <= kDepthC; ++j) if (t + j < N) step(t + j, R[j], R[(j + kDepthC) % (kDepthC + 1)]);
  } else {
    if constexpr (kChunk) load_chunk(0, 0, ck0); else load_step(0, ra);
    prime();
    for (int t = 0; t < N; ++t) {
      step(t, ra, ra);
      if (__builtin_amdgcn_ballot_w64(alive) == 0ull) {
        __hip_atomic_store(s_cons, kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (active) d.t_steps[ti] = fail_t;
        return;
      }
    }
  }
  wait_ge(&s_prod, N + kRing + 1);
  K4_TIME_END(1, N);
  if (active) d.t_steps[ti] = fail_t;
  if (alive && s_pstat[lane] <= N) alive = false;
  if (!alive) return;
  if constexpr (NC > 1) run_cost = s_prun[lane];
  const double cost_new = run_cost + s_pcost[lane];   // + l_f(x_N)
  const double *evb = d.ev + GI((size_t)a * N, 2 * Cons::NSEG, 0);
  const size_t tstride = (size_t)d.NB * (2 * Cons::NSEG) * kLS;
  if constexpr (NC > 1) {   // the first object's parked |g + s| terms, in t order (what the one-consumer kernel accumulates on the fly)
    const double *q = evb + (size_t)Cons::NSEG * kLS;
    int t = 0;
    for (; t + 3 < N; t += 4) {
      const double v0 = q[(size_t)t * tstride], v1 = q[(size_t)(t + 1) * tstride], v2 = q[(size_t)(t + 2) * tstride], v3 = q[(size_t)(t + 3) * tstride];
      ev_total0 += v0; ev_total0 += v1; ev_total0 += v2; ev_total0 += v3;
    }
    for (; t < N; ++t) ev_total0 += q[(size_t)t * tstride];
  }
  double total = ev_total0, mer = cost_new;
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
  if constexpr (TERM) {   // terminal equality rows, stacked in constraint order (term_reductions, dev_terminal.hpp)
    double n1 = 0.0, ninf = 0.0, dp = 0.0;
    for (int c = 0; c < P->n_term; ++c) {
      const TermDev &td = P->terms[c];
      if (td.kind != CDDP_HIP_TERM_EQUALITY) continue;
      for (int r = 0; r < td.dim; ++r) {
        const double h = s_xN[r * 64 + lane] - P->pool[td.off_target + r];
        n1 += l2norm ? h * h : fabs(h);
        ninf = dmax(ninf, fabs(h));
      }
    }
    total += n1; ev_max = dmax(ev_max, ninf);
    for (int c = 0; c < P->n_term; ++c) {
      const TermDev &td = P->terms[c];
      if (td.kind != CDDP_HIP_TERM_EQUALITY) continue;
      for (int r = 0; r < td.dim; ++r) {
        const int j = td.offset + r;
        const double h = s_xN[r * 64 + lane] - P->pool[td.off_target + r];
        const double lam = d.LamT[(size_t)j * d.Bp + b] + a_pr * d.dLamT[(size_t)j * d.Bp + b];
        if (!dfinite(lam)) return;
        d.LamTt[((size_t)a * kPTMax + j) * d.Bp + b] = lam;
        dp += lam * h;
      }
    }
    mer += dp;
  }
  const double th = l2norm ? sqrt(total) : total;
  const double theta_new = dmax(th, ev_max), phi_new = mer, ipr = ev_max, icomp = ev_icomp;
  if (!dfinite(phi_new) || !dfinite(theta_new) || !dfinite(ipr) || !dfinite(icomp)) return;
  bool accept = false;
  {   // filter acceptance, ipddp_solver.cpp:1793-1834
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
  d.t_ysmin[ti] = ys_lo; d.t_ysmax[ti] = ys_hi;
  d.t_success[ti] = accept ? 1 : 0;
}

#undef GI
}  // namespace cddp_dev
