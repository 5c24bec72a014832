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
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_condense(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M;
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
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
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  double x[NX], u[NU], y[M], s[M], g[M], Qyx[M * NX], Qyu[M * NU];
  ld<NX>(Xc + GI(t, NX, 0), kLS, x);
  ld<NU>(Uc + GI(t, NU, 0), kLS, u);
  ld<M>(Yc + GI(t, M, 0), kLS, y);
  ld<M>(Sc + GI(t, M, 0), kLS, s);
  ld<M>(Gc + GI(t, M, 0), kLS, g);
#pragma unroll
  for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
  Cons::template jac<NX, NU>(P, x, Qyx, Qyu);
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
  double *o = d.cst + GI(t, L::SIZE, 0);
  st<NX>(o + (size_t)L::CX * kLS, kLS, cx);
  st<NU>(o + (size_t)L::CU * kLS, kLS, cu);
  st<NU * NU>(o + (size_t)L::WQYU * kLS, kLS, WQyu);
  st<NU>(o + (size_t)L::QYUSIR * kLS, kLS, QyuSir);
  o[(size_t)L::IPR * kLS] = ipr;
  o[(size_t)L::ICOMP * kLS] = icomp;
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
    st<NU * NX>(o + (size_t)L::WQYX * kLS, kLS, WQyx);
    st<NX>(o + (size_t)L::QYXSIR * kLS, kLS, QyxSir);
    st<NX * NX>(o + (size_t)L::WXQYX * kLS, kLS, WxQyx);
  }
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
    struct StepIn { double A[NX * NX], Bm[NX * NU], c[CST]; };
    auto load_step = [&](int tt, StepIn &r) {
      ld<NX * NX>(d.A + GI(tt, NX * NX, 0), kLS, r.A);
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bm);
      ld<CST>(d.cst + GI(tt, CST, 0), kLS, r.c);
    };
    StepIn nxt;
    load_step(N - 1, nxt);
    for (int t = N - 1; t >= 0; --t) {
      StepIn cs = nxt;
      if (t > 0) load_step(t - 1, nxt);
      PIPELINE_FENCE();
      double (&A)[NX * NX] = cs.A; double (&Bm)[NX * NU] = cs.Bm;
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
      q_blocks<NX, NU>(P, A, Bm, Vx, Vxx, Qxx, Qux, Quu);
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
        if (!f.ok) { fail = true; break; }
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
    }
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
    conv = (inf_pr < tol && inf_du < tol && inf_comp < tol && asn < o.tolerance * 10.0);
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
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
  double x[NX], y[M], s[M], g[M], Qyx[M * NX], Qyu[M * NU], kk[NU], KK[NU * NX], dx[NX];
  ld<NX>(Xc + GI(t, NX, 0), kLS, x);
  ld<M>(Yc + GI(t, M, 0), kLS, y);
  ld<M>(Sc + GI(t, M, 0), kLS, s);
  ld<M>(Gc + GI(t, M, 0), kLS, g);
  ld<NU>(d.k + GI(t, NU, 0), kLS, kk);
  ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
  ld<NX>(d.dX + GI(t, NX, 0), kLS, dx);
#pragma unroll
  for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
  Cons::template jac<NX, NU>(P, x, Qyx, Qyu);
  double ky[M], ksv[M], Ky[M * NX], Ksm[M * NX];
  double apr = 1.0, adu = 1.0;
#pragma unroll
  for (int r = 0; r < M; ++r) {
    const double ss = dmax(s[r], s_floor);
    const double YSr = clip_pos(y[r], ss);
    const double rp = g[r] + s[r];
    const double rc = y[r] * s[r] - mu;
    const double rhat = y[r] * rp - rc;
    double temp = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) temp += Qyu[r * NU + i] * kk[i];
    ky[r] = clip_sgn(rhat + y[r] * temp, ss);
    ksv[r] = (-rp) - temp;
    double a = 0.0, c = 0.0;
#pragma unroll
    for (int cc = 0; cc < NX; ++cc) {
      double s2 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) s2 += Qyu[r * NU + i] * KK[i * NX + cc];
      const double inner = Qyx[r * NX + cc] + s2;
      Ky[r * NX + cc] = dmin(dmax(YSr * inner, -kMaxBarrierRatio), kMaxBarrierRatio);
      Ksm[r * NX + cc] = (-Qyx[r * NX + cc]) - s2;
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) { a += Ksm[r * NX + j] * dx[j]; c += Ky[r * NX + j] * dx[j]; }
    const double ds = ksv[r] + a;
    const double dy = dmin(dmax(ky[r] + c, -kMaxBarrierRatio), kMaxBarrierRatio);
    if (ds < 0.0) apr = dmin(apr, -tau * s[r] / ds);
    if (dy < 0.0) adu = dmin(adu, -tau * y[r] / dy);
  }
  st<M>(d.ky + GI(t, M, 0), kLS, ky);
  st<M>(d.ks + GI(t, M, 0), kLS, ksv);
  st<M * NX>(d.Ky + GI(t, M * NX, 0), kLS, Ky);
  st<M * NX>(d.Ks + GI(t, M * NX, 0), kLS, Ksm);
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
template <class Model>
__global__ __launch_bounds__(64) void k_costate(DevBuf d, int a0, int na, int phase_req, int force, int first_only) {
  constexpr int NX = Model::NX;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  if (!force && d.phase[b] != phase_req) return;
  const int cur = d.cur[b];
  double xo[NX], lo[NX], vx[NX], vxx[NX * NX];
  ld<NX>(d.X + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, xo);
  ld<NX>(d.Lam + (size_t)cur * d.planeX + GI(t, NX, 0), kLS, lo);
  ld<NX>(d.Vx + GI(t, NX, 0), kLS, vx);
  ld<NX * NX>(d.Vxx + GI(t, NX * NX, 0), kLS, vxx);
  for (int a = a0; a < a0 + na; ++a) {
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
      for (int j = 0; j < NX; ++j) s += vxx[i * NX + j] * (xn[j] - xo[j]);
      lam[i] = (lo[i] + a_pr * vx[i]) + s;
      finite = finite && dfinite(lam[i]);
    }
    if (!finite) { d.t_success[ti] = 0; continue; }
    st<NX>(d.Lam + (size_t)slot * d.planeX + GI(t, NX, 0), kLS, lam);
    if (first_only) break;
  }
}

#undef GI
}  // namespace cddp_dev
