// Lane-cooperative form of the terminal-equality reduced-LQR backward pass (IPDDPSolver::backwardPass,
// ipddp_solver.cpp:1120-1353; solveTerminalEqualityLQR :478-639; solveSequentialLQR :413-476).
//
// The one-lane-per-trajectory form (te_backward, kernels.hpp) keeps the nx x nx blocks of a 7-joint arm in scratch
// memory on 64 wavefronts: 665 ms per sweep at BASELINE config 5 (profiles/r01_f).  Here G lanes share a
// trajectory exactly as in kernels_coop.hpp (lane q owns column q of P_t, operands in LDS), and the p+1 gradient
// recursions of the reduced LQR -- the nominal one and one per unit terminal direction -- ride along ONE PER LANE
// (p + 1 <= G): lane v carries p_v(t) in registers, reads A_t, B_t, Q_ux, K, K^T Q_uu from LDS, and later rolls its
// own closed-loop variant forward.  Passes of one sweep:
//
//   k_te_condense      (batch x N)   the per-step LQ model terms that do not depend on P: q, r, R, residual maxima
//   k_backward_te_coop (G lanes per trajectory)
//        P1  backward matrix recursion + p+1 gradient recursions           -> K, V_xx, te_k, te_p
//        P2  linear rollout of the p+1 closed loops (lane v = variant v)   -> x_T variants (LDS)
//        P3  p x p regularised normal equations for the multiplier step (lane 0; operands in LDS)
//        P4  recombination k = k_0 + sum_v lambda_v (k_v - k_0), same for V_x  (elements spread over the lanes)
//        P5  linear-policy rollout dX with the final gains (row per lane)
//   k_te_post          (batch x N)   max |Q_u|, slack / dual gains k_s, k_y (+ Y S^-1), step caps; the last step
//                                     of a trajectory to finish applies the early-convergence test
//
// Every number is produced by the same expression, in the same order, as te_backward (the parity tests compare
// both against the oracle).  Layouts with state-dependent path rows (HAS_X) or terminal inequalities keep the
// one-lane kernel.
#pragma once
#include "kernels_coop.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

// NGRP groups of LDS operands, group g + 1 fetched (ldg) before group g is reduced (cmp), a scheduling barrier in between: left to
// itself the compiler issues one ds_read, waits for it and multiplies -- one LDS round trip per operand (profiles/r05_big2_roles.md)
#ifndef CDDP_TE_STAGED
#define CDDP_TE_STAGED 1
#endif
// reduced terminal system on the lanes of the group (TeCfg::kP3Par) / recombination rows fetched four trips ahead; 0: the one-lane
// system and one row per trip (the forms these are held bitwise to)
#ifndef CDDP_TE_P3PAR
#define CDDP_TE_P3PAR 1
#endif
#ifndef CDDP_TE_P4UNR
#define CDDP_TE_P4UNR 4
#endif
template <int NGRP, int BN, class LD, class CMP> DEV void lds_pipe(LD &&ldg, CMP &&cmp) {
  double b0[BN], b1[BN];
  ldg(0, b0);
#pragma unroll
  for (int g = 0; g < NGRP; ++g) {
    if (g + 1 < NGRP) { if ((g & 1) == 0) ldg(g + 1, b1); else ldg(g + 1, b0); }
    __builtin_amdgcn_sched_barrier(0);
    if ((g & 1) == 0) cmp(g, b0); else cmp(g, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class Model, class Cons>
struct TeCfg {
  static constexpr int NX = Model::NX, NU = Model::NU, G = CoopCfg<Model>::G, TPW = 64 / G;
  // one lane per gradient variant (pT + 1 <= G); a terminal equality selects states, so more than NX rows never occur
  static constexpr int PMAX0 = (G - 1 < kPTMax) ? G - 1 : kPTMax;
  static constexpr int PMAX = PMAX0 < NX ? PMAX0 : NX;
  static constexpr int VP = 16;                                     // variant stride of the te_k / te_p stacks
  static_assert(PMAX + 1 <= VP, "a row holds every variant");
  // per-step record written by k_te_condense
  static constexpr int cQ = 0, cR = NX, cRR = NX + NU, cIPR = cRR + NU * NU, cICOMP = cIPR + 1, REC = cICOMP + 1;
  static constexpr int NA = (NX * NX + G - 1) / G, NB = (NX * NU + G - 1) / G, NC = (REC + G - 1) / G, NK = (NU * NX + G - 1) / G,
                       NQ = (NU * NU + G - 1) / G;
  // LDS map of one trajectory.  The kernel is LDS-bound in OCCUPANCY: at 16 KB per trajectory (round 1: A, B and the record
  // double-buffered, every phase with its own area) a 4-trajectory workgroup took 64 KB, two workgroups = two wavefronts
  // per CU, and the 1024 wavefronts of the C5 batch ran in two rounds on half the SIMDs.  Now <= 40 KB per workgroup
  // (four per CU, one wavefront per SIMD, one round): A_t, B_t and the record are single-buffered (the prefetched slices
  // wait in registers and land after round 3, when nothing reads the old ones), the staging area of the gradient
  // variants overlays B^T P and Q_uu (dead once the gain column exists), and the later phases (variant rollouts P2,
  // reduced system P3, x_T block) overlay the sweep's areas; only h_T, the multipliers, dx and the reduction row persist.
  static constexpr bool kPvOverlay = NU > 1;                        // (NU = 1 solves against Q_uu itself in the variant pass)
  static constexpr int oA = 0, oB = oA + NX * NX, oM = oB + NX * NU, oT2 = oM + NX * NX, oQuu = oT2 + NU * NX;
  static constexpr int oPv = kPvOverlay ? oT2 : oQuu + NU * NU;
  static constexpr int T2_END = (oPv + NX * G > oQuu + NU * NU) ? oPv + NX * G : oQuu + NU * NU;
  static constexpr int oKK = T2_END, oQux = oKK + NU * NX, oKtQ = oQux + NU * NX, oC = oKtQ + NX * NU, oF = oC + REC,
                       SWEEP_END = oF + NU * NU + NU;
  // P2: closed-loop rollout of the variants -- its own double-buffered A, B, K and staging area, from 0
  static constexpr int rA = 0, rB = rA + 2 * NX * NX, rK = rB + 2 * NX * NU, rPv = rK + 2 * NU * NX, P2_END = rPv + NX * G;
  // P3: reduced-system work area from 0; the x_T block of the variants sits right behind it (written after P2's last read)
  static constexpr int P3 = 4 * PMAX * PMAX + 6 * PMAX;
  static constexpr int oXT = P3, XT_END = oXT + (PMAX + 1) * NX;
  static constexpr int M1 = SWEEP_END > P2_END ? SWEEP_END : P2_END;
  static constexpr int oPersist = M1 > XT_END ? M1 : XT_END;
  static constexpr int oH = oPersist, oLam = oH + PMAX, oBest = oLam + PMAX, oDx = oBest + PMAX, oRed = oDx + NX, RAW = oRed + G;
  static constexpr int STRIDE = (RAW + 31) / 32 * 32 + 4;
  static_assert(2 * (NU + NU * NX) <= oPersist, "gain buffers of the dX rollout fit below the persistent block");
};

DEV void atomic_max_pos(double *addr, double v) {   // v >= 0: the IEEE bit pattern orders like an unsigned integer
  if (!(v >= 0.0)) return;
  atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// ================================================================================ LQ model terms, (batch x N)
// q = l_x + G_x^T (y + S^-1 rhat), r = l_u + G_u^T (y + S^-1 rhat), R = sym(l_uu + G_u^T Y S^-1 G_u)
// (ipddp_solver.cpp:1143-1245); Q and the cross term do not depend on the step for layouts without G_x.
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_te_condense(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt, int force) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  typedef Objective<NX, NU> Obj;
  typedef TeCfg<Model, Cons> C;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  const double *Uc = d.U + (size_t)cur * d.planeU;
  const double mu = d.mu[b];
  const double s_floor = dmax(mu * 1e-3, kEpsSlack);
  double x[NX], u[NU], q[NX], r[NU], R[NU * NU];
  ld<NX>(Xc + GI(t, NX, 0), kLS, x);
  ld<NU>(Uc + GI(t, NU, 0), kLS, u);
  const double *Rd = P->pool + P->off_Rdt;
#pragma unroll
  for (int i = 0; i < NU; ++i)
#pragma unroll
    for (int c = 0; c < NU; ++c) R[i * NU + c] = 0.5 * ((2.0 * Rd[i * NU + c]) + (2.0 * Rd[c * NU + i]));
  Obj::lx(P, xrt, t, x, q);
  Obj::lu(P, u, r);
  double ipr = 0.0, icomp = 0.0;
  if constexpr (M > 0) {
    const double *Sc = d.S + (size_t)cur * d.planeM;
    const double *Yc = d.Y + (size_t)cur * d.planeM;
    const double *Gc = d.G + (size_t)cur * d.planeM;
    double y[MM], s[MM], g[MM], Qyx[MM * NX], Qyu[MM * NU], YS[MM], ypS[MM];
    ld<M>(Yc + GI(t, M, 0), kLS, y);
    ld<M>(Sc + GI(t, M, 0), kLS, s);
    ld<M>(Gc + GI(t, M, 0), kLS, g);
#pragma unroll
    for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
    Cons::template jac<NX, NU>(P, x, u, Qyx, Qyu);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const double ss = dmax(s[i], s_floor);
      YS[i] = clip_pos(y[i], ss);
      const double rp = g[i] + s[i], rc = y[i] * s[i] - mu;
      const double rhat = y[i] * rp - rc;
      ypS[i] = y[i] + clip_sgn(rhat, ss);
      ipr = dmax(ipr, fabs(rp)); icomp = dmax(icomp, fabs(rc));
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { double a = 0.0;
#pragma unroll
      for (int rr = 0; rr < M; ++rr) a += Qyx[rr * NX + i] * ypS[rr];
      q[i] += a; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
      for (int rr = 0; rr < M; ++rr) a += Qyu[rr * NU + i] * ypS[rr];
      r[i] += a; }
    double Rn[NU * NU];
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
      for (int c = 0; c < NU; ++c) { double a = 0.0;
#pragma unroll
        for (int rr = 0; rr < M; ++rr) a += (Qyu[rr * NU + i] * YS[rr]) * Qyu[rr * NU + c];
        Rn[i * NU + c] = R[i * NU + c] + a; }
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
      for (int c = 0; c < NU; ++c) R[i * NU + c] = 0.5 * (Rn[i * NU + c] + Rn[c * NU + i]);
  }
  double *rec = d.te_cst + GT(t, C::REC, 0);
  const size_t ts = TSTRIDE;
  st<NX>(rec + (size_t)C::cQ * ts, ts, q);
  st<NU>(rec + (size_t)C::cR * ts, ts, r);
  st<NU * NU>(rec + (size_t)C::cRR * ts, ts, R);
  rec[(size_t)C::cIPR * ts] = ipr;
  rec[(size_t)C::cICOMP * ts] = icomp;
}

// ---------------------------------------------------------------------------------------------
// Reduced-system helpers on memory operands (LDS): the steps of LDLTd<NMAX>::compute / solve and of
// singular_minmax (kernels.hpp) with an explicit leading dimension; run-time size n.
DEV bool ldlt_mem_compute(double *m, double *trd, double *temp, int n, int ld) {
  if (n <= 1) { if (n == 1) trd[0] = 0.0; return true; }
  bool found_zero_pivot = false, ret = true;
  for (int k = 0; k < n; ++k) {
    int big = k;
    double bigv = fabs(m[k * ld + k]);
    for (int i = k + 1; i < n; ++i) { const double v = fabs(m[i * ld + i]); if (v > bigv) { bigv = v; big = i; } }
    trd[k] = (double)big;
    if (k != big) {
      const int s = n - big - 1;
      for (int j = 0; j < k; ++j) { const double t = m[k * ld + j]; m[k * ld + j] = m[big * ld + j]; m[big * ld + j] = t; }
      for (int i = 0; i < s; ++i) { const double t = m[(big + 1 + i) * ld + k]; m[(big + 1 + i) * ld + k] = m[(big + 1 + i) * ld + big]; m[(big + 1 + i) * ld + big] = t; }
      { const double t = m[k * ld + k]; m[k * ld + k] = m[big * ld + big]; m[big * ld + big] = t; }
      for (int i = k + 1; i < big; ++i) { const double t = m[i * ld + k]; m[i * ld + k] = m[big * ld + i]; m[big * ld + i] = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = m[j * ld + j] * m[k * ld + j];
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += m[k * ld + j] * temp[j];
      m[k * ld + k] -= s;
      for (int i = 0; i < rs; ++i) {
        double t = 0.0;
        for (int j = 0; j < k; ++j) t += m[(k + 1 + i) * ld + j] * temp[j];
        m[(k + 1 + i) * ld + k] -= t;
      }
    }
    const double akk = m[k * ld + k];
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < n; ++j) {
        trd[j] = (double)j;
        for (int i = j + 1; i < n; ++i) ret = ret && (m[i * ld + j] == 0.0);
      }
      return ret;
    }
    if (rs > 0 && valid) { for (int i = 0; i < rs; ++i) m[(k + 1 + i) * ld + k] /= akk; }
    else if (rs > 0) { for (int i = 0; i < rs; ++i) ret = ret && (m[(k + 1 + i) * ld + k] == 0.0); }
    if (found_zero_pivot && valid) ret = false;
    else if (!valid) found_zero_pivot = true;
  }
  return ret;
}
DEV void ldlt_mem_solve(const double *m, const double *trd, int n, int ld, double *x) {
  for (int k = 0; k < n; ++k) { const int t = (int)trd[k]; if (t != k) { const double v = x[k]; x[k] = x[t]; x[t] = v; } }
  for (int i = 0; i < n; ++i) { double s = x[i]; for (int kk = 0; kk < i; ++kk) s -= m[i * ld + kk] * x[kk]; x[i] = s; }
  for (int i = 0; i < n; ++i) { const double dd = m[i * ld + i]; x[i] = (fabs(dd) > DBL_MIN) ? x[i] / dd : 0.0; }
  for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int kk = i + 1; kk < n; ++kk) s -= m[kk * ld + i] * x[kk]; x[i] = s; }
  for (int k = n - 1; k >= 0; --k) { const int t = (int)trd[k]; if (t != k) { const double v = x[k]; x[k] = x[t]; x[t] = v; } }
}
// ldlt_mem_compute on the G lanes of a trajectory (lane q; gmask = the group's lanes): the pivot search reads the diagonal in one
// batch and compares in the one-lane order; the elements of a transposition, the entries of temp and the rows below the pivot
// (their update sum and their division) are spread over the lanes; m_kk - s is formed by every lane (each needs the pivot).  Every
// entry goes through the operations of the one-lane loop in the same order: same bits.  The return value is the same on every lane.
template <int G, int PM>
DEV bool ldlt_mem_compute_coop(double *m, double *trd, double *temp, int n, int ld, int q, unsigned long long gmask) {
  if (n <= 1) { if (n == 1 && q == 0) trd[0] = 0.0; return true; }
  bool found_zero_pivot = false, ret = true, lret = true;
  for (int k = 0; k < n; ++k) {
    int big = k;
    {
      double dv[PM];
#pragma unroll
      for (int i = 0; i < PM; ++i) { const int ii = i < n ? i : 0; dv[i] = m[ii * ld + ii]; }
      double bigv = 0.0;
#pragma unroll
      for (int i = 0; i < PM; ++i) {
        const double v = fabs(dv[i]);
        if (i == k) bigv = v;
        else if (i > k && i < n && v > bigv) { bigv = v; big = i; }
      }
    }
    if (q == 0) trd[k] = (double)big;
    if (k != big) {
      const int s = n - big - 1, mid = big - k - 1, total = k + s + 1 + mid;
      for (int e = q; e < total; e += G) {
        int ia, ib;
        if (e < k) { ia = k * ld + e; ib = big * ld + e; }
        else if (e < k + s) { const int i = e - k; ia = (big + 1 + i) * ld + k; ib = (big + 1 + i) * ld + big; }
        else if (e == k + s) { ia = k * ld + k; ib = big * ld + big; }
        else { const int i = k + 1 + (e - k - s - 1); ia = i * ld + k; ib = big * ld + i; }
        const double t = m[ia]; m[ia] = m[ib]; m[ib] = t;
      }
      lds_sync();
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = q; j < k; j += G) temp[j] = m[j * ld + j] * m[k * ld + j];
      lds_sync();
    }
    double akk = m[k * ld + k];
    if (k > 0) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += m[k * ld + j] * temp[j];
      akk -= s;
    }
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < n; ++j) {
        if (q == 0) trd[j] = (double)j;
        for (int i = j + 1; i < n; ++i) ret = ret && (m[i * ld + j] == 0.0);
      }
      lds_sync();
      return ret;
    }
    for (int i = q; i < rs; i += G) {
      double v = m[(k + 1 + i) * ld + k];
      if (k > 0) {
        double t = 0.0;
        for (int j = 0; j < k; ++j) t += m[(k + 1 + i) * ld + j] * temp[j];
        v -= t;
      }
      if (valid) v /= akk; else lret = lret && (v == 0.0);
      m[(k + 1 + i) * ld + k] = v;
    }
    if (k > 0 && q == 0) m[k * ld + k] = akk;
    lds_sync();
    if (found_zero_pivot && valid) ret = false;
    else if (!valid) found_zero_pivot = true;
  }
  if (__ballot(!lret) & gmask) ret = false;
  return ret;
}
DEV void singular_minmax_mem(double *U, const double *A, int n, int ld, double &smax, double &smin) {   // one-sided Jacobi
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) U[i * ld + j] = A[i * ld + j];
  for (int sweep = 0; sweep < 80; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < n; ++i) { alpha += U[i * ld + p] * U[i * ld + p]; beta += U[i * ld + q] * U[i * ld + q]; gamma += U[i * ld + p] * U[i * ld + q]; }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < n; ++i) { const double up = U[i * ld + p], uq = U[i * ld + q]; U[i * ld + p] = cs * up - sn * uq; U[i * ld + q] = sn * up + cs * uq; }
      }
    if (!rotated) break;
  }
  smax = 0.0; smin = INFINITY;
  for (int j = 0; j < n; ++j) { double s2 = 0; for (int i = 0; i < n; ++i) s2 += U[i * ld + j] * U[i * ld + j]; const double sv = sqrt(s2); smax = dmax(smax, sv); smin = dmin(smin, sv); }
  if (n == 0) { smax = 0.0; smin = 0.0; }
}

// LDLTs<N>::solve with the factor read from memory: F = m[N * N] followed by the transpositions (as doubles)
template <int N>
DEV void ldlt_lds_solve(const double *F, double *x) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int tk = (int)F[N * N + k];
#pragma unroll
    for (int B = k + 1; B < N; ++B) { const bool sw = tk == B; const double a = x[k], b = x[B]; x[k] = sw ? b : a; x[B] = sw ? a : b; }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { double sacc = x[i];
#pragma unroll
    for (int kk = 0; kk < i; ++kk) sacc -= F[i * N + kk] * x[kk];
    x[i] = sacc; }
#pragma unroll
  for (int i = 0; i < N; ++i) { const double dd = F[i * N + i]; x[i] = (fabs(dd) > DBL_MIN) ? x[i] / dd : 0.0; }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) { double sacc = x[i];
#pragma unroll
    for (int kk = i + 1; kk < N; ++kk) sacc -= F[kk * N + i] * x[kk];
    x[i] = sacc; }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const int tk = (int)F[N * N + k];
#pragma unroll
    for (int B = k + 1; B < N; ++B) { const bool sw = tk == B; const double a = x[k], b = x[B]; x[k] = sw ? b : a; x[B] = sw ? a : b; }
  }
}

// ================================================================================ cooperative sweep
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_te_coop(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                         int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  typedef TeCfg<Model, Cons> C;
  constexpr int G = C::G, REC = C::REC, VP = C::VP;
  __shared__ double lds[C::TPW * C::STRIDE];
  __shared__ double ldsQ[NX * NX];   // sym(2 Q dt): the LQ model's Q for layouts without G_x (loop-invariant)
  __shared__ int ldsCol[kPTMax];     // state index selected by each stacked terminal-equality row
  const int lane = threadIdx.x;
  {   // every lane of the wavefront takes part, BEFORE the per-trajectory early exits
    const double *Qp = Pk->pool + Pk->off_Qdt;
    for (int e = lane; e < NX * NX; e += 64) { const int i = e / NX, c = e % NX; ldsQ[e] = 0.5 * ((2.0 * Qp[i * NX + c]) + (2.0 * Qp[c * NX + i])); }
    if (lane < kPTMax) ldsCol[lane] = (lane < Pk->pT) ? term_eq_col(Pk, lane) : 0;
    lds_sync();
  }
  const int q = lane % G, tl = lane / G;
  const int qc = q < NX ? q : NX - 1;
  const int b = coop_group<C::TPW>((int)blockIdx.x, d.xcd_map) * C::TPW + tl;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  double *Ls = lds + tl * C::STRIDE;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N, pT = P->pT;
  const bool hasv = q <= pT;           // this lane carries gradient variant v = q
  const int v = q;
  const unsigned long long gmask = ((G == 64) ? ~0ull : ((1ull << G) - 1ull)) << (tl * G);
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  if (count_iter && q == 0) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double inf_pr = 0, inf_comp = 0, step_norm = 0;
  double *tek = d.te_k, *tep = d.te_p;
  const size_t Bp = d.Bp;
  struct InAB { double a[C::NA], bm[C::NB], c[C::NC]; };
  auto loadAB = [&](int tt, InAB &r) {   // this lane's slices of A_t, B_t and of the step's LQ record
#pragma unroll
    for (int j = 0; j < C::NA; ++j) { const int e = q + G * j; r.a[j] = d.A[GT(tt, NX * NX, e < NX * NX ? e : NX * NX - 1)]; }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) { const int e = q + G * j; r.bm[j] = d.Bm[GT(tt, NX * NU, e < NX * NU ? e : NX * NU - 1)]; }
#pragma unroll
    for (int j = 0; j < C::NC; ++j) { const int e = q + G * j; r.c[j] = d.te_cst[GT(tt, REC, e < REC ? e : REC - 1)]; }
  };
  auto storeAB = [&](const InAB &r) {
    double *La = Ls + C::oA, *Lb = Ls + C::oB, *Lc = Ls + C::oC;
#pragma unroll
    for (int j = 0; j < C::NA; ++j) { const int e = q + G * j; if (e < NX * NX) La[e] = r.a[j]; }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) { const int e = q + G * j; if (e < NX * NU) Lb[e] = r.bm[j]; }
#pragma unroll
    for (int j = 0; j < C::NC; ++j) { const int e = q + G * j; if (e < REC) Lc[e] = r.c[j]; }
  };
#ifndef TE_EXP
#define TE_EXP 0   // 9: phase timers (profiles/r05_big2_roles.md); the product is 0
#endif
  unsigned long long tk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_last = 0;
  auto tick = [&](const int k) {
    if (TE_EXP == 9) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now = __builtin_readcyclecounter(); tk_acc[k] += now - tk_last; tk_last = now; __builtin_amdgcn_sched_barrier(0); }
  };
  if (TE_EXP == 9) tk_last = __builtin_readcyclecounter();
  for (;;) {
    ++nb;
    inf_pr = 0; inf_comp = 0; step_norm = 0;
    double Vc[NX], pv[NX];
    {   // ---- terminal data: h_T, previous multipliers, P_N, p_v(N)
      double xN[NX], VxN[NX];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      Obj::final_grad(P, xN, VxN);
#pragma unroll
      for (int i = 0; i < NX; ++i) Ls[C::oDx + i] = xN[i];
      lds_sync();
      if (q < pT) {
        double h = 0.0;
        for (int c = 0; c < P->n_term; ++c) {
          const TermDev &td = P->terms[c];
          if (td.kind != CDDP_HIP_TERM_EQUALITY) continue;
          if (q >= td.offset && q < td.offset + td.dim) { const int r = q - td.offset; h = Ls[C::oDx + r] - P->pool[td.off_target + r]; }
        }
        Ls[C::oH + q] = h;
        Ls[C::oLam + q] = d.LamT[(size_t)q * Bp + b];
      }
      lds_sync();
      for (int r = 0; r < pT; ++r) inf_pr = dmax(inf_pr, fabs(Ls[C::oH + r]));
      const double *Qf = P->pool + P->off_Qf;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        // V_xx(N) = sym(2 Qf), then P_N = sym(V_xx(N)) (te_backward prologue)
        const double vic = 0.5 * ((2.0 * Qf[i * NX + qc]) + (2.0 * Qf[qc * NX + i]));
        const double vci = 0.5 * ((2.0 * Qf[qc * NX + i]) + (2.0 * Qf[i * NX + qc]));
        Vc[i] = 0.5 * (vic + vci);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = VxN[i];
        double add = 0.0;                                   // (H_T^T lambda_prev)_i, rows ascending
        for (int r = 0; r < pT; ++r) add += ((ldsCol[r] == i) ? 1.0 : 0.0) * Ls[C::oLam + r];
        a += add;
        if (hasv && v > 0 && ldsCol[v - 1] == i) a += 1.0;
        pv[i] = a;
      }
      if (hasv) {
#pragma unroll
        for (int i = 0; i < NX; ++i) tep[(((size_t)N * Bp + b) * NX + i) * VP + v] = pv[i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) d.Vxx[GI(N, NX * NX, i * NX + qc)] = Vc[i];
    }
    bool fail = false;
    // ---- P1: matrix recursion + gradient variants
    auto step = [&](const int t, InAB &nab) -> bool {
      const int tp = t > 0 ? t - 1 : 0;
      const double *La = Ls + C::oA, *Lb = Ls + C::oB, *Lc = Ls + C::oC;
      bool bad = false;
      double Aq[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) Aq[j] = La[j * NX + qc];
      // round 1: column qc of T1 = A^T P and of BtP = B^T P
      {   // row groups software-pipelined: the operands of group g + 1 leave LDS before group g is reduced
        constexpr int GR = 2, NGRP = (NX + GR - 1) / GR, RWP = NX;
        double b0[GR * RWP], b1[GR * RWP];
        auto ldg = [&](const int g, double (&buf)[GR * RWP]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int i = g * GR + r; if (i < NX) {
#pragma unroll
            for (int k = 0; k < NX; ++k) buf[r * RWP + k] = La[k * NX + i];
          } }
        };
        auto cmp = [&](const int g, const double (&buf)[GR * RWP]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int i = g * GR + r; if (i < NX) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) s += buf[r * RWP + k] * Vc[k];
            Ls[C::oM + i * NX + qc] = s;
          } }
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
      if constexpr (CDDP_TE_STAGED) {
        lds_pipe<NU, NX>([&](const int u, auto &buf) {
#pragma unroll
          for (int k = 0; k < NX; ++k) buf[k] = Lb[k * NU + u];
        }, [&](const int u, const auto &buf) { double s = 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) s += buf[k] * Vc[k];
          Ls[C::oT2 + u * NX + qc] = s; });
      } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Lb[k * NU + u] * Vc[k];
        Ls[C::oT2 + u * NX + qc] = s; }
      }
      lds_sync();
      tick(8);
      // round 2a: Q + A^T P A (in place over T1, row by row), Q_ux column; the entries of Q_uu spread over the lanes
      {   // row groups software-pipelined: the operands of group g + 1 leave LDS before group g is reduced
        constexpr int GR = 2, NGRP = (NX + GR - 1) / GR, RWP = NX;
        double b0[GR * RWP], b1[GR * RWP];
        auto ldg = [&](const int g, double (&buf)[GR * RWP]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int i = g * GR + r; if (i < NX) {
#pragma unroll
            for (int j = 0; j < NX; ++j) buf[r * RWP + j] = Ls[C::oM + i * NX + j];
          } }
        };
        auto cmp = [&](const int g, const double (&buf)[GR * RWP]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int i = g * GR + r; if (i < NX) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) s += buf[r * RWP + j] * Aq[j];
            Ls[C::oM + i * NX + qc] = ldsQ[i * NX + qc] + s;
          } }
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
      double Quxq[NU];
      if constexpr (CDDP_TE_STAGED) {
        lds_pipe<NU, NX>([&](const int u, auto &buf) {
#pragma unroll
          for (int j = 0; j < NX; ++j) buf[j] = Ls[C::oT2 + u * NX + j];
        }, [&](const int u, const auto &buf) { double s = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) s += buf[j] * Aq[j];
          Quxq[u] = s + 0.0;            // + M^T, M = 0 without G_x
          Ls[C::oQux + u * NX + qc] = Quxq[u]; });
        lds_pipe<C::NQ, 2 * NX + 2>([&](const int j, auto &buf) {
          const int e = q + G * j;
          const int ee = e < NU * NU ? e : NU * NU - 1;
          const int u = ee / NU, w = ee - u * NU;
#pragma unroll
          for (int k = 0; k < NX; ++k) { buf[k] = Ls[C::oT2 + u * NX + k]; buf[NX + k] = Lb[k * NU + w]; }
          buf[2 * NX] = Lc[C::cRR + u * NU + w]; buf[2 * NX + 1] = Lc[C::cRR + w * NU + u];
        }, [&](const int j, const auto &buf) {
          const int e = q + G * j;
          const int ee = e < NU * NU ? e : NU * NU - 1;
          const int u = ee / NU, w = ee - u * NU;
          double s = 0.0;               // (B^T P) B; (B^T P^T) B is the same number: P is exactly symmetric
#pragma unroll
          for (int k = 0; k < NX; ++k) s += buf[k] * buf[NX + k];
          double ruw = buf[2 * NX], rwu = buf[2 * NX + 1];
          if (u == w) { ruw += reg; rwu += reg; }
          if (e < NU * NU) Ls[C::oQuu + e] = 0.5 * (((ruw + s) + rwu) + s);
        });
      } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Ls[C::oT2 + u * NX + j] * Aq[j];
        Quxq[u] = s + 0.0;            // + M^T, M = 0 without G_x
        Ls[C::oQux + u * NX + qc] = Quxq[u]; }
#pragma unroll
      for (int j = 0; j < C::NQ; ++j) {
        const int e = q + G * j;
        const int ee = e < NU * NU ? e : NU * NU - 1;
        const int u = ee / NU, w = ee - u * NU;
        double s = 0.0;               // (B^T P) B; (B^T P^T) B is the same number: P is exactly symmetric
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Ls[C::oT2 + u * NX + k] * Lb[k * NU + w];
        double ruw = Lc[C::cRR + u * NU + w], rwu = Lc[C::cRR + w * NU + u];
        if (u == w) { ruw += reg; rwu += reg; }
        if (e < NU * NU) Ls[C::oQuu + e] = 0.5 * (((ruw + s) + rwu) + s);
      }
      }
      lds_sync();
      tick(9);
      // round 2b: factor (replicated; lane 0 parks it in LDS for the variant solves), K column, row of K^T Q_uu
      double KKc[NU];
      if constexpr (NU == 1) {
        KKc[0] = -ldlt1_solve(Ls[C::oQuu], Quxq[0]);
      } else {
        LDLTs<NU> f;
        {
          double Quu[NU * NU];
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) Quu[i] = Ls[C::oQuu + i];
          f.compute(Quu, NU);
        }
        if (!f.ok) bad = true;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Quxq[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) KKc[i] = -col[i];
        if (q == 0) {
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) Ls[C::oF + i] = f.m[i];
#pragma unroll
          for (int i = 0; i < NU; ++i) Ls[C::oF + NU * NU + i] = (double)f.tr[i];
        }
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) bad = bad || !dfinite(KKc[i]);
      if constexpr (CDDP_TE_STAGED) {
        lds_pipe<NU, NU>([&](const int j, auto &buf) {
#pragma unroll
          for (int u = 0; u < NU; ++u) buf[u] = Ls[C::oQuu + u * NU + j];
        }, [&](const int j, const auto &buf) { double s = 0.0;     // row qc of K^T Q_uu
#pragma unroll
          for (int u = 0; u < NU; ++u) s += KKc[u] * buf[u];
          Ls[C::oKtQ + qc * NU + j] = s; });
      } else {
#pragma unroll
      for (int j = 0; j < NU; ++j) { double s = 0.0;     // row qc of K^T Q_uu
#pragma unroll
        for (int u = 0; u < NU; ++u) s += KKc[u] * Ls[C::oQuu + u * NU + j];
        Ls[C::oKtQ + qc * NU + j] = s; }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) Ls[C::oKK + u * NX + qc] = KKc[u];
      lds_sync();
      tick(10);
      // next step's A, B, record: fetched behind the factorisation (its registers are free again), landed in LDS at
      // the end of the step -- the gradient variant and round 3 cover the latency
      loadAB(tp, nab);
      PIPELINE_FENCE();
      inf_pr = dmax(inf_pr, Lc[C::cIPR]); inf_comp = dmax(inf_comp, Lc[C::cICOMP]);
      // gradient variant of this lane
      if (hasv) {
        double drift[NX], Qu[NU], kk[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) drift[i] = pv[i] + 0.0;     // + P * 0 (no affine dynamics term)
        if constexpr (CDDP_TE_STAGED) {
          lds_pipe<NU, NX + 1>([&](const int i, auto &buf) {
#pragma unroll
            for (int k = 0; k < NX; ++k) buf[k] = Lb[k * NU + i];
            buf[NX] = Lc[C::cR + i];
          }, [&](const int i, const auto &buf) { double a = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) a += buf[k] * drift[k];
            Qu[i] = buf[NX] + a; });
        } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) a += Lb[k * NU + i] * drift[k];
          Qu[i] = Lc[C::cR + i] + a; }
        }
        if constexpr (NU == 1) kk[0] = -ldlt1_solve(Ls[C::oQuu], Qu[0]);
        else {
          double col[NU];
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qu[i];
          ldlt_lds_solve<NU>(Ls + C::oF, col);
#pragma unroll
          for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        }
        if constexpr (CDDP_TE_STAGED) {
          lds_pipe<NX, NX + 3 * NU + 1>([&](const int i, auto &buf) {
#pragma unroll
            for (int k = 0; k < NX; ++k) buf[k] = La[k * NX + i];
#pragma unroll
            for (int j = 0; j < NU; ++j) { buf[NX + j] = Ls[C::oQux + j * NX + i]; buf[NX + NU + j] = Ls[C::oKK + j * NX + i]; buf[NX + 2 * NU + j] = Ls[C::oKtQ + i * NU + j]; }
            buf[NX + 3 * NU] = Lc[C::cQ + i];
          }, [&](const int i, const auto &buf) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) a += buf[k] * drift[k];
            const double Qx = buf[NX + 3 * NU] + a;
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) { a1 += buf[NX + j] * kk[j]; a2 += buf[NX + NU + j] * Qu[j]; a3 += buf[NX + 2 * NU + j] * kk[j]; }
            Ls[C::oPv + i * G + q] = ((Qx + a1) + a2) + a3;
          });
        } else {
#pragma unroll 4
        for (int i = 0; i < NX; ++i) {
          double a = 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) a += La[k * NX + i] * drift[k];
          const double Qx = Lc[C::cQ + i] + a;
          double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { a1 += Ls[C::oQux + j * NX + i] * kk[j]; a2 += Ls[C::oKK + j * NX + i] * Qu[j]; a3 += Ls[C::oKtQ + i * NU + j] * kk[j]; }
          Ls[C::oPv + i * G + q] = ((Qx + a1) + a2) + a3;
        }
        }
        lds_sync();
#pragma unroll
        for (int i = 0; i < NX; ++i) { pv[i] = Ls[C::oPv + i * G + q]; bad = bad || !dfinite(pv[i]); }
#pragma unroll
        for (int i = 0; i < NU; ++i) bad = bad || !dfinite(kk[i]);
#pragma unroll
        for (int i = 0; i < NU; ++i) tek[(((size_t)t * Bp + b) * NU + i) * VP + v] = kk[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) tep[(((size_t)t * Bp + b) * NX + i) * VP + v] = pv[i];
      }
      tick(11);
      // round 3: P_t column (in place over the lane's own Q + A^T P A column)
      {   // row groups software-pipelined: the operands of group g + 1 leave LDS before group g is reduced
        constexpr int GR = 2, NGRP = (NX + GR - 1) / GR, RWP = 3 * NU + 1;
        double b0[GR * RWP], b1[GR * RWP];
        auto ldg = [&](const int g, double (&buf)[GR * RWP]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int i = g * GR + r; if (i < NX) {
#pragma unroll
            for (int j = 0; j < NU; ++j) { buf[r * RWP + j] = Ls[C::oQux + j * NX + i]; buf[r * RWP + NU + j] = Ls[C::oKK + j * NX + i]; buf[r * RWP + 2 * NU + j] = Ls[C::oKtQ + i * NU + j]; }
            buf[r * RWP + 3 * NU] = Ls[C::oM + i * NX + qc];
          } }
        };
        auto cmp = [&](const int g, const double (&buf)[GR * RWP]) {
#pragma unroll
          for (int r = 0; r < GR; ++r) { const int i = g * GR + r; if (i < NX) {
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) { a1 += buf[r * RWP + j] * KKc[j]; a2 += buf[r * RWP + NU + j] * Quxq[j]; a3 += buf[r * RWP + 2 * NU + j] * KKc[j]; }
            Ls[C::oM + i * NX + qc] = ((buf[r * RWP + 3 * NU] + a1) + a2) + a3;
          } }
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
      tick(12);
      storeAB(nab);   // A_{t-1}, B_{t-1}, record: nothing reads the step's own copies any more
      lds_sync();
#pragma unroll
      for (int i = 0; i < NX; ++i) { Vc[i] = 0.5 * (Ls[C::oM + i * NX + qc] + Ls[C::oM + qc * NX + i]); bad = bad || !dfinite(Vc[i]); }
      lds_sync();
      if (__ballot(bad) & gmask) return false;
#pragma unroll
      for (int u = 0; u < NU; ++u) d.K[GI(t, NU * NX, u * NX + qc)] = KKc[u];
      if (d.t4) {   // the copy the variant rollouts (P2) and the dX rollout (P5) re-read
#pragma unroll
        for (int u = 0; u < NU; ++u) d.Kt[G4(t, NU * NX + NU, u * NX + qc)] = KKc[u];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) d.Vxx[GI(t, NX * NX, i * NX + qc)] = Vc[i];
      tick(13);
      return true;
    };
    {
      InAB nab;
      loadAB(N - 1, nab);
      storeAB(nab);
      lds_sync();
      tick(0);
      for (int t = N - 1; t >= 0; --t)
        if (!step(t, nab)) { fail = true; break; }
      tick(1);
    }
    if (!fail) {
      // ---- P2: closed-loop linear rollout of every variant (lane v), dx0 = 0 (rolloutLinearPolicy :368-392)
      struct RIn { double a[C::NA], bm[C::NB], ks[C::NK], kf[NU]; };
      auto load_r = [&](int tt, RIn &r) {
#pragma unroll
        for (int j = 0; j < C::NA; ++j) { const int e = q + G * j; r.a[j] = d.A[GT(tt, NX * NX, e < NX * NX ? e : NX * NX - 1)]; }
#pragma unroll
        for (int j = 0; j < C::NB; ++j) { const int e = q + G * j; r.bm[j] = d.Bm[GT(tt, NX * NU, e < NX * NU ? e : NX * NU - 1)]; }
#pragma unroll
        for (int j = 0; j < C::NK; ++j) { const int e = q + G * j; const int ee = e < NU * NX ? e : NU * NX - 1; r.ks[j] = d.t4 ? d.Kt[G4(tt, NU * NX + NU, ee)] : d.K[GI(tt, NU * NX, ee)]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) r.kf[i] = tek[(((size_t)tt * Bp + b) * NU + i) * VP + (hasv ? v : 0)];
      };
      auto store_r = [&](int buf, const RIn &r) {
        double *La = Ls + C::rA + buf * NX * NX, *Lb = Ls + C::rB + buf * NX * NU, *Lk = Ls + C::rK + buf * NU * NX;
#pragma unroll
        for (int j = 0; j < C::NA; ++j) { const int e = q + G * j; if (e < NX * NX) La[e] = r.a[j]; }
#pragma unroll
        for (int j = 0; j < C::NB; ++j) { const int e = q + G * j; if (e < NX * NU) Lb[e] = r.bm[j]; }
#pragma unroll
        for (int j = 0; j < C::NK; ++j) { const int e = q + G * j; if (e < NU * NX) Lk[e] = r.ks[j]; }
      };
      double dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = 0.0;
      RIn rc, rn;
      load_r(0, rc);
      store_r(0, rc);
      lds_sync();
      for (int t = 0; t < N; ++t) {
        const int tn = t + 1 < N ? t + 1 : t;
        load_r(tn, rn);
        PIPELINE_FENCE();
        const double *La = Ls + C::rA + (t & 1) * NX * NX, *Lb = Ls + C::rB + (t & 1) * NX * NU, *Lk = Ls + C::rK + (t & 1) * NU * NX;
        double du[NU];
        if constexpr (CDDP_TE_STAGED) {
          lds_pipe<NU, NX>([&](const int i, auto &buf) {
#pragma unroll
            for (int j = 0; j < NX; ++j) buf[j] = Lk[i * NX + j];
          }, [&](const int i, const auto &buf) { double a = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += buf[j] * dx[j];
            du[i] = rc.kf[i] + a; });
          lds_pipe<NX, NX + NU>([&](const int i, auto &buf) {
#pragma unroll
            for (int j = 0; j < NX; ++j) buf[j] = La[i * NX + j];
#pragma unroll
            for (int j = 0; j < NU; ++j) buf[NX + j] = Lb[i * NU + j];
          }, [&](const int i, const auto &buf) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += buf[j] * dx[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) c += buf[NX + j] * du[j];
            Ls[C::rPv + i * G + q] = (a + c) + 0.0;
          });
        } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) a += Lk[i * NX + j] * dx[j];
          du[i] = rc.kf[i] + a; }
#pragma unroll 4
        for (int i = 0; i < NX; ++i) {
          double a = 0.0, c = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) a += La[i * NX + j] * dx[j];
#pragma unroll
          for (int j = 0; j < NU; ++j) c += Lb[i * NU + j] * du[j];
          Ls[C::rPv + i * G + q] = (a + c) + 0.0;
        }
        }
        store_r((t & 1) ^ 1, rn);
        lds_sync();
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = Ls[C::rPv + i * G + q];
#pragma unroll
        for (int i = 0; i < NU; ++i) rc.kf[i] = rn.kf[i];
      }
      if (hasv) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Ls[C::oXT + v * NX + i] = dx[i];
      }
      lds_sync();
      tick(2);
      // ---- P3: reduced terminal system (:550-617); operands in LDS (overlaying the sweep area)
      if constexpr (CDDP_TE_P3PAR != 0) {
        // (round 5) the lanes of the group share the system: every entry of A_s, A_s^T A_s, A_s^T b, of the shifted matrix and of
        // the residual is its own sequential sum (entries / rows spread over the lanes), the factorisation runs with one lane per
        // row (ldlt_mem_compute_coop); the scales stay in sequence, every accept / skip decision is taken by all lanes on the
        // same LDS values.  Same work area as the one-lane form below, which it is held bitwise to.
        const int p = pT, ld_ = pT;
        double *As = Ls, *AtA = As + p * p, *Sh = AtA + p * p, *Uw = Sh + p * p, *rhs = Uw + p * p, *Atb = rhs + p,
               *lam = Atb + p, *best = lam + p, *temp = best + p, *trd = temp + p;
        const double *xT = Ls + C::oXT;
        for (int e = q; e < p * p; e += G) {
          const int r = e / p, i = e - r * p;
          const int cr = ldsCol[r];
          double a = 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) a += ((k == cr) ? 1.0 : 0.0) * (xT[(i + 1) * NX + k] - xT[k]);
          As[r * ld_ + i] = a;
        }
        for (int r = q; r < p; r += G) {
          const int cr = ldsCol[r];
          double hx = 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) hx += ((k == cr) ? 1.0 : 0.0) * xT[k];
          rhs[r] = (-Ls[C::oH + r]) - hx;
        }
        lds_sync();
        tick(6);
        for (int e = q; e < p * p; e += G) {
          const int i = e / p, c = e - i * p;
          double a = 0.0; for (int k = 0; k < p; ++k) a += As[k * ld_ + i] * As[k * ld_ + c];
          AtA[i * ld_ + c] = a;
        }
        for (int i = q; i < p; i += G) { double a = 0.0; for (int k = 0; k < p; ++k) a += As[k * ld_ + i] * rhs[k]; Atb[i] = a; best[i] = 0.0; }
        lds_sync();
        tick(7);
        double tr = 0.0;
        for (int i = 0; i < p; ++i) tr += AtA[i * ld_ + i];
        const double trace_term = (tr > 1.0 ? tr / (p > 1 ? p : 1) : 1.0);
        const double base_floor = dmax(1e-10, o.ipddp_jacobian_regularization_value * solver_pow(dmax(mu, 0.0), o.ipddp_jacobian_regularization_exponent));
        const double regv = dmax(base_floor, 1e-6 * trace_term);
        double svd_reg = 0.0;   // (only a non-finite system runs the Jacobi sweeps: see the one-lane form below)
        if (!dfinite(tr)) {
          if (q == 0) {
            double smax, smin;
            singular_minmax_mem(Uw, As, p, ld_, smax, smin);
            temp[0] = dmax(1e-8 * smax - smin, 0.0);
          }
          lds_sync();
          svd_reg = temp[0];
          lds_sync();
        }
        const double reg_base = dmax(regv, svd_reg);
        double rn2 = 0.0; for (int r = 0; r < p; ++r) rn2 += rhs[r] * rhs[r];
        const double cap = 100.0 * (1.0 + sqrt(rn2));
        double best_res = INFINITY; bool found = false;
        tick(14);
        for (int sc = 0; sc < 5; ++sc) {
          const double scale = (sc == 0) ? 1.0 : (sc == 1) ? 10.0 : (sc == 2) ? 100.0 : (sc == 3) ? 1e3 : 1e4;
          const double reg_i = dmax(reg_base * scale, 1e-12);
          for (int e = q; e < p * p; e += G) { const int i = e / p, c = e - i * p; Sh[i * ld_ + c] = AtA[i * ld_ + c] + reg_i * ((i == c) ? 1.0 : 0.0); }
          for (int i = q; i < p; i += G) lam[i] = Atb[i];
          lds_sync();
          if (!ldlt_mem_compute_coop<G, C::PMAX>(Sh, trd, temp, p, ld_, q, gmask)) continue;
          if (q == 0) ldlt_mem_solve(Sh, trd, p, ld_, lam);
          lds_sync();
          bool fin = true; double ln = 0.0;
          for (int i = 0; i < p; ++i) { const double li = lam[i]; fin = fin && dfinite(li); ln += li * li; }
          if (!fin) continue;
          ln = sqrt(ln);
          if (ln > cap) {
            const double f2 = cap / dmax(ln, 1e-12);
            for (int i = q; i < p; i += G) lam[i] = lam[i] * f2;
            lds_sync();
          }
          for (int r = q; r < p; r += G) { double a = 0.0; for (int i = 0; i < p; ++i) a += As[r * ld_ + i] * lam[i]; const double e = a - rhs[r]; Uw[r] = e * e; }
          lds_sync();
          double res = 0.0;
          for (int r = 0; r < p; ++r) res += Uw[r];
          res = sqrt(res);
          if (!dfinite(res)) continue;
          if (!found || res < best_res) { for (int i = q; i < p; i += G) best[i] = lam[i]; best_res = res; found = true; }
        }
        lds_sync();
        tick(15);
        for (int i = q; i < p; i += G) { const double bv = found ? best[i] : 0.0; d.dLamT[(size_t)i * Bp + b] = bv; Ls[C::oBest + i] = bv; }
      } else if (q == 0) {
        const int p = pT, ld_ = pT;
        double *As = Ls, *AtA = As + p * p, *Sh = AtA + p * p, *Uw = Sh + p * p, *rhs = Uw + p * p, *Atb = rhs + p,
               *lam = Atb + p, *best = lam + p, *temp = best + p, *trd = temp + p;
        const double *xT = Ls + C::oXT;
        for (int r = 0; r < p; ++r) {
          const int cr = ldsCol[r];
          for (int i = 0; i < p; ++i) {
            double a = 0.0;
            for (int k = 0; k < NX; ++k) a += ((k == cr) ? 1.0 : 0.0) * (xT[(i + 1) * NX + k] - xT[k]);
            As[r * ld_ + i] = a;
          }
          double hx = 0.0;
          for (int k = 0; k < NX; ++k) hx += ((k == cr) ? 1.0 : 0.0) * xT[k];
          rhs[r] = (-Ls[C::oH + r]) - hx;
        }
        double tr = 0.0;
        for (int i = 0; i < p; ++i) {
          for (int c = 0; c < p; ++c) { double a = 0.0; for (int k = 0; k < p; ++k) a += As[k * ld_ + i] * As[k * ld_ + c]; AtA[i * ld_ + c] = a; }
          double a = 0.0; for (int k = 0; k < p; ++k) a += As[k * ld_ + i] * rhs[k]; Atb[i] = a;
        }
        for (int i = 0; i < p; ++i) tr += AtA[i * ld_ + i];
        const double trace_term = (tr > 1.0 ? tr / (p > 1 ? p : 1) : 1.0);
        const double base_floor = dmax(1e-10, o.ipddp_jacobian_regularization_value * solver_pow(dmax(mu, 0.0), o.ipddp_jacobian_regularization_exponent));
        const double regv = dmax(base_floor, 1e-6 * trace_term);
        // te_backward adds svd_reg = max(1e-8 smax - smin, 0) of the singular values of A_s (80 Jacobi sweeps) and takes
        // reg_base = max(regv, svd_reg).  Whenever tr = ||A_s||_F^2 is finite that maximum is regv, whatever the
        // sweeps return: smax <= ||A_s||_F = sqrt(tr), so svd_reg <= 1e-8 sqrt(tr), while regv >= 1e-6 max(1, tr / p) --
        // for tr <= 1 that is 1e-8 against 1e-6, for tr > 1 the ratio is <= 1e-2 p / sqrt(tr) <= 0.16 (p <= 16).  The
        // sweeps therefore only run for a non-finite system, where they reproduce te_backward's NaN / inf handling.
        double svd_reg = 0.0;
        if (!dfinite(tr)) {
          double smax, smin;
          singular_minmax_mem(Uw, As, p, ld_, smax, smin);
          svd_reg = dmax(1e-8 * smax - smin, 0.0);
        }
        const double reg_base = dmax(regv, svd_reg);
        double rn2 = 0.0; for (int r = 0; r < p; ++r) rn2 += rhs[r] * rhs[r];
        const double cap = 100.0 * (1.0 + sqrt(rn2));
        for (int i = 0; i < p; ++i) best[i] = 0.0;
        double best_res = INFINITY; bool found = false;
        for (int sc = 0; sc < 5; ++sc) {
          const double scale = (sc == 0) ? 1.0 : (sc == 1) ? 10.0 : (sc == 2) ? 100.0 : (sc == 3) ? 1e3 : 1e4;
          const double reg_i = dmax(reg_base * scale, 1e-12);
          for (int i = 0; i < p; ++i) for (int c = 0; c < p; ++c) Sh[i * ld_ + c] = AtA[i * ld_ + c] + reg_i * ((i == c) ? 1.0 : 0.0);
          if (!ldlt_mem_compute(Sh, trd, temp, p, ld_)) continue;
          for (int i = 0; i < p; ++i) lam[i] = Atb[i];
          ldlt_mem_solve(Sh, trd, p, ld_, lam);
          bool fin = true; double ln = 0.0;
          for (int i = 0; i < p; ++i) { fin = fin && dfinite(lam[i]); ln += lam[i] * lam[i]; }
          if (!fin) continue;
          ln = sqrt(ln);
          if (ln > cap) { const double f2 = cap / dmax(ln, 1e-12); for (int i = 0; i < p; ++i) lam[i] = lam[i] * f2; }
          double res = 0.0;
          for (int r = 0; r < p; ++r) { double a = 0.0; for (int i = 0; i < p; ++i) a += As[r * ld_ + i] * lam[i]; const double e = a - rhs[r]; res += e * e; }
          res = sqrt(res);
          if (!dfinite(res)) continue;
          if (!found || res < best_res) { for (int i = 0; i < p; ++i) best[i] = lam[i]; best_res = res; found = true; }
        }
        if (!found) for (int i = 0; i < p; ++i) best[i] = 0.0;
        for (int i = 0; i < p; ++i) { d.dLamT[(size_t)i * Bp + b] = best[i]; Ls[C::oBest + i] = best[i]; }
      }
      lds_sync();
      tick(3);
      // ---- P4: recombination (:619-634); elements (t, i) spread over the lanes of the group
      double sn = 0.0;
      {
        double bw[C::PMAX];                 // multiplier step, 0 beyond pT (those terms are skipped below)
#pragma unroll
        for (int w = 0; w < C::PMAX; ++w) bw[w] = (w < pT) ? Ls[C::oBest + w] : 0.0;
        // one element (t, i) per lane and trip: its VP = 16 variant values are one 128-byte row, fetched whole
        auto combine = [&](const double *row) {
          double rv[VP];
#pragma unroll
          for (int w = 0; w < VP; ++w) rv[w] = row[w];
          const double k0 = rv[0];
          double ko = k0;
#pragma unroll
          for (int w = 0; w < C::PMAX; ++w) if (w < pT) ko += bw[w] * (rv[w + 1] - k0);
          return ko;
        };
        if constexpr (CDDP_TE_P4UNR > 1) {
          // the rows of UNR trips leave memory before the first is combined (one row per trip: a memory round trip per trip)
          constexpr int UNR = CDDP_TE_P4UNR;
          auto combine_r = [&](const double (&rv)[VP]) {
            const double k0 = rv[0];
            double ko = k0;
#pragma unroll
            for (int w = 0; w < C::PMAX; ++w) if (w < pT) ko += bw[w] * (rv[w + 1] - k0);
            return ko;
          };
          const int nk = N * NU, np = (N + 1) * NX;
          for (int idx0 = q; idx0 < nk; idx0 += G * UNR) {
            double rv[UNR][VP];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
              const int idx = idx0 + G * j < nk ? idx0 + G * j : idx0;
              const double *row = tek + ((size_t)(idx / NU) * Bp * NU + (size_t)b * NU + (idx % NU)) * VP;
#pragma unroll
              for (int w = 0; w < VP; ++w) rv[j][w] = row[w];
            }
            PIPELINE_FENCE();
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
              const int idx = idx0 + G * j;
              if (idx < nk) {
                const int t = idx / NU, i = idx - t * NU;
                const double ko = combine_r(rv[j]);
                d.k[GI(t, NU, i)] = ko;
                if (d.t4) d.Kt[G4(t, NU * NX + NU, NU * NX + i)] = ko;
                sn = dmax(sn, fabs(ko));
              }
            }
          }
          for (int idx0 = q; idx0 < np; idx0 += G * UNR) {
            double rv[UNR][VP];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
              const int idx = idx0 + G * j < np ? idx0 + G * j : idx0;
              const double *row = tep + ((size_t)(idx / NX) * Bp * NX + (size_t)b * NX + (idx % NX)) * VP;
#pragma unroll
              for (int w = 0; w < VP; ++w) rv[j][w] = row[w];
            }
            PIPELINE_FENCE();
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
              const int idx = idx0 + G * j;
              if (idx < np) { const int t = idx / NX, i = idx - t * NX; d.Vx[GI(t, NX, i)] = combine_r(rv[j]); }
            }
          }
        } else {
        for (int idx = q; idx < N * NU; idx += G) {
          const int t = idx / NU, i = idx - t * NU;
          const double ko = combine(tek + (((size_t)t * Bp + b) * NU + i) * VP);
          d.k[GI(t, NU, i)] = ko;
          if (d.t4) d.Kt[G4(t, NU * NX + NU, NU * NX + i)] = ko;
          sn = dmax(sn, fabs(ko));
        }
        for (int idx = q; idx < (N + 1) * NX; idx += G) {
          const int t = idx / NX, i = idx - t * NX;
          d.Vx[GI(t, NX, i)] = combine(tep + (((size_t)t * Bp + b) * NX + i) * VP);
        }
        }
      }
      Ls[C::oRed + q] = sn;
      lds_sync();
#pragma unroll
      for (int j = 0; j < G; ++j) step_norm = dmax(step_norm, Ls[C::oRed + j]);
      tick(4);
      // ---- P5: linear-policy rollout dX with the final gains (ipddp_solver.cpp:1511-1520); lane qc = row qc
      if constexpr (G == 16 && CDDP_TE_STAGED) {
        // (as the epilogue of k_backward_ipddp_coop_big2: dx and du by row broadcast, the rows of K, A, B two steps ahead)
        struct RIn5 { double Kr[NX], kq, Ar[NX], Br[NU]; };
        const size_t tstr = (size_t)d.NB * 64;
        const int ui = q < NU ? q : NU - 1;
        constexpr int EK = NU * NX + NU;
        const double *baseK = d.t4 ? d.Kt + G4(0, EK, ui * NX) : d.K + GI(0, NU * NX, ui * NX);
        const double *basek = d.t4 ? d.Kt + G4(0, EK, NU * NX + ui) : d.k + GI(0, NU, ui);
        const size_t sK = d.t4 ? tstr * EK : tstr * (NU * NX), sk = d.t4 ? tstr * EK : tstr * NU, se = d.t4 ? 4 : 64;
        const double *baseA = d.A + GT(0, NX * NX, qc * NX), *baseB = d.Bm + GT(0, NX * NU, qc * NU);
        const size_t sa = TSTRIDE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // P4's k (this wavefront's own stores) is re-read below
        auto load5 = [&](int tt, RIn5 &r) {
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
        auto step5 = [&](const int t, const RIn5 &rc5, RIn5 &rl5) {   // rc5: the rows of step t; rl5 takes the rows of step t + 2
          if (t >= N) return;
          load5(t + 2, rl5);
          PIPELINE_FENCE();
          pdX[(size_t)t * tstr * NX] = dxq;
          if (t < N - 1) {
            double dxb[NX], du[NU];
            static_for<NX>([&](auto J) { constexpr int jj = decltype(J)::value; dxb[jj] = row_bcast<jj>(dxq); });
            double au = 0.0;
#pragma unroll
            for (int jj = 0; jj < NX; ++jj) au += rc5.Kr[jj] * dxb[jj];
            const double du_own = rc5.kq + au;
            static_for<NU>([&](auto U) { constexpr int u = decltype(U)::value; du[u] = row_bcast<u>(du_own); });
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int jj = 0; jj < NX; ++jj) a += rc5.Ar[jj] * dxb[jj];
#pragma unroll
            for (int jj = 0; jj < NU; ++jj) c += rc5.Br[jj] * du[jj];
            dxq = (a + c) + 0.0;
          }
        };
        RIn5 r0, r1, r2;
        load5(0, r0); load5(1, r1);
        for (int t = 0; t < N; t += 3) { step5(t, r0, r2); step5(t + 1, r1, r0); step5(t + 2, r2, r1); }
      } else {
        double dxr[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dxr[i] = 0.0;
        struct GIn { double ks[C::NK], kq, Aq[NX], Bq[NU]; };
        constexpr int GS = NU + NU * NX;
        static_assert(2 * GS <= 2 * NX * NX + 2 * NX * NU, "gain buffers fit the A/B area");
        auto load_g = [&](int tt, GIn &r) {
#pragma unroll
          for (int j = 0; j < C::NK; ++j) { const int e = q + G * j; const int ee = e < NU * NX ? e : NU * NX - 1; r.ks[j] = d.t4 ? d.Kt[G4(tt, NU * NX + NU, ee)] : d.K[GI(tt, NU * NX, ee)]; }
          r.kq = d.t4 ? d.Kt[G4(tt, NU * NX + NU, NU * NX + (q < NU ? q : NU - 1))] : d.k[GI(tt, NU, q < NU ? q : NU - 1)];
#pragma unroll
          for (int j = 0; j < NX; ++j) r.Aq[j] = d.A[GT(tt, NX * NX, qc * NX + j)];
#pragma unroll
          for (int j = 0; j < NU; ++j) r.Bq[j] = d.Bm[GT(tt, NX * NU, qc * NU + j)];
        };
        auto store_g = [&](int buf, const GIn &r) {
          double *Lg = Ls + C::oA + buf * GS;
          if (q < NU) Lg[q] = r.kq;
#pragma unroll
          for (int j = 0; j < C::NK; ++j) { const int e = q + G * j; if (e < NU * NX) Lg[NU + e] = r.ks[j]; }
        };
        GIn gc, gn;
#pragma unroll
        for (int i = 0; i < NX; ++i) Ls[C::oDx + i] = 0.0;
        load_g(0, gc);
        store_g(0, gc);
        lds_sync();
        for (int t = 0; t < N; ++t) {
          const int tn = t + 1 < N - 1 ? t + 1 : t;
          load_g(tn, gn);
          PIPELINE_FENCE();
          d.dX[GI(t, NX, qc)] = Ls[C::oDx + qc];
          if (t < N - 1) {
            const double *Lg = Ls + C::oA + (t & 1) * GS;
            double du[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
              for (int j = 0; j < NX; ++j) a += Lg[NU + i * NX + j] * dxr[j];
              du[i] = Lg[i] + a; }
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a += gc.Aq[j] * dxr[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) c += gc.Bq[j] * du[j];
            const double dxq = (a + c) + 0.0;
            lds_sync();
            Ls[C::oDx + qc] = dxq;
            store_g((t & 1) ^ 1, gn);
            lds_sync();
#pragma unroll
            for (int i = 0; i < NX; ++i) dxr[i] = Ls[C::oDx + i];
          }
          gc = gn;
        }
      }
      tick(5);
      if (TE_EXP == 9 && q == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) d.k[GI(j, NU, 0)] = (double)tk_acc[j];
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
  d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;
  d.te_cnt[b] = ok ? 0 : -1;       // k_te_post: steps of this trajectory finished so far / "no post-processing"
  if (ok) {
    d.dV0[b] = 0.0; d.dV1[b] = 0.0; d.inf_du[b] = 0.0; d.step_norm[b] = step_norm;
    d.inf_pr[b] = inf_pr; d.inf_comp[b] = inf_comp;
  }
  if (force) return;
  if (!ok) { d.status[b] = CDDP_HIP_STATUS_REG_LIMIT; d.phase[b] = PH_DONE; }
}

// ================================================================================ post-processing, (batch x N)
// max_t |r_t + B_t^T V_x(t+1)| (:1260-1266), slack / dual gains with the final k, K (:1270-1312), the directions
// dS = k_s + K_s dX, dY = clamp(k_y + K_y dX) and computeMaxStepSizes (:1522-1532, 2939-2988).  The lane that
// completes a trajectory's N-th step applies checkEarlyConvergence (:925-958).
template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_te_post(DevBuf d, const ProblemDev *__restrict__ Pk, int force, int tstep) {
  constexpr int NX = Model::NX, NU = Model::NU, M = Cons::M, MM = (M > 0 ? M : 1);
  typedef TeCfg<Model, Cons> C;
  const int b = blockIdx.x * 64 + threadIdx.x;
  // a lane walks tstep consecutive steps and publishes ONE max / min / count per quantity: the device-scope atomics of 150 blocks on
  // the same 64 words, not the 1.6 GB of rows, set the pace of the one-step-per-block form at C5's size
  const int t0 = blockIdx.y * tstep, t1 = (t0 + tstep < d.N) ? t0 + tstep : d.N;
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  if (d.te_cnt[b] < 0) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double mu = d.mu[b];
  // accumulated as the bit patterns the atomics compare (atomic_max_pos / atomic_min_pos: unsigned order of non-negative doubles), so
  // that a lane's several steps combine exactly as their separate atomics did
  unsigned long long idu_all = 0ull, apr_all = (unsigned long long)__double_as_longlong(1.0), adu_all = apr_all;
#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
  {
    // B^T p with B streamed a row at a time (row k enters every a_i as its k-th term: the sums run in the same order as column by
    // column, with nu accumulators instead of the nx nu block in registers)
    double r[NU], pn[NX], acc[NU];
    ld<NU>(d.te_cst + GT(t, C::REC, C::cR), TSTRIDE, r);
    ld<NX>(d.Vx + GI(t + 1, NX, 0), kLS, pn);
#pragma unroll
    for (int i = 0; i < NU; ++i) acc[i] = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      double Bk[NU];
      ld<NU>(d.Bm + GT(t, NX * NU, k * NU), TSTRIDE, Bk);
#pragma unroll
      for (int i = 0; i < NU; ++i) acc[i] += Bk[i] * pn[k];
    }
    double idu = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) idu = dmax(idu, fabs(r[i] + acc[i]));
    if (idu >= 0.0) { const unsigned long long u = (unsigned long long)__double_as_longlong(idu); idu_all = u > idu_all ? u : idu_all; }
  }
  if constexpr (M > 0) {
    const double *Xc = d.X + (size_t)cur * d.planeX;
    const double *Sc = d.S + (size_t)cur * d.planeM;
    const double *Yc = d.Y + (size_t)cur * d.planeM;
    const double *Gc = d.G + (size_t)cur * d.planeM;
    const double s_floor = dmax(mu * 1e-3, kEpsSlack);
    const double tau = dmax(o.barrier_min_fraction_to_boundary, 1.0 - mu);
    // control box only (UDiag): row rr of G_u K is one entry of G_u times ONE row of K, so K is streamed a row at a time (the second
    // box half re-reads the rows out of L2) instead of held whole -- 98 doubles at (14, 7), which with the rest made this a
    // one-wave-per-SIMD kernel
    constexpr bool kStreamK = UDiag<Cons>::value;
    double x[NX], y[MM], s[MM], g[MM], Qyx[MM * NX], Qyu[MM * NU], kk[NU], KK[kStreamK ? NX : NU * NX], dx[NX];
    ld<NX>(Xc + GI(t, NX, 0), kLS, x);
    ld<M>(Yc + GI(t, M, 0), kLS, y); ld<M>(Sc + GI(t, M, 0), kLS, s); ld<M>(Gc + GI(t, M, 0), kLS, g);
    ld<NU>(d.k + GI(t, NU, 0), kLS, kk);
    if constexpr (!kStreamK) ld<NU * NX>(d.K + GI(t, NU * NX, 0), kLS, KK);
    ld<NX>(d.dX + GI(t, NX, 0), kLS, dx);
#pragma unroll
    for (int i = 0; i < M * NX; ++i) Qyx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < M * NU; ++i) Qyu[i] = 0.0;
    typename Cons::Ctx cctx;
    if constexpr (UDiag<Cons>::value) Cons::load(P, cctx);     // control box only: one entry per row of G_u (see k_forward_ipddp_pc)
    else {
      double uj[NU];
      if constexpr (Cons::NEEDS_U) ld<NU>(d.U + (size_t)cur * d.planeU + GI(t, NU, 0), kLS, uj);
      Cons::template jac<NX, NU>(P, x, uj, Qyx, Qyu);
    }
    double apr = 1.0, adu = 1.0;
#pragma unroll
    for (int rr = 0; rr < M; ++rr) {
      const double ss = dmax(s[rr], s_floor);
      const double YSr = clip_pos(y[rr], ss);
      const double rp = g[rr] + s[rr], rc = y[rr] * s[rr] - mu;
      const double rhat = y[rr] * rp - rc;
      double temp = 0.0;
      if constexpr (UDiag<Cons>::value) temp = 0.0 + UDiag<Cons>::val(cctx, rr) * kk[UDiag<Cons>::col(rr)];
      else {
#pragma unroll
        for (int i = 0; i < NU; ++i) temp += Qyu[rr * NU + i] * kk[i];
      }
      const double kyr = clip_sgn(rhat + y[rr] * temp, ss);
      const double ksr = (-rp) - temp;
      double a = 0.0, c = 0.0;
      if constexpr (kStreamK) ld<NX>(d.K + GI(t, NU * NX, UDiag<Cons>::col(rr) * NX), kLS, KK);
#pragma unroll
      for (int cc = 0; cc < NX; ++cc) {
        double s2 = 0.0, gx = 0.0;
        if constexpr (UDiag<Cons>::value) s2 = 0.0 + UDiag<Cons>::val(cctx, rr) * KK[cc];
        else {
          gx = Qyx[rr * NX + cc];
#pragma unroll
          for (int i = 0; i < NU; ++i) s2 += Qyu[rr * NU + i] * KK[i * NX + cc];
        }
        const double Kyv = dmin(dmax(YSr * (gx + s2), -kMaxBarrierRatio), kMaxBarrierRatio);
        const double Ksv = (-gx) - s2;
        a += Ksv * dx[cc]; c += Kyv * dx[cc];
      }
      d.ky[GI(t, M, rr)] = kyr;
      d.ks[GI(t, M, rr)] = ksr;
      d.ys[GI(t, M, rr)] = YSr;     // the rollout consumer rebuilds the rows of K_s, K_y from K and Y S^-1 (see k_post)
      const double ds = ksr + a;
      const double dy = dmin(dmax(kyr + c, -kMaxBarrierRatio), kMaxBarrierRatio);
      if (ds < 0.0) apr = dmin(apr, -tau * s[rr] / ds);
      if (dy < 0.0) adu = dmin(adu, -tau * y[rr] / dy);
    }
    if (apr < 1.0) { const unsigned long long u = (unsigned long long)__double_as_longlong(apr >= 0.0 ? apr : 0.0); apr_all = u < apr_all ? u : apr_all; }
    if (adu < 1.0) { const unsigned long long u = (unsigned long long)__double_as_longlong(adu >= 0.0 ? adu : 0.0); adu_all = u < adu_all ? u : adu_all; }
  }
  }
  const unsigned long long one_bits = (unsigned long long)__double_as_longlong(1.0);
  if (idu_all != 0ull) atomicMax((unsigned long long *)(d.inf_du + b), idu_all);
  if (apr_all != one_bits) atomicMin((unsigned long long *)(d.apr_max + b), apr_all);
  if (adu_all != one_bits) atomicMin((unsigned long long *)(d.adu_max + b), adu_all);
  __threadfence();
  const int done = atomicAdd(d.te_cnt + b, t1 - t0);
  if (done + (t1 - t0) != N) return;
  // ---- last step of this trajectory: every contribution is in; early-convergence test
  __threadfence();
  const double inf_du = __longlong_as_double((long long)atomicMax((unsigned long long *)(d.inf_du + b), 0ull));
  d.inf_du[b] = inf_du;
  if (force) return;
  const double inf_pr = d.inf_pr[b], inf_comp = d.inf_comp[b], step_norm = d.step_norm[b];
  bool conv;
  // computeScaledDualInfeasibility (ipddp_solver.cpp:931, 2725-2776), as every other sweep's early-convergence test uses it
  // (this kernel is only instantiated for layouts without state-dependent path rows, where it returns inf_du unchanged)
  const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, d.cur[b], inf_du);
  if (M == 0) conv = (inf_pr < o.tolerance && sdu_early < o.tolerance);   // no barrier terms (terminal equality only)
  else {
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
  }
  if (conv) { d.status[b] = CDDP_HIP_STATUS_OPTIMAL; d.phase[b] = PH_DONE; hist_push(d, b, mu); return; }
  d.phase[b] = PH_FWD1;
}

#undef GI
}  // namespace cddp_dev
