// Stack-fed mode (host plugins): arbitrary DynamicalSystem / Objective / Constraint subclasses cannot run on the GPU, so
// the caller evaluates them on the host and hands over the (N x batch) derivative stacks; the GPU runs the backward
// pass on them -- the north_star's "coalesced HBM loads of the (N x batch) stacks of f_x/f_u/l_xx/l_uu/l_ux".
// Bound to a handle (cddp_hip_stacks_*): device buffers persist between calls, stacks are uploaded once per iterate,
// the sweep is ONE launch.  Three branches of the reference:
//   CDDP_HIP_STACKS_CLDDP      clddp_solver.cpp:79-204 without bounds (PD test, dense inverse, reg only in the factor)
//   CDDP_HIP_STACKS_IPDDP      ipddp_solver.cpp:1048-1118 (unconstrained: LDLT, reg kept in the value update)
//   CDDP_HIP_STACKS_LOGDDP     logddp_solver.cpp:470-575 (the host folds the relaxed-log-barrier gradients / Hessians of
//                              barrier.hpp:95-262 into the cost stacks: reg added THEN symmetrised, LDLT, the value update with
//                              the un-regularised Q_uu in CLDDP's association order, raw max |Q_u|)
//   CDDP_HIP_STACKS_MSIPDDP    msipddp_solver.cpp:1112-1208 (no constraints): the IPDDP recursion with the multiple-shooting defect
//                              d_t = f(x_t, u_t) - x_{t+1} entering as V_x + V_xx d_t in Q_x, Q_u (defect stack required)
//   CDDP_HIP_STACKS_IPDDP_PATH ipddp_solver.cpp:1355-1568 (path constraints condensed: y, s, g, G_x, G_u stacks; gains of
//                              the slack / dual directions, linear-policy rollout, dS, dY, computeMaxStepSizes :2939-2988)
// One trajectory per lane, batch-minor stacks [t][e][Bp]: every wavefront load is one coalesced 512-B row.
//   bytes read / trajectory  = 8 * (N*(nx^2 + nx*nu + nx + nu + nx^2 + nu^2 + nu*nx [+ 3m + m*nx + m*nu]) + nx + nx^2)
//   bytes written            = 8 * (N*(nu*nx + nu + nx + nx^2 [+ 2m + 2m*nx]) + nx + nx^2 + 2)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dev_linalg.hpp"
#include "dev_boxqp.hpp"
#include "../../include/cddp_hip.h"

using namespace cddp_dev;

extern "C" int cddp_hip_internal_set_error(int code, const char *msg);   // capi.hip (thread-local last-error string)

namespace {

constexpr double kEpsSlackS = 1e-10;          // EPS_SLACK            ipddp_solver.cpp:36
constexpr double kMaxRatioS = 1e6;            // MAX_BARRIER_RATIO    ipddp_solver.cpp:38

struct StackArgs {
  int B, Bp, N, branch;
  int t4;                                      // 1: the (N x batch) stacks are tile-minor, [t][b / 4][e][b % 4] -- the step record of the FOUR trajectories a
                                               // cooperative workgroup owns is contiguous (E x 32 B: whole 128-B lines per load instead of sixteen 32-B pieces
                                               // of lines that four workgroups share); the handles whose default sweep is the cooperative one (nx > 8).
                                               // 0: [t][e][Bp], one coalesced 512-B row per element for the one-lane kernels.  [1][E][Bp] inputs stay plain.
  double reg_factor, reg_max, tau_min;
  const double *reg_in, *mu;                   // [Bp]
  const double *fx, *fu, *lx, *lu, *lxx, *luu, *lux, *VxN, *VxxN;
  const double *y, *s, *g, *Gx, *Gu;           // path-constraint stacks (branch IPDDP_PATH)
  const double *Fxx, *Fuu, *Fux;               // dt-scaled dynamics Hessian tensors (full DDP, use_ilqr = false); NULL = Gauss-Newton
  const double *dfc;                           // MSIPDDP defects d_t [N][nx]
  double *QuuF;                                // MSIPDDP factor cache [N][nu*nu] (cddp_hip_stacks_factor_cache), or NULL
  int *fvalid;                                 // [N][Bp] 1 = step t has a cached matrix
  const double *U, *lo, *up;                   // CLDDP control box (clddp_solver.cpp:147-178): current controls [N][nu], bounds [nu]; lo = NULL: none
  cddp_hip_options opt;                        // BoxQP parameters
  double *K, *k, *Vx, *Vxx, *dV;
  double *ky, *Ky, *ks, *Ks, *dX;              // IPDDP_PATH outputs
  double *scal;                                // [6][Bp]: reg used, inf_du, inf_pr, inf_comp, step_norm, (unused)
  double *caps;                                // [2][Bp]: alpha_pr_max, alpha_du_max
  int *ok;
};

// (T4: a compile-time parameter of the one-lane kernel -- the tile-minor instantiations exist for the nx > 8 shapes only, as the bitwise
//  cross-check of the cooperative sweep on its own layout; written as a run-time select in every index, the nx = 6 / m = 6 instantiation
//  came out of ROCm 7.2 reading mu wrong)
#define SI(t, E, e) (T4 ? ((((size_t)(t)) * (size_t)(a.Bp >> 2) + (size_t)(b >> 2)) * (E) + (e)) * 4 + (size_t)(b & 3) \
                        : ((((size_t)(t)) * (E) + (e)) * (size_t)a.Bp + (size_t)b))

DEV double clipp(double num, double den) { return dclamp(num / den, 0.0, kMaxRatioS); }
DEV double clips(double num, double den) { return dclamp(num / den, -kMaxRatioS, kMaxRatioS); }

// One backward sweep at regularisation `reg`; returns false where the reference's backwardPass returns false.
template <int NX, int NU, int M, bool T4>
DEV bool sweep(const StackArgs &a, int b, double reg, double mu, double &dV0, double &dV1, double &inf_du, double &inf_pr,
               double &inf_comp, double &step_norm) {
  constexpr int MM = M > 0 ? M : 1;
  const int N = a.N;
  const bool lg = a.branch == CDDP_HIP_STACKS_LOGDDP;
  const bool msp = a.branch == CDDP_HIP_STACKS_MSIPDDP_PATH;    // MSIPDDP's condensation (plain ratios) + defects
  const bool ms = a.branch == CDDP_HIP_STACKS_MSIPDDP || msp;   // defects enter Q_x, Q_u
  const bool ip = a.branch != CDDP_HIP_STACKS_CLDDP && !lg;
  double Vx[NX], Vxx[NX * NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) Vx[i] = a.VxN[(size_t)i * a.Bp + b];
#pragma unroll
  for (int i = 0; i < NX * NX; ++i) Vxx[i] = a.VxxN[(size_t)i * a.Bp + b];
  if (ip || lg) {   // V_xx = symmetrize(V_xx)  (ipddp_solver.cpp:992, logddp_solver.cpp:475)
    double T[NX * NX];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) T[i] = Vxx[i];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (T[i * NX + c] + T[c * NX + i]);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) a.Vx[SI(N, NX, i)] = Vx[i];
#pragma unroll
  for (int i = 0; i < NX * NX; ++i) a.Vxx[SI(N, NX * NX, i)] = Vxx[i];
  dV0 = dV1 = 0.0; inf_du = inf_pr = inf_comp = step_norm = 0.0;
  double norm_Vx = 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) norm_Vx += fabs(Vx[i]);
  for (int t = N - 1; t >= 0; --t) {
    double A[NX * NX], Bm[NX * NU], Qx[NX], Qu[NU], Qxx[NX * NX], Quu[NU * NU], Qux[NU * NX];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) A[i] = a.fx[SI(t, NX * NX, i)];
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Bm[i] = a.fu[SI(t, NX * NU, i)];
#pragma unroll
    for (int i = 0; i < NX; ++i) Qx[i] = a.lx[SI(t, NX, i)];
#pragma unroll
    for (int i = 0; i < NU; ++i) Qu[i] = a.lu[SI(t, NU, i)];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Qxx[i] = a.lxx[SI(t, NX * NX, i)];
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Quu[i] = a.luu[SI(t, NU * NU, i)];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) Qux[i] = a.lux[SI(t, NU * NX, i)];
    double y[MM], s[MM], g[MM], Gx[MM * NX], Gu[MM * NU];
    if constexpr (M > 0) {
#pragma unroll
      for (int i = 0; i < M; ++i) { y[i] = a.y[SI(t, M, i)]; s[i] = a.s[SI(t, M, i)]; g[i] = a.g[SI(t, M, i)]; }
#pragma unroll
      for (int i = 0; i < M * NX; ++i) Gx[i] = a.Gx[SI(t, M * NX, i)];
#pragma unroll
      for (int i = 0; i < M * NU; ++i) Gu[i] = a.Gu[SI(t, M * NU, i)];
      // Q_x = l_x + Q_yx^T y + A^T V_x ; Q_u = l_u + Q_yu^T y + B^T V_x   (:1391-1392)
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += Gx[r * NX + i] * y[r];
        Qx[i] = Qx[i] + s1; }
#pragma unroll
      for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += Gu[r * NU + i] * y[r];
        Qu[i] = Qu[i] + s1; }
    }
    double w[NX];   // V_x, or V_x + V_xx d_t under multiple shooting (msipddp_solver.cpp:1144-1145)
#pragma unroll
    for (int i = 0; i < NX; ++i) w[i] = Vx[i];
    if (ms) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s1 += Vxx[i * NX + k] * a.dfc[SI(t, NX, k)];
        w[i] = Vx[i] + s1; }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += A[k * NX + i] * w[k];
      Qx[i] = Qx[i] + s1; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += Bm[k * NU + i] * w[k];
      Qu[i] = Qu[i] + s1; }
    double T1[NX * NX], T2[NU * NX], P1[NX * NX], P2[NU * NX], P3[NU * NU];
    mm_tn<NX, NX, NX>(A, Vxx, T1);
    mm_tn<NU, NX, NX>(Bm, Vxx, T2);
    mm_nn<NX, NX, NX>(T1, A, P1);
    mm_nn<NU, NX, NX>(T2, A, P2);
    mm_nn<NU, NX, NU>(T2, Bm, P3);
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Qxx[i] = Qxx[i] + P1[i];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) Qux[i] = Qux[i] + P2[i];
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Quu[i] = Quu[i] + P3[i];
    if (a.Fxx) {   // full DDP: Q_xx += V_x(i) F_xx[t][i], Q_ux += V_x(i) F_ux[t][i], Q_uu += V_x(i) F_uu[t][i], V_x of step t + 1
                   // (ipddp_solver.cpp:1070-1082, 1396-1408; logddp_solver.cpp:505-515; CLDDP has no such terms and the entry
                   // point refuses the combination); rolled: the tensors are streamed
      for (int i = 0; i < NX; ++i) {
        const double w = Vx[i];
#pragma unroll
        for (int e = 0; e < NX * NX; ++e) Qxx[e] = Qxx[e] + w * a.Fxx[SI(t, NX * NX * NX, i * NX * NX + e)];
#pragma unroll
        for (int e = 0; e < NU * NX; ++e) Qux[e] = Qux[e] + w * a.Fux[SI(t, NX * NU * NX, i * NU * NX + e)];
#pragma unroll
        for (int e = 0; e < NU * NU; ++e) Quu[e] = Quu[e] + w * a.Fuu[SI(t, NX * NU * NU, i * NU * NU + e)];
      }
    }
    double kk[NU], KK[NU * NX];
    // ---------------------------------------------------------------- gains
    double YS[MM], rp[MM], rhat[MM], ssafe[MM], Sir[MM];
    if constexpr (M > 0) {
      const double s_floor = dmax(mu * 1e-3, kEpsSlackS);
#pragma unroll
      for (int r = 0; r < M; ++r) {
        ssafe[r] = msp ? s[r] : dmax(s[r], s_floor);
        YS[r] = msp ? y[r] / s[r] : clipp(y[r], ssafe[r]);       // msipddp_solver.cpp:1312-1316: YSinv(i, i) = y(i) / s(i)
        rp[r] = g[r] + s[r];
        const double rc = y[r] * s[r] - mu;
        rhat[r] = y[r] * rp[r] - rc;
        Sir[r] = msp ? rhat[r] / s[r] : clips(rhat[r], ssafe[r]);
        inf_pr = dmax(inf_pr, fabs(rp[r])); inf_comp = dmax(inf_comp, fabs(rc));
      }
      // Q_uu_reg = sym(Q_uu) + Q_yu^T YS Q_yu + reg I ; rhs = [Q_u + Q_yu^T S^-1 rhat | Q_ux + Q_yu^T YS Q_yx]   (:1424-1448)
      double Qr[NU * NU], rhs_u[NU], rhs_x[NU * NX];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += (Gu[r * NU + i] * YS[r]) * Gu[r * NU + c];
          Qr[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]) + s1;
        }
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += Gu[r * NU + i] * Sir[r];
        rhs_u[i] = Qu[i] + s1;
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s2 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s2 += (Gu[r * NU + i] * YS[r]) * Gx[r * NX + c];
          rhs_x[i * NX + c] = Qux[i * NX + c] + s2;
        }
      }
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Qr[0], rhs_u[0]);
#pragma unroll
        for (int c = 0; c < NX; ++c) KK[c] = -ldlt1_solve(Qr[0], rhs_x[c]);
      } else {
        LDLTd<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = rhs_u[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        for (int c = 0; c < NX; ++c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = rhs_x[i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
        }
      }
      // slack / dual direction gains (:1458-1472)
#pragma unroll
      for (int r = 0; r < M; ++r) {
        double temp = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) temp += Gu[r * NU + i] * kk[i];
        a.ky[SI(t, M, r)] = msp ? (rhat[r] + y[r] * temp) / s[r] : clips(rhat[r] + y[r] * temp, ssafe[r]);
        a.ks[SI(t, M, r)] = (-rp[r]) - temp;
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s2 = 0.0;
#pragma unroll
          for (int i = 0; i < NU; ++i) s2 += Gu[r * NU + i] * KK[i * NX + c];
          const double inner = Gx[r * NX + c] + s2;
          a.Ky[SI(t, M * NX, r * NX + c)] = msp ? YS[r] * inner : dclamp(YS[r] * inner, -kMaxRatioS, kMaxRatioS);
          a.Ks[SI(t, M * NX, r * NX + c)] = (-Gx[r * NX + c]) - s2;
        }
      }
      // condensed terms into the Q blocks (:1488-1492): un-regularised, un-symmetrised Q_uu
#pragma unroll
      for (int i = 0; i < NU; ++i) Qu[i] = rhs_u[i];
#pragma unroll
      for (int i = 0; i < NX; ++i) { double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += Gx[r * NX + i] * Sir[r];
        Qx[i] = Qx[i] + s1; }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) { double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += (Gx[r * NX + i] * YS[r]) * Gx[r * NX + c];
          Qxx[i * NX + c] = Qxx[i * NX + c] + s1; }
      if (msp) {   // msipddp_solver.cpp:1398: Q_ux += Q_yx^T YS^-1 Q_yu, an (nx x nu) product on the (nu x nx) block: the transpose for
                   // nu = 1 (same linear layout), elementwise for nx = nu; the entry point refuses every other shape
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            const int pi = (NU == 1) ? c : i, pc = (NU == 1) ? 0 : c;   // entry (pi, pc) of the product
            double s1 = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) s1 += (Gx[r * NX + pi] * YS[r]) * Gu[r * NU + (pc < NU ? pc : 0)];
            Qux[i * NX + c] = Qux[i * NX + c] + s1;
          }
      } else {
#pragma unroll
      for (int i = 0; i < NU * NX; ++i) Qux[i] = rhs_x[i];
      }
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) { double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += (Gu[r * NU + i] * YS[r]) * Gu[r * NU + c];
          Quu[i * NU + c] = Quu[i * NU + c] + s1; }
    } else if (ip) {
      // unconstrained IPDDP: Q_uu = sym(Q_uu) + reg I, kept in the value update (:1084-1101)
      double Qs[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]);
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] = Qs[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Quu[i * NU + i] += reg;
      double Qf[NU * NU];   // the matrix that is factored: this sweep's, or the cached one of the step (MSIPDDP, msipddp_solver.cpp:1169-1185)
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qf[i] = Quu[i];
      const bool caching = ms && a.QuuF != nullptr;
      const bool cached = caching && a.fvalid[(size_t)t * a.Bp + b] != 0;
      if (cached) {
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qf[i] = a.QuuF[SI(t, NU * NU, i)];
      }
      if (NU == 1) {
        if (caching && !cached) { a.QuuF[SI(t, 1, 0)] = Qf[0]; a.fvalid[(size_t)t * a.Bp + b] = 1; }
        kk[0] = -ldlt1_solve(Qf[0], Qu[0]);
#pragma unroll
        for (int c = 0; c < NX; ++c) KK[c] = -ldlt1_solve(Qf[0], Qux[c]);
      } else {
        LDLTd<NU> f;
        f.compute(Qf, NU);
        if (!f.ok) { if (caching) a.fvalid[(size_t)t * a.Bp + b] = 0; return false; }
        if (caching && !cached) {
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) a.QuuF[SI(t, NU * NU, i)] = Qf[i];
          a.fvalid[(size_t)t * a.Bp + b] = 1;
        }
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
    } else if (lg) {
      // LogDDP (logddp_solver.cpp:524-548): Q_uu_reg = Q_uu + reg I, THEN symmetrised; LDLT; [k | K] = -solve([Q_u | Q_ux]);
      // Q_uu itself stays as it is for the value update
      double Qr[NU * NU], Qs[NU * NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = Quu[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Qr[i * NU + c] + Qr[c * NU + i]);
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Qs[0], Qu[0]);
#pragma unroll
        for (int c = 0; c < NX; ++c) KK[c] = -ldlt1_solve(Qs[0], Qux[c]);
      } else {
        LDLTd<NU> f;
        f.compute(Qs, NU);
        if (!f.ok) return false;
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
      // CLDDP without bounds: PD test, H = (Q_uu + reg I)^-1, reg only in the factor (clddp_solver.cpp:130-145)
      double Qr[NU * NU], H[NU * NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = Quu[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += reg;
      if (min_real_eig<NU>(Qr) <= 0) return false;
      if (a.lo) {   // control-limited step: BoxQP on [lower - u_t, upper - u_t], warm-started with the previous k_t; feedback on the
                    // free directions only (clddp_solver.cpp:147-178)
        double lb[NU], ub[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { const double ut = a.U[SI(t, NU, i)]; lb[i] = a.lo[i] - ut; ub[i] = a.up[i] - ut; kk[i] = a.k[SI(t, NU, i)]; }
        int free_[NU];
        LDLTd<NU> Hfree;
        const int stq = boxqp_solve<NU>(a.opt, Qr, Qu, lb, ub, kk, free_, Hfree);
        if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
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
      } else {
      inverse_pplu<NU>(Qr, H);
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) s1 += (-H[i * NU + j]) * Qu[j];
        kk[i] = s1;
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s2 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) s2 += (-H[i * NU + j]) * Qux[j * NX + c];
          KK[i * NX + c] = s2;
        }
      }
      }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) a.k[SI(t, NU, i)] = kk[i];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) a.K[SI(t, NU * NX, i)] = KK[i];
    double Quuk[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s1 = 0.0;
#pragma unroll
      for (int j = 0; j < NU; ++j) s1 += Quu[i * NU + j] * kk[j];
      Quuk[i] = s1; }
    { double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) { s0 += Qu[i] * kk[i]; s1 += kk[i] * Quuk[i]; }
      dV0 += s0; dV1 += 0.5 * s1; }
    double KtQ[NX * NU];
    mm_tn<NX, NU, NU>(KK, Quu, KtQ);
    double Vn[NX * NX];
    if (ip) {   // IPDDP association order (ipddp_solver.cpp:1098-1101, 1497-1500)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += KK[j * NX + i] * Qu[j]; q += Qux[j * NX + i] * kk[j]; r += KtQ[i * NU + j] * kk[j]; }
        Vx[i] = ((Qx[i] + p) + q) + r;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { p += KK[j * NX + i] * Qux[j * NX + c]; q += Qux[j * NX + i] * KK[j * NX + c]; r += KtQ[i * NU + j] * KK[j * NX + c]; }
          Vn[i * NX + c] = ((Qxx[i * NX + c] + p) + q) + r;
        }
    } else {    // CLDDP association order (clddp_solver.cpp:188-191); LogDDP writes the same expression (logddp_solver.cpp:565-569)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += KtQ[i * NU + j] * kk[j]; q += Qux[j * NX + i] * kk[j]; r += KK[j * NX + i] * Qu[j]; }
        Vx[i] = ((Qx[i] + p) + q) + r;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { p += KtQ[i * NU + j] * KK[j * NX + c]; q += Qux[j * NX + i] * KK[j * NX + c]; r += KK[j * NX + i] * Qux[j * NX + c]; }
          Vn[i * NX + c] = ((Qxx[i * NX + c] + p) + q) + r;
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (Vn[i * NX + c] + Vn[c * NX + i]);
#pragma unroll
    for (int i = 0; i < NX; ++i) a.Vx[SI(t, NX, i)] = Vx[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) a.Vxx[SI(t, NX * NX, i)] = Vxx[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(Qu[i])); step_norm = dmax(step_norm, fabs(kk[i])); }
#pragma unroll
    for (int i = 0; i < NX; ++i) norm_Vx += fabs(Vx[i]);
  }
  if (!ip && !lg) {   // CLDDP scales inf_du (clddp_solver.cpp:194-201); the caller passes termination_scaling_max_factor in tau_min
    double sc = a.tau_min;
    sc = dmax(sc, norm_Vx / (double)(N * NX)) / sc;
    inf_du = inf_du / sc;
  }
  return true;
}

template <int NX, int NU, int M, bool T4>
__global__ __launch_bounds__(64) void k_stacks_backward(StackArgs a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const double mu = a.mu ? a.mu[b] : 0.0;
  double reg = a.reg_in[b];
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  bool ok = false;
  for (;;) {   // "retry with larger regularisation" loop of cddp_solver_base.cpp:93-111 (reg_factor <= 1: a single attempt)
    ok = sweep<NX, NU, M, T4>(a, b, reg, mu, dV0, dV1, inf_du, inf_pr, inf_comp, step_norm);
    if (ok || !(a.reg_factor > 1.0)) break;
    reg = reg * a.reg_factor;
    if (!(reg > 0.0)) reg = (a.opt.reg_min_value > 0.0) ? a.opt.reg_min_value : a.reg_max;   // 0 is a fixed point of reg * f (kernels.hpp::reg_increase)
    reg = dmin(reg, a.reg_max);
    if (reg >= a.reg_max) break;
  }
  a.ok[b] = ok ? 1 : 0;
  a.dV[(size_t)0 * a.Bp + b] = dV0; a.dV[(size_t)1 * a.Bp + b] = dV1;
  a.scal[(size_t)0 * a.Bp + b] = reg; a.scal[(size_t)1 * a.Bp + b] = inf_du; a.scal[(size_t)2 * a.Bp + b] = inf_pr;
  a.scal[(size_t)3 * a.Bp + b] = inf_comp; a.scal[(size_t)4 * a.Bp + b] = step_norm;
  double apr = 1.0, adu = 1.0;
  if constexpr (M > 0) {
    if (ok && a.branch != CDDP_HIP_STACKS_MSIPDDP_PATH) {   // rolloutLinearPolicy from dx0 = 0 (:1511-1520), dS / dY (:1522-1532), computeMaxStepSizes (:2939-2988)
      const int N = a.N;
      const double tau = dmax(a.tau_min, 1.0 - mu);
      double dx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = 0.0;
      for (int t = 0; t < N; ++t) {
#pragma unroll
        for (int i = 0; i < NX; ++i) a.dX[SI(t, NX, i)] = dx[i];
        for (int r = 0; r < M; ++r) {
          double p = 0.0, q = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) { p += a.Ks[SI(t, M * NX, r * NX + j)] * dx[j]; q += a.Ky[SI(t, M * NX, r * NX + j)] * dx[j]; }
          const double ds = a.ks[SI(t, M, r)] + p;
          const double dy = dclamp(a.ky[SI(t, M, r)] + q, -kMaxRatioS, kMaxRatioS);
          if (ds < 0.0) apr = dmin(apr, -tau * a.s[SI(t, M, r)] / ds);
          if (dy < 0.0) adu = dmin(adu, -tau * a.y[SI(t, M, r)] / dy);
        }
        double du[NU], dxn[NX];
#pragma unroll
        for (int i = 0; i < NU; ++i) { double p = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) p += a.K[SI(t, NU * NX, i * NX + j)] * dx[j];
          du[i] = a.k[SI(t, NU, i)] + p; }
#pragma unroll
        for (int i = 0; i < NX; ++i) { double p = 0.0, q = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) p += a.fx[SI(t, NX * NX, i * NX + j)] * dx[j];
#pragma unroll
          for (int j = 0; j < NU; ++j) q += a.fu[SI(t, NX * NU, i * NU + j)] * du[j];
          dxn[i] = (p + q) + 0.0; }
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) a.dX[SI(N, NX, i)] = dx[i];
      apr = dclamp(apr, 0.0, 1.0); adu = dclamp(adu, 0.0, 1.0);
    }
  }
  a.caps[(size_t)0 * a.Bp + b] = apr; a.caps[(size_t)1 * a.Bp + b] = adu;
}

int sfail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  return cddp_hip_internal_set_error(code, buf);
}
#define SCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return sfail(-10, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

#include "stacks_coop.hpp"
#include "stacks_te.hpp"

// the shapes pick_coop() instantiates (kept next to it)
constexpr bool sc_has_coop(int nx, int nu, int m) {
  return (nx == 4 && nu == 1 && (m == 0 || m == 2)) || (nx == 3 && nu == 2 && (m == 0 || m == 5)) || (nx == 6 && nu == 3 && (m == 0 || m == 6)) ||
         (nx == 12 && nu == 4 && (m == 0 || m == 8)) || (nx == 13 && nu == 4 && (m == 0 || m == 8)) || (nx == 14 && nu == 7 && (m == 0 || m == 14));
}
template <int NX, int NU, int M>
void launch(const StackArgs &a, hipStream_t s) {
  if constexpr (sc_has_coop(NX, NU, M)) {   // (tile-minor stacks exist only where a cooperative kernel does)
    if (a.t4) { hipLaunchKernelGGL((k_stacks_backward<NX, NU, M, true>), dim3((a.B + 63) / 64), dim3(64), 0, s, a); return; }
  }
  hipLaunchKernelGGL((k_stacks_backward<NX, NU, M, false>), dim3((a.B + 63) / 64), dim3(64), 0, s, a);
}
typedef void (*LaunchFn)(const StackArgs &, hipStream_t);

LaunchFn pick(int nx, int nu, int m) {
#define PICK(X, U, MM) if (nx == X && nu == U && m == MM) return &launch<X, U, MM>;
#ifndef CDDP_STACKS_DEV_SHAPE   // (kernel-development builds define it: one cooperative shape, seconds per compile; never set by the Makefile)
  PICK(1, 1, 0) PICK(1, 1, 1) PICK(1, 1, 2) PICK(2, 1, 0) PICK(2, 1, 2) PICK(3, 1, 0) PICK(3, 1, 2) PICK(4, 1, 0) PICK(4, 1, 2)
  PICK(3, 2, 0) PICK(3, 2, 4) PICK(3, 2, 5) PICK(4, 2, 0) PICK(4, 2, 4) PICK(6, 3, 0) PICK(6, 3, 6) PICK(8, 3, 0) PICK(8, 3, 6)
  PICK(12, 4, 0) PICK(12, 4, 8) PICK(13, 4, 0) PICK(13, 4, 8) PICK(14, 7, 0)
#endif
#undef PICK
  return nullptr;
}

// lane-cooperative form (stacks_coop.hpp): the default for nx > 8, where the one-lane kernel runs from scratch memory; the small
// shapes are instantiated for the bitwise cross-check of the two forms (CDDP_HIP_STACKS_SWEEP=coop | lane overrides the default)
LaunchFn pick_coop(int nx, int nu, int m) {
#define PICK(X, U, MM) static_assert(sc_has_coop(X, U, MM), "sc_has_coop() lists the cooperative shapes"); if (nx == X && nu == U && m == MM) return &launch_coop<X, U, MM>;
#ifdef CDDP_STACKS_DEV_SHAPE
  PICK(12, 4, 8)
#else
  PICK(4, 1, 0) PICK(4, 1, 2) PICK(3, 2, 0) PICK(3, 2, 5) PICK(6, 3, 0) PICK(6, 3, 6)
  PICK(12, 4, 0) PICK(12, 4, 8) PICK(13, 4, 0) PICK(13, 4, 8) PICK(14, 7, 0) PICK(14, 7, 14)
#endif
#undef PICK
  return nullptr;
}

// batch-major [b][r] <-> stack [r][Bp] (r = t * E + e), one thread per element, b fastest: the stack side is coalesced
__global__ void k_stack_transpose(double *aos, double *soa, int B, int Bp, int R, int to_stack, int E, int t4) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (b >= B) return;
  size_t si = (size_t)r * Bp + b;
  if (t4) { const int t = r / E, e = r - t * E; si = (((size_t)t * (size_t)(Bp >> 2) + (size_t)(b >> 2)) * E + e) * 4 + (size_t)(b & 3); }
  if (to_stack) soa[si] = aos[(size_t)b * R + r];
  else aos[(size_t)b * R + r] = soa[si];
}
}  // namespace

struct cddp_hip_stack_handle {
  int device = 0, B = 0, Bp = 0, nx = 0, nu = 0, m = 0, N = 0;
  LaunchFn fn = nullptr;        // one lane per trajectory
  LaunchFn fn_coop = nullptr;   // sixteen lanes per trajectory (stacks_coop.hpp)
  int used_coop = 0;            // form of the last sweep
  bool coop_default = false;    // the handle's default sweep (and stack layout) is the cooperative one
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<void *> allocs;
  StackArgs a{};
  double *d_reg = nullptr, *d_mu = nullptr;
  double *d_Fxx = nullptr, *d_Fuu = nullptr, *d_Fux = nullptr;   // allocated by the first cddp_hip_set_hessian_stacks
  double *d_U = nullptr, *d_lo = nullptr, *d_up = nullptr;      // allocated by the first cddp_hip_set_control_box
  double *d_dfc = nullptr;                                      // allocated by the first cddp_hip_set_defect_stack
  double *d_QuuF = nullptr; int *d_fvalid = nullptr;            // allocated by the first cddp_hip_stacks_factor_cache(h, 1)
  bool factor_cache = false;
  bool have_dyn = false, have_con = false, swept = false;
  StackTeArgs te{};             // terminal-equality branch (stacks_te.hpp): buffers of the last cddp_hip_set_terminal_equality
  int te_cap = 0;               // rows the te buffers were sized for
  bool have_te = false, swept_te = false;
  double last_ms = 0.0;
  double *d_stage = nullptr; size_t stage_cap = 0;   // batch-major staging buffer of upload() / download()
};

namespace {
int salloc(cddp_hip_stack_handle *h, double **p, size_t n) {
  void *q = nullptr;
  SCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(double)));
  SCHK(hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(double), h->stream));
  h->allocs.push_back(q);
  *p = (double *)q;
  return 0;
}
// Host arrays are batch-major [b][t][e], the stacks [t][e][Bp].  Round 6: the transposition runs on the DEVICE (k_stack_transpose) around one
// contiguous copy -- the host loop it replaces (a write stride of Bp * 8 bytes per element) moved 1.1 GB/s and was 94 % of a plug-in solve
// (bench.py's plug-in line: 4096 pendulum trajectories, 30 sweeps: 4.35 s of 4.65 s; profiles/r06_plugin_route.md).
int stage_reserve(cddp_hip_stack_handle *h, size_t n) {
  if (n <= h->stage_cap) return 0;
  if (h->d_stage) { SCHK(hipStreamSynchronize(h->stream)); SCHK(hipFree(h->d_stage)); h->d_stage = nullptr; h->stage_cap = 0; }
  void *q = nullptr;
  SCHK(hipMalloc(&q, n * sizeof(double)));
  h->d_stage = (double *)q; h->stage_cap = n;
  return 0;
}
int upload(cddp_hip_stack_handle *h, const double *src, double *dst, int T, int E) {
  if (!src) return 0;
  const size_t n = (size_t)T * E * h->B;
  if (n == 0) return 0;
  { int rc = stage_reserve(h, n); if (rc) return rc; }
  SCHK(hipMemcpyAsync(h->d_stage, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_stack_transpose, dim3((unsigned)((h->B + 255) / 256), (unsigned)(T * E)), dim3(256), 0, h->stream, h->d_stage, dst, h->B, h->Bp, T * E, 1, E,
                     (h->a.t4 && T > 1) ? 1 : 0);
  SCHK(hipGetLastError());
  SCHK(hipStreamSynchronize(h->stream));   // the caller may reuse src; the staging buffer is reused by the next upload
  return 0;
}
int download(cddp_hip_stack_handle *h, const double *src, double *dst, int T, int E) {
  if (!dst) return 0;
  const size_t n = (size_t)T * E * h->B;
  if (n == 0) return 0;
  { int rc = stage_reserve(h, n); if (rc) return rc; }
  hipLaunchKernelGGL(k_stack_transpose, dim3((unsigned)((h->B + 255) / 256), (unsigned)(T * E)), dim3(256), 0, h->stream, h->d_stage, const_cast<double *>(src), h->B, h->Bp, T * E, 0, E,
                     (h->a.t4 && T > 1) ? 1 : 0);
  SCHK(hipGetLastError());
  SCHK(hipMemcpyAsync(dst, h->d_stage, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  SCHK(hipStreamSynchronize(h->stream));
  return 0;
}
}  // namespace

extern "C" {

int cddp_hip_stacks_create_abi(int abi_version, int options_bytes, int device, int batch, int nx, int nu, int m, int horizon, cddp_hip_stack_handle **out) {
  if (!out) return sfail(-1, "null argument");
  if (abi_version != CDDP_HIP_ABI_VERSION || options_bytes != (int)sizeof(cddp_hip_options))
    return sfail(-2, "ABI mismatch: caller built against version %d with a %d-byte cddp_hip_options, library has version %d and %d bytes",
                 abi_version, options_bytes, CDDP_HIP_ABI_VERSION, (int)sizeof(cddp_hip_options));
  if (batch <= 0 || nx <= 0 || nu <= 0 || m < 0 || horizon <= 0) return sfail(-1, "bad dimensions batch=%d nx=%d nu=%d m=%d N=%d", batch, nx, nu, m, horizon);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sfail(-20, "no HIP device available: the stack-fed sweep has no CPU fallback");
  if (device < 0 || device >= ndev) return sfail(-1, "device %d out of range (%d devices)", device, ndev);
  LaunchFn fn = pick(nx, nu, m), fn_coop = pick_coop(nx, nu, m);
  if (!fn && !fn_coop) return sfail(-4, "no stack-fed instantiation for nx=%d nu=%d m=%d", nx, nu, m);
  SCHK(hipSetDevice(device));
  cddp_hip_stack_handle *h = new cddp_hip_stack_handle();
  h->device = device; h->B = batch; h->Bp = (batch + 63) / 64 * 64; h->nx = nx; h->nu = nu; h->m = m; h->N = horizon; h->fn = fn; h->fn_coop = fn_coop;
  {   // tile-minor stacks where the cooperative sweep is the default (CDDP_HIP_STACKS_LAYOUT=plain | t4 overrides: the cross-check of the two)
    // Which sweep is the handle's default: the cooperative one (sixteen lanes per trajectory: 16 x the wavefronts of the one-lane form) from
    // nx = 6 on, where the one-lane kernel runs from scratch memory (nx 6 / m 6: 10.2 -> 1.0 ms at 4096 trajectories, 21 -> 15 ms at 65536), and
    // -- round 6 -- for the smaller shapes WITH PATH ROWS while the batch leaves the chip mostly empty under the one-lane form (B / 64 wavefronts
    // on 1024 SIMDs): C2 shape 0.92 -> 0.49 ms at 4096, 0.98 -> 0.65 at 8192 (1.02 -> 1.27 at 16384: one-lane from there), C3 shape 3.2 -> 1.4 ms
    // at 8192, 3.3 -> 2.9 at 16384.  The CLDDP / unconstrained sweeps of those shapes stay one-lane (0.19 vs 0.27 ms).  profiles/r06_plugin_route.md
    h->coop_default = fn_coop && (nx >= 6 || !fn || (m > 0 && batch <= (nx <= 3 ? 16384 : 8192)));
    h->a.t4 = h->coop_default ? 1 : 0;
    if (const char *e = std::getenv("CDDP_HIP_STACKS_LAYOUT")) { if (!std::strcmp(e, "plain")) h->a.t4 = 0; else if (!std::strcmp(e, "t4") && fn_coop) h->a.t4 = 1; }
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return sfail(-10, "hipStreamCreate failed"); }
  hipEventCreate(&h->e0); hipEventCreate(&h->e1);
  const size_t Bp = h->Bp, N = horizon;
  StackArgs &a = h->a;
  a.B = batch; a.Bp = h->Bp; a.N = horizon;
  struct { double **p; size_t n; } bufs[] = {
      {(double **)&a.fx, N * nx * nx * Bp}, {(double **)&a.fu, N * nx * nu * Bp}, {(double **)&a.lx, N * nx * Bp}, {(double **)&a.lu, N * nu * Bp},
      {(double **)&a.lxx, N * nx * nx * Bp}, {(double **)&a.luu, N * nu * nu * Bp}, {(double **)&a.lux, N * nu * nx * Bp},
      {(double **)&a.VxN, (size_t)nx * Bp}, {(double **)&a.VxxN, (size_t)nx * nx * Bp},
      {(double **)&a.y, N * m * Bp}, {(double **)&a.s, N * m * Bp}, {(double **)&a.g, N * m * Bp}, {(double **)&a.Gx, N * m * nx * Bp}, {(double **)&a.Gu, N * m * nu * Bp},
      {&a.K, N * nu * nx * Bp}, {&a.k, N * nu * Bp}, {&a.Vx, (N + 1) * nx * Bp}, {&a.Vxx, (N + 1) * nx * nx * Bp}, {&a.dV, 2 * Bp},
      {&a.ky, N * m * Bp}, {&a.Ky, N * m * nx * Bp}, {&a.ks, N * m * Bp}, {&a.Ks, N * m * nx * Bp}, {&a.dX, (N + 1) * nx * Bp},
      {&a.scal, 6 * Bp}, {&a.caps, 2 * Bp}, {&h->d_reg, Bp}, {&h->d_mu, Bp}};
  for (auto &bf : bufs) { int rc = salloc(h, bf.p, bf.n); if (rc) { cddp_hip_stacks_destroy(h); return rc; } }
  { void *q = nullptr; if (hipMalloc(&q, Bp * sizeof(int)) != hipSuccess) { cddp_hip_stacks_destroy(h); return sfail(-10, "hipMalloc failed"); } h->allocs.push_back(q); a.ok = (int *)q; }
  a.reg_in = h->d_reg; a.mu = h->d_mu;
  if (hipStreamSynchronize(h->stream) != hipSuccess) { cddp_hip_stacks_destroy(h); return sfail(-10, "device initialisation failed"); }
  *out = h;
  return 0;
}

int cddp_hip_stacks_destroy(cddp_hip_stack_handle *h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (void *q : h->allocs) hipFree(q);
  if (h->d_stage) hipFree(h->d_stage);
  if (h->e0) { hipEventDestroy(h->e0); hipEventDestroy(h->e1); }
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int cddp_hip_set_stacks(cddp_hip_stack_handle *h, const double *fx, const double *fu, const double *lx, const double *lu,
                        const double *lxx, const double *luu, const double *lux, const double *VxN, const double *VxxN) {
  if (!h) return sfail(-1, "null handle");
  SCHK(hipSetDevice(h->device));
  if (!h->have_dyn && !(fx && fu && lx && lu && lxx && luu && lux && VxN && VxxN))
    return sfail(-1, "the first cddp_hip_set_stacks call must supply every stack (later calls may pass NULL to keep one)");
  const int N = h->N, nx = h->nx, nu = h->nu;
  struct { const double *src; const double *dst; int T, E; } items[] = {
      {fx, h->a.fx, N, nx * nx}, {fu, h->a.fu, N, nx * nu}, {lx, h->a.lx, N, nx}, {lu, h->a.lu, N, nu}, {lxx, h->a.lxx, N, nx * nx},
      {luu, h->a.luu, N, nu * nu}, {lux, h->a.lux, N, nu * nx}, {VxN, h->a.VxN, 1, nx}, {VxxN, h->a.VxxN, 1, nx * nx}};
  for (auto &it : items) { int rc = upload(h, it.src, (double *)it.dst, it.T, it.E); if (rc) return rc; }
  h->have_dyn = true; h->swept = false;
  return 0;
}

int cddp_hip_set_defect_stack(cddp_hip_stack_handle *h, const double *defects) {
  if (!h) return sfail(-1, "null handle");
  SCHK(hipSetDevice(h->device));
  if (!defects) { h->a.dfc = nullptr; h->swept = false; return 0; }
  if (!h->d_dfc) { int rc = salloc(h, &h->d_dfc, (size_t)h->N * h->nx * h->Bp); if (rc) return rc; }
  int rc = upload(h, defects, h->d_dfc, h->N, h->nx); if (rc) return rc;
  h->a.dfc = h->d_dfc; h->swept = false;
  return 0;
}

int cddp_hip_stacks_factor_cache(cddp_hip_stack_handle *h, int enable) {
  if (!h) return sfail(-1, "null handle");
  SCHK(hipSetDevice(h->device));
  if (!enable) { h->factor_cache = false; return 0; }
  const size_t nq = (size_t)h->N * h->nu * h->nu * h->Bp, nv = (size_t)h->N * h->Bp;
  if (!h->d_QuuF) {
    int rc = salloc(h, &h->d_QuuF, nq); if (rc) return rc;
    void *q = nullptr; SCHK(hipMalloc(&q, nv * sizeof(int))); h->allocs.push_back(q); h->d_fvalid = (int *)q;
  }
  SCHK(hipMemsetAsync(h->d_fvalid, 0, nv * sizeof(int), h->stream));
  SCHK(hipStreamSynchronize(h->stream));
  h->factor_cache = true;
  return 0;
}

int cddp_hip_set_control_box(cddp_hip_stack_handle *h, const double *lower, const double *upper, const double *U) {
  if (!h) return sfail(-1, "null handle");
  SCHK(hipSetDevice(h->device));
  if (!lower && !upper && !U) { h->a.lo = h->a.up = h->a.U = nullptr; h->swept = false; return 0; }   // no bounds
  if (!h->a.lo && !(lower && upper && U)) return sfail(-1, "the first cddp_hip_set_control_box call needs lower, upper and the control stack U");
  if ((lower == nullptr) != (upper == nullptr)) return sfail(-1, "lower and upper bounds come together");
  const int N = h->N, nu = h->nu;
  if (!h->d_U) {
    int rc = salloc(h, &h->d_U, (size_t)N * nu * h->Bp); if (rc) return rc;
    rc = salloc(h, &h->d_lo, nu); if (rc) return rc;
    rc = salloc(h, &h->d_up, nu); if (rc) return rc;
  }
  if (lower) {
    for (int i = 0; i < nu; ++i) if (!(lower[i] <= upper[i])) return sfail(-2, "control bound %d: lower %g > upper %g", i, lower[i], upper[i]);
    SCHK(hipMemcpyAsync(h->d_lo, lower, sizeof(double) * nu, hipMemcpyHostToDevice, h->stream));
    SCHK(hipMemcpyAsync(h->d_up, upper, sizeof(double) * nu, hipMemcpyHostToDevice, h->stream));
    SCHK(hipStreamSynchronize(h->stream));
  }
  if (U) { int rc = upload(h, U, h->d_U, N, nu); if (rc) return rc; }
  h->a.lo = h->d_lo; h->a.up = h->d_up; h->a.U = h->d_U;
  h->swept = false;
  return 0;
}

int cddp_hip_set_hessian_stacks(cddp_hip_stack_handle *h, const double *Fxx, const double *Fuu, const double *Fux) {
  if (!h) return sfail(-1, "null handle");
  SCHK(hipSetDevice(h->device));
  if (!Fxx && !Fuu && !Fux) { h->a.Fxx = h->a.Fuu = h->a.Fux = nullptr; h->swept = false; return 0; }   // back to Gauss-Newton
  if (!(Fxx && Fuu && Fux)) return sfail(-1, "cddp_hip_set_hessian_stacks needs F_xx, F_uu and F_ux together (or three NULLs to drop them)");
  const int N = h->N, nx = h->nx, nu = h->nu;
  if (!h->d_Fxx) {
    int rc = salloc(h, &h->d_Fxx, (size_t)N * nx * nx * nx * h->Bp); if (rc) return rc;
    rc = salloc(h, &h->d_Fuu, (size_t)N * nx * nu * nu * h->Bp); if (rc) return rc;
    rc = salloc(h, &h->d_Fux, (size_t)N * nx * nu * nx * h->Bp); if (rc) return rc;
  }
  struct { const double *src; double *dst; int T, E; } items[] = {
      {Fxx, h->d_Fxx, N, nx * nx * nx}, {Fuu, h->d_Fuu, N, nx * nu * nu}, {Fux, h->d_Fux, N, nx * nu * nx}};
  for (auto &it : items) { int rc = upload(h, it.src, it.dst, it.T, it.E); if (rc) return rc; }
  h->a.Fxx = h->d_Fxx; h->a.Fuu = h->d_Fuu; h->a.Fux = h->d_Fux;
  h->swept = false;
  return 0;
}

int cddp_hip_set_constraint_stacks(cddp_hip_stack_handle *h, const double *y, const double *s, const double *g, const double *Gx, const double *Gu) {
  if (!h) return sfail(-1, "null handle");
  if (h->m <= 0) return sfail(-1, "this stack handle was created without path constraints (m = 0)");
  SCHK(hipSetDevice(h->device));
  if (!h->have_con && !(y && s && g && Gx && Gu)) return sfail(-1, "the first cddp_hip_set_constraint_stacks call must supply y, s, g, G_x and G_u");
  const int N = h->N, nx = h->nx, nu = h->nu, m = h->m;
  struct { const double *src; const double *dst; int T, E; } items[] = {
      {y, h->a.y, N, m}, {s, h->a.s, N, m}, {g, h->a.g, N, m}, {Gx, h->a.Gx, N, m * nx}, {Gu, h->a.Gu, N, m * nu}};
  for (auto &it : items) { int rc = upload(h, it.src, (double *)it.dst, it.T, it.E); if (rc) return rc; }
  h->have_con = true; h->swept = false;
  return 0;
}

// Terminal-equality data of the reduced-LQR branch (stacks_te.hpp): H_T [B][pT][nx] (dense rows of the stacked terminal-equality Jacobian),
// b_T [B][pT] = -h_T(x_N), lambda_prev [B][pT] (Lambda_T_eq_ of the iterate), floor [B] = max(1e-10, jacobian_regularization_value *
// pow(max(mu, 0), jacobian_regularization_exponent)) evaluated by the caller.
int cddp_hip_set_terminal_equality(cddp_hip_stack_handle *h, int pT, const double *HT, const double *bT, const double *lambda_prev, const double *reg_floor) {
  if (!h || !HT || !bT || !lambda_prev || !reg_floor) return sfail(-1, "null argument");
  if (pT < 1 || pT > kPTS) return sfail(-3, "terminal-equality rows must be 1 .. %d on the stack-fed route (got %d)", kPTS, pT);
  if (h->m != 0) return sfail(-1, "the terminal-equality branch takes the path constraints condensed into the LQ stacks: use a handle created with m = 0");
  if (!pick_te(h->nx, h->nu)) return sfail(-3, "no terminal-equality stack kernel for nx = %d, nu = %d", h->nx, h->nu);
  if (h->a.t4 && h->nx < 6) return sfail(-3, "the terminal-equality stack kernel of this shape reads [t][e][batch] stacks: the handle was created tile-minor (CDDP_HIP_STACKS_LAYOUT=t4)");
  SCHK(hipSetDevice(h->device));
  const int N = h->N, nx = h->nx, nu = h->nu, Bp = h->Bp;
  if (pT > h->te_cap) {
    int rc;
    double *p = nullptr;
    if ((rc = salloc(h, &p, (size_t)pT * nx * Bp))) return rc; h->te.HT = p;
    if ((rc = salloc(h, &p, (size_t)pT * Bp))) return rc; h->te.bT = p;
    if ((rc = salloc(h, &p, (size_t)pT * Bp))) return rc; h->te.lam_prev = p;
    if ((rc = salloc(h, &p, (size_t)Bp))) return rc; h->te.floor_ = p;
    if ((rc = salloc(h, &h->te.te_p, (size_t)(pT + 1) * (N + 1) * nx * Bp))) return rc;
    if ((rc = salloc(h, &h->te.te_k, (size_t)(pT + 1) * N * nu * Bp))) return rc;
    if ((rc = salloc(h, &h->te.dlam, (size_t)pT * Bp))) return rc;
    if (!h->te.dX && (rc = salloc(h, &h->te.dX, (size_t)(N + 1) * nx * Bp))) return rc;
    h->te_cap = pT;
  }
  h->te.pT = pT;
  int rc;
  if ((rc = upload(h, HT, const_cast<double *>(h->te.HT), pT, nx))) return rc;
  if ((rc = upload(h, bT, const_cast<double *>(h->te.bT), 1, pT))) return rc;
  if ((rc = upload(h, lambda_prev, const_cast<double *>(h->te.lam_prev), 1, pT))) return rc;
  if ((rc = upload(h, reg_floor, const_cast<double *>(h->te.floor_), 1, 1))) return rc;
  h->have_te = true;
  return 0;
}

int cddp_hip_stacks_get_terminal(cddp_hip_stack_handle *h, double *dlambda, double *dX) {
  if (!h) return sfail(-1, "null handle");
  if (!h->swept_te) return sfail(-1, "no terminal-equality sweep result");
  SCHK(hipSetDevice(h->device));
  int rc;
  if ((rc = download(h, h->te.dlam, dlambda, 1, h->te.pT))) return rc;
  if ((rc = download(h, h->te.dX, dX, h->N + 1, h->nx))) return rc;
  return 0;
}

int cddp_hip_stacks_backward(cddp_hip_stack_handle *h, int branch, const cddp_hip_options *opt, const double *reg, const double *mu,
                             int retry, int32_t *ok) {
  if (!h || !opt || !reg) return sfail(-1, "null argument");
  if (branch == CDDP_HIP_STACKS_IPDDP_TERM_EQ) {   // reduced LQR with terminal equality rows (stacks_te.hpp)
    if (!h->have_dyn) return sfail(-1, "cddp_hip_set_stacks must be called before cddp_hip_stacks_backward");
    if (!h->have_te) return sfail(-1, "cddp_hip_set_terminal_equality must be called before the terminal-equality sweep");
    if (h->a.Fxx) return sfail(-1, "the terminal-equality branch takes the second-order terms folded into its Q / R / M stacks (ipddp_solver.cpp:1160-1178): drop the Hessian stacks");
    for (int b = 0; b < h->B; ++b) if (!(reg[b] >= 0.0)) return sfail(-2, "regularisation of trajectory %d must be non-negative (got %g)", b, reg[b]);
    if (retry && (!(opt->reg_update_factor > 1.0) || !(opt->reg_max_value > 0.0)))
      return sfail(-2, "retry needs regularization.update_factor > 1 and max_value > 0 (got %g, %g)", opt->reg_update_factor, opt->reg_max_value);
    SCHK(hipSetDevice(h->device));
    SCHK(hipMemcpyAsync(h->d_reg, reg, sizeof(double) * h->B, hipMemcpyHostToDevice, h->stream));
    StackArgs a = h->a;
    a.branch = branch; a.opt = *opt; a.mu = nullptr; a.lo = a.up = a.U = nullptr; a.QuuF = nullptr; a.fvalid = nullptr;
    a.reg_factor = retry ? opt->reg_update_factor : 0.0; a.reg_max = opt->reg_max_value; a.tau_min = opt->barrier_min_fraction_to_boundary;
    SCHK(hipEventRecord(h->e0, h->stream));
    pick_te(h->nx, h->nu)(a, h->te, h->stream);
    SCHK(hipEventRecord(h->e1, h->stream));
    SCHK(hipGetLastError());
    if (ok) SCHK(hipMemcpyAsync(ok, a.ok, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
    SCHK(hipStreamSynchronize(h->stream));
    float ms = 0; hipEventElapsedTime(&ms, h->e0, h->e1);
    h->last_ms = ms; h->swept = true; h->swept_te = true; h->used_coop = 0;
    return 0;
  }
  if (branch != CDDP_HIP_STACKS_CLDDP && branch != CDDP_HIP_STACKS_IPDDP && branch != CDDP_HIP_STACKS_IPDDP_PATH && branch != CDDP_HIP_STACKS_LOGDDP &&
      branch != CDDP_HIP_STACKS_MSIPDDP && branch != CDDP_HIP_STACKS_MSIPDDP_PATH)
    return sfail(-2, "unknown stack-fed branch %d", branch);
  if (branch == CDDP_HIP_STACKS_MSIPDDP_PATH) {
    if (!(h->nu == 1 || h->nx == h->nu))   // msipddp_solver.cpp:1398 adds an (nx x nu) product to the (nu x nx) block Q_ux
      return sfail(-1, "the path-constrained MSIPDDP recursion is only defined for nu = 1 or nx = nu (msipddp_solver.cpp:1398 adds an (nx x nu) product to the (nu x nx) block Q_ux); got nx = %d, nu = %d", h->nx, h->nu);
    if (!h->a.dfc) return sfail(-1, "cddp_hip_set_defect_stack must be called before the MSIPDDP sweep");
  }
  if (branch == CDDP_HIP_STACKS_MSIPDDP) {
    if (h->m > 0)   // msipddp_solver.cpp:1398 adds an (nx x nu) product to the (nu x nx) block Q_ux: not a defined recursion for nx != nu
      return sfail(-1, "the MSIPDDP branch covers the unconstrained recursion (msipddp_solver.cpp:1112-1208); handle with m = 0");
    if (!h->a.dfc) return sfail(-1, "cddp_hip_set_defect_stack must be called before the MSIPDDP sweep");
    if (h->a.Fxx) return sfail(-1, "the MSIPDDP branch is Gauss-Newton here: its second-order terms weigh the Hessians with the costates (msipddp_solver.cpp:1151-1163); drop the Hessian stacks");
  }
  if (!h->have_dyn) return sfail(-1, "cddp_hip_set_stacks must be called before cddp_hip_stacks_backward");
  if (branch == CDDP_HIP_STACKS_IPDDP_PATH || branch == CDDP_HIP_STACKS_MSIPDDP_PATH) {
    if (h->m <= 0) return sfail(-1, "the path-constrained branch needs a handle created with m > 0");
    if (!h->have_con) return sfail(-1, "cddp_hip_set_constraint_stacks must be called before the path-constrained sweep");
    if (!mu) return sfail(-1, "the path-constrained branch needs the barrier parameter mu[b]");
    for (int b = 0; b < h->B; ++b) if (!(mu[b] > 0.0)) return sfail(-2, "barrier parameter of trajectory %d must be positive (got %g)", b, mu[b]);
  } else if (h->m > 0) {
    return sfail(-1, "this handle carries path-constraint stacks (m = %d): use CDDP_HIP_STACKS_IPDDP_PATH / CDDP_HIP_STACKS_MSIPDDP_PATH, or a handle with m = 0", h->m);
  }
  if (branch == CDDP_HIP_STACKS_CLDDP && h->a.Fxx)
    return sfail(-1, "CLDDPSolver::backwardPass has no second-order dynamics terms (clddp_solver.cpp:79-204): drop the Hessian stacks for this branch");
  for (int b = 0; b < h->B; ++b) if (!(reg[b] >= 0.0)) return sfail(-2, "regularisation of trajectory %d must be non-negative (got %g)", b, reg[b]);
  if (retry && (!(opt->reg_update_factor > 1.0) || !(opt->reg_max_value > 0.0)))
    return sfail(-2, "retry needs regularization.update_factor > 1 and max_value > 0 (got %g, %g)", opt->reg_update_factor, opt->reg_max_value);
  SCHK(hipSetDevice(h->device));
  SCHK(hipMemcpyAsync(h->d_reg, reg, sizeof(double) * h->B, hipMemcpyHostToDevice, h->stream));
  if (mu) SCHK(hipMemcpyAsync(h->d_mu, mu, sizeof(double) * h->B, hipMemcpyHostToDevice, h->stream));
  StackArgs a = h->a;
  a.branch = branch;
  a.opt = *opt;
  if (branch != CDDP_HIP_STACKS_CLDDP) a.lo = a.up = a.U = nullptr;   // the box belongs to the CLDDP branch
  a.mu = mu ? h->d_mu : nullptr;
  a.QuuF = (h->factor_cache && branch == CDDP_HIP_STACKS_MSIPDDP) ? h->d_QuuF : nullptr;
  a.fvalid = a.QuuF ? h->d_fvalid : nullptr;
  a.reg_factor = retry ? opt->reg_update_factor : 0.0;
  a.reg_max = opt->reg_max_value;
  a.tau_min = (branch == CDDP_HIP_STACKS_CLDDP) ? opt->termination_scaling_max_factor : opt->barrier_min_fraction_to_boundary;
  SCHK(hipEventRecord(h->e0, h->stream));
  {   // ONE launch; nx > 8 defaults to the cooperative form
    LaunchFn f = (h->coop_default && h->fn_coop) ? h->fn_coop : (h->fn ? h->fn : h->fn_coop);
    if (const char *e = std::getenv("CDDP_HIP_STACKS_SWEEP")) {
      if (!std::strcmp(e, "coop") && h->fn_coop) f = h->fn_coop;
      else if (!std::strcmp(e, "lane") && h->fn) f = h->fn;
    }
    if (f == h->fn_coop && !a.t4) {
      // the cooperative form addresses an element of a step record through a 32-bit byte offset behind the record's base (stacks_coop.hpp)
      const long long nx = h->nx, nu = h->nu, m = h->m > 0 ? h->m : 1;
      long long emax = nx * nx > m * nx ? nx * nx : m * nx;
      if (a.Fxx) emax = nx * nx * nx;
      if ((emax + 16) * (long long)h->Bp * 8 >= (1ll << 32)) {
        if (h->fn) f = h->fn;
        else return sfail(-4, "cooperative stack-fed sweep: batch %d too large for its 32-bit record offsets (nx=%d%s) -- split the batch", h->B, h->nx, a.Fxx ? ", with Hessian stacks" : "");
      }
    }
    h->used_coop = (f == h->fn_coop) ? 1 : 0;
    f(a, h->stream);
  }
  SCHK(hipEventRecord(h->e1, h->stream));
  SCHK(hipGetLastError());
  if (ok) SCHK(hipMemcpyAsync(ok, a.ok, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
  SCHK(hipStreamSynchronize(h->stream));
  float ms = 0; hipEventElapsedTime(&ms, h->e0, h->e1);
  h->last_ms = ms; h->swept = true;
  return 0;
}

double cddp_hip_stacks_last_kernel_ms(cddp_hip_stack_handle *h) { return h ? h->last_ms : -1.0; }
int cddp_hip_stacks_last_sweep_form(cddp_hip_stack_handle *h) { return h ? h->used_coop : -1; }

int cddp_hip_stacks_get_gains(cddp_hip_stack_handle *h, double *K, double *k, double *Vx, double *Vxx, double *dV) {
  if (!h) return sfail(-1, "null handle");
  if (!h->swept) return sfail(-1, "no sweep result: call cddp_hip_stacks_backward first");
  SCHK(hipSetDevice(h->device));
  const int N = h->N, nx = h->nx, nu = h->nu;
  int rc;
  if ((rc = download(h, h->a.K, K, N, nu * nx))) return rc;
  if ((rc = download(h, h->a.k, k, N, nu))) return rc;
  if ((rc = download(h, h->a.Vx, Vx, N + 1, nx))) return rc;
  if ((rc = download(h, h->a.Vxx, Vxx, N + 1, nx * nx))) return rc;
  if ((rc = download(h, h->a.dV, dV, 1, 2))) return rc;
  return 0;
}

int cddp_hip_stacks_get_constraint_gains(cddp_hip_stack_handle *h, double *k_y, double *K_y, double *k_s, double *K_s, double *dX) {
  if (!h) return sfail(-1, "null handle");
  if (!h->swept || h->m <= 0) return sfail(-1, "no path-constrained sweep result");
  SCHK(hipSetDevice(h->device));
  const int N = h->N, nx = h->nx, m = h->m;
  int rc;
  if ((rc = download(h, h->a.ky, k_y, N, m))) return rc;
  if ((rc = download(h, h->a.Ky, K_y, N, m * nx))) return rc;
  if ((rc = download(h, h->a.ks, k_s, N, m))) return rc;
  if ((rc = download(h, h->a.Ks, K_s, N, m * nx))) return rc;
  if ((rc = download(h, h->a.dX, dX, N + 1, nx))) return rc;
  return 0;
}

int cddp_hip_stacks_get_scalars(cddp_hip_stack_handle *h, double *reg, double *inf_du, double *inf_pr, double *inf_comp, double *step_norm,
                                double *alpha_pr_max, double *alpha_du_max) {
  if (!h) return sfail(-1, "null handle");
  if (!h->swept) return sfail(-1, "no sweep result: call cddp_hip_stacks_backward first");
  SCHK(hipSetDevice(h->device));
  std::vector<double> sc((size_t)6 * h->Bp), cp((size_t)2 * h->Bp);
  SCHK(hipMemcpy(sc.data(), h->a.scal, sc.size() * sizeof(double), hipMemcpyDeviceToHost));
  SCHK(hipMemcpy(cp.data(), h->a.caps, cp.size() * sizeof(double), hipMemcpyDeviceToHost));
  double *outs[5] = {reg, inf_du, inf_pr, inf_comp, step_norm};
  for (int i = 0; i < 5; ++i) if (outs[i]) for (int b = 0; b < h->B; ++b) outs[i][b] = sc[(size_t)i * h->Bp + b];
  if (alpha_pr_max) for (int b = 0; b < h->B; ++b) alpha_pr_max[b] = cp[b];
  if (alpha_du_max) for (int b = 0; b < h->B; ++b) alpha_du_max[b] = cp[(size_t)h->Bp + b];
  return 0;
}

// One-shot form of round 1 (kept for callers that sweep a set of stacks exactly once): create, upload, ONE launch,
// download, destroy.  reg is a scalar, no retry.
int cddp_hip_backward_stacks(int device, int batch, int nx, int nu, int horizon, const double *fx, const double *fu, const double *lx,
                             const double *lu, const double *lxx, const double *luu, const double *lux, const double *VxN,
                             const double *VxxN, double reg, int reg_in_value, double *K, double *k, double *Vx, double *Vxx,
                             double *dV, int32_t *ok, double *kernel_ms) {
  if (!(fx && fu && lx && lu && lxx && luu && lux && VxN && VxxN)) return sfail(-1, "null input stack");
  cddp_hip_stack_handle *h = nullptr;
  int rc = cddp_hip_stacks_create(device, batch, nx, nu, 0, horizon, &h);
  if (rc) return rc;
  cddp_hip_options opt; cddp_hip_default_options(&opt);
  std::vector<double> regv((size_t)batch, reg);
  rc = cddp_hip_set_stacks(h, fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN);
  if (!rc) rc = cddp_hip_stacks_backward(h, reg_in_value ? CDDP_HIP_STACKS_IPDDP : CDDP_HIP_STACKS_CLDDP, &opt, regv.data(), nullptr, 0, ok);
  if (!rc) rc = cddp_hip_stacks_get_gains(h, K, k, Vx, Vxx, dV);
  if (!rc && kernel_ms) *kernel_ms = cddp_hip_stacks_last_kernel_ms(h);
  cddp_hip_stacks_destroy(h);
  return rc;
}

#ifdef SC_TIMING
int cddp_hip_debug_sc_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sc_times), sizeof(unsigned long long) * (size_t)n);
}
#endif

}  // extern "C"
