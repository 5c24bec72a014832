// Element-ownership IPDDP Riccati sweep for small plants (nx <= 4, nu <= 2): 16 lanes per trajectory, lane (i, j) owns V_xx[i][j].
//
// The column-ownership sweep (kernels_coop.hpp, G = 4 at nx <= 4) runs 256 wavefronts for the 4096-trajectory C2 batch --
// one SIMD in four -- and each of them issues ~340 dependent-latency-bound f64 instructions per step (126 us per launch,
// profiles/r02_kernel_stats_cartpole_ipddp.md).  Here a trajectory is spread over a 4 x 4 lane grid: 1024 wavefronts (every SIMD
// gets one), each lane carries ONE element of every nx x nx product, and the exchanges between lanes go through the crossbar
// (ds_bpermute: no LDS memory, no write / wait / read / wait round trip):
//
//   column gather   V_xx[:, j]                 (lanes (k, j))        -> T1[i][j] = (A^T V_xx)[i][j], T2[u][j] = (B^T V_xx)[u][j]
//   row gather      T1[i, :], T2[u, :]         (lanes (i, k))        -> Q_xx[i][j], Q_ux[u][j], Q_uu[u][v]
//   (nu = 2)        Q_uu entries, Q_ux[:, j]                          -> factor, k (replicated), K[:, j]
//   diagonal fetch  K[:, i], Q_ux[:, i]        (lane (i, i))         -> Vn[i][j]
//   transpose       Vn[j][i]                   (lane (j, i))         -> V_xx[i][j] = (Vn[i][j] + Vn[j][i]) / 2
//   row gather      V_x[:]                     (lanes (i, k))
//
// Every output element is accumulated by ONE lane with the sums and the association of k_backward_ipddp_lean /
// k_backward_ipddp_coop (ipddp_solver.cpp:1392-1508), so the sweep stays bit-identical to both
// (tests/test_gpu_parity.py::test_cooperative_and_lane_sweeps_agree_bitwise).  Lanes with i >= nx or j >= nx shadow row / column
// nx - 1 (same values, stores predicated off).  The 16 lanes of a trajectory take every branch together (all decisions are
// computed redundantly from replicated values), so a group never diverges internally.
#pragma once
#include "kernels_coop.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

// the value `v` holds in lane `src` of the wavefront
DEV double lane_get(double v, int src) {
  const int a = src << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// a[idx] for a lane-varying idx < N without a dynamically indexed register array (value selects)
template <int N> DEV double pick(const double *a, int idx) {
  double v = a[0];
#pragma unroll
  for (int k = 1; k < N; ++k) v = (idx == k) ? a[k] : v;
  return v;
}

template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_ipddp_elem(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                            int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  static_assert(NX <= 4 && NU <= 2, "element-ownership sweep: 4 x 4 lane grid, nu <= 2");
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  constexpr int CST = L::SIZE, TPW = 4;
  const int lane = threadIdx.x;
  const int gb = lane & ~15;                       // first lane of this trajectory's group
  const int i = (lane >> 2) & 3, j = lane & 3;
  const int ic = i < NX ? i : NX - 1, jc = j < NX ? j : NX - 1;
  const int ui = i < NU ? i : NU - 1, vj = j < NU ? j : NU - 1;
  const bool own_elem = i < NX && j < NX;          // this lane stores V_xx[i][j]
  const bool own_col = i == 0 && j < NX;           // ... V_x[j]
  const bool own_gain = i < NU && j < NX;          // ... K[i][j]
  const bool lead = (lane & 15) == 0;
  const int b = coop_group<TPW>((int)blockIdx.x, d.xcd_map) * TPW + (lane >> 4);
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  if (count_iter && lead) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  // source lanes of the exchanges
  const int src_row = gb + 4 * i, src_col = gb + j, src_diag = gb + 5 * i, src_tr = gb + 4 * j + i;
  // loop-invariant constants: (Q dt)[ic][jc] per lane, R dt (uniform)
  const double Qe = P->pool[P->off_Qdt + ic * NX + jc];
  double Rr[NU * NU];
  {
    const double *Rp = P->pool + P->off_Rdt;
#pragma unroll
    for (int e = 0; e < NU * NU; ++e) Rr[e] = Rp[e];
  }
  const double Re = pick<NU * NU>(Rr, ui * NU + vj);
  for (;;) {
    ++nb;
    double Vx[NX], V;    // V_x (replicated), V_xx[ic][jc]
    {
      double xN[NX];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      Obj::final_grad(P, xN, Vx);
      const double *Qf = P->pool + P->off_Qf;
      V = 0.5 * ((2.0 * Qf[ic * NX + jc]) + (2.0 * Qf[jc * NX + ic]));
    }
    dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0;
    if (own_col) d.Vx[GI(N, NX, j)] = pick<NX>(Vx, jc);
    if (own_elem) d.Vxx[GI(N, NX * NX, i * NX + j)] = V;
    bool fail = false;
    struct In {
      double Aci[NX], Acj[NX], Bf[NX * NU], cxj, cu[NU], WQyu[NU * NU], QyuSir[NU], ipr, icomp;
      double wqyx, qyxsir, wxqyx;   // HAS_X only
    };
    auto load = [&](int tt, In &r) {
      const double *Ab = d.A + GI(tt, NX * NX, 0);
#pragma unroll
      for (int k = 0; k < NX; ++k) { r.Aci[k] = Ab[(size_t)(k * NX + ic) * kLS]; r.Acj[k] = Ab[(size_t)(k * NX + jc) * kLS]; }
      ld<NX * NU>(d.Bm + GI(tt, NX * NU, 0), kLS, r.Bf);
      const double *c = d.cst + GI(tt, CST, 0);
      r.cxj = c[(size_t)(L::CX + jc) * kLS];
      ld<NU>(c + (size_t)L::CU * kLS, kLS, r.cu);
      ld<NU * NU>(c + (size_t)L::WQYU * kLS, kLS, r.WQyu);
      ld<NU>(c + (size_t)L::QYUSIR * kLS, kLS, r.QyuSir);
      r.ipr = c[(size_t)L::IPR * kLS]; r.icomp = c[(size_t)L::ICOMP * kLS];
      if constexpr (Cons::HAS_X) {
        r.wqyx = c[(size_t)(L::WQYX + ui * NX + jc) * kLS];
        r.qyxsir = c[(size_t)(L::QYXSIR + jc) * kLS];
        r.wxqyx = c[(size_t)(L::WXQYX + ic * NX + jc) * kLS];
      }
    };
    auto step = [&](const int t, const In &c, In &nxt) -> bool {
      const int tp = t > 0 ? t - 1 : 0;   // unconditional (clamped) prefetch
      load(tp, nxt);
      PIPELINE_FENCE();
      // columns ui / vj of B by value select (lane-varying, nu <= 2)
      double Bu[NX], Bv[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) { Bu[k] = pick<NU>(c.Bf + k * NU, ui); Bv[k] = pick<NU>(c.Bf + k * NU, vj); }
      // ---- column jc of V_xx; T1[ic][jc], T2[ui][jc], Q_x[jc], Q_u (replicated)
      double Vcol[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) Vcol[k] = lane_get(V, src_col + 4 * k);
      double T1, T2;
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += c.Aci[k] * Vcol[k];
        T1 = s; }
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bu[k] * Vcol[k];
        T2 = s; }
      double Qxj;
      { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += c.Acj[k] * Vx[k];
        Qxj = c.cxj + s2; }
      double Qu[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) { double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += c.Bf[k * NU + u] * Vx[k];
        Qu[u] = c.cu[u] + s2; }
      // ---- rows ic of T1 and ui of T2; Q_xx[ic][jc], Q_ux[ui][jc], Q_uu[ui][vj]
      double T1r[NX], T2r[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) { T1r[k] = lane_get(T1, src_row + k); T2r[k] = lane_get(T2, src_row + k); }
      double Qxx, Qux, Quu_e;
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += T1r[k] * c.Acj[k];
        Qxx = (2.0 * Qe) + s; }
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += T2r[k] * c.Acj[k];
        Qux = s; }
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += T2r[k] * Bv[k];
        Quu_e = (2.0 * Re) + s; }
      double Quu[NU * NU];
      if constexpr (NU == 1) Quu[0] = Quu_e;
      else {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int v = 0; v < NU; ++v) Quu[u * NU + v] = lane_get(Quu_e, gb + 4 * u + v);
      }
      double Qr[NU * NU];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int e = 0; e < NU; ++e) Qr[a * NU + e] = 0.5 * (Quu[a * NU + e] + Quu[e * NU + a]) + c.WQyu[a * NU + e];
#pragma unroll
      for (int a = 0; a < NU; ++a) Qr[a * NU + a] += reg;
      // condensed Q_ux entry: the right-hand side of the K solve IS the condensed value (:1453, :1491)
      double quxq = Qux;
      if constexpr (Cons::HAS_X) quxq = quxq + c.wqyx;
      double kk[NU], Kc[NU], Qc[NU];     // k (replicated), K[:, jc], condensed Q_ux[:, jc]
      if constexpr (NU == 1) {
        Qc[0] = quxq;
        kk[0] = -ldlt1_solve(Qr[0], Qu[0] + c.QyuSir[0]);
        Kc[0] = -ldlt1_solve(Qr[0], Qc[0]);
      } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) Qc[u] = lane_get(quxq, src_col + 4 * u);
        LDLTs<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) return false;
        double col[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) col[a] = Qu[a] + c.QyuSir[a];
        f.solve(col);
#pragma unroll
        for (int a = 0; a < NU; ++a) kk[a] = -col[a];
#pragma unroll
        for (int a = 0; a < NU; ++a) col[a] = Qc[a];
        f.solve(col);
#pragma unroll
        for (int a = 0; a < NU; ++a) Kc[a] = -col[a];
      }
      if (lead) st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      if (own_gain) d.K[GI(t, NU * NX, i * NX + j)] = pick<NU>(Kc, ui);
      // ---- condensed, un-regularised blocks (ipddp_solver.cpp:1488-1492); dV
#pragma unroll
      for (int a = 0; a < NU; ++a) Qu[a] += c.QyuSir[a];
      if constexpr (Cons::HAS_X) { Qxj += c.qyxsir; Qxx += c.wxqyx; }
#pragma unroll
      for (int e = 0; e < NU * NU; ++e) Quu[e] += c.WQyu[e];
      inf_pr = dmax(inf_pr, c.ipr); inf_comp = dmax(inf_comp, c.icomp);
      double Quuk[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) { double s1 = 0.0;
#pragma unroll
        for (int e = 0; e < NU; ++e) s1 += Quu[a * NU + e] * kk[e];
        Quuk[a] = s1; }
      { double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int a = 0; a < NU; ++a) { s0 += kk[a] * Qu[a]; s1 += kk[a] * Quuk[a]; }
        dV0 += s0; dV1 += 0.5 * s1; }
      // ---- column ic of K and of the condensed Q_ux from the diagonal lane; value update
      double Ki[NU], Qi[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) { Ki[u] = lane_get(Kc[u], src_diag); Qi[u] = lane_get(Qc[u], src_diag); }
      double KtQi[NU], KtQj[NU];   // rows ic and jc of K^T Q_uu (mm_tn's expression)
#pragma unroll
      for (int e = 0; e < NU; ++e) {
        double s = 0.0, s2 = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) { s += Ki[u] * Quu[u * NU + e]; s2 += Kc[u] * Quu[u * NU + e]; }
        KtQi[e] = s; KtQj[e] = s2;
      }
      double Vxq;
      {
        double a = 0.0, bb = 0.0, cc = 0.0;
#pragma unroll
        for (int e = 0; e < NU; ++e) { a += Kc[e] * Qu[e]; bb += Qc[e] * kk[e]; }
#pragma unroll
        for (int e = 0; e < NU; ++e) cc += KtQj[e] * kk[e];
        Vxq = ((Qxj + a) + bb) + cc;
      }
      double Vn;
      {
        double a = 0.0, bb = 0.0, e2 = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) { a += Ki[u] * Qc[u]; bb += Qi[u] * Kc[u]; e2 += KtQi[u] * Kc[u]; }
        Vn = ((Qxx + a) + bb) + e2;
      }
      const double VnT = lane_get(Vn, src_tr);
      V = 0.5 * (Vn + VnT);
#pragma unroll
      for (int k = 0; k < NX; ++k) Vx[k] = lane_get(Vxq, src_row + k);
      if (own_col) d.Vx[GI(t, NX, j)] = Vxq;
      if (own_elem) d.Vxx[GI(t, NX * NX, i * NX + j)] = V;
#pragma unroll
      for (int a = 0; a < NU; ++a) { inf_du = dmax(inf_du, fabs(Qu[a])); step_norm = dmax(step_norm, fabs(kk[a])); }
      return true;
    };
    In ra, rb;
    load(N - 1, ra);
    int t = N - 1;
    for (; t >= 1; t -= 2) {
      if (!step(t, ra, rb)) { fail = true; break; }
      if (!step(t - 1, rb, ra)) { fail = true; break; }
    }
    if (!fail && t == 0) fail = !step(0, ra, rb);
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
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): lane (i, j) computes row jc of dx_{t+1}; the rows of the
      // 4 x 4 grid repeat each other (a row gather hands every lane the whole dx)
      double dx[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) dx[k] = 0.0;
      double dxq = 0.0;
      struct RIn { double kk[NU], KK[NU * NX], Aq[NX], Bq[NU]; };
      auto load_r = [&](int tt, RIn &r) {
        ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
        ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
#pragma unroll
        for (int k = 0; k < NX; ++k) r.Aq[k] = d.A[GI(tt, NX * NX, jc * NX + k)];
#pragma unroll
        for (int k = 0; k < NU; ++k) r.Bq[k] = d.Bm[GI(tt, NX * NU, jc * NU + k)];
      };
      auto rstep = [&](const int t, const RIn &rc, RIn &rn) {
        const int tn = t + 1 < N - 1 ? t + 1 : t;
        load_r(tn, rn);
        PIPELINE_FENCE();
        if (own_col) d.dX[GI(t, NX, j)] = dxq;
        if (t < N - 1) {
          double du[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) { double a = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) a += rc.KK[u * NX + k] * dx[k];
            du[u] = rc.kk[u] + a; }
          double a = 0.0, c2 = 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) a += rc.Aq[k] * dx[k];
#pragma unroll
          for (int k = 0; k < NU; ++k) c2 += rc.Bq[k] * du[k];
          dxq = (a + c2) + 0.0;
#pragma unroll
          for (int k = 0; k < NX; ++k) dx[k] = lane_get(dxq, src_row + k);
        }
      };
      RIn ra, rb;
      load_r(0, ra);
      int t = 0;
      for (; t + 1 < N; t += 2) { rstep(t, ra, rb); rstep(t + 1, rb, ra); }
      if (t < N) rstep(t, ra, rb);
    }
  }
  if (!lead) return;
  d.reg[b] = reg;
  d.n_bwd[b] += nb;
  d.bwd_ok[b] = ok ? 1 : 0;
  d.apr_max[b] = 1.0; d.adu_max[b] = 1.0;        // K3 lowers them by atomic min
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
