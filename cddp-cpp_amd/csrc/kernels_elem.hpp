// Element-ownership IPDDP Riccati sweep for small plants (nx <= 4, nu <= 2): 16 lanes per trajectory, lane (i, j) owns V_xx[i][j].
//
// The column-ownership sweep (kernels_coop.hpp, G = 4 at nx <= 4) runs 256 wavefronts for the 4096-trajectory C2 batch --
// one SIMD in four -- and each of them issues ~340 f64-rate instructions per step (a wave64 f64 operation occupies the SIMD for
// 4-5 cycles whatever the instruction-level parallelism): 126 us per launch (profiles/r02_kernel_stats_cartpole_ipddp.md).
// Here a trajectory is spread over a 4 x 4 lane grid, lane l = 4 j + i: the four lanes of a QUAD hold one column of V_xx.  1024
// wavefronts (every SIMD gets one), each lane carries ONE element of every nx x nx product:
//
//   column j of V_xx          quad broadcasts (DPP quad_perm, no LDS)  -> T1[i][j] = (A^T V_xx)[i][j], T2[u][j] = (B^T V_xx)[u][j]
//   LDS round 1               rows T1[i, :], T2[u, :]                  -> Q_xx[i][j], Q_ux[u][j], Q_uu[u][v]
//   (nu = 2: LDS round 1b     Q_uu entries, Q_ux[:, j])                -> factor, k (replicated), K[:, j]
//   LDS round 2               K[:, i], condensed Q_ux[:, i]            -> Vn[i][j], V_x[j]
//   LDS round 3               Vn[j][i], V_x[:]                         -> V_xx[i][j] = (Vn[i][j] + Vn[j][i]) / 2
//
// -- the three LDS rounds of the column form, a third of its arithmetic per lane.
// MEASURED (MI355X, C2, B = 4096; profiles/r03_element_sweep.md): NOT faster.  Exchanges through ds_bpermute: 206 us per launch;
// LDS rounds with per-lane global loads: 195 us (16 rows touched per load instruction); LDS rounds with the cooperative fetch
// below: 165 us -- against 126 us for the column form.  With one wavefront per SIMD an LDS round trip costs ~300 cycles and this
// form has five of them per step (input record, three exchanges, the parked prefetch) where the column form has three, so the
// shorter instruction stream does not pay.  Kept opt-in (CDDP_HIP_SWEEP=elem) with its bitwise test; the column form
// (kernels_coop.hpp, now with quad-broadcast exchanges) stays the default.
// Every output element is accumulated by ONE lane with the sums and the association of k_backward_ipddp_lean /
// k_backward_ipddp_coop (ipddp_solver.cpp:1392-1508), so the sweep stays bit-identical to both
// (tests/test_gpu_parity.py::test_element_sweep_agrees_bitwise).  Lanes with i >= nx or j >= nx shadow row / column nx - 1 (same
// values, global stores predicated off).  The 16 lanes of a trajectory take every branch together (all decisions are computed
// redundantly from replicated values); the wavefront is its own workgroup, so an LDS round needs lgkmcnt(0), no barrier.
#pragma once
#include "kernels_coop.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

// a[idx] for a lane-varying idx < N without a dynamically indexed register array (value selects)
template <int N> DEV double pick(const double *a, int idx) {
  double v = a[0];
#pragma unroll
  for (int k = 1; k < N; ++k) v = (idx == k) ? a[k] : v;
  return v;
}

template <int TOT>
struct ElemCfg {
  // LDS doubles per trajectory: T1 4x4 | T2 2x4 | K 2x4 | condensed Q_ux 2x4 | Vn 4x4 | V_x 4 | Q_uu 2x2 | two input records
  static constexpr int oT1 = 0, oT2 = 16, oK = 24, oQ = 32, oVn = 40, oVx = 56, oQuu = 60, oIn = 64, RAW = oIn + 2 * TOT;
  static constexpr int STRIDE = (RAW + 1) / 2 * 2 + 2;   // even, consecutive trajectories start 4 banks apart mod 8-bank rows
  static constexpr int TPW = 4;
  static constexpr int NL = (TOT + 15) / 16;              // cooperative fetch: doubles per lane and step
};

template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_ipddp_elem(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                            int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  static_assert(NX <= 4 && NU <= 2, "element-ownership sweep: 4 x 4 lane grid, nu <= 2");
  typedef Objective<NX, NU> Obj;
  typedef CstLayout<Model, Cons> L;
  constexpr int CST = L::SIZE, EA = NX * NX, EB = NX * NU, TOT = EA + EB + CST;   // the step record: A_t | B_t | condensed terms
  typedef ElemCfg<TOT> C;
  constexpr int TPW = C::TPW;
  __shared__ double lds[TPW * C::STRIDE];
  const int lane = threadIdx.x;
  const int i = lane & 3, j = (lane >> 2) & 3;     // lane l = 4 j + i of its group: quad = column j, position in the quad = row i
  const int ic = i < NX ? i : NX - 1, jc = j < NX ? j : NX - 1;
  const int ui = i < NU ? i : NU - 1, vj = j < NU ? j : NU - 1;
  const bool own_elem = i < NX && j < NX;          // this lane stores V_xx[i][j]
  const bool own_col = i == 0 && j < NX;           // ... V_x[j], writes column j's K / Q_ux to LDS
  const bool own_gain = i < NU && j < NX;          // ... K[i][j]
  const bool lead = (lane & 15) == 0;
  const int b = coop_group<TPW>((int)blockIdx.x, d.xcd_map) * TPW + (lane >> 4);
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  double *Ls = lds + (lane >> 4) * C::STRIDE;
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
    // The step record (A_t, B_t, condensed terms: TOT doubles per trajectory) is fetched COOPERATIVELY -- lane q of the group
    // takes elements q, q + 16, ... -- one step ahead, parked in registers during the step and dropped into the other LDS input
    // buffer at its end; every lane then reads the entries it needs from LDS.  (Per-lane global loads of "its" entries touched
    // 16 different 512-B rows per instruction and four times the instructions of the column form: the first LDS version of this
    // kernel spent its time in the vector memory pipeline, 195 us per launch.)
    struct Raw { double v[C::NL]; };
    const int q16 = lane & 15;
    auto fetch = [&](int tt, Raw &r) {
      const double *Ab = d.A + GI(tt, EA, 0), *Bb = d.Bm + GI(tt, EB, 0), *Cb = d.cst + GI(tt, CST, 0);
#pragma unroll
      for (int jj = 0; jj < C::NL; ++jj) {
        const int e = q16 + 16 * jj;
        const int ec = e - EA - EB;
        const double *pa = Ab + (size_t)(e < EA ? e : EA - 1) * kLS;
        const double *pb = Bb + (size_t)(e - EA < EB ? (e - EA > 0 ? e - EA : 0) : EB - 1) * kLS;
        const double *pc = Cb + (size_t)(ec < CST ? (ec > 0 ? ec : 0) : CST - 1) * kLS;
        const double *p = e < EA ? pa : (e < EA + EB ? pb : pc);
        r.v[jj] = *p;
      }
    };
    auto park = [&](int buf, const Raw &r) {
      double *Li = Ls + C::oIn + buf * TOT;
#pragma unroll
      for (int jj = 0; jj < C::NL; ++jj) { const int e = q16 + 16 * jj; if (e < TOT) Li[e] = r.v[jj]; }
    };
    auto take = [&](int buf, In &r) {   // this lane's view of the step record
      const double *Li = Ls + C::oIn + buf * TOT;
      const double *c = Li + EA + EB;
#pragma unroll
      for (int k = 0; k < NX; ++k) { r.Aci[k] = Li[k * NX + ic]; r.Acj[k] = Li[k * NX + jc]; }
#pragma unroll
      for (int e = 0; e < EB; ++e) r.Bf[e] = Li[EA + e];
      r.cxj = c[L::CX + jc];
#pragma unroll
      for (int e = 0; e < NU; ++e) { r.cu[e] = c[L::CU + e]; r.QyuSir[e] = c[L::QYUSIR + e]; }
#pragma unroll
      for (int e = 0; e < NU * NU; ++e) r.WQyu[e] = c[L::WQYU + e];
      r.ipr = c[L::IPR]; r.icomp = c[L::ICOMP];
      if constexpr (Cons::HAS_X) {
        r.wqyx = c[L::WQYX + ui * NX + jc];
        r.qyxsir = c[L::QYXSIR + jc];
        r.wxqyx = c[L::WXQYX + ic * NX + jc];
      }
    };
    auto step = [&](const int t, Raw &nxt) -> bool {
      const int tp = t > 0 ? t - 1 : 0;   // unconditional (clamped) prefetch
      fetch(tp, nxt);
      PIPELINE_FENCE();
      In c;
      take(t & 1, c);
      // columns ui / vj of B by value select (lane-varying, nu <= 2)
      double Bu[NX], Bv[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) { Bu[k] = pick<NU>(c.Bf + k * NU, ui); Bv[k] = pick<NU>(c.Bf + k * NU, vj); }
      // ---- column jc of V_xx (the lanes of this quad); T1[ic][jc], T2[ui][jc], Q_x[jc], Q_u (replicated)
      double Vcol[NX];
      quad_gather<NX>(V, Vcol);
      double T1, T2;
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += c.Aci[k] * Vcol[k];
        T1 = s; }
      { double s = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s += Bu[k] * Vcol[k];
        T2 = s; }
      Ls[C::oT1 + i * 4 + j] = T1;
      if (i < NU) Ls[C::oT2 + i * 4 + j] = T2;
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
      lds_sync();
      // ---- LDS round 1: rows ic of T1 and ui of T2; Q_xx[ic][jc], Q_ux[ui][jc], Q_uu[ui][vj]
      double T1r[NX], T2r[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) { T1r[k] = Ls[C::oT1 + i * 4 + k]; T2r[k] = Ls[C::oT2 + ui * 4 + k]; }
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
      // condensed Q_ux entry: the right-hand side of the K solve IS the condensed value (:1453, :1491)
      double quxq = Qux;
      if constexpr (Cons::HAS_X) quxq = quxq + c.wqyx;
      double Quu[NU * NU], Qc[NU];     // Q_uu (replicated), condensed Q_ux[:, jc]
      if constexpr (NU == 1) { Quu[0] = Quu_e; Qc[0] = quxq; }
      else {   // LDS round 1b: the other entries of Q_uu and of column jc of Q_ux
        if (i < NU && j < NU) Ls[C::oQuu + i * NU + j] = Quu_e;
        if (i < NU) Ls[C::oQ + i * 4 + j] = quxq;
        lds_sync();
#pragma unroll
        for (int e = 0; e < NU * NU; ++e) Quu[e] = Ls[C::oQuu + e];
#pragma unroll
        for (int u = 0; u < NU; ++u) Qc[u] = Ls[C::oQ + u * 4 + j];
      }
      double Qr[NU * NU];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int e = 0; e < NU; ++e) Qr[a * NU + e] = 0.5 * (Quu[a * NU + e] + Quu[e * NU + a]) + c.WQyu[a * NU + e];
#pragma unroll
      for (int a = 0; a < NU; ++a) Qr[a * NU + a] += reg;
      double kk[NU], Kc[NU];     // k (replicated), K[:, jc]
      if constexpr (NU == 1) {
        kk[0] = -ldlt1_solve(Qr[0], Qu[0] + c.QyuSir[0]);
        Kc[0] = -ldlt1_solve(Qr[0], Qc[0]);
      } else {
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
      // ---- LDS round 2: column jc of K (and, nu = 1, of the condensed Q_ux) published by the quad's first lane
      if (i == 0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) Ls[C::oK + u * 4 + j] = Kc[u];
        if constexpr (NU == 1) Ls[C::oQ + j] = Qc[0];
      }
      if (lead) st<NU>(d.k + GI(t, NU, 0), kLS, kk);
      if (own_gain) d.K[GI(t, NU * NX, i * NX + j)] = pick<NU>(Kc, ui);
      // condensed, un-regularised blocks (ipddp_solver.cpp:1488-1492); dV
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
      double KtQj[NU];   // row jc of K^T Q_uu (mm_tn's expression)
#pragma unroll
      for (int e = 0; e < NU; ++e) { double s2 = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) s2 += Kc[u] * Quu[u * NU + e];
        KtQj[e] = s2; }
      double Vxq;
      {
        double a = 0.0, bb = 0.0, cc = 0.0;
#pragma unroll
        for (int e = 0; e < NU; ++e) { a += Kc[e] * Qu[e]; bb += Qc[e] * kk[e]; }
#pragma unroll
        for (int e = 0; e < NU; ++e) cc += KtQj[e] * kk[e];
        Vxq = ((Qxj + a) + bb) + cc;
      }
      lds_sync();
      double Ki[NU], Qi[NU];   // column ic of K and of the condensed Q_ux
#pragma unroll
      for (int u = 0; u < NU; ++u) { Ki[u] = Ls[C::oK + u * 4 + ic]; Qi[u] = Ls[C::oQ + u * 4 + ic]; }
      double KtQi[NU];   // row ic of K^T Q_uu
#pragma unroll
      for (int e = 0; e < NU; ++e) { double s = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) s += Ki[u] * Quu[u * NU + e];
        KtQi[e] = s; }
      double Vn;
      {
        double a = 0.0, bb = 0.0, e2 = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) { a += Ki[u] * Qc[u]; bb += Qi[u] * Kc[u]; e2 += KtQi[u] * Kc[u]; }
        Vn = ((Qxx + a) + bb) + e2;
      }
      // ---- LDS round 3: Vn[jc][ic] for the symmetrisation, the whole V_x; the next step's record goes to the other input buffer
      Ls[C::oVn + i * 4 + j] = Vn;
      if (i == 0) Ls[C::oVx + j] = Vxq;
      park((t & 1) ^ 1, nxt);
      lds_sync();
      V = 0.5 * (Vn + Ls[C::oVn + j * 4 + i]);
#pragma unroll
      for (int k = 0; k < NX; ++k) Vx[k] = Ls[C::oVx + k];
      if (own_col) d.Vx[GI(t, NX, j)] = Vxq;
      if (own_elem) d.Vxx[GI(t, NX * NX, i * NX + j)] = V;
#pragma unroll
      for (int a = 0; a < NU; ++a) { inf_du = dmax(inf_du, fabs(Qu[a])); step_norm = dmax(step_norm, fabs(kk[a])); }
      return true;
    };
    Raw rw;
    fetch(N - 1, rw);
    park((N - 1) & 1, rw);
    lds_sync();
    for (int t = N - 1; t >= 0; --t)
      if (!step(t, rw)) { fail = true; break; }
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
      // rolloutLinearPolicy, dx0 = 0 (ipddp_solver.cpp:1511-1520): lane (i, j) computes ROW ic of dx_{t+1}, so that the four
      // lanes of a quad hold the whole new dx and a quad broadcast hands it to each of them -- no LDS on this chain
      double dx[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) dx[k] = 0.0;
      double dxq = 0.0;
      struct RIn { double kk[NU], KK[NU * NX], Aq[NX], Bq[NU]; };
      auto load_r = [&](int tt, RIn &r) {
        ld<NU>(d.k + GI(tt, NU, 0), kLS, r.kk);
        ld<NU * NX>(d.K + GI(tt, NU * NX, 0), kLS, r.KK);
#pragma unroll
        for (int k = 0; k < NX; ++k) r.Aq[k] = d.A[GI(tt, NX * NX, ic * NX + k)];
#pragma unroll
        for (int k = 0; k < NU; ++k) r.Bq[k] = d.Bm[GI(tt, NX * NU, ic * NU + k)];
      };
      auto rstep = [&](const int t, const RIn &rc, RIn &rn) {
        const int tn = t + 1 < N - 1 ? t + 1 : t;
        load_r(tn, rn);
        PIPELINE_FENCE();
        if (j == 0 && i < NX) d.dX[GI(t, NX, i)] = dxq;
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
          quad_gather<NX>(dxq, dx);
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
