// MFMA form of the path-constrained IPDDP sweep (IPDDPSolver::backwardPass, ipddp_solver.cpp:1355-1568) for plants whose
// state dimension fills most of a 16 x 16 f64 matrix-core tile (8 < nx <= 15: the quadrotor shapes nx = 12 / 13, the
// 7-joint arm nx = 14) -- the north_star's "MFMA only if a state/control dimension actually fills an MFMA tile".
//
// ONE wavefront per trajectory.  Every matrix of a step lives in the C/D register layout of v_mfma_f64_16x16x4_f64:
//   lane l = 16 g + c, register v  <->  element (row g + 4 v, column c)            ("D layout", 4 doubles per lane)
// and the products are chained without moving data (identities checked on the device by profiles/ubench/mfma_f64.hip):
//   * a D-layout tile is, register by register, the B operand of k-step v when the four k-steps run over the
//     interleaved index sets {g + 4 v : g = 0..3};
//   * for a symmetric tile (V_xx) the same registers are also the A operand (lane (g, c) needs X[c][g+4v] = X[g+4v][c]);
//   * A_t / B_t loaded in that layout serve as A^T / B^T on the left AND as A / B on the right.
// Per step: T1 = V A; R = [T1 | V_x] (V_x rides in the spare column nx); A^T R -> Q_xx, A^T V_x; B^T R -> Q_ux, B^T V_x;
// T2 = V B; B^T T2 -> Q_uu: 20 MFMAs (K = 16 each), then the nu x nu LDLT on every lane (its own right-hand-side
// column), and the value update K^T Q_ux + Q_ux^T K + K^T Q_uu K as 4 more short products.  LDS carries only the step
// record (A, B, condensed terms; double-buffered prefetch) and three 16 x 16 scratch tiles (column gather, transpose).
//
// Rounding: the matrix core accumulates fused and in its own order, so the results are NOT bitwise those of the
// reference-order kernels (kernels_coop.hpp); they agree to ~1e-13 relative per sweep (tests/test_mfma_sweep.py holds the
// gains to 1e-8 against the oracle and reports the decision-flip rate of whole solves next to the libm-noise yardstick).
// The condensed Q_uu enters the value update through its transpose (it is symmetric up to that rounding).
#pragma once
#include "kernels_coop.hpp"

namespace cddp_dev {

#define GI(t, E, e) (((((size_t)(t)) * (size_t)d.NB + (size_t)(b >> 6)) * (E) + (e)) * 64 + (size_t)(b & 63))

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

struct Tile {          // 16 x 16 f64 in D layout
  double r[4];
};

// acc = L * R with L given as A-operand registers, R as B-operand registers, over k-steps [0, nk)
template <int NK = 4>
DEV Tile tmul(const Tile &Lop, const Tile &Rop) {
  mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int v = 0; v < NK; ++v) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop.r[v], Rop.r[v], acc, 0, 0, 0);
  Tile o;
#pragma unroll
  for (int v = 0; v < 4; ++v) o.r[v] = acc[v];
  return o;
}

DEV double sel4(double a0, double a1, double a2, double a3, int g) { return g == 0 ? a0 : (g == 1 ? a1 : (g == 2 ? a2 : a3)); }

// two independent products issued alternately: the matrix core's 64-cycle dependent latency of one chain is covered by the other
template <int NK = 4>
DEV void tmul2(const Tile &L1, const Tile &R1, Tile &O1, const Tile &L2, const Tile &R2, Tile &O2) {
  mfma_d4 a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int v = 0; v < NK; ++v) {
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(L1.r[v], R1.r[v], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(L2.r[v], R2.r[v], a2, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) { O1.r[v] = a1[v]; O2.r[v] = a2[v]; }
}
template <int NK = 4>
DEV void tmul3(const Tile &L1, const Tile &R1, Tile &O1, const Tile &L2, const Tile &R2, Tile &O2, const Tile &L3, const Tile &R3, Tile &O3) {
  mfma_d4 a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0}, a3 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int v = 0; v < NK; ++v) {
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(L1.r[v], R1.r[v], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(L2.r[v], R2.r[v], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(L3.r[v], R3.r[v], a3, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) { O1.r[v] = a1[v]; O2.r[v] = a2[v]; O3.r[v] = a3[v]; }
}

// The step record [A_t | B_t | cst_t] is staged into ZERO-PADDED 16 x 16 tiles in LDS, so every operand is read in D layout
// with one ds_read per register and no masks: products of zero-padded operands are zero outside the valid blocks.
//   TA  A_t                                   TB  B_t (columns < nu)
//   TW  G_u^T YS^-1 G_u (nu x nu)             TX  column CV: c_x          [HAS_X: TXW = G_x^T YS^-1 G_x | column CV: G_x^T S^-1 rhat]
//   TU  column CV: c_u  [HAS_X: columns < nx: G_u^T YS^-1 G_x]            TS  column CV: G_u^T S^-1 rhat
template <class Model, class Cons>
struct MfmaCfg {
  static constexpr int NX = Model::NX, NU = Model::NU;
  typedef CstLayout<Model, Cons> L;
  static constexpr int CST = L::SIZE;
  static constexpr int oB = NX * NX, oC = oB + NX * NU, REC = oC + CST;                  // record element ranges: A | B | cst
  static constexpr int NLD = (REC + 63) / 64;                                            // global loads per lane per step
  static constexpr int CV = NX;                                                          // the spare column that carries V_x
  static constexpr int TA = 0, TB = 256, TW = 512, TX = 768, TU = 1024, TS = 1280, TXW = 1536;
  static constexpr int NT = Cons::HAS_X ? 7 : 6;
  static constexpr int oSC = NT * 256;                                                   // ipr, icomp (+ a dump slot for clamped slices)
  static constexpr int oS1 = oSC + 16, oS2 = oS1 + 256, oS3 = oS2 + 256, LDSD = oS3 + 256;
  static constexpr int NKU = (NU + 3) / 4;                                               // k-steps of a product over the control index
  static_assert(NX > 8 && NX <= 15 && NU <= 8, "MFMA sweep: 8 < nx <= 15 (one spare column), nu <= 8");
  // LDS position of record element e
  static DEV int pos(int e) {
    if (e < oB) return TA + (e / NX) * 16 + e % NX;
    if (e < oC) { const int f = e - oB; return TB + (f / NU) * 16 + f % NU; }
    const int f = e - oC;
    if (f < L::CU) return TX + (f - L::CX) * 16 + CV;
    if (f < L::WQYU) return TU + (f - L::CU) * 16 + CV;
    if (f < L::QYUSIR) { const int h = f - L::WQYU; return TW + (h / NU) * 16 + h % NU; }
    if (f < L::IPR) return TS + (f - L::QYUSIR) * 16 + CV;
    if (f == L::IPR) return oSC;
    if (f == L::ICOMP) return oSC + 1;
    if constexpr (Cons::HAS_X) {
      if (f < L::QYXSIR) { const int h = f - L::WQYX; return TU + (h / NX) * 16 + h % NX; }
      if (f < L::WXQYX) return TXW + (f - L::QYXSIR) * 16 + CV;
      const int h = f - L::WXQYX; return TXW + (h / NX) * 16 + h % NX;
    }
    return oSC + 2;
  }
};

template <class Model, class Cons>
__global__ __launch_bounds__(64) void k_backward_ipddp_mfma(DevBuf d, const ProblemDev *__restrict__ Pk, const double *__restrict__ xrt,
                                                            int force, int count_iter) {
  constexpr int NX = Model::NX, NU = Model::NU;
  typedef Objective<NX, NU> Obj;
  typedef MfmaCfg<Model, Cons> C;
  typedef typename C::L L;
  constexpr int CV = C::CV;
  __shared__ double lds[C::LDSD];
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  // XCD-aware mapping: consecutive workgroups go to consecutive XCDs, and the 64 trajectories of a tile share the 512-B
  // rows of every stack, so a tile's wavefronts are kept on ONE XCD (one L2): block i -> XCD i % 8, tile = 8 (i / 512) + xcd
  const int bi = blockIdx.x, xcd = bi & 7, bj = bi >> 3;
  const int b = (((bj >> 6) * 8 + xcd) << 6) + (bj & 63);
  if (b >= d.B) return;
  if (!force && d.phase[b] != PH_ACTIVE) return;
  const ProblemDev *__restrict__ P = Pk;
  const cddp_hip_options &o = P->opt;
  const int N = d.N;
  const int cur = d.cur[b];
  const double *Xc = d.X + (size_t)cur * d.planeX;
  if (count_iter && lane == 0) d.iter[b] += 1;
  double reg = d.reg[b];
  const double mu = d.mu[b];
  bool ok = false;
  int nb = 0;
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  double *S1 = lds + C::oS1, *S2 = lds + C::oS2, *S3 = lds + C::oS3;
  double *sink = d.sink + ((size_t)(b & 1023) << 6) + lane;
  const int tpos = g * 16 + c;                   // D-layout position of register 0 in a tile (register v: + 64 v)
  // loop-invariant tiles: 2 Q dt, 2 R dt, reg on the control diagonal
  Tile Q2, R2;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = g + 4 * v;
    Q2.r[v] = (r < NX && c < NX) ? 2.0 * P->pool[P->off_Qdt + r * NX + c] : 0.0;
    R2.r[v] = (r < NU && c < NU) ? 2.0 * P->pool[P->off_Rdt + r * NU + c] : 0.0;
  }
  // ---- step record: slice j of this lane is record element lane + 64 j; its source offset and LDS position are fixed
  struct RecIn { double e[C::NLD]; };
  int lpos[C::NLD];
#pragma unroll
  for (int j = 0; j < C::NLD; ++j) { const int e = lane + 64 * j; lpos[j] = e < C::REC ? C::pos(e) : C::oSC + 2; }
  auto load_rec = [&](int tt, RecIn &r) {
#pragma unroll
    for (int j = 0; j < C::NLD; ++j) {
      int e = lane + 64 * j;
      if (e >= C::REC) e = C::REC - 1;
      const double *src = e < C::oB ? d.A + GI(tt, NX * NX, e) : (e < C::oC ? d.Bm + GI(tt, NX * NU, e - C::oB) : d.cst + GI(tt, C::CST, e - C::oC));
      r.e[j] = *src;
    }
  };
  auto store_rec = [&](const RecIn &r) {
#pragma unroll
    for (int j = 0; j < C::NLD; ++j) lds[lpos[j]] = r.e[j];
  };
  auto rd = [&](int tile) { Tile t_; 
#pragma unroll
    for (int v = 0; v < 4; ++v) t_.r[v] = lds[tile + tpos + 64 * v];
    return t_; };
  // zero the padded tiles once
#pragma unroll
  for (int i = 0; i < (C::oSC + 16 + 63) / 64; ++i) { const int e = lane + 64 * i; if (e < C::oSC + 16) lds[e] = 0.0; }
  lds_sync();

  for (;;) {
    ++nb;
    // terminal value: V_xx = sym(2 Q_f), V_x = 2 Q_f (x_N - x_ref)
    Tile Vt, vx;
    {
      double xN[NX], Vx0[NX];
      ld<NX>(Xc + GI(N, NX, 0), kLS, xN);
      Obj::final_grad(P, xN, Vx0);
      const double *Qf = P->pool + P->off_Qf;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = g + 4 * v;
        Vt.r[v] = (r < NX && c < NX) ? 0.5 * ((2.0 * Qf[r * NX + c]) + (2.0 * Qf[c * NX + r])) : 0.0;
        double xv = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) xv = (i == r) ? Vx0[i] : xv;
        vx.r[v] = (c == CV) ? xv : 0.0;
        *((r < NX && c < NX) ? d.Vxx + GI(N, NX * NX, (r < NX ? r : 0) * NX + (c < NX ? c : 0)) : sink) = Vt.r[v];
        *((r < NX && c == CV) ? d.Vx + GI(N, NX, r < NX ? r : 0) : sink) = xv;
      }
    }
    dV0 = 0; dV1 = 0; inf_du = 0; inf_pr = 0; inf_comp = 0; step_norm = 0;
    bool fail = false;
    RecIn rn;
    load_rec(N - 1, rn);
    store_rec(rn);
    lds_sync();
    for (int t = N - 1; t >= 0; --t) {
      // every operand of this step out of the (single) record buffer, then the prefetched record of step t - 1 goes in
      const Tile At = rd(C::TA), Bt = rd(C::TB), Wt = rd(C::TW), Xt = rd(C::TX), Ut = rd(C::TU), St = rd(C::TS);
      Tile XWt;
      if constexpr (Cons::HAS_X) XWt = rd(C::TXW);
      const double ipr_t = lds[C::oSC], icomp_t = lds[C::oSC + 1];
      lds_sync();
      load_rec(t > 0 ? t - 1 : 0, rn);            // HBM latency of the next record under this step's work; it is stored into
      PIPELINE_FENCE();                            // the (single) record buffer at the bottom of the step, all tile reads done
      Tile T1, T2, QA, QB, QU;
      tmul2(Vt, At, T1, Vt, Bt, T2);            // V A, V B
      Tile R;
#pragma unroll
      for (int v = 0; v < 4; ++v) R.r[v] = (c == CV) ? vx.r[v] : T1.r[v];
      tmul3(At, R, QA, Bt, R, QB, Bt, T2, QU);  // A^T [V A | V_x], B^T [V A | V_x], B^T V B
      Tile quu, quc, qb, qxx;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        quu.r[v] = R2.r[v] + QU.r[v];                                     // Q_uu = l_uu + B^T V B
        quc.r[v] = quu.r[v] + Wt.r[v];                                    // condensed Q_uu (:1492)
        qb.r[v] = (Ut.r[v] + QB.r[v]) + St.r[v];                          // [ Q_ux (+ G_u^T YS^-1 G_x) | (c_u + B^T V_x) + G_u^T S^-1 rhat ]
        double qx = (Q2.r[v] + Xt.r[v]) + QA.r[v];                        // [ Q_xx | c_x + A^T V_x ]
        if constexpr (Cons::HAS_X) qx += XWt.r[v];
        qxx.r[v] = qx;
      }
      // ---- sym(Q_uu) + W + reg I in D layout (transpose through LDS), then every lane gathers the nu x nu matrix and
      // its own right-hand-side column
#pragma unroll
      for (int v = 0; v < 4; ++v) { S1[tpos + 64 * v] = quu.r[v]; S2[tpos + 64 * v] = qb.r[v]; }
      lds_sync();
      {
        Tile qr;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = g + 4 * v;
          const double qt = S1[c * 16 + r];
          qr.r[v] = (0.5 * (quu.r[v] + qt) + Wt.r[v]) + ((r == c && r < NU) ? reg : 0.0);   // (:1424-1426)
        }
        lds_sync();
#pragma unroll
        for (int v = 0; v < 4; ++v) S1[tpos + 64 * v] = qr.r[v];
        lds_sync();
      }
      double Qr[NU * NU], rhs[NU];
      const int col = c <= CV ? c : CV;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
#pragma unroll
        for (int j = 0; j < NU; ++j) Qr[i * NU + j] = S1[i * 16 + j];
        rhs[i] = S2[i * 16 + col];
      }
      double sol[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) sol[i] = rhs[i];
      if (NU == 1) sol[0] = ldlt1_solve(Qr[0], sol[0]);
      else {
        LDLTs<NU> f;
        f.compute(Qr, NU);
        if (!f.ok) { fail = true; break; }      // wave-uniform: every lane factors the same matrix
        f.solve(sol);
      }
      double Kc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) Kc[i] = (i < NU) ? -sol[i < NU ? i : 0] : 0.0;   // column `col` of [K | k]
      Tile Kt;                                    // [K | k] in D layout (rows u)
      Kt.r[0] = sel4(Kc[0], Kc[1], Kc[2], Kc[3], g);
      Kt.r[1] = sel4(Kc[4], Kc[5], Kc[6], Kc[7], g);
      Kt.r[2] = 0.0; Kt.r[3] = 0.0;
      if (c > CV) { Kt.r[0] = 0.0; Kt.r[1] = 0.0; }
      // ---- value update: Vn = [Q_xx | Q_x] + K^T [Q_ux | Q_u] + [Q_ux]^T [K | k] + K^T Q_uu [K | k]   (:1497-1500)
      Tile P1, P2, Mt, P3;
      tmul3<C::NKU>(Kt, qb, P1, qb, Kt, P2, quc, Kt, Mt);   // (Mt: (condensed Q_uu)^T [K | k])
      P3 = tmul<C::NKU>(Kt, Mt);
      Tile Vn;
#pragma unroll
      for (int v = 0; v < 4; ++v) Vn.r[v] = ((qxx.r[v] + P1.r[v]) + P2.r[v]) + P3.r[v];
      // symmetrise through LDS: V[r][c] = (Vn[r][c] + Vn[c][r]) / 2; column CV of Mt = Q_uu k goes along for dV1
#pragma unroll
      for (int v = 0; v < 4; ++v) { S3[tpos + 64 * v] = Vn.r[v]; S2[tpos + 64 * v] = Mt.r[v]; }
      lds_sync();
      {   // scalars (meaningful on the lanes of column CV, which hold k and the condensed Q_u)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          s0 += Kc[i] * rhs[i]; s1 += Kc[i] * S2[i * 16 + CV];
          inf_du = dmax(inf_du, fabs(rhs[i])); step_norm = dmax(step_norm, fabs(Kc[i]));
        }
        dV0 += s0; dV1 += 0.5 * s1;
        inf_pr = dmax(inf_pr, ipr_t); inf_comp = dmax(inf_comp, icomp_t);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = g + 4 * v;
        const double vt = S3[c * 16 + r];
        const bool in = (r < NX && c < NX);
        Vt.r[v] = in ? 0.5 * (Vn.r[v] + vt) : 0.0;
        vx.r[v] = (r < NX && c == CV) ? Vn.r[v] : 0.0;
        // every lane stores every time (lanes without a destination write to the sink): stores inside divergent branches
        // make the waitcnt pass wait for a store acknowledge every step (see kernels_lean.hpp)
        const int rx = r < NX ? r : 0, cx2 = c < NX ? c : 0, ru = r < NU ? r : 0;
        *(in ? d.Vxx + GI(t, NX * NX, rx * NX + cx2) : sink) = Vt.r[v];
        *((r < NX && c == CV) ? d.Vx + GI(t, NX, rx) : sink) = vx.r[v];
        if (v < C::NKU) {
          *((r < NU && c < NX) ? d.K + GI(t, NU * NX, ru * NX + cx2) : sink) = Kt.r[v];
          *((r < NU && c == CV) ? d.k + GI(t, NU, ru) : sink) = Kt.r[v];
        }
      }
      store_rec(rn);                              // record of step t - 1
      lds_sync();
    }
    if (!fail) { ok = true; break; }
    if (force == 2) break;
    reg = reg_increase(o, reg);
    if (reg >= o.reg_max_value) break;
  }
  // the scalars live on the lanes of column CV; lane CV (g = 0) publishes them through LDS for the whole wavefront
  if (lane == CV) { S1[0] = dV0; S1[1] = dV1; S1[2] = inf_du; S1[3] = step_norm; }
  lds_sync();
  dV0 = S1[0]; dV1 = S1[1]; inf_du = S1[2]; step_norm = S1[3];
  bool conv = false;
  if (ok) {
    const double tol = dmax(o.tolerance, o.ipddp_barrier_tol_mult * mu);
    const double asn = fabs(d.alpha_pr[b]) * step_norm;
    const double sdu_early = scaled_inf_du_v<Model, Cons>(d, b, cur, inf_du);   // computeScaledDualInfeasibility (:931)
    conv = (inf_pr < tol && sdu_early < tol && inf_comp < tol && asn < o.tolerance * 10.0);
  }
  if (lane != 0) return;
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

// rolloutLinearPolicy from dx0 = 0 (ipddp_solver.cpp:1511-1520) for the trajectories the MFMA sweep just finished (phase
// PH_FWD1, or bwd_ok under `force`): the dX stack K3 (k_post) reads.  One wavefront per trajectory, same block map; lane
// c < nx computes row c of dx_{t+1}; the step's K, k, A, B are staged in LDS cooperatively, prefetched one step ahead.
template <class Model>
__global__ __launch_bounds__(64) void k_dx_rollout_wave(DevBuf d, int force) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr int oK = 0, ok_ = oK + NU * NX, oAr = ok_ + NU, oBr = oAr + NX * NX, RREC = oBr + NX * NU, RLD = (RREC + 63) / 64, RP = RLD * 64;
  __shared__ double lds[2 * RP + 64];
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int bi = blockIdx.x, xcd = bi & 7, bj = bi >> 3;
  const int b = (((bj >> 6) * 8 + xcd) << 6) + (bj & 63);
  if (b >= d.B) return;
  if (force ? !d.bwd_ok[b] : d.phase[b] != PH_FWD1) return;
  const int N = d.N;
  double *S3 = lds + 2 * RP;
  double *sink = d.sink + ((size_t)(b & 1023) << 6) + lane;
  struct RIn { double e[RLD]; };
  auto load_r = [&](int tt, RIn &r) {
#pragma unroll
    for (int j = 0; j < RLD; ++j) {
      int e = lane + 64 * j;
      if (e >= RREC) e = RREC - 1;
      const double *src = e < ok_ ? d.K + GI(tt, NU * NX, e) : (e < oAr ? d.k + GI(tt, NU, e - ok_) : (e < oBr ? d.A + GI(tt, NX * NX, e - oAr) : d.Bm + GI(tt, NX * NU, e - oBr)));
      r.e[j] = *src;
    }
  };
  auto store_r = [&](int buf, const RIn &r) {
#pragma unroll
    for (int j = 0; j < RLD; ++j) lds[buf * RP + lane + 64 * j] = r.e[j];
  };
  double dx[NX], mine = 0.0;                 // mine = this lane's own row of dx_t
#pragma unroll
  for (int i = 0; i < NX; ++i) dx[i] = 0.0;
  const int qc = c < NX ? c : NX - 1;
  RIn rr;
  load_r(0, rr);
  store_r(0, rr);
  lds_sync();
  for (int t = 0; t < N; ++t) {
    const int tn = t + 1 < N - 1 ? t + 1 : t;
    load_r(tn, rr);
    PIPELINE_FENCE();
    *((g == 0 && c < NX) ? d.dX + GI(t, NX, qc) : sink) = mine;
    if (t < N - 1) {
      const double *Lg = lds + (t & 1) * RP;
      double du[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) { double a = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) a += Lg[oK + i * NX + j] * dx[j];
        du[i] = Lg[ok_ + i] + a; }
      double a = 0.0, cc = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) a += Lg[oAr + qc * NX + j] * dx[j];
#pragma unroll
      for (int j = 0; j < NU; ++j) cc += Lg[oBr + qc * NU + j] * du[j];
      const double dxq = (a + cc) + 0.0;
      mine = dxq;
      S3[lane] = dxq;
      store_r((t & 1) ^ 1, rr);
      lds_sync();
#pragma unroll
      for (int i = 0; i < NX; ++i) dx[i] = S3[i];
      lds_sync();
    }
  }
}

#undef GI
}  // namespace cddp_dev
