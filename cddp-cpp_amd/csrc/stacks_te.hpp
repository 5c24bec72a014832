// Stack-fed terminal-equality sweep (round 6): the reduced-LQR branch of IPDDPSolver::backwardPass (ipddp_solver.cpp:1120-1353) for HOST
// plug-ins.  The caller (plugin_solve.hip) hands over the per-step LQ model the reference builds at :1143-1245 -- Q_t, q_t, R_t (without
// the regularisation), r_t, M_t, A_t, B_t, with the path constraints already condensed into them and Q_N, q_N carrying the terminal
// inequality terms -- through the ordinary stack slots (fx = A, fu = B, lx = q, lu = r, lxx = Q, luu = R, lux = M as nx x nu, VxN = q_N,
// VxxN = Q_N), plus the DENSE terminal Jacobian H_T (p x nx), b_T = -h_T and the previous multipliers.  One lane per trajectory runs
// solveTerminalEqualityLQR (:478-639): the p + 1 sequential LQR sweeps (solveSequentialLQR :413-476) share every matrix quantity and differ
// in the gradient recursion only, so the matrices are carried once and the p + 1 gradient variants side by side (the arrangement of
// kernels.hpp::te_backward, which serves the built-in plants with their selector-shaped H_T); then the closed-loop variant rollouts, the
// p x p regularised normal equations over five regularisation scales, the recombination of k and p, max |r + B^T p_{t+1}|, max |k| and the
// linear-policy rollout dX (:1252-1268).  Every sum in the reference's order.  The pow() of the regularisation floor (:577-578) is
// evaluated by the caller in host arithmetic (te_floor): the plug-in route keeps glibc's elementary functions.
#pragma once

namespace {

constexpr int kPTS = 8;   // terminal-equality rows of a plug-in problem (stack-fed route)

struct StackTeArgs {
  int pT;
  const double *HT;       // [pT][nx][Bp]
  const double *bT;       // [pT][Bp]   = -h_T
  const double *lam_prev; // [pT][Bp]
  const double *floor_;   // [Bp] max(1e-10, reg_scale * pow(max(mu, 0), reg_exponent))
  double *te_p;           // [(pT+1)][(N+1)][nx][Bp]
  double *te_k;           // [(pT+1)][N][nu][Bp]
  double *dlam;           // [pT][Bp]
  double *dX;             // [(N+1)][nx][Bp]
};

template <int NMAXP>
DEV void te_singular_minmax(const double *A, int n, double &smax, double &smin) {   // one-sided Jacobi (kernels.hpp::singular_minmax)
  double U[NMAXP * NMAXP];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) U[i * NMAXP + j] = A[i * NMAXP + j];
  for (int sweep = 0; sweep < 80; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < n; ++i) { alpha += U[i * NMAXP + p] * U[i * NMAXP + p]; beta += U[i * NMAXP + q] * U[i * NMAXP + q]; gamma += U[i * NMAXP + p] * U[i * NMAXP + q]; }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < n; ++i) { double up = U[i * NMAXP + p], uq = U[i * NMAXP + q]; U[i * NMAXP + p] = cs * up - sn * uq; U[i * NMAXP + q] = sn * up + cs * uq; }
      }
    if (!rotated) break;
  }
  smax = 0.0; smin = INFINITY;
  for (int j = 0; j < n; ++j) { double s2 = 0; for (int i = 0; i < n; ++i) s2 += U[i * NMAXP + j] * U[i * NMAXP + j]; double sv = sqrt(s2); smax = dmax(smax, sv); smin = dmin(smin, sv); }
  if (n == 0) { smax = 0.0; smin = 0.0; }
}

// One pass at regularisation `reg`; false where the reference's backwardPass returns false (failed factorisation, non-finite recursion).
template <int NX, int NU, bool T4>   // T4: SI() on tile-minor stacks (the nx = 6 handles, whose default sweep is the cooperative one)
DEV bool te_sweep(const StackArgs &a, const StackTeArgs &e, int b, double reg, double &inf_du, double &step_norm) {
  const int N = a.N, pT = e.pT;
#define TP(v, t, i) e.te_p[((((size_t)(v)) * (N + 1) + (t)) * NX + (i)) * (size_t)a.Bp + (size_t)b]
#define TK(v, t, i) e.te_k[((((size_t)(v)) * N + (t)) * NU + (i)) * (size_t)a.Bp + (size_t)b]
  double HT[kPTS * NX], lam_prev[kPTS];
  for (int r = 0; r < pT; ++r) { lam_prev[r] = e.lam_prev[(size_t)r * a.Bp + b]; for (int i = 0; i < NX; ++i) HT[r * NX + i] = e.HT[((size_t)r * NX + i) * a.Bp + b]; }
  double Pm[NX * NX];
  {
    double QN[NX * NX];
    for (int i = 0; i < NX * NX; ++i) QN[i] = a.VxxN[(size_t)i * a.Bp + b];
    for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) Pm[i * NX + c] = 0.5 * (QN[i * NX + c] + QN[c * NX + i]);   // P[T] = sym(Q[T])
    // q_base[T] = q[T] + H_T^T lambda_prev; variant v > 0 adds row v - 1 of H_T (:509-530)
    for (int v = 0; v <= pT; ++v)
      for (int i = 0; i < NX; ++i) {
        double add = 0.0;
        for (int r = 0; r < pT; ++r) add += HT[r * NX + i] * lam_prev[r];
        double q = a.VxN[(size_t)i * a.Bp + b] + add;
        if (v > 0) q += HT[(v - 1) * NX + i];
        TP(v, N, i) = q;
      }
    for (int i = 0; i < NX * NX; ++i) a.Vxx[SI(N, NX * NX, i)] = Pm[i];
  }
  for (int t = N - 1; t >= 0; --t) {
    double A[NX * NX], Bm[NX * NU], Q[NX * NX], q[NX], R[NU * NU], r[NU], Mm[NX * NU];
    for (int i = 0; i < NX * NX; ++i) { A[i] = a.fx[SI(t, NX * NX, i)]; Q[i] = a.lxx[SI(t, NX * NX, i)]; }
    for (int i = 0; i < NX * NU; ++i) { Bm[i] = a.fu[SI(t, NX * NU, i)]; Mm[i] = a.lux[SI(t, NU * NX, i)]; }
    for (int i = 0; i < NX; ++i) q[i] = a.lx[SI(t, NX, i)];
    for (int i = 0; i < NU; ++i) r[i] = a.lu[SI(t, NU, i)];
    for (int i = 0; i < NU * NU; ++i) R[i] = a.luu[SI(t, NU * NU, i)];
    for (int i = 0; i < NU; ++i) R[i * NU + i] += reg;                        // R[t].diagonal() += regularization (:1247)
    double BtP[NU * NX], Quu[NU * NU], Qux[NU * NX];
    mm_tn<NU, NX, NX>(Bm, Pm, BtP);
    {
      double T1[NU * NU], T2[NU * NU], BtPt[NU * NX];
      mm_nn<NU, NX, NU>(BtP, Bm, T1);
      for (int i = 0; i < NU; ++i) for (int c = 0; c < NX; ++c) { double s = 0.0; for (int k = 0; k < NX; ++k) s += Bm[k * NU + i] * Pm[c * NX + k]; BtPt[i * NX + c] = s; }
      mm_nn<NU, NX, NU>(BtPt, Bm, T2);
      for (int i = 0; i < NU; ++i) for (int c = 0; c < NU; ++c) Quu[i * NU + c] = 0.5 * (((R[i * NU + c] + T1[i * NU + c]) + R[c * NU + i]) + T2[i * NU + c]);
    }
    {
      double T3[NU * NX];
      mm_nn<NU, NX, NX>(BtP, A, T3);
      for (int i = 0; i < NU; ++i) for (int c = 0; c < NX; ++c) Qux[i * NX + c] = T3[i * NX + c] + Mm[c * NU + i];
    }
    LDLTd<NU> f;
    f.compute(Quu, NU);
    if (!f.ok) return false;
    double KK[NU * NX], col[NU];
    for (int c = 0; c < NX; ++c) { for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c]; f.solve(col); for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i]; }
    double KtQ[NX * NU];
    mm_tn<NX, NU, NU>(KK, Quu, KtQ);
    bool fin = true;
    for (int i = 0; i < NU * NX; ++i) fin = fin && dfinite(KK[i]);
    for (int v = 0; v <= pT; ++v) {   // gradient variants (drift = p_next + P_next * 0)
      double pv[NX], drift[NX], Qx[NX], Qu[NU], kk[NU], pn[NX];
      for (int i = 0; i < NX; ++i) pv[i] = TP(v, t + 1, i);
      for (int i = 0; i < NX; ++i) { double s = 0.0; for (int k = 0; k < NX; ++k) s += Pm[i * NX + k] * 0.0; drift[i] = pv[i] + s; }
      for (int i = 0; i < NX; ++i) { double s = 0.0; for (int k = 0; k < NX; ++k) s += A[k * NX + i] * drift[k]; Qx[i] = q[i] + s; }
      for (int i = 0; i < NU; ++i) { double s = 0.0; for (int k = 0; k < NX; ++k) s += Bm[k * NU + i] * drift[k]; Qu[i] = r[i] + s; }
      for (int i = 0; i < NU; ++i) col[i] = Qu[i];
      f.solve(col);
      for (int i = 0; i < NU; ++i) kk[i] = -col[i];
      for (int i = 0; i < NX; ++i) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int j = 0; j < NU; ++j) { a1 += Qux[j * NX + i] * kk[j]; a2 += KK[j * NX + i] * Qu[j]; a3 += KtQ[i * NU + j] * kk[j]; }
        pn[i] = ((Qx[i] + a1) + a2) + a3;
        fin = fin && dfinite(pn[i]);
      }
      for (int i = 0; i < NU; ++i) { fin = fin && dfinite(kk[i]); TK(v, t, i) = kk[i]; }
      for (int i = 0; i < NX; ++i) TP(v, t, i) = pn[i];
    }
    double T1[NX * NX], AtPA[NX * NX], Pn[NX * NX];
    mm_tn<NX, NX, NX>(A, Pm, T1);
    mm_nn<NX, NX, NX>(T1, A, AtPA);
    for (int i = 0; i < NX; ++i)
      for (int c = 0; c < NX; ++c) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int j = 0; j < NU; ++j) { a1 += Qux[j * NX + i] * KK[j * NX + c]; a2 += KK[j * NX + i] * Qux[j * NX + c]; a3 += KtQ[i * NU + j] * KK[j * NX + c]; }
        Pn[i * NX + c] = (((Q[i * NX + c] + AtPA[i * NX + c]) + a1) + a2) + a3;
      }
    for (int i = 0; i < NX; ++i) for (int c = 0; c < NX; ++c) { Pm[i * NX + c] = 0.5 * (Pn[i * NX + c] + Pn[c * NX + i]); fin = fin && dfinite(Pm[i * NX + c]); }
    if (!fin) return false;
    for (int i = 0; i < NU * NX; ++i) a.K[SI(t, NU * NX, i)] = KK[i];
    for (int i = 0; i < NX * NX; ++i) a.Vxx[SI(t, NX * NX, i)] = Pm[i];
  }
  // closed-loop rollouts of the variants, dx0 = 0 (rolloutLinearPolicy :368-392)
  double xT[(kPTS + 1) * NX];
  for (int v = 0; v <= pT; ++v) {
    double dx[NX];
    for (int i = 0; i < NX; ++i) dx[i] = 0.0;
    for (int t = 0; t < N; ++t) {
      double du[NU], dxn[NX];
      for (int i = 0; i < NU; ++i) { double s = 0.0; for (int j = 0; j < NX; ++j) s += a.K[SI(t, NU * NX, i * NX + j)] * dx[j]; du[i] = TK(v, t, i) + s; }
      for (int i = 0; i < NX; ++i) { double s = 0.0, c = 0.0; for (int j = 0; j < NX; ++j) s += a.fx[SI(t, NX * NX, i * NX + j)] * dx[j]; for (int j = 0; j < NU; ++j) c += a.fu[SI(t, NX * NU, i * NU + j)] * du[j]; dxn[i] = (s + c) + 0.0; }
      for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
    }
    for (int i = 0; i < NX; ++i) xT[v * NX + i] = dx[i];
  }
  // reduced terminal system (:550-617): A_small = H_T S, rhs = b_T - H_T x_T^0
  double As[kPTS * kPTS], rhs[kPTS], AtA[kPTS * kPTS], Atb[kPTS];
  for (int r = 0; r < pT; ++r) {
    for (int i = 0; i < pT; ++i) { double s = 0.0; for (int k = 0; k < NX; ++k) s += HT[r * NX + k] * (xT[(i + 1) * NX + k] - xT[k]); As[r * kPTS + i] = s; }
    double hx = 0.0;
    for (int k = 0; k < NX; ++k) hx += HT[r * NX + k] * xT[k];
    rhs[r] = e.bT[(size_t)r * a.Bp + b] - hx;
  }
  double tr = 0.0;
  for (int i = 0; i < pT; ++i) {
    for (int c = 0; c < pT; ++c) { double s = 0.0; for (int k = 0; k < pT; ++k) s += As[k * kPTS + i] * As[k * kPTS + c]; AtA[i * kPTS + c] = s; }
    double s = 0.0; for (int k = 0; k < pT; ++k) s += As[k * kPTS + i] * rhs[k]; Atb[i] = s;
  }
  for (int i = 0; i < pT; ++i) tr += AtA[i * kPTS + i];
  const double trace_term = (tr > 1.0 ? tr / (pT > 1 ? pT : 1) : 1.0);
  const double regv = dmax(e.floor_[b], 1e-6 * trace_term);
  double smax, smin;
  te_singular_minmax<kPTS>(As, pT, smax, smin);
  const double svd_reg = dmax(1e-8 * smax - smin, 0.0);
  const double reg_base = dmax(regv, svd_reg);
  double rn = 0.0; for (int r = 0; r < pT; ++r) rn += rhs[r] * rhs[r];
  const double cap = 100.0 * (1.0 + sqrt(rn));
  const double scales[5] = {1.0, 10.0, 100.0, 1e3, 1e4};
  double best[kPTS]; for (int i = 0; i < pT; ++i) best[i] = 0.0;
  double best_res = INFINITY; bool found = false;
  for (int sc = 0; sc < 5; ++sc) {
    const double reg_i = dmax(reg_base * scales[sc], 1e-12);
    double Sh[kPTS * kPTS];
    for (int i = 0; i < pT; ++i) for (int c = 0; c < pT; ++c) Sh[i * kPTS + c] = AtA[i * kPTS + c] + reg_i * ((i == c) ? 1.0 : 0.0);
    LDLTd<kPTS> f;
    f.compute(Sh, pT);
    if (!f.ok) continue;
    double lam[kPTS]; for (int i = 0; i < pT; ++i) lam[i] = Atb[i];
    f.solve(lam);
    bool fin = true; double ln = 0.0;
    for (int i = 0; i < pT; ++i) { fin = fin && dfinite(lam[i]); ln += lam[i] * lam[i]; }
    if (!fin) continue;
    ln = sqrt(ln);
    if (ln > cap) { const double f2 = cap / dmax(ln, 1e-12); for (int i = 0; i < pT; ++i) lam[i] = lam[i] * f2; }
    double res = 0.0;
    for (int r = 0; r < pT; ++r) { double s = 0.0; for (int i = 0; i < pT; ++i) s += As[r * kPTS + i] * lam[i]; const double d = s - rhs[r]; res += d * d; }
    res = sqrt(res);
    if (!dfinite(res)) continue;
    if (!found || res < best_res) { for (int i = 0; i < pT; ++i) best[i] = lam[i]; best_res = res; found = true; }
  }
  if (!found) for (int i = 0; i < pT; ++i) best[i] = 0.0;
  for (int i = 0; i < pT; ++i) e.dlam[(size_t)i * a.Bp + b] = best[i];
  // recombination (:619-634); inf_du, step_norm (:1260-1266)
  inf_du = 0.0; step_norm = 0.0;
  for (int t = 0; t <= N; ++t) {
    if (t < N) {
      double ko[NU];
      for (int i = 0; i < NU; ++i) ko[i] = TK(0, t, i);
      for (int v = 0; v < pT; ++v) for (int i = 0; i < NU; ++i) ko[i] += best[v] * (TK(v + 1, t, i) - TK(0, t, i));
      for (int i = 0; i < NU; ++i) { a.k[SI(t, NU, i)] = ko[i]; step_norm = dmax(step_norm, fabs(ko[i])); }
    }
    double po[NX];
    for (int i = 0; i < NX; ++i) po[i] = TP(0, t, i);
    for (int v = 0; v < pT; ++v) for (int i = 0; i < NX; ++i) po[i] += best[v] * (TP(v + 1, t, i) - TP(0, t, i));
    for (int i = 0; i < NX; ++i) a.Vx[SI(t, NX, i)] = po[i];
  }
  for (int t = 0; t < N; ++t)
    for (int i = 0; i < NU; ++i) { double s = 0.0; for (int k = 0; k < NX; ++k) s += a.fu[SI(t, NX * NU, k * NU + i)] * a.Vx[SI(t + 1, NX, k)]; inf_du = dmax(inf_du, fabs(a.lu[SI(t, NU, i)] + s)); }
  {   // rolloutLinearPolicy with the recombined gains (:1268)
    double dx[NX];
    for (int i = 0; i < NX; ++i) dx[i] = 0.0;
    for (int t = 0; t < N; ++t) {
      for (int i = 0; i < NX; ++i) e.dX[SI(t, NX, i)] = dx[i];
      double du[NU], dxn[NX];
      for (int i = 0; i < NU; ++i) { double s = 0.0; for (int j = 0; j < NX; ++j) s += a.K[SI(t, NU * NX, i * NX + j)] * dx[j]; du[i] = a.k[SI(t, NU, i)] + s; }
      for (int i = 0; i < NX; ++i) { double s = 0.0, c = 0.0; for (int j = 0; j < NX; ++j) s += a.fx[SI(t, NX * NX, i * NX + j)] * dx[j]; for (int j = 0; j < NU; ++j) c += a.fu[SI(t, NX * NU, i * NU + j)] * du[j]; dxn[i] = (s + c) + 0.0; }
      for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
    }
    for (int i = 0; i < NX; ++i) e.dX[SI(N, NX, i)] = dx[i];
  }
#undef TP
#undef TK
  return true;
}

template <int NX, int NU, bool T4>
__global__ __launch_bounds__(64) void k_stacks_te(StackArgs a, StackTeArgs e) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  double reg = a.reg_in[b];
  double inf_du = 0, step_norm = 0;
  bool ok = false;
  for (;;) {   // "increase the regularisation and retry" (cddp_solver_base.cpp:93-111)
    ok = te_sweep<NX, NU, T4>(a, e, b, reg, inf_du, step_norm);
    if (ok || !(a.reg_factor > 1.0)) break;
    reg = reg * a.reg_factor;
    if (!(reg > 0.0)) reg = (a.opt.reg_min_value > 0.0) ? a.opt.reg_min_value : a.reg_max;
    reg = dmin(reg, a.reg_max);
    if (reg >= a.reg_max) break;
  }
  a.ok[b] = ok ? 1 : 0;
  a.dV[(size_t)0 * a.Bp + b] = 0.0; a.dV[(size_t)1 * a.Bp + b] = 0.0;     // dV_ stays zero in this branch (:994)
  a.scal[(size_t)0 * a.Bp + b] = reg; a.scal[(size_t)1 * a.Bp + b] = inf_du; a.scal[(size_t)2 * a.Bp + b] = 0.0;
  a.scal[(size_t)3 * a.Bp + b] = 0.0; a.scal[(size_t)4 * a.Bp + b] = step_norm;
  a.caps[(size_t)0 * a.Bp + b] = 1.0; a.caps[(size_t)1 * a.Bp + b] = 1.0;
}

template <int NX, int NU>
void launch_te(const StackArgs &a, const StackTeArgs &e, hipStream_t s) {
  if constexpr (NX >= 6) {
    if (a.t4) { hipLaunchKernelGGL((k_stacks_te<NX, NU, true>), dim3((a.B + 63) / 64), dim3(64), 0, s, a, e); return; }
  }
  hipLaunchKernelGGL((k_stacks_te<NX, NU, false>), dim3((a.B + 63) / 64), dim3(64), 0, s, a, e);
}
typedef void (*LaunchTeFn)(const StackArgs &, const StackTeArgs &, hipStream_t);

LaunchTeFn pick_te(int nx, int nu) {
#define PICK(X, U) if (nx == X && nu == U) return &launch_te<X, U>;
  PICK(1, 1) PICK(2, 1) PICK(2, 2) PICK(3, 1) PICK(3, 2) PICK(4, 1) PICK(4, 2) PICK(6, 3)
#undef PICK
  return nullptr;
}

}  // namespace
