// Lane-cooperative form of the stack-fed sweep (stacks.hip) for plug-ins with a large state: sixteen lanes per trajectory,
// every matrix of a step in LDS, each output element owned by ONE lane that accumulates it in exactly the order the
// one-lane kernel `sweep<NX, NU, M>` uses (k ascending from 0.0) -- the two kernels are bitwise interchangeable
// (tests/test_stack_fed_coop.py).  The one-lane form keeps ~6 nx^2 doubles per lane live: at nx = 12 it runs from scratch
// memory (10-18 KB per lane) and takes 155-200 ms per sweep of 2048 trajectories (profiles/r02_stackfed_sweeps.md); here a
// trajectory's step state is 12-17 KB of LDS shared by its sixteen lanes and nothing spills.
//
// Same branches as the one-lane kernel (clddp_solver.cpp:79-204, ipddp_solver.cpp:1048-1118 / 1355-1568, logddp_solver.cpp:470-575,
// msipddp_solver.cpp:1112-1208).  The small replicated pieces (nu x nu factorisation, BoxQP, the scalar reductions) are done by every
// lane of the group on identical inputs, so "failed" is group-uniform without a vote.
// Included by stacks.hip inside its anonymous namespace (uses StackArgs, SI, clipp, clips).
#pragma once

DEV void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// blocks of one 64-trajectory tile on one XCD: the sixteen single-wave workgroups of a tile read 32-B pieces of the same
// 128-B lines of the batch-minor stacks (same reasoning as kernels_coop.hpp::coop_group, profiles/r03_pmc_calibration_32B.md)
DEV int sc_group(int bid) {
  const int sup = bid >> 7, r = bid & 127;
  return (sup * 8 + (r & 7)) * 16 + (r >> 3);
}
inline unsigned sc_grid(int B) { return (unsigned)(((B + 3) / 4 + 127) / 128 * 128); }

template <int NX, int NU, int M>
struct SCfg {
  static constexpr int MM = M > 0 ? M : 1;
  enum : int {
    oVxx = 0, oVx = oVxx + NX * NX, oA = oVx + NX, oB = oA + NX * NX, oW = oB + NX * NU, oQx = oW + NX, oQu = oQx + NX,
    oQxx = oQu + NU, oQux = oQxx + NX * NX, oQuu = oQux + NU * NX, oT1 = oQuu + NU * NU, oT2 = oT1 + NX * NX, oKK = oT2 + NU * NX,
    okk = oKK + NU * NX, oKtQ = okk + NU, oQr = oKtQ + NX * NU, oRu = oQr + NU * NU, oRx = oRu + NU,
    oY = oRx + NU * NX, oS = oY + MM, oGg = oS + MM, oGx = oGg + MM, oGu = oGx + MM * NX, oYS = oGu + MM * NU, oSir = oYS + MM,
    oRhat = oSir + MM, oRp = oRhat + MM, oSs = oRp + MM, oRed = oSs + MM, SIZE0 = oRed + 32,
    STRIDE = SIZE0 | 1   // odd: the four trajectories of a wavefront start in different LDS banks
  };
  static constexpr int oVn = oT1;   // V_xx before symmetrisation overlays T1 (dead after the A-products)
};

// "for e in my elements of [0, E)": element e of a step quantity belongs to lane e mod 16 of the trajectory's group
#define SC_EACH(E, e) _Pragma("unroll") for (int e##_it = 0; e##_it < ((E) + 15) / 16; ++e##_it) if (const int e = e##_it * 16 + gl; e < (E))

template <int NU>
struct SCFactor {   // the factorisation the one-lane kernel uses for this size, on register copies
  LDLTs<NU> f;
  DEV bool compute(const double *Q) { f.compute(Q, NU); return f.ok; }
  DEV void solve(double *x) const { f.solve(x); }
};
template <>
struct SCFactor<2> {
  LDLTd<2> f;
  DEV bool compute(const double *Q) { f.compute(Q, 2); return f.ok; }
  DEV void solve(double *x) const { f.solve(x); }
};
template <>
struct SCFactor<1> {
  double d;
  DEV bool compute(const double *Q) { d = Q[0]; return true; }
  DEV void solve(double *x) const { x[0] = ldlt1_solve(d, x[0]); }
};

template <int NX, int NU, int M>
DEV bool sweep_coop(const StackArgs &a, const int b, const int gl, double *__restrict__ L, const double reg, const double mu, double &dV0,
                    double &dV1, double &inf_du, double &inf_pr, double &inf_comp, double &step_norm) {
  typedef SCfg<NX, NU, M> C;
  const int N = a.N;
  const bool lg = a.branch == CDDP_HIP_STACKS_LOGDDP;
  const bool ms = a.branch == CDDP_HIP_STACKS_MSIPDDP;
  const bool ip = a.branch != CDDP_HIP_STACKS_CLDDP && !lg;
  SC_EACH(NX, i) L[C::oVx + i] = a.VxN[(size_t)i * a.Bp + b];
  if (ip || lg) {
    SC_EACH(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      L[C::oVxx + e] = 0.5 * (a.VxxN[(size_t)(i * NX + c) * a.Bp + b] + a.VxxN[(size_t)(c * NX + i) * a.Bp + b]);
    }
  } else {
    SC_EACH(NX * NX, e) L[C::oVxx + e] = a.VxxN[(size_t)e * a.Bp + b];
  }
  lds_sync();
  SC_EACH(NX, i) a.Vx[SI(N, NX, i)] = L[C::oVx + i];
  SC_EACH(NX * NX, e) a.Vxx[SI(N, NX * NX, e)] = L[C::oVxx + e];
  dV0 = dV1 = 0.0; inf_du = inf_pr = inf_comp = step_norm = 0.0;
  double norm_Vx = 0.0;
  if (!ip && !lg) {
#pragma unroll
    for (int i = 0; i < NX; ++i) norm_Vx += fabs(L[C::oVx + i]);
  }
  for (int t = N - 1; t >= 0; --t) {
    // ---------------------------------------------------------------- step record -> LDS
    SC_EACH(NX * NX, e) L[C::oA + e] = a.fx[SI(t, NX * NX, e)];
    SC_EACH(NX * NU, e) L[C::oB + e] = a.fu[SI(t, NX * NU, e)];
    SC_EACH(NX, e) L[C::oQx + e] = a.lx[SI(t, NX, e)];
    SC_EACH(NU, e) L[C::oQu + e] = a.lu[SI(t, NU, e)];
    SC_EACH(NX * NX, e) L[C::oQxx + e] = a.lxx[SI(t, NX * NX, e)];
    SC_EACH(NU * NU, e) L[C::oQuu + e] = a.luu[SI(t, NU * NU, e)];
    SC_EACH(NU * NX, e) L[C::oQux + e] = a.lux[SI(t, NU * NX, e)];
    if constexpr (M > 0) {
      SC_EACH(M, e) { L[C::oY + e] = a.y[SI(t, M, e)]; L[C::oS + e] = a.s[SI(t, M, e)]; L[C::oGg + e] = a.g[SI(t, M, e)]; }
      SC_EACH(M * NX, e) L[C::oGx + e] = a.Gx[SI(t, M * NX, e)];
      SC_EACH(M * NU, e) L[C::oGu + e] = a.Gu[SI(t, M * NU, e)];
    }
    // w = V_x, or V_x + V_xx d_t under multiple shooting (V of step t + 1 is already in LDS)
    if (ms) {
      SC_EACH(NX, i) {
        double s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s1 += L[C::oVxx + i * NX + k] * a.dfc[SI(t, NX, k)];
        L[C::oW + i] = L[C::oVx + i] + s1;
      }
    } else {
      SC_EACH(NX, i) L[C::oW + i] = L[C::oVx + i];
    }
    lds_sync();
    // ---------------------------------------------------------------- Q_x, Q_u, T1 = A^T V_xx, T2 = B^T V_xx
    SC_EACH(NX, i) {
      double q = L[C::oQx + i];
      if constexpr (M > 0) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGx + r * NX + i] * L[C::oY + r];
        q = q + s1;
      }
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s2 += L[C::oA + k * NX + i] * L[C::oW + k];
      L[C::oQx + i] = q + s2;
    }
    SC_EACH(NU, i) {
      double q = L[C::oQu + i];
      if constexpr (M > 0) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGu + r * NU + i] * L[C::oY + r];
        q = q + s1;
      }
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s2 += L[C::oB + k * NU + i] * L[C::oW + k];
      L[C::oQu + i] = q + s2;
    }
    SC_EACH(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += L[C::oA + k * NX + i] * L[C::oVxx + k * NX + c];
      L[C::oT1 + e] = s1;
    }
    SC_EACH(NU * NX, e) {
      const int i = e / NX, c = e - i * NX;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += L[C::oB + k * NU + i] * L[C::oVxx + k * NX + c];
      L[C::oT2 + e] = s1;
    }
    lds_sync();
    // ---------------------------------------------------------------- Q_xx += T1 A, Q_ux += T2 A, Q_uu += T2 B (+ tensor terms)
    SC_EACH(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += L[C::oT1 + i * NX + k] * L[C::oA + k * NX + c];
      double q = L[C::oQxx + e] + s1;
      if (a.Fxx) for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * a.Fxx[SI(t, NX * NX * NX, j * NX * NX + e)];
      L[C::oQxx + e] = q;
    }
    SC_EACH(NU * NX, e) {
      const int i = e / NX, c = e - i * NX;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += L[C::oT2 + i * NX + k] * L[C::oA + k * NX + c];
      double q = L[C::oQux + e] + s1;
      if (a.Fxx) for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * a.Fux[SI(t, NX * NU * NX, j * NU * NX + e)];
      L[C::oQux + e] = q;
    }
    SC_EACH(NU * NU, e) {
      const int i = e / NU, c = e - i * NU;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += L[C::oT2 + i * NX + k] * L[C::oB + k * NU + c];
      double q = L[C::oQuu + e] + s1;
      if (a.Fxx) for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * a.Fuu[SI(t, NX * NU * NU, j * NU * NU + e)];
      L[C::oQuu + e] = q;
    }
    lds_sync();
    // ---------------------------------------------------------------- gains
    double kk[NU];
    if constexpr (M > 0) {
      const double s_floor = dmax(mu * 1e-3, kEpsSlackS);
      SC_EACH(M, r) {
        const double y = L[C::oY + r], s = L[C::oS + r], g = L[C::oGg + r];
        const double ssafe = dmax(s, s_floor);
        const double rp = g + s;
        const double rc = y * s - mu;
        const double rhat = y * rp - rc;
        L[C::oSs + r] = ssafe; L[C::oYS + r] = clipp(y, ssafe); L[C::oRp + r] = rp; L[C::oRhat + r] = rhat; L[C::oSir + r] = clips(rhat, ssafe);
        inf_pr = dmax(inf_pr, fabs(rp)); inf_comp = dmax(inf_comp, fabs(rc));
      }
      lds_sync();
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += (L[C::oGu + r * NU + i] * L[C::oYS + r]) * L[C::oGu + r * NU + c];
        double q = 0.5 * (L[C::oQuu + i * NU + c] + L[C::oQuu + c * NU + i]) + s1;
        if (i == c) q += reg;
        L[C::oQr + e] = q;
      }
      SC_EACH(NU, i) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGu + r * NU + i] * L[C::oSir + r];
        L[C::oRu + i] = L[C::oQu + i] + s1;
      }
      SC_EACH(NU * NX, e) {
        const int i = e / NX, c = e - i * NX;
        double s2 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s2 += (L[C::oGu + r * NU + i] * L[C::oYS + r]) * L[C::oGx + r * NX + c];
        L[C::oRx + e] = L[C::oQux + e] + s2;
      }
      lds_sync();
      {
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQr + i];
        SCFactor<NU> f;
        if (!f.compute(Qr)) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = L[C::oRu + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = L[C::oRx + i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = -col[i];
        }
      }
      lds_sync();
      // slack / dual direction gains (:1458-1472)
      SC_EACH(M, r) {
        double temp = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) temp += L[C::oGu + r * NU + i] * kk[i];
        a.ky[SI(t, M, r)] = clips(L[C::oRhat + r] + L[C::oY + r] * temp, L[C::oSs + r]);
        a.ks[SI(t, M, r)] = (-L[C::oRp + r]) - temp;
      }
      SC_EACH(M * NX, e) {
        const int r = e / NX, c = e - r * NX;
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) s2 += L[C::oGu + r * NU + i] * L[C::oKK + i * NX + c];
        const double gx = L[C::oGx + e];
        const double inner = gx + s2;
        a.Ky[SI(t, M * NX, e)] = dclamp(L[C::oYS + r] * inner, -kMaxRatioS, kMaxRatioS);
        a.Ks[SI(t, M * NX, e)] = (-gx) - s2;
      }
      // condensed terms into the Q blocks (:1488-1492)
      SC_EACH(NX, i) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGx + r * NX + i] * L[C::oSir + r];
        L[C::oQx + i] = L[C::oQx + i] + s1;
      }
      SC_EACH(NX * NX, e) {
        const int i = e / NX, c = e - i * NX;
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += (L[C::oGx + r * NX + i] * L[C::oYS + r]) * L[C::oGx + r * NX + c];
        L[C::oQxx + e] = L[C::oQxx + e] + s1;
      }
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += (L[C::oGu + r * NU + i] * L[C::oYS + r]) * L[C::oGu + r * NU + c];
        L[C::oQuu + e] = L[C::oQuu + e] + s1;
      }
      SC_EACH(NU, i) L[C::oQu + i] = L[C::oRu + i];
      SC_EACH(NU * NX, e) L[C::oQux + e] = L[C::oRx + e];
    } else if (ip || lg) {
      // IPDDP: Q_uu = sym(Q_uu) + reg I, kept (:1084-1101).  LogDDP: factor sym(Q_uu + reg I), Q_uu itself untouched (:524-548)
      double qs[(NU * NU + 15) / 16];
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double p = L[C::oQuu + i * NU + c], q = L[C::oQuu + c * NU + i];
        if (lg) { if (i == c) { p += reg; q += reg; } qs[e_it] = 0.5 * (p + q); }
        else { double v = 0.5 * (p + q); if (i == c) v += reg; qs[e_it] = v; }
      }
      lds_sync();
      SC_EACH(NU * NU, e) { L[C::oQr + e] = qs[e_it]; if (!lg) L[C::oQuu + e] = qs[e_it]; }
      lds_sync();
      {
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQr + i];
        SCFactor<NU> f;
        if (!f.compute(Qr)) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = L[C::oQu + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = L[C::oQux + i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = -col[i];
        }
      }
    } else {
      // CLDDP: PD test on Q_uu + reg I, BoxQP or dense inverse (clddp_solver.cpp:130-178); every lane of the group repeats it
      double Qr[NU * NU], Qu[NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQuu + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) { Qr[i * NU + i] += reg; Qu[i] = L[C::oQu + i]; }
      if (min_real_eig<NU>(Qr) <= 0) return false;
      if (a.lo) {
        double lb[NU], ub[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { const double ut = a.U[SI(t, NU, i)]; lb[i] = a.lo[i] - ut; ub[i] = a.up[i] - ut; kk[i] = a.k[SI(t, NU, i)]; }
        int free_[NU];
        LDLTd<NU> Hfree;
        const int stq = boxqp_solve<NU>(a.opt, Qr, Qu, lb, ub, kk, free_, Hfree);
        if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
        int free_idx[NU]; int nf = 0;
        for (int i = 0; i < NU; ++i) if (free_[i]) free_idx[nf++] = i;
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = 0.0;
          if (nf > 0) {
            double col[NU];
            for (int i = 0; i < nf; ++i) col[i] = L[C::oQux + free_idx[i] * NX + c];
            Hfree.solve(col);
            for (int i = 0; i < nf; ++i) L[C::oKK + free_idx[i] * NX + c] = -col[i];
          }
        }
      } else {
        double H[NU * NU];
        inverse_pplu<NU>(Qr, H);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double s1 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) s1 += (-H[i * NU + j]) * Qu[j];
          kk[i] = s1;
        }
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) s2 += (-H[i * NU + j]) * L[C::oQux + j * NX + c];
            L[C::oKK + i * NX + c] = s2;
          }
        }
      }
    }
    lds_sync();
    SC_EACH(NU, i) a.k[SI(t, NU, i)] = kk[i];
    SC_EACH(NU * NX, e) a.K[SI(t, NU * NX, e)] = L[C::oKK + e];
    {   // expected-decrease terms: the scalar chain every lane repeats
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) q += L[C::oQuu + i * NU + j] * kk[j];
        s0 += L[C::oQu + i] * kk[i]; s1 += kk[i] * q;
      }
      dV0 += s0; dV1 += 0.5 * s1;
#pragma unroll
      for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(L[C::oQu + i])); step_norm = dmax(step_norm, fabs(kk[i])); }
    }
    SC_EACH(NX * NU, e) {   // K^T Q_uu
      const int i = e / NU, j = e - i * NU;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NU; ++k) s1 += L[C::oKK + k * NX + i] * L[C::oQuu + k * NU + j];
      L[C::oKtQ + e] = s1;
    }
    lds_sync();
    // ---------------------------------------------------------------- value update
    double vxn[(NX + 15) / 16];
    SC_EACH(NX, i) {
      double p = 0.0, q = 0.0, r = 0.0;
      if (ip) {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKK + j * NX + i] * L[C::oQu + j]; q += L[C::oQux + j * NX + i] * kk[j]; r += L[C::oKtQ + i * NU + j] * kk[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKtQ + i * NU + j] * kk[j]; q += L[C::oQux + j * NX + i] * kk[j]; r += L[C::oKK + j * NX + i] * L[C::oQu + j]; }
      }
      vxn[i_it] = ((L[C::oQx + i] + p) + q) + r;
    }
    SC_EACH(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      double p = 0.0, q = 0.0, r = 0.0;
      if (ip) {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKK + j * NX + i] * L[C::oQux + j * NX + c]; q += L[C::oQux + j * NX + i] * L[C::oKK + j * NX + c]; r += L[C::oKtQ + i * NU + j] * L[C::oKK + j * NX + c]; }
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKtQ + i * NU + j] * L[C::oKK + j * NX + c]; q += L[C::oQux + j * NX + i] * L[C::oKK + j * NX + c]; r += L[C::oKK + j * NX + i] * L[C::oQux + j * NX + c]; }
      }
      L[C::oVn + e] = ((L[C::oQxx + e] + p) + q) + r;
    }
    SC_EACH(NX, i) { L[C::oVx + i] = vxn[i_it]; a.Vx[SI(t, NX, i)] = vxn[i_it]; }
    lds_sync();
    SC_EACH(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      const double v = 0.5 * (L[C::oVn + i * NX + c] + L[C::oVn + c * NX + i]);
      L[C::oVxx + e] = v; a.Vxx[SI(t, NX * NX, e)] = v;
    }
    if (!ip && !lg) {
#pragma unroll
      for (int i = 0; i < NX; ++i) norm_Vx += fabs(L[C::oVx + i]);
    }
    lds_sync();
  }
  if constexpr (M > 0) {   // the lanes hold partial maxima over their constraint rows
    L[C::oRed + gl] = inf_pr; L[C::oRed + 16 + gl] = inf_comp;
    lds_sync();
    inf_pr = 0.0; inf_comp = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { inf_pr = dmax(inf_pr, L[C::oRed + i]); inf_comp = dmax(inf_comp, L[C::oRed + 16 + i]); }
    lds_sync();
  }
  if (!ip && !lg) {
    double sc = a.tau_min;
    sc = dmax(sc, norm_Vx / (double)(N * NX)) / sc;
    inf_du = inf_du / sc;
  }
  return true;
}

template <int NX, int NU, int M>
__global__ __launch_bounds__(64) void k_stacks_backward_coop(StackArgs a) {
  typedef SCfg<NX, NU, M> C;
  extern __shared__ double sc_lds[];
  const int tl = threadIdx.x >> 4, gl = threadIdx.x & 15;
  const int b = sc_group((int)blockIdx.x) * 4 + tl;
  if (b >= a.B) return;
  double *L = sc_lds + tl * C::STRIDE;
  const double mu = a.mu ? a.mu[b] : 0.0;
  double reg = a.reg_in[b];
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  bool ok = false;
  for (;;) {
    ok = sweep_coop<NX, NU, M>(a, b, gl, L, reg, mu, dV0, dV1, inf_du, inf_pr, inf_comp, step_norm);
    if (ok || !(a.reg_factor > 1.0)) break;
    reg = reg * a.reg_factor;
    if (!(reg > 0.0)) reg = (a.opt.reg_min_value > 0.0) ? a.opt.reg_min_value : a.reg_max;
    reg = dmin(reg, a.reg_max);
    if (reg >= a.reg_max) break;
    lds_sync();
  }
  double apr = 1.0, adu = 1.0;
  if constexpr (M > 0) {
    if (ok) {   // rolloutLinearPolicy from dx0 = 0, dS / dY, computeMaxStepSizes (ipddp_solver.cpp:1511-1532, 2939-2988)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // the gains were written through other lanes of this group
      const int N = a.N;
      const double tau = dmax(a.tau_min, 1.0 - mu);
      SC_EACH(NX, i) L[C::oW + i] = 0.0;
      lds_sync();
      for (int t = 0; t < N; ++t) {
        SC_EACH(NX, i) a.dX[SI(t, NX, i)] = L[C::oW + i];
        SC_EACH(M, r) {
          double p = 0.0, q = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) { const double dxj = L[C::oW + j]; p += a.Ks[SI(t, M * NX, r * NX + j)] * dxj; q += a.Ky[SI(t, M * NX, r * NX + j)] * dxj; }
          const double ds = a.ks[SI(t, M, r)] + p;
          const double dy = dclamp(a.ky[SI(t, M, r)] + q, -kMaxRatioS, kMaxRatioS);
          if (ds < 0.0) apr = dmin(apr, -tau * a.s[SI(t, M, r)] / ds);
          if (dy < 0.0) adu = dmin(adu, -tau * a.y[SI(t, M, r)] / dy);
        }
        SC_EACH(NU, i) {
          double p = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) p += a.K[SI(t, NU * NX, i * NX + j)] * L[C::oW + j];
          L[C::oQu + i] = a.k[SI(t, NU, i)] + p;
        }
        lds_sync();
        double dxn[(NX + 15) / 16];
        SC_EACH(NX, i) {
          double p = 0.0, q = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) p += a.fx[SI(t, NX * NX, i * NX + j)] * L[C::oW + j];
#pragma unroll
          for (int j = 0; j < NU; ++j) q += a.fu[SI(t, NX * NU, i * NU + j)] * L[C::oQu + j];
          dxn[i_it] = (p + q) + 0.0;
        }
        lds_sync();
        SC_EACH(NX, i) L[C::oW + i] = dxn[i_it];
        lds_sync();
      }
      SC_EACH(NX, i) a.dX[SI(N, NX, i)] = L[C::oW + i];
      L[C::oRed + gl] = apr; L[C::oRed + 16 + gl] = adu;
      lds_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) { apr = dmin(apr, L[C::oRed + i]); adu = dmin(adu, L[C::oRed + 16 + i]); }
      apr = dclamp(apr, 0.0, 1.0); adu = dclamp(adu, 0.0, 1.0);
    }
  }
  if (gl == 0) {
    a.ok[b] = ok ? 1 : 0;
    a.dV[(size_t)0 * a.Bp + b] = dV0; a.dV[(size_t)1 * a.Bp + b] = dV1;
    a.scal[(size_t)0 * a.Bp + b] = reg; a.scal[(size_t)1 * a.Bp + b] = inf_du; a.scal[(size_t)2 * a.Bp + b] = inf_pr;
    a.scal[(size_t)3 * a.Bp + b] = inf_comp; a.scal[(size_t)4 * a.Bp + b] = step_norm;
    a.caps[(size_t)0 * a.Bp + b] = apr; a.caps[(size_t)1 * a.Bp + b] = adu;
  }
}

template <int NX, int NU, int M>
void launch_coop(const StackArgs &a, hipStream_t s) {
  typedef SCfg<NX, NU, M> C;
  constexpr size_t lds_bytes = (size_t)4 * C::STRIDE * sizeof(double);
  static_assert(lds_bytes <= 160 * 1024, "step state of four trajectories must fit the CU's LDS");
  static bool once = false;
  if (!once && lds_bytes > 64 * 1024) {
    hipFuncSetAttribute((const void *)k_stacks_backward_coop<NX, NU, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    once = true;
  }
  hipLaunchKernelGGL((k_stacks_backward_coop<NX, NU, M>), dim3(sc_grid(a.B)), dim3(64), lds_bytes, s, a);
}
