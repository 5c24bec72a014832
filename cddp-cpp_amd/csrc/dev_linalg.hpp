// Device-side small dense linear algebra for the batched CLDDP / IPDDP kernels (gfx950).
//
// Execution model: ONE TRAJECTORY PER LANE.  Every matrix on the Riccati path is at most
// 14 x 14 fp64, far below an MFMA tile for the single-GPU configs (nx <= 4), so each lane of a
// 64-wide wavefront owns a complete trajectory and keeps its Q/V blocks in VGPRs; the batch
// index is the fastest-varying memory index, so every load/store of a wavefront is one fully
// coalesced 512-byte transaction (see DESIGN.md "Data layout").  All routines below are
// therefore plain per-lane code on small fixed-size arrays with compile-time dimensions
// (fully unrolled by hipcc); no cross-lane traffic is needed.
//
// Semantics follow the reference's linear-algebra dependency where solver decisions depend
// on it (Eigen 3.4.0 LDLT with diagonal pivoting, PartialPivLU inverse) -- see the call sites
// cited at each routine.
#pragma once
#include <cfloat>
#ifndef CDDP_HOST_MODELS   // host_models.cpp compiles the plants (dev_models.hpp) for the host with DEV = inline
#include <hip/hip_runtime.h>
#define DEV __device__ __forceinline__
#define DEV_NOINLINE __device__ __attribute__((noinline))   // large, cold device routines (second-order duals with many seeds)
#else
#include <cmath>
#define DEV_NOINLINE
#endif

namespace cddp_dev {

DEV double dmax(double a, double b) { return (a < b) ? b : a; } // == std::max(a,b), NaN behaviour included
DEV double dmin(double a, double b) { return (b < a) ? b : a; } // == std::min(a,b)
DEV double dclamp(double v, double lo, double hi) { return dmin(dmax(v, lo), hi); }  // std::clamp
DEV bool dfinite(double v) { return fabs(v) <= DBL_MAX; }       // false for NaN and +-Inf
// std::copysign(1.0, dJ) of the reference's accept test (clddp_solver.cpp:254, ipddp_solver.cpp:1848, msipddp_solver.cpp:1702) with the
// zero case written out: dJ = cost - J_new of an unchanged trial is an exact +0 (x - x in round-to-nearest), for which copysign gives
// +1; the library is built with -fno-signed-zeros, under which the sign BIT of a zero difference is not something the compiler has
// to preserve, so the comparison form is used for every non-NaN value (identical to copysign for every non-zero dJ and for +0).
// NaN keeps copysign's sign-bit rule (documented as not comparable across x86 and gfx950).
DEV double sign_of_reduction(double dJ) { return (dJ != dJ) ? copysign(1.0, dJ) : ((dJ < 0.0) ? -1.0 : 1.0); }
// Wave-uniform read-only data (reference trajectory, problem pool): route the address through SGPRs and load
// through the CONSTANT address space so the compiler emits scalar (SMEM) loads.  A generic-pointer load becomes a
// flat/global VECTOR load, which sits on the in-order vmcnt queue in front of the software-pipelined prefetch and
// forces it to drain every step (measured: +270 us per K4 launch at C2).
#ifndef CDDP_HOST_MODELS
typedef const double __attribute__((address_space(4))) *cptr_t;
DEV cptr_t uniform_ptr(const double *p) {
  unsigned long long v = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffull));
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (cptr_t)(((unsigned long long)hi << 32) | lo);
}
#endif
// a*b + c with two roundings (no FMA contraction).  Used where an accept/reject decision sits exactly on a
// rounding knife-edge: the fraction-to-boundary rule caps alpha at -tau*s/ds, so the trial slack
// s + alpha*ds lands ON the bound (1-tau)*s and `s_new < (1-tau)*s` is decided by the last bit.  The
// reference is built without FMA (plain -O3 x86-64, CMakeLists.txt:34-40), so the product must round first.
DEV double madd_2r(double a, double b, double c) {
#pragma clang fp contract(off)
  double p = a * b;
  return p + c;
}
// (base + a*k) + sum_j K[j]*dx[j], every product rounded before it is added (same knife-edge as above)
template <int N>
DEV double affine_2r(double base, double a, double k, const double *Krow, const double *dx) {
#pragma clang fp contract(off)
  double p = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) { double m = Krow[j] * dx[j]; p = p + m; }
  double q = a * k;
  double r = base + q;
  return r + p;
}

// Eigen::LDLT<MatrixXd>, restated (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked
// and LDLT::_solve_impl).  Reference call sites: boxqp.cpp:105,147; ipddp_solver.cpp:456,583,
// 1087,1428.  `n` is the runtime size (<= NMAX) so BoxQP can factor a free sub-block.
template <int NMAX>
struct LDLTd {
  double m[NMAX * NMAX];
  int tr[NMAX];
  int n;
  bool ok;

  DEV void compute(const double *A, int n_) {
    n = n_;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) m[i * NMAX + j] = A[i * NMAX + j];
    ok = true;
    if (n <= 1) { if (n == 1) tr[0] = 0; return; }
    bool found_zero_pivot = false;
    bool ret = true;
    double temp[NMAX];
    for (int k = 0; k < n; ++k) {
      int big = k;
      double bigv = fabs(m[k * NMAX + k]);
      for (int i = k + 1; i < n; ++i) {
        double v = fabs(m[i * NMAX + i]);
        if (v > bigv) { bigv = v; big = i; }
      }
      tr[k] = big;
      if (k != big) {
        int s = n - big - 1;
        for (int j = 0; j < k; ++j) { double t = m[k * NMAX + j]; m[k * NMAX + j] = m[big * NMAX + j]; m[big * NMAX + j] = t; }
        for (int i = 0; i < s; ++i) {
          double t = m[(big + 1 + i) * NMAX + k]; m[(big + 1 + i) * NMAX + k] = m[(big + 1 + i) * NMAX + big]; m[(big + 1 + i) * NMAX + big] = t;
        }
        { double t = m[k * NMAX + k]; m[k * NMAX + k] = m[big * NMAX + big]; m[big * NMAX + big] = t; }
        for (int i = k + 1; i < big; ++i) { double t = m[i * NMAX + k]; m[i * NMAX + k] = m[big * NMAX + i]; m[big * NMAX + i] = t; }
      }
      int rs = n - k - 1;
      if (k > 0) {
        for (int j = 0; j < k; ++j) temp[j] = m[j * NMAX + j] * m[k * NMAX + j];
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += m[k * NMAX + j] * temp[j];
        m[k * NMAX + k] -= s;
        for (int i = 0; i < rs; ++i) {
          double t = 0.0;
          for (int j = 0; j < k; ++j) t += m[(k + 1 + i) * NMAX + j] * temp[j];
          m[(k + 1 + i) * NMAX + k] -= t;
        }
      }
      double akk = m[k * NMAX + k];
      bool valid = fabs(akk) > 0.0;
      if (k == 0 && !valid) {
        for (int j = 0; j < n; ++j) {
          tr[j] = j;
          for (int i = j + 1; i < n; ++i) ret = ret && (m[i * NMAX + j] == 0.0);
        }
        ok = ret;
        return;
      }
      if (rs > 0 && valid) { for (int i = 0; i < rs; ++i) m[(k + 1 + i) * NMAX + k] /= akk; }
      else if (rs > 0) { for (int i = 0; i < rs; ++i) ret = ret && (m[(k + 1 + i) * NMAX + k] == 0.0); }
      if (found_zero_pivot && valid) ret = false;
      else if (!valid) found_zero_pivot = true;
    }
    ok = ret;
  }

  // in-place solve of one right-hand side column x (length n, stride 1)
  DEV void solve(double *x) const {
    for (int k = 0; k < n; ++k) { int t = tr[k]; if (t != k) { double v = x[k]; x[k] = x[t]; x[t] = v; } }
    for (int i = 0; i < n; ++i) { double s = x[i]; for (int kk = 0; kk < i; ++kk) s -= m[i * NMAX + kk] * x[kk]; x[i] = s; }
    for (int i = 0; i < n; ++i) { double d = m[i * NMAX + i]; x[i] = (fabs(d) > DBL_MIN) ? x[i] / d : 0.0; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int kk = i + 1; kk < n; ++kk) s -= m[kk * NMAX + i] * x[kk]; x[i] = s; }
    for (int k = n - 1; k >= 0; --k) { int t = tr[k]; if (t != k) { double v = x[k]; x[k] = x[t]; x[t] = v; } }
  }
};

// Full-size (n == N) LDLT with STATIC register indexing: the pivot position is a run-time value, so every
// pivot-dependent access of LDLTd is rewritten as "for each compile-time candidate B: if (big == B) ...".  Same
// arithmetic, same pivot rule, same D^+ as LDLTd<N>::compute / solve with n = N; nothing is dynamically indexed, so
// nothing lands in scratch memory (the generic form costs ~100 scratch VMEM operations per factorisation).
template <int N>
struct LDLTs {
  double m[N * N];
  int tr[N];
  bool ok;

  DEV void compute(const double *A, int /*n == N*/) {
#pragma unroll
    for (int i = 0; i < N * N; ++i) m[i] = A[i];
    ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) tr[i] = i;
    if (N <= 1) return;
    bool found_zero_pivot = false, ret = true, done = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (done) continue;
      int big = k;
      double bigv = fabs(m[k * N + k]);
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const double v = fabs(m[i * N + i]);
        if (v > bigv) { bigv = v; big = i; }
      }
      tr[k] = big;
#pragma unroll
      for (int B = k + 1; B < N; ++B) {
        if (big == B) {
#pragma unroll
          for (int j = 0; j < k; ++j) { const double t = m[k * N + j]; m[k * N + j] = m[B * N + j]; m[B * N + j] = t; }
#pragma unroll
          for (int i = B + 1; i < N; ++i) { const double t = m[i * N + k]; m[i * N + k] = m[i * N + B]; m[i * N + B] = t; }
          { const double t = m[k * N + k]; m[k * N + k] = m[B * N + B]; m[B * N + B] = t; }
#pragma unroll
          for (int i = k + 1; i < B; ++i) { const double t = m[i * N + k]; m[i * N + k] = m[B * N + i]; m[B * N + i] = t; }
        }
      }
      if (k > 0) {
        double temp[N];
#pragma unroll
        for (int j = 0; j < k; ++j) temp[j] = m[j * N + j] * m[k * N + j];
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) sacc += m[k * N + j] * temp[j];
        m[k * N + k] -= sacc;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
          double t = 0.0;
#pragma unroll
          for (int j = 0; j < k; ++j) t += m[i * N + j] * temp[j];
          m[i * N + k] -= t;
        }
      }
      const double akk = m[k * N + k];
      const bool valid = fabs(akk) > 0.0;
      if (k == 0 && !valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          tr[j] = j;
#pragma unroll
          for (int i = j + 1; i < N; ++i) ret = ret && (m[i * N + j] == 0.0);
        }
        done = true;
        continue;
      }
      if (k + 1 < N) {
        if (valid) {
#pragma unroll
          for (int i = k + 1; i < N; ++i) m[i * N + k] /= akk;
        } else {
#pragma unroll
          for (int i = k + 1; i < N; ++i) ret = ret && (m[i * N + k] == 0.0);
        }
      }
      if (found_zero_pivot && valid) ret = false;
      else if (!valid) found_zero_pivot = true;
    }
    ok = ret;
  }

  // (the transpositions are applied as value selects: written as "if (tr[k] == B) swap(x[k], x[B])" the optimiser
  //  re-rolls the chain into x[tr[k]], i.e. a dynamically indexed array in scratch memory)
  DEV void solve(double *x) const {
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int B = k + 1; B < N; ++B) { const bool sw = tr[k] == B; const double a = x[k], b = x[B]; x[k] = sw ? b : a; x[B] = sw ? a : b; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { double sacc = x[i];
#pragma unroll
      for (int kk = 0; kk < i; ++kk) sacc -= m[i * N + kk] * x[kk];
      x[i] = sacc; }
#pragma unroll
    for (int i = 0; i < N; ++i) { const double dd = m[i * N + i]; x[i] = (fabs(dd) > DBL_MIN) ? x[i] / dd : 0.0; }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) { double sacc = x[i];
#pragma unroll
      for (int kk = i + 1; kk < N; ++kk) sacc -= m[kk * N + i] * x[kk];
      x[i] = sacc; }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
#pragma unroll
      for (int B = k + 1; B < N; ++B) { const bool sw = tr[k] == B; const double a = x[k], b = x[B]; x[k] = sw ? b : a; x[B] = sw ? a : b; }
    }
  }
};

// NMAX = 2 with static indexing only (the generic form's pivot / transposition arrays are dynamically indexed and
// land in scratch memory): the same steps as LDLTd<NMAX>::compute / solve written out for n in {0, 1, 2}.
template <>
struct LDLTd<2> {
  double m[4];
  int n;
  bool ok, swapped;

  DEV void compute(const double *A, int n_) {
    n = n_;
    m[0] = A[0]; m[1] = A[1]; m[2] = A[2]; m[3] = A[3];
    ok = true; swapped = false;
    if (n <= 1) return;
    swapped = fabs(m[3]) > fabs(m[0]);                    // pivot = first largest |diagonal|
    if (swapped) { const double t = m[0]; m[0] = m[3]; m[3] = t; }
    const double a00 = m[0];
    if (!(fabs(a00) > 0.0)) { swapped = false; ok = (m[2] == 0.0); return; }
    m[2] /= a00;
    const double temp0 = m[0] * m[2];
    m[3] -= m[2] * temp0;
  }
  DEV void solve(double *x) const {
    if (n <= 0) return;
    if (n == 1) { x[0] = (fabs(m[0]) > DBL_MIN) ? x[0] / m[0] : 0.0; return; }
    if (swapped) { const double v = x[0]; x[0] = x[1]; x[1] = v; }
    x[1] = x[1] - m[2] * x[0];
    x[0] = (fabs(m[0]) > DBL_MIN) ? x[0] / m[0] : 0.0;
    x[1] = (fabs(m[3]) > DBL_MIN) ? x[1] / m[3] : 0.0;
    x[0] = x[0] - m[2] * x[1];
    if (swapped) { const double v = x[0]; x[0] = x[1]; x[1] = v; }
  }
};

// scalar specialisation: 1x1 LDLT always reports Success; D^+ with tolerance DBL_MIN.
DEV double ldlt1_solve(double d, double x) { return (fabs(d) > DBL_MIN) ? x / d : 0.0; }

// MatrixXd::inverse() (PartialPivLU) -- reference clddp_solver.cpp:143.
template <int N>
DEV void inverse_pplu(const double *A, double *inv) {
  if (N == 1) { inv[0] = 1.0 / A[0]; return; }
  double lu[N * N];
  int perm[N];
  for (int i = 0; i < N * N; ++i) lu[i] = A[i];
  for (int i = 0; i < N; ++i) perm[i] = i;
  for (int k = 0; k < N; ++k) {
    int piv = k; double best = fabs(lu[k * N + k]);
    for (int i = k + 1; i < N; ++i) if (fabs(lu[i * N + k]) > best) { best = fabs(lu[i * N + k]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < N; ++j) { double t = lu[k * N + j]; lu[k * N + j] = lu[piv * N + j]; lu[piv * N + j] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < N; ++i) {
      lu[i * N + k] /= lu[k * N + k];
      for (int j = k + 1; j < N; ++j) lu[i * N + j] -= lu[i * N + k] * lu[k * N + j];
    }
  }
  for (int c = 0; c < N; ++c) {
    double y[N];
    for (int i = 0; i < N; ++i) { double s = (perm[i] == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= lu[i * N + k] * y[k]; y[i] = s; }
    for (int i = N - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < N; ++k) s -= lu[i * N + k] * inv[k * N + c]; inv[i * N + c] = s / lu[i * N + i]; }
  }
}

// min Re(eig) of Q_uu_reg -- EigenSolver(...).eigenvalues().real().minCoeff(), clddp_solver.cpp:133.
// 1x1 / 2x2: closed form for a general real matrix; larger: cyclic Jacobi on the symmetric part
// (Q_uu_reg is symmetric up to rounding; deviation noted in DESIGN.md).
template <int N>
DEV double min_real_eig(const double *M) {
  if (N == 1) return M[0];
  if (N == 2) {
    double a = M[0], b = M[1], c = M[2], d = M[3];
    double tr = a + d, det = a * d - b * c;
    double disc = 0.25 * tr * tr - det;
    if (disc < 0) return 0.5 * tr;
    return 0.5 * tr - sqrt(disc);
  }
  double S[N * N];
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) S[i * N + j] = 0.5 * (M[i * N + j] + M[j * N + i]);
  for (int sweep = 0; sweep < 60; ++sweep) {
    // Every caller only asks for the SIGN (`min_real_eig(Q) <= 0`, clddp_solver.cpp:133-140).  Once the current iterate certifies it -- all
    // Gershgorin discs right of +margin (positive definite), or a diagonal entry = a Rayleigh quotient below -margin (an eigenvalue below it)
    // -- the remaining sweeps (the reference-shaped loop runs on until the off-diagonal sum of squares underflows 1e-300) can only move the
    // answer by O(N eps |S|), six orders below the margin: same decision, a fraction of the rotations (round 6: the stack-fed CLDDP sweep at
    // nx = 12, nu = 4 repeated this test in every lane of a trajectory, 25 ms per sweep against 5 ms for the IPDDP form).
    {
      double lo = INFINITY, dmn = INFINITY, scale = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) if (j != i) r += fabs(S[i * N + j]);
        lo = dmin(lo, S[i * N + i] - r); dmn = dmin(dmn, S[i * N + i]); scale = dmax(scale, fabs(S[i * N + i]) + r);
      }
      const double margin = 1e-8 * scale;
      if (lo > margin) return lo;
      if (dmn < -margin) return dmn;
    }
    double off = 0;
    for (int i = 0; i < N; ++i) for (int j = i + 1; j < N; ++j) off += S[i * N + j] * S[i * N + j];
    if (off < 1e-300) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (S[p * N + q] == 0.0) continue;
        double theta = (S[q * N + q] - S[p * N + p]) / (2.0 * S[p * N + q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < N; ++k) { double skp = S[k * N + p], skq = S[k * N + q]; S[k * N + p] = cs * skp - sn * skq; S[k * N + q] = sn * skp + cs * skq; }
        for (int k = 0; k < N; ++k) { double spk = S[p * N + k], sqk = S[q * N + k]; S[p * N + k] = cs * spk - sn * sqk; S[q * N + k] = sn * spk + cs * sqk; }
      }
  }
  double mn = S[0];
  for (int i = 1; i < N; ++i) mn = dmin(mn, S[i * N + i]);
  return mn;
}

// ---- tiny GEMM helpers, row-major, compile-time sizes, k-ascending accumulation ----------
// C(RxC) = A^T(RxK) * B(KxC) where A is stored KxR
template <int R, int K, int C>
DEV void mm_tn(const double *A, const double *B, double *Cm) {
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) s += A[k * R + i] * B[k * C + j];
      Cm[i * C + j] = s;
    }
}
// C(RxC) = A(RxK) * B(KxC)
template <int R, int K, int C>
DEV void mm_nn(const double *A, const double *B, double *Cm) {
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * C + j];
      Cm[i * C + j] = s;
    }
}

}  // namespace cddp_dev
