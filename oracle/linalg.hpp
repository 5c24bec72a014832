// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Tiny fixed-capacity dense linear algebra used by the CPU restatement of cddp-cpp's
// CLDDP / IPDDP solvers.  It stands in for the third-party dependency the reference uses
// for every matrix operation: Eigen 3.4.0 (reference CMakeLists.txt:65-97, FetchContent tag
// 3.4.0), which is NOT vendored under /root/reference and not installed in this image.
// The pieces whose *semantics* matter for solver decisions are restated from Eigen 3.4's
// published algorithms:
//   * LDLT  : Eigen/src/Cholesky/LDLT.h  ldlt_inplace<Lower>::unblocked  (diagonal pivoting on
//             the largest |D_ii|, "Success" unless a valid pivot follows a zero pivot) and
//             LDLT::_solve_impl (pseudo-inverse of D with tolerance = numeric_limits::min()).
//   * inverse(): dynamic-size MatrixXd::inverse() goes through PartialPivLU.
//   * EigenSolver(...).eigenvalues().real().minCoeff(): restated for symmetric input
//             (Jacobi); 1x1 and 2x2 general matrices use the closed form.
//   * JacobiSVD singular values: one-sided Jacobi (Hestenes).
// parity unpinned: no golden vector of the reference pins these numerics.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

namespace oracle {

// Capacity (rows*cols) of every Mat / Vec.  512 covers any layout on the path (m x nx <= 32 x 16) and is what the parity
// build uses.  sizeof(Mat) = 8 + 8 * cap: at 512 every std::vector<Vec> trajectory copy of a line-search trial is a
// >128 KB allocation, i.e. an mmap + page faults per trial, and 256 host threads serialise in the kernel's address-space
// lock (round 1: 8.3x speed-up on 256 cores).  The timing build of bench.py's cpu_baseline therefore compiles with
// -DORACLE_MAT_CAP=<the largest matrix of the workload> (a Mat that does not fit aborts, it never truncates).
#ifndef ORACLE_MAT_CAP
#define ORACLE_MAT_CAP 512
#endif
constexpr int kMatCap = ORACLE_MAT_CAP;

// Summation-ORDER knob (test infrastructure, default 0 = every sum serial, left to right -- the order the HIP kernels keep).
// assoc_mode() == 1 models the association Eigen 3.4 gives the reference's sums in a plain x86-64 Release build (SSE2, 2-wide Packet2d,
// no FMA; the reference cannot be built here, so this is a reading of Eigen's kernels, not a measurement -- DESIGN.md section 5):
//   * redux (Redux.h, LinearVectorizedTraversal, no unrolling: .dot, .sum, .squaredNorm / .norm, .lpNorm<1>, inner products v^T w):
//     two Packet2d accumulators over the indices {0,1},{4,5},... and {2,3},{6,7},..., added to each other, a leftover whole packet added,
//     horizontal add lane 0 + lane 1, then the odd tail element -- redux_order();
//   * row-major matrix x vector (GeneralMatrixVector.h, RowMajor: X.transpose() * v and v.transpose() * X of column-major X): ONE Packet2d
//     accumulator per output row, started from zero, over the index pairs in order, horizontal add, then the scalar tail -- gemv_row_order();
//   * column-major matrix x vector, and every matrix x matrix product (lazy coefficient-based below the GEMM threshold, the GEBP kernel above
//     it): each output accumulates over the inner index in order -- serial, as in mode 0.
// What is NOT modelled: runtime pointer alignment of block expressions (first_default_aligned can shift the packets by one element),
// expression templates that fuse differently from the restated statement, dynamic-size heuristics above 128 columns.
inline int &assoc_mode() { static int v = 0; return v; }
inline double serial_order(const double *p, int n) { double s = 0.0; for (int i = 0; i < n; ++i) s += p[i]; return s; }
inline double redux_order(const double *p, int n) {
  const int ps = 2;
  const int alignedSize2 = (n / (2 * ps)) * (2 * ps), alignedSize = (n / ps) * ps;
  if (!alignedSize) { double res = n > 0 ? p[0] : 0.0; for (int i = 1; i < n; ++i) res += p[i]; return res; }
  double a0 = p[0], a1 = p[1];
  if (alignedSize > ps) {
    double b0 = p[2], b1 = p[3];
    for (int i = 2 * ps; i < alignedSize2; i += 2 * ps) { a0 += p[i]; a1 += p[i + 1]; b0 += p[i + 2]; b1 += p[i + 3]; }
    a0 += b0; a1 += b1;
    if (alignedSize > alignedSize2) { a0 += p[alignedSize2]; a1 += p[alignedSize2 + 1]; }
  }
  double res = a0 + a1;
  for (int i = alignedSize; i < n; ++i) res += p[i];
  return res;
}
inline double gemv_row_order(const double *p, int n) {
  double c0 = 0.0, c1 = 0.0;
  int j = 0;
  for (; j + 2 <= n; j += 2) { c0 = p[j] + c0; c1 = p[j + 1] + c1; }
  double cc = c0 + c1;
  for (; j < n; ++j) cc += p[j];
  return 0.0 + cc;   // res[i] += alpha * cc0 on a zeroed destination
}

struct Mat {
  int r = 0, c = 0;
  bool tv = false;   // this value is X.transpose() of a column-major X used as an operand (set by T(), consumed by operator*; copies drop it)
  double a[kMatCap];
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_) {
    if (r * c > kMatCap) { std::fprintf(stderr, "oracle::Mat capacity exceeded %dx%d\n", r, c); std::abort(); }
    for (int i = 0; i < r * c; ++i) a[i] = 0.0;
  }
  Mat(const Mat &o) : r(o.r), c(o.c) { for (int i = 0; i < r * c; ++i) a[i] = o.a[i]; }
  Mat &operator=(const Mat &o) { r = o.r; c = o.c; tv = false; for (int i = 0; i < r * c; ++i) a[i] = o.a[i]; return *this; }
  double &operator()(int i, int j) { return a[i * c + j]; }
  double operator()(int i, int j) const { return a[i * c + j]; }
  double &operator()(int i) { return a[i]; }           // vector access (c==1 or r==1)
  double operator()(int i) const { return a[i]; }
  int size() const { return r * c; }
  static Mat Zero(int r, int c = 1) { return Mat(r, c); }
  static Mat Ones(int r, int c = 1) { Mat m(r, c); for (int i = 0; i < r * c; ++i) m.a[i] = 1.0; return m; }
  static Mat Identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  static Mat FromPtr(const double *p, int r, int c = 1) {
    Mat m(r, c); for (int i = 0; i < r * c; ++i) m.a[i] = p[i]; return m;
  }
  bool allFinite() const { for (int i = 0; i < r * c; ++i) if (!std::isfinite(a[i])) return false; return true; }
  Mat T() const { Mat m(c, r); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m(j, i) = (*this)(i, j); m.tv = !tv; return m; }
  Mat plain() const { Mat m(*this); m.tv = false; return m; }   // a transposed value stored in a matrix of its own (MatrixXd X = Y.transpose())
  double lpNormInf() const { double v = 0; for (int i = 0; i < r * c; ++i) v = std::max(v, std::fabs(a[i])); return v; }
  double lpNorm1() const { if (assoc_mode() == 1) { double p[kMatCap]; for (int i = 0; i < r * c; ++i) p[i] = std::fabs(a[i]); return redux_order(p, r * c); }
                           double v = 0; for (int i = 0; i < r * c; ++i) v += std::fabs(a[i]); return v; }
  double squaredNorm() const { if (assoc_mode() == 1) { double p[kMatCap]; for (int i = 0; i < r * c; ++i) p[i] = a[i] * a[i]; return redux_order(p, r * c); }
                               double v = 0; for (int i = 0; i < r * c; ++i) v += a[i] * a[i]; return v; }
  double norm() const { return std::sqrt(squaredNorm()); }
  double dot(const Mat &o) const { if (assoc_mode() == 1) { double p[kMatCap]; for (int i = 0; i < r * c; ++i) p[i] = a[i] * o.a[i]; return redux_order(p, r * c); }
                                   double v = 0; for (int i = 0; i < r * c; ++i) v += a[i] * o.a[i]; return v; }
  double trace() const { double v = 0; for (int i = 0; i < std::min(r, c); ++i) v += (*this)(i, i); return v; }
  double minCoeff() const { double v = a[0]; for (int i = 1; i < r * c; ++i) v = std::min(v, a[i]); return v; }
  double maxCoeff() const { double v = a[0]; for (int i = 1; i < r * c; ++i) v = std::max(v, a[i]); return v; }
  Mat row(int i) const { Mat m(1, c); for (int j = 0; j < c; ++j) m(0, j) = (*this)(i, j); return m; }
  Mat col(int j) const { Mat m(r, 1); for (int i = 0; i < r; ++i) m(i, 0) = (*this)(i, j); return m; }
  Mat block(int i0, int j0, int nr, int nc) const {
    Mat m(nr, nc); for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) m(i, j) = (*this)(i0 + i, j0 + j); return m;
  }
  void setBlock(int i0, int j0, const Mat &b) {
    for (int i = 0; i < b.r; ++i) for (int j = 0; j < b.c; ++j) (*this)(i0 + i, j0 + j) = b(i, j);
  }
  Mat segment(int i0, int n) const { Mat m(n, 1); for (int i = 0; i < n; ++i) m.a[i] = a[i0 + i]; return m; }
  void setSegment(int i0, const Mat &v) { for (int i = 0; i < v.size(); ++i) a[i0 + i] = v.a[i]; }
};
using Vec = Mat;  // column vector: c == 1

inline Mat operator+(const Mat &A, const Mat &B) {
  assert(A.r == B.r && A.c == B.c);
  Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = A.a[i] + B.a[i]; return m;
}
inline Mat operator-(const Mat &A, const Mat &B) {
  assert(A.r == B.r && A.c == B.c);
  Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = A.a[i] - B.a[i]; return m;
}
inline Mat operator-(const Mat &A) { Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = -A.a[i]; return m; }
inline Mat operator*(double s, const Mat &A) { Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = s * A.a[i]; return m; }
inline Mat operator*(const Mat &A, double s) { return s * A; }
inline Mat operator/(const Mat &A, double s) { Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = A.a[i] / s; return m; }
inline Mat &operator+=(Mat &A, const Mat &B) { assert(A.r == B.r && A.c == B.c); for (int i = 0; i < A.size(); ++i) A.a[i] += B.a[i]; return A; }
inline Mat &operator-=(Mat &A, const Mat &B) { assert(A.r == B.r && A.c == B.c); for (int i = 0; i < A.size(); ++i) A.a[i] -= B.a[i]; return A; }
// Summation-order noise knob (test infrastructure, default off): with matmul_noise() != 0 every entry of a matrix product
// with an inner dimension >= 2 is moved by -1 / 0 / +1 ulp (hash of the value) -- what a different accumulation order
// (Eigen's vectorised kernels, a matrix-core instruction, FMA contraction) does to a dot product.  The oracle-vs-noisy-
// oracle decision-flip rate is the yardstick for kernels that cannot keep the reference's summation order
// (cddp-cpp_amd/csrc/kernels_mfma.hpp; tests/test_mfma_sweep.py).
inline int &matmul_noise() { static int v = 0; return v; }
inline double matmul_perturb(double s) {
  unsigned long long h; static_assert(sizeof(h) == sizeof(s), "");
  __builtin_memcpy(&h, &s, sizeof(h));
  h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  switch (h % 3u) { case 0: return s; case 1: return std::nextafter(s, std::numeric_limits<double>::infinity());
                    default: return std::nextafter(s, -std::numeric_limits<double>::infinity()); }
}
// Plain triple loop, k innermost, ascending: sum_k A(i,k) B(k,j).
inline Mat operator*(const Mat &A, const Mat &B) {
  assert(A.c == B.r);
  Mat m(A.r, B.c);
  const bool noisy = matmul_noise() != 0 && A.c >= 2;
  // assoc_mode 1: which Eigen kernel the product runs through (see assoc_mode()): 0 serial, 1 redux (inner product), 2 row-major gemv
  int kind = 0;
  if (assoc_mode() == 1 && A.c >= 2) {
    if (A.r == 1 && B.c == 1) kind = 1;
    else if (B.c == 1) kind = A.tv ? 2 : 0;          // X^T v: row-major lhs; X v: column-major lhs
    else if (A.r == 1) kind = B.tv ? 0 : 2;          // v^T X = (X^T v)^T
  }
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < B.c; ++j) {
      double s = 0.0;
      if (kind == 0) { for (int k = 0; k < A.c; ++k) s += A(i, k) * B(k, j); }
      else {
        double p[kMatCap];
        for (int k = 0; k < A.c; ++k) p[k] = A(i, k) * B(k, j);
        s = kind == 1 ? redux_order(p, A.c) : gemv_row_order(p, A.c);
      }
      m(i, j) = noisy ? matmul_perturb(s) : s;
    }
  return m;
}
inline Mat cwiseProduct(const Mat &A, const Mat &B) { Mat m(A.r, A.c); for (int i = 0; i < m.size(); ++i) m.a[i] = A.a[i] * B.a[i]; return m; }
inline Mat symmetrize(const Mat &A) { return 0.5 * (A + A.T()); }   // ipddp_solver.cpp:217-220
inline Mat diagTimes(const Mat &d, const Mat &A) {  // diag(d) * A
  Mat m(A.r, A.c); for (int i = 0; i < A.r; ++i) for (int j = 0; j < A.c; ++j) m(i, j) = d.a[i] * A(i, j); return m;
}

// ---------------------------------------------------------------------------------------
// Eigen::LDLT<MatrixXd> (Lower), restated from Eigen 3.4.0 Eigen/src/Cholesky/LDLT.h.
// Call sites in the reference: boxqp.cpp:105,147; ipddp_solver.cpp:456,583,1087,1428.
// ---------------------------------------------------------------------------------------
struct LDLT {
  Mat m;                    // packed L (strict lower) and D (diagonal)
  std::vector<int> transp;  // transpositions
  bool ok = false;          // info() == Success
  int n = 0;

  LDLT() {}
  explicit LDLT(const Mat &A) { compute(A); }

  void compute(const Mat &A) {
    n = A.r;
    m = A;
    transp.assign(n, 0);
    ok = true;
    if (n == 0) return;
    if (n == 1) { transp[0] = 0; ok = true; return; }  // unblocked(): size<=1 -> true
    bool found_zero_pivot = false;
    bool ret = true;
    std::vector<double> temp(n);
    for (int k = 0; k < n; ++k) {
      // largest |diagonal| in the remaining corner (first maximum wins, as maxCoeff(&idx))
      int big = k;
      double bigv = std::fabs(m(k, k));
      for (int i = k + 1; i < n; ++i) {
        double v = std::fabs(m(i, i));
        if (v > bigv) { bigv = v; big = i; }
      }
      transp[k] = big;
      if (k != big) {
        int s = n - big - 1;
        for (int j = 0; j < k; ++j) std::swap(m(k, j), m(big, j));          // row heads
        for (int i = 0; i < s; ++i) std::swap(m(big + 1 + i, k), m(big + 1 + i, big));  // col tails
        std::swap(m(k, k), m(big, big));
        for (int i = k + 1; i < big; ++i) {
          double tmp = m(i, k);
          m(i, k) = m(big, i);
          m(big, i) = tmp;
        }
      }
      int rs = n - k - 1;
      if (k > 0) {
        for (int j = 0; j < k; ++j) temp[j] = m(j, j) * m(k, j);   // D(0:k) .* A10^T
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += m(k, j) * temp[j];
        m(k, k) -= s;
        for (int i = 0; i < rs; ++i) {
          double t = 0.0;
          for (int j = 0; j < k; ++j) t += m(k + 1 + i, j) * temp[j];
          m(k + 1 + i, k) -= t;
        }
      }
      double realAkk = m(k, k);
      bool pivot_is_valid = (std::fabs(realAkk) > 0.0);
      if (k == 0 && !pivot_is_valid) {
        // the entire diagonal is zero: success iff the matrix is strictly-lower zero
        for (int j = 0; j < n; ++j) {
          transp[j] = j;
          for (int i = j + 1; i < n; ++i) ret = ret && (m(i, j) == 0.0);
        }
        ok = ret;
        return;
      }
      if (rs > 0 && pivot_is_valid) {
        for (int i = 0; i < rs; ++i) m(k + 1 + i, k) /= realAkk;
      } else if (rs > 0) {
        for (int i = 0; i < rs; ++i) ret = ret && (m(k + 1 + i, k) == 0.0);
      }
      if (found_zero_pivot && pivot_is_valid) ret = false;  // factorisation failed
      else if (!pivot_is_valid) found_zero_pivot = true;
    }
    ok = ret;
  }

  // X = A^{-1} B  (B: n x c).  dst = P^T L^-T D^+ L^-1 P b
  Mat solve(const Mat &B) const {
    Mat X = B;
    const int cols = B.c;
    for (int k = 0; k < n; ++k) {  // apply transpositions P
      int t = transp[k];
      if (t != k) for (int j = 0; j < cols; ++j) std::swap(X(k, j), X(t, j));
    }
    for (int j = 0; j < cols; ++j)  // L^-1 (unit lower)
      for (int i = 0; i < n; ++i) {
        double s = X(i, j);
        for (int kk = 0; kk < i; ++kk) s -= m(i, kk) * X(kk, j);
        X(i, j) = s;
      }
    const double tol = std::numeric_limits<double>::min();
    for (int i = 0; i < n; ++i) {   // pseudo-inverse of D
      double d = m(i, i);
      if (std::fabs(d) > tol) { for (int j = 0; j < cols; ++j) X(i, j) /= d; }
      else { for (int j = 0; j < cols; ++j) X(i, j) = 0.0; }
    }
    for (int j = 0; j < cols; ++j)  // L^-T
      for (int i = n - 1; i >= 0; --i) {
        double s = X(i, j);
        for (int kk = i + 1; kk < n; ++kk) s -= m(kk, i) * X(kk, j);
        X(i, j) = s;
      }
    for (int k = n - 1; k >= 0; --k) {  // P^T
      int t = transp[k];
      if (t != k) for (int j = 0; j < cols; ++j) std::swap(X(k, j), X(t, j));
    }
    return X;
  }
};

// MatrixXd::inverse() for dynamic sizes: PartialPivLU, then solve against identity
// (reference clddp_solver.cpp:143; manipulator.cpp:45 M.inverse()).
inline Mat inversePartialPivLU(const Mat &A) {
  const int n = A.r;
  Mat lu = A;
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(lu(k, k));
    for (int i = k + 1; i < n; ++i) if (std::fabs(lu(i, k)) > best) { best = std::fabs(lu(i, k)); piv = i; }
    if (piv != k) { for (int j = 0; j < n; ++j) std::swap(lu(k, j), lu(piv, j)); std::swap(perm[k], perm[piv]); }
    for (int i = k + 1; i < n; ++i) {
      lu(i, k) /= lu(k, k);
      for (int j = k + 1; j < n; ++j) lu(i, j) -= lu(i, k) * lu(k, j);
    }
  }
  Mat inv(n, n);
  for (int c = 0; c < n; ++c) {
    std::vector<double> y(n);
    for (int i = 0; i < n; ++i) {  // forward (unit lower) on permuted identity column
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= lu(i, k) * y[k];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < n; ++k) s -= lu(i, k) * inv(k, c);
      inv(i, c) = s / lu(i, i);
    }
  }
  return inv;
}

// min over Re(eigenvalues) -- EigenSolver<MatrixXd>(M).eigenvalues().real().minCoeff()
// (reference clddp_solver.cpp:133-134).  1x1 / 2x2: closed form for a general real matrix;
// larger: cyclic Jacobi on the symmetric part (Q_uu_reg is symmetric up to rounding).
inline double minRealEigenvalue(const Mat &M) {
  const int n = M.r;
  if (n == 1) return M(0, 0);
  if (n == 2) {
    double a = M(0, 0), b = M(0, 1), c = M(1, 0), d = M(1, 1);
    double tr = a + d, det = a * d - b * c;
    double disc = 0.25 * tr * tr - det;
    if (disc < 0) return 0.5 * tr;
    return 0.5 * tr - std::sqrt(disc);
  }
  Mat S = symmetrize(M);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += S(i, j) * S(i, j);
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (S(p, q) == 0.0) continue;
        double theta = (S(q, q) - S(p, p)) / (2.0 * S(p, q));
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k) {
          double skp = S(k, p), skq = S(k, q);
          S(k, p) = cs * skp - sn * skq;
          S(k, q) = sn * skp + cs * skq;
        }
        for (int k = 0; k < n; ++k) {
          double spk = S(p, k), sqk = S(q, k);
          S(p, k) = cs * spk - sn * sqk;
          S(q, k) = sn * spk + cs * sqk;
        }
      }
  }
  double mn = S(0, 0);
  for (int i = 1; i < n; ++i) mn = std::min(mn, S(i, i));
  return mn;
}

// Singular values of A (JacobiSVD<MatrixXd>(A).singularValues(), ipddp_solver.cpp:566-569):
// one-sided Jacobi on the columns.
inline std::vector<double> singularValues(const Mat &A) {
  Mat U = A;
  const int rr = U.r, cc = U.c;
  for (int sweep = 0; sweep < 80; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < cc; ++p)
      for (int q = p + 1; q < cc; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < rr; ++i) { alpha += U(i, p) * U(i, p); beta += U(i, q) * U(i, q); gamma += U(i, p) * U(i, q); }
        if (std::fabs(gamma) <= 1e-300 || std::fabs(gamma) <= 1e-16 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < rr; ++i) {
          double up = U(i, p), uq = U(i, q);
          U(i, p) = cs * up - sn * uq;
          U(i, q) = sn * up + cs * uq;
        }
      }
    if (!rotated) break;
  }
  std::vector<double> sv(cc);
  for (int j = 0; j < cc; ++j) { double s = 0; for (int i = 0; i < rr; ++i) s += U(i, j) * U(i, j); sv[j] = std::sqrt(s); }
  return sv;
}

}  // namespace oracle
