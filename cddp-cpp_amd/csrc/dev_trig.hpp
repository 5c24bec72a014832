// sin / cos for the plants' rollouts and Jacobians: branch-free, several angles per call.
//
// The rollout producer is ONE dependent f64 instruction stream per wavefront.  The device libm's
// sincos() hides a divergent branch (|x| >= 2^30: Payne-Hanek), so consecutive calls land in separate basic blocks and the
// scheduler cannot interleave the three Euler angles of the quadrotor or the thirteen joint angles of the 7-joint arm;
// inlined 52 times per RK4 step it is also most of the producer's code.  Here every angle of a call goes through the same
// straight-line code (one basic block); arguments outside the fast range are redone with the libm behind ONE wave-uniform
// branch (a ballot, so no exec-mask region is opened around the library call).
//
//   reduction   dn = rint(x * 2/pi);  r + y = x - dn * pi/2 in double-double with pi/2 = P1 + P2 + P3 (53 + 53 + 53 bits):
//               fma(-dn, P1, x) is exact (both terms are multiples of 2^-52 and the difference is below 1), the P2 product
//               is split exactly with a second fma, P3 enters the tail; valid far beyond the limit used here
//   kernels     the Sun / FreeBSD msun k_sin / k_cos minimax polynomials on [-pi/4, pi/4] with the tail correction
//               (public-domain algorithm; coefficients S1..S6, C1..C6), error < 1 ulp -- the bound the glibc and device
//               libm results the parity tests already bridge (tests/test_oracle_trig_noise.py) satisfy as well
//   quadrant    value selects on dn mod 4
//
// Written with explicit __builtin_fma where a fused operation is REQUIRED (the library is built with -ffp-contract=off)
// and plain mul / add elsewhere (the polynomials are specified unfused).  Host-compilable: tests/cpp/test_dev_trig.cpp
// measures the error against long-double libm (max 0.78 ulp over 4e6 arguments up to 1e9).
#pragma once
#include <cmath>

#ifndef DEV
#define DEV inline
#define CDDP_TRIG_HOST 1
#endif

namespace cddp_dev {

constexpr double kTrigFastLimit = 1.0e9;   // |dn| < 6.4e8 < 2^30: the quadrant fits an int

struct SinCosPair { double s, c; };

DEV SinCosPair sincos_fast(double x) {
  const double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
  const double P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54, P3 = -0x1.f1976b7ed8fbcp-110;
  const double dn = __builtin_rint(x * TWO_OVER_PI);
  const double r0 = __builtin_fma(-dn, P1, x);          // exact
  const double p = dn * P2;
  const double pe = __builtin_fma(dn, P2, -p);          // exact rounding error of p
  const double r = r0 - p;
  const double bb = r - r0;                             // two-sum of r0 + (-p)
  const double e1 = (r0 - (r - bb)) + (-p - bb);
  const double y = __builtin_fma(-dn, P3, e1 - pe);     // tail: r + y = x - dn * pi/2 to ~2^-110 |dn|
  const double z = r * r, w = z * z;
  // k_sin
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double rs = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
  const double v = z * r;
  const double ks = r - ((z * (0.5 * y - v * rs) - y) - v * S1);
  // k_cos
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double rc = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  const double hz = 0.5 * z;
  const double w1 = 1.0 - hz;
  const double kc = w1 + (((1.0 - w1) - hz) + (z * rc - r * y));
  const int n = (int)dn;
  const bool swap = (n & 1) != 0;
  double s = swap ? kc : ks, c = swap ? ks : kc;
  if (n & 2) s = -s;
  if ((n + 1) & 2) c = -c;
  SinCosPair o; o.s = s; o.c = c;
  return o;
}

#ifndef CDDP_TRIG_HOST
// The libm fallback as a real function call: inlined N times it put N copies of the Payne-Hanek reduction into the calling
// kernel and its register demand (the largest of the whole rollout kernel) set the kernel's allocation.
__device__ __attribute__((noinline)) inline void sincos_libm(double a, double *s, double *c) { sincos(a, s, c); }
#endif

// sin and cos of N angles.  All fast paths first (one basic block); when ANY lane of the wavefront holds an argument outside
// the fast range (or a non-finite one) every lane evaluates the libm and keeps its result where needed.
template <int N>
DEV void sincos_n(const double *a, double *s, double *c) {
  bool slow = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const SinCosPair r = sincos_fast(a[i]);
    s[i] = r.s; c[i] = r.c;
    slow = slow || !(__builtin_fabs(a[i]) < kTrigFastLimit);   // also true for NaN / inf
  }
#ifdef CDDP_TRIG_HOST
  if (slow)
    for (int i = 0; i < N; ++i)
      if (!(__builtin_fabs(a[i]) < kTrigFastLimit)) { s[i] = std::sin(a[i]); c[i] = std::cos(a[i]); }
#else
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0ull, 0)) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double ts, tc;
      sincos_libm(a[i], &ts, &tc);
      const bool out = !(__builtin_fabs(a[i]) < kTrigFastLimit);
      s[i] = out ? ts : s[i];
      c[i] = out ? tc : c[i];
    }
  }
#endif
}

DEV void sincos_1(double a, double *s, double *c) { sincos_n<1>(&a, s, c); }

}  // namespace cddp_dev
