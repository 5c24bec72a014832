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

// ---- log, exp, pow with the same property: straight-line code from IEEE basic operations, host-compilable -----------------
// The solver core itself calls three more libm routines: log (barrier merit, ipddp_solver.cpp:2850-2880), pow (barrier update
// mu^1.2, :2569-2614; terminal-equality regularisation mu^0.25, :556-617).  The device libm and glibc agree on them to an ulp,
// not to the bit, and over hundreds of iterations one ulp in mu moves a line-search decision (seen on the reference's N = 400
// quadrotor test: 449 vs 445 iterations).  The parity build therefore routes them through the routines below as well.
//   log   x = 2^k (1 + f), 1 + f in [sqrt(2)/2, sqrt(2)); s = f / (2 + f); log(1 + f) = 2 s + s R(s^2) with the Sun / FreeBSD
//         msun e_log.c minimax coefficients Lg1..Lg7 (public-domain algorithm), result assembled as in that routine; < 1 ulp
//   exp   k = rint(x / ln 2), r = x - k ln2_hi - k ln2_lo, e^r = 1 + (r c / (2 - c) - lo + hi), c = r - r^2 P(r^2) (msun e_exp.c
//         coefficients P1..P5), scaled by 2^k through the exponent field; < 1 ulp for |x| <= 700
//   pow   exp(y log x): relative error ~ |y log x| 2^-53 (a few ulp at mu^1.2, mu >= 1e-10) -- enough for a barrier parameter and,
//         above all, the SAME value on both sides
// Arguments outside the straight-line range (x <= 0, subnormal, non-finite, |y log x| > 700) fall back to the libm.
DEV double bits_to_double(unsigned long long u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
DEV unsigned long long double_to_bits(double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; }

DEV bool log_fast_ok(double x) { return x >= 0x1p-1022 && x <= 0x1.fffffffffffffp+1023; }   // positive, normal, finite
DEV double log_fast_k(double x, int k0) {   // log(x) + k0 ln 2 for positive normal finite x (k0: the power of two a caller scaled x by)
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  unsigned long long u = double_to_bits(x);
  unsigned hx = (unsigned)(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;                       // mantissa range [sqrt(2)/2, sqrt(2))
  const int k = (int)(hx >> 20) - 0x3ff + k0;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  u = ((unsigned long long)hx << 32) | (u & 0xffffffffull);
  const double f = bits_to_double(u) - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double dk = (double)k;
  return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}
DEV double log_fast(double x) { return log_fast_k(x, 0); }

DEV bool exp_fast_ok(double x) { return __builtin_fabs(x) <= 700.0; }   // result normal, 2^k applied through the exponent field
DEV double exp_fast(double x) {
  const double invln2 = 1.44269504088896338700e+00, ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const double dk = __builtin_rint(invln2 * x);
  const double hi = x - dk * ln2hi;          // exact for |k| < 2^11: ln2hi carries 32 significant bits
  const double lo = dk * ln2lo;
  const double r = hi - lo;
  const double xx = r * r;
  const double c = r - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
  const double y = 1.0 + (r * c / (2.0 - c) - lo + hi);
  const int k = (int)dk;
  // y in (0.7, 1.42) and |k| <= 1010: y * 2^k stays normal; multiply by the power of two built in the exponent field
  return y * bits_to_double((unsigned long long)(0x3ff + k) << 52);
}

// asin on |x| < 0.5 (the car's steering kinematics: asin(sin(delta) h v / wheelbase), |argument| << 0.5): the Sun / FreeBSD msun e_asin.c
// rational approximation asin(x) = x + x R(x^2), R = p / q (public-domain algorithm, coefficients pS0..pS5, qS1..qS4); < 1 ulp.
DEV bool asin_fast_ok(double x) { return __builtin_fabs(x) < 0.5; }
DEV double asin_fast(double x) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  if (__builtin_fabs(x) < 0x1p-26) return x;
  const double t = x * x;
  const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  return x + x * (p / q);
}

#ifdef CDDP_TRIG_HOST
inline double libm_asin(double x) { return std::asin(x); }
inline double libm_log(double x) { return std::log(x); }
inline double libm_pow(double x, double y) { return std::pow(x, y); }
#else
// Real function calls, like sincos_libm above: inlined, the device libm's log / pow (argument reduction tables, special-case
// branches) joined the register allocation of every kernel that takes a logarithm -- in the nx >= 12 rollout consumers that
// meant 896 B of scratch per lane and a 3 - 4 x slower launch (round 4, profiles/r04_trig_ab.md) for a path no finite positive
// argument ever takes.
__device__ __attribute__((noinline)) inline double libm_asin(double x) { return asin(x); }
__device__ __attribute__((noinline)) inline double libm_log(double x) { return log(x); }
__device__ __attribute__((noinline)) inline double libm_pow(double x, double y) { return pow(x, y); }
#endif
DEV double asin_shared(double x) { return asin_fast_ok(x) ? asin_fast(x) : libm_asin(x); }
// log over the whole real line WITHOUT a libm fallback (round 4): subnormal arguments are scaled by 2^54 as msun's e_log.c does,
// zero / negative / non-finite arguments get their IEEE answers through selects.  Straight-line code on both sides (the oracle's
// trig_mode 1 includes this header), and no call or libm body in the register allocation of the rollout consumers that take a
// logarithm per slack entry: with the `ok ? fast : libm` form the nx >= 12 consumers of the shared-arithmetic build spilled 896 B
// per lane and ran 3 - 4 x slower than the device-libm build (profiles/r04_trig_ab.md).
DEV double log_shared(double x) {
  const bool tiny = x < 0x1p-1022;                      // subnormal, zero or negative; false for NaN
  const double xs = tiny ? x * 0x1p54 : x;
  double r = log_fast_k(xs, tiny ? -54 : 0);            // meaningless for non-positive / non-finite x, replaced below
  r = (x == 0.0) ? -__builtin_inf() : r;
  r = (x < 0.0) ? __builtin_nan("") : r;
  r = !(x <= 0x1.fffffffffffffp+1023) ? x + x : r;      // +inf, NaN
  return r;
}
DEV double pow_shared(double x, double y) {
  if (log_fast_ok(x)) {
    const double t = y * log_fast(x);
    if (exp_fast_ok(t)) return exp_fast(t);
  }
  return libm_pow(x, y);
}

// What the solver core calls (kernels*.hpp, dev_terminal.hpp): the device libm in the product build, the shared routines in
// the parity build.
#ifndef CDDP_TRIG_HOST
#ifdef CDDP_TRIG_SHARED
DEV double solver_log(double x) { return log_shared(x); }
DEV double solver_pow(double x, double y) { return pow_shared(x, y); }
#else
DEV double solver_log(double x) { return log(x); }
DEV double solver_pow(double x, double y) { return pow(x, y); }
#endif
#endif

// sin / cos as the REFERENCE's own plants (pendulum, cart-pole, unicycle, quadrotor-13, 3-DOF manipulator) evaluate them.
// Default build: the device libm.  Parity build (-DCDDP_TRIG_SHARED, lib/libcddp_hip_sharedtrig.so, selected with the
// environment variable CDDP_HIP_TRIG=shared): the routine above -- which also compiles for the host (CDDP_TRIG_HOST), so a CPU
// checker can evaluate the very same routine and both sides run the SAME IEEE add / mul / fma sequence for every sine and cosine and
// the knife-edge plants (central-difference Jacobians amplify a last-bit libm difference 2.5e4 x) become bit-comparable.
#ifndef CDDP_TRIG_HOST
#ifdef CDDP_TRIG_SHARED
DEV void plant_sincos(double a, double *s, double *c) { sincos_1(a, s, c); }
DEV double plant_sin(double a) { double s, c; sincos_1(a, &s, &c); return s; }
DEV double plant_cos(double a) { double s, c; sincos_1(a, &s, &c); return c; }
DEV double plant_tan(double a) { double s, c; sincos_1(a, &s, &c); return s / c; }   // bicycle steering: sin / cos of the shared routine
DEV double plant_asin(double a) { return asin_shared(a); }
#else
DEV void plant_sincos(double a, double *s, double *c) { sincos(a, s, c); }
DEV double plant_sin(double a) { return sin(a); }
DEV double plant_cos(double a) { return cos(a); }
DEV double plant_tan(double a) { return tan(a); }
DEV double plant_asin(double a) { return asin(a); }
#endif
#elif defined(CDDP_HOST_MODELS)   // host_models.cpp: the plants compiled for the host (the host libm, or the shared routine)
#ifdef CDDP_TRIG_SHARED
DEV void plant_sincos(double a, double *s, double *c) { sincos_1(a, s, c); }
DEV double plant_sin(double a) { double s, c; sincos_1(a, &s, &c); return s; }
DEV double plant_cos(double a) { double s, c; sincos_1(a, &s, &c); return c; }
DEV double plant_tan(double a) { double s, c; sincos_1(a, &s, &c); return s / c; }
DEV double plant_asin(double a) { return asin_shared(a); }
#else
DEV void plant_sincos(double a, double *s, double *c) { *s = std::sin(a); *c = std::cos(a); }
DEV double plant_sin(double a) { return std::sin(a); }
DEV double plant_cos(double a) { return std::cos(a); }
DEV double plant_tan(double a) { return std::tan(a); }
DEV double plant_asin(double a) { return std::asin(a); }
#endif
#endif

}  // namespace cddp_dev
