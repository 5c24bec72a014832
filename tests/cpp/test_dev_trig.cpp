// Accuracy of cddp-cpp_amd/csrc/dev_trig.hpp (host build of the same source) against long-double libm.
// usage: test_dev_trig [n]   -> prints "max_ulp_sin max_ulp_cos n ok" and exits 0 when both are < 1.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include "../../cddp-cpp_amd/csrc/dev_trig.hpp"

static double ulp_of(double v) {
  double a = std::fabs(v);
  if (a == 0) return 4.9e-324;
  return std::nextafter(a, INFINITY) - a;
}
static uint64_t rng_state = 0x243F6A8885A308D3ull;
static double urand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
  long n = argc > 1 ? std::atol(argv[1]) : 500000;
  double ms = 0, mc = 0, worst_s = 0, worst_c = 0;
  auto check = [&](double x) {
    double s, c; cddp_dev::sincos_1(x, &s, &c);
    long double rs = sinl((long double)x), rc = cosl((long double)x);
    double es = (double)(fabsl((long double)s - rs) / (long double)ulp_of((double)rs));
    double ec = (double)(fabsl((long double)c - rc) / (long double)ulp_of((double)rc));
    if (es > ms) { ms = es; worst_s = x; }
    if (ec > mc) { mc = ec; worst_c = x; }
  };
  const double ranges[] = {0.8, 3.2, 7.0, 30.0, 1000.0, 1.0e6, 9.9e8};
  for (double R : ranges) for (long i = 0; i < n; ++i) check((2.0 * urand() - 1.0) * R);
  // neighbourhoods of the multiples of pi/2 (cancellation in the reduction) and tiny arguments
  for (int k = -4000; k <= 4000; ++k) for (int j = -8; j <= 8; ++j) {
    double x = k * 1.5707963267948966; for (int q = 0; q < (j < 0 ? -j : j); ++q) x = std::nextafter(x, j < 0 ? -INFINITY : INFINITY);
    check(x);
  }
  for (int e = -300; e < 0; e += 3) { check(std::ldexp(1.1, e)); check(-std::ldexp(1.7, e)); }
  // out-of-range and non-finite arguments take the libm path
  double s, c; cddp_dev::sincos_1(1.0e12, &s, &c);
  bool ok = s == std::sin(1.0e12) && c == std::cos(1.0e12);
  cddp_dev::sincos_1(INFINITY, &s, &c); ok = ok && std::isnan(s) && std::isnan(c);
  cddp_dev::sincos_1(NAN, &s, &c); ok = ok && std::isnan(s) && std::isnan(c);
  cddp_dev::sincos_1(0.0, &s, &c); ok = ok && s == 0.0 && c == 1.0;
  // batched form equals the single form
  double a3[3] = {0.3, -2.9, 1234.5}, s3[3], c3[3]; cddp_dev::sincos_n<3>(a3, s3, c3);
  for (int i = 0; i < 3; ++i) { cddp_dev::sincos_1(a3[i], &s, &c); ok = ok && s == s3[i] && c == c3[i]; }
  std::printf("%.4f %.4f %ld %d worst_at %.17g %.17g\n", ms, mc, n, ok ? 1 : 0, worst_s, worst_c);
  return (ms < 1.0 && mc < 1.0 && ok) ? 0 : 1;
}
