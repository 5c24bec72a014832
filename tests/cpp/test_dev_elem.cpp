// Accuracy of the shared log / exp / pow of cddp-cpp_amd/csrc/dev_trig.hpp (host build of the same source) against long-double libm.
// usage: test_dev_elem [n] -> prints "max_ulp_log max_ulp_exp max_rel_pow_in_ulp n ok max_ulp_asin" ; exit 0 when log, exp, asin < 1 ulp and pow < 64 ulp
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <initializer_list>
#include <cmath>
#include "../../cddp-cpp_amd/csrc/dev_trig.hpp"

static double ulp_of(double v) { double a = std::fabs(v); if (a == 0) return 4.9e-324; return std::nextafter(a, INFINITY) - a; }
static uint64_t rng_state = 0x13198A2E03707344ull;
static double urand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
  long n = argc > 1 ? std::atol(argv[1]) : 500000;
  double ml = 0, me = 0, mp = 0, ma = 0;
  for (long i = 0; i < n; ++i) {
    // log: slack-like magnitudes 1e-10 .. 1e6, values around 1 (cancellation), the whole normal range
    const double xs[3] = {std::pow(10.0, -10.0 + 16.0 * urand()), 0.5 + 1.5 * urand(), std::ldexp(1.0 + urand(), (int)(urand() * 2040) - 1020)};
    for (double x : xs) {
      const long double r = logl((long double)x);
      const double e = (double)(fabsl((long double)cddp_dev::log_shared(x) - r) / (long double)ulp_of((double)r));
      if (e > ml) ml = e;
    }
    const double t = (2.0 * urand() - 1.0) * 700.0, t2 = (2.0 * urand() - 1.0) * 2.0;
    for (double x : {t, t2}) {
      const long double r = expl((long double)x);
      const double e = (double)(fabsl((long double)cddp_dev::exp_fast(x) - r) / (long double)ulp_of((double)r));
      if (e > me) me = e;
    }
    // asin on the fast range |x| < 0.5 (uniform, and log-uniform down to 1e-12: the car's argument is sin(delta) h v / wheelbase)
    for (double x : {(2.0 * urand() - 1.0) * 0.4999999, std::pow(10.0, -12.0 + 11.7 * urand()) * (urand() < 0.5 ? -1.0 : 1.0)}) {
      const long double r = asinl((long double)x);
      const double e = (double)(fabsl((long double)cddp_dev::asin_shared(x) - r) / (long double)ulp_of((double)r));
      if (e > ma) ma = e;
    }
    // pow as the solver uses it: mu in [1e-10, 10], exponents 1.2 and 0.25 (and a random one in (0, 2))
    const double mu = std::pow(10.0, -10.0 + 11.0 * urand());
    for (double y : {1.2, 0.25, 2.0 * urand()}) {
      const long double r = powl((long double)mu, (long double)y);
      const double e = (double)(fabsl((long double)cddp_dev::pow_shared(mu, y) - r) / (long double)ulp_of((double)r));
      if (e > mp) mp = e;
    }
  }
  bool ok = true;
  // out-of-range arguments take the libm path
  ok = ok && cddp_dev::log_shared(0.0) == std::log(0.0) && std::isnan(cddp_dev::log_shared(-1.0)) && cddp_dev::log_shared(INFINITY) == INFINITY;
  ok = ok && cddp_dev::log_shared(4.9e-324) == std::log(4.9e-324);
  ok = ok && cddp_dev::pow_shared(0.0, 0.25) == 0.0 && cddp_dev::pow_shared(1.0, 1.2) == 1.0 && cddp_dev::log_shared(1.0) == 0.0;
  ok = ok && cddp_dev::pow_shared(1e300, 3.0) == std::pow(1e300, 3.0) && cddp_dev::pow_shared(1e-300, 3.0) == std::pow(1e-300, 3.0);
  ok = ok && cddp_dev::asin_shared(0.75) == std::asin(0.75) && cddp_dev::asin_shared(1.0) == std::asin(1.0) && std::isnan(cddp_dev::asin_shared(1.5)) && cddp_dev::asin_shared(0.0) == 0.0;
  std::printf("%.4f %.4f %.4f %ld %d %.4f\n", ml, me, mp, n, ok ? 1 : 0, ma);
  return (ml < 1.0 && me < 1.0 && mp < 64.0 && ma < 1.0 && ok) ? 0 : 1;
}
